#!/usr/bin/env python3
"""what a live mask on one layer costs the compositor (it sends the stack through the general kernel): 8K, 9 and 32 layers"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
dev = torch.device("cuda", 0)
stack, modes, opac = bench.synth_stack(torch, dev, w, h, 32, seed=0x5EED0002)
mask = torch.randint(0, 256, (h, w), dtype=torch.uint8, device=dev)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
for n in (9, 32):
    ptrs = [stack[k].data_ptr() for k in range(n)]
    info = [(k, float(opac[k]), True, int(modes[k]) if modes[k] != 14 else 1) for k in range(n)]
    for label, masks in (("no mask", None), ("mask on layer 3", [mask.data_ptr() if k == 3 else 0 for k in range(n)])):
        for _ in range(10): r.flatten_dev(ptrs, info, w, h, flat.data_ptr(), masks)
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(20): r.flatten_dev(ptrs, info, w, h, flat.data_ptr(), masks)
        torch.cuda.synchronize(); r.timing_enable(False)
        print(f"{n} layers, {label}: {r.timing_read('flatten')[0] / 20:.4f} ms")
