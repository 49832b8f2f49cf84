#!/usr/bin/env python3
"""where the elimination kernel starts to pay: S2's first n layers (the Overwrite reset is layer 14), elimination forced on (dle_min_layers=0) / off (99)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
dev = torch.device("cuda", 0)
stack, modes, opac = bench.synth_stack(torch, dev, w, h, 32, seed=0x5EED0002)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
for _ in range(100): r.flatten_dev([stack[k].data_ptr() for k in range(9)], [(k, 1.0, True, 1) for k in range(9)], w, h, flat.data_ptr())
for n, allnormal in [(n, a) for a in (False, True) for n in (10, 12, 14, 15, 16, 20, 32)]:
    ptrs = [stack[k].data_ptr() for k in range(n)]
    info = [(k, float(opac[k]), True, 0 if allnormal else int(modes[k])) for k in range(n)]
    res = []
    for gate in (0, 99):
        r.tune("dle_min_layers", gate)
        for _ in range(10): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(20): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
        torch.cuda.synchronize(); r.timing_enable(False)
        res.append(r.timing_read("flatten")[0] / 20)
    r.tune("dle_min_layers", 16)
    tag = " all Normal" if allnormal else ""
    print(f"{n} layers{tag}: elimination {res[0]:.4f} ms   plain {res[1]:.4f} ms   ratio {res[0] / res[1]:.3f}")
