#!/usr/bin/env python3
"""tools/lab/skew_stack.py — does the compositor care how the 32 layer buffers are aligned against each other?  Same S2 stack, layers as slices of one
allocation whose starts differ by a whole number of 2 MiB slots plus l * skew bytes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h, n = 7680, 4320, 32
dev = torch.device("cuda", 0)
stack, modes, opac = bench.synth_stack(torch, dev, w, h, n, seed=0x5EED0002)
size = w * h * 4
slot = ((size + (2 << 20) - 1) // (2 << 20)) * (2 << 20) + (4 << 20)
big = torch.empty(slot * n + (8 << 20), dtype=torch.uint8, device=dev)
base = big.data_ptr(); base_al = (base + (2 << 20) - 1) // (2 << 20) * (2 << 20)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); ref = torch.empty_like(flat)
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
r.flatten_dev([stack[k].data_ptr() for k in range(n)], info, w, h, ref.data_ptr())
for rnd in range(2):
    for skew in (None, 0, 256, 4352, 66048, 768, 33024, 2304):
        if skew is None:
            ptrs = [stack[k].data_ptr() for k in range(n)]; tag = "torch tensors"
        else:
            ptrs = []
            for k in range(n):
                off = (base_al - base) + slot * k + skew * k
                big[off:off + size].copy_(stack[k].reshape(-1)); ptrs.append(base + off)
            tag = f"skew {skew}"
        for _ in range(10): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(40): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
        torch.cuda.synchronize(); r.timing_enable(False)
        ok = bool(torch.equal(flat, ref))
        print(f"{tag:16s} flatten {r.timing_read('flatten')[0] / 40:.4f} ms  same result: {ok}", flush=True)
