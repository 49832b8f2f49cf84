#!/usr/bin/env python3
"""tools/lab/box_prefix_ab.py — two-pass box blur at 8K: sliding-window horizontal pass (box_prefix_from = 0) against the prefix-sum one (forced from radius 1)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
S, D = src.data_ptr(), dst.data_ptr()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1: r.box_blur_dev(S, D, w, h, 9.0)
torch.cuda.synchronize()
def t(rad):
    for _ in range(8): r.box_blur_dev(S, D, w, h, rad)
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(20): r.box_blur_dev(S, D, w, h, rad)
    torch.cuda.synchronize(); r.timing_enable(False)
    return r.timing_read("box_blur")[0] / 20
for rad in (5.0, 9.0, 12.0, 16.0, 24.0, 32.0, 48.0, 100.0, 300.0):
    res = []
    for rep in range(3):
        for frm in (0, 1):
            r.tune("box_prefix_from", frm); res.append(t(rad))
    print(f"r={rad:g}  sliding {min(res[0::2]):.4f}  prefix {min(res[1::2]):.4f}", flush=True)
