#!/usr/bin/env python3
"""tools/lab/box_sweep.py — two-pass box blur at 8K: outputs per lane of the horizontal (box_px) and vertical (box_py) pass, per radius"""
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
    ms, c = r.timing_read("box_blur")
    return ms / 20
for rad in (5.0, 9.0, 16.0, 24.0, 48.0, 100.0):
    best = {}
    for rep in range(2):
        for px in (4, 8, 16):
            for py in (16, 32, 64, 128):
                r.tune("box_px", px); r.tune("box_py", py)
                v = t(rad); best[(px, py)] = min(best.get((px, py), 1e9), v)
    r.tune("box_px", 0); r.tune("box_py", 0)
    auto = min(t(rad), t(rad))
    row = "  ".join(f"{px}/{py}:{best[(px, py)]:.4f}" for px in (4, 8, 16) for py in (16, 32, 64, 128))
    print(f"r={rad:g} auto {auto:.4f} | {row}", flush=True)
