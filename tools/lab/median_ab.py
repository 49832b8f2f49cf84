#!/usr/bin/env python3
"""tools/lab/median_ab.py — median at 8K, radii 2 .. 8: which kernel family per radius (pfx_tune median_bits_min = first radius of the bit-plane select; below it the
shared-column networks (<= 4) or the value search)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
S, D = src.data_ptr(), dst.data_ptr()
def t(rad):
    for _ in range(4): r.median_dev(S, D, w, h, rad)
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(8): r.median_dev(S, D, w, h, rad)
    torch.cuda.synchronize(); r.timing_enable(False)
    return r.timing_read("median")[0] / 8
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1: r.median_dev(S, D, w, h, 2)
for rad in (2, 3, 4, 5, 6, 8):
    row = []
    for mn in (2, 3, 4, 5, 9):
        r.tune("median_bits_min", mn)
        row.append(f"bits_from={mn}:{min(t(rad), t(rad)):.4f}")
    print(f"r={rad} " + "  ".join(row), flush=True)
