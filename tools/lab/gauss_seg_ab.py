#!/usr/bin/env python3
"""tools/lab/gauss_seg_ab.py — row segments per strip of the matrix-core Gaussian (pfx_tune gauss_mfma_segments, 0 = automatic) at 8K and on band shapes"""
import os, sys, time, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, H = 7680, 4320
src = torch.randint(0, 256, (H, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
S, D = src.data_ptr(), dst.data_ptr()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1: r.gaussian_blur_dev(S, D, w, H, 16.0)
torch.cuda.synchronize()
random.seed(3)
for (rows, sigma) in ((4320, 16.0), (2272, 16.0), (1184, 16.0), (672, 16.0), (4320, 4.0), (2160, 16.0)):
    cand = [0, 1, 2, 3, 4, 6]
    res = {c: [] for c in cand}
    for rep in range(5):
        order = cand[:]; random.shuffle(order)
        for seg in order:
            r.tune("gauss_mfma_segments", seg)
            for _ in range(10): r.gaussian_blur_dev(S, D, w, rows, sigma)
            torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
            for _ in range(30): r.gaussian_blur_dev(S, D, w, rows, sigma)
            torch.cuda.synchronize(); r.timing_enable(False)
            ms, c = r.timing_read("gauss_mfma")
            res[seg].append(ms / c)
    r.tune("gauss_mfma_segments", 0)
    print(f"rows {rows} sigma {sigma:g}: " + "  ".join(f"{c}:{sorted(res[c])[len(res[c]) // 2]:.4f}" for c in cand), flush=True)
