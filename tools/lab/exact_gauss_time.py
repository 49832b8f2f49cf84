#!/usr/bin/env python3
"""tools/lab/exact_gauss_time.py — the bit-exact Gaussian (pfx_ctx_set_exact; what sharpen / glow / drop shadow and the batch pipeline run since round 5) at 8K per sigma,
next to the default (matrix-core) mode"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
for kv in sys.argv[1:]:
    k, v = kv.split("="); r.tune(k, int(v))
w, h = 7680, 4320
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
for sigma in (0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 5.0, 5.33, 6.0, 8.0, 16.0):
    row = []
    for exact in (False, True):
        r.set_exact(exact)
        for _ in range(3): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, sigma)
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(8): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, sigma)
        torch.cuda.synchronize(); r.timing_enable(False)
        t = {k: r.timing_read(k)[0] / 8 for k in ("gauss_mfma", "gauss_h", "gauss_v", "gauss_fused")}
        row.append(("exact  " if exact else "default") + " " + "  ".join(f"{k} {v:.3f}" for k, v in t.items() if v > 0) + f"  = {sum(t.values()):.3f} ms")
    r.set_exact(False)
    print(f"sigma {sigma:5.1f} (r = {int(-(-3 * sigma // 1))}): " + "   |   ".join(row))
