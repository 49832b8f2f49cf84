#!/usr/bin/env python3
"""tools/time_gauss_sigma.py — Gaussian blur at 8K over the reference dialog's sigma range (0.1 .. 100): which kernel family serves which sigma"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
r.tune("gauss_v_cfg", cfg)
for sigma in ((0.5, 1.0, 2.0, 4.0, 8.0, 10.0, 16.0) if cfg == 0 else ()) + (17.0, 24.0, 32.0, 40.0, 50.0, 64.0, 75.0, 100.0):
    try:
        for _ in range(3): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, sigma)
    except Exception as e:
        print(f'v_cfg={cfg} sigma={sigma}: {e}'); continue
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(5): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, sigma)
    torch.cuda.synchronize(); r.timing_enable(False)
    t = {k: r.timing_read(k)[0] / 5 for k in ("gauss_mfma", "gauss_h", "gauss_v", "gaussian")}
    print(f"v_cfg={cfg} sigma={sigma}: " + "  ".join(f"{k} {v:.3f}" for k, v in t.items() if v > 0))
