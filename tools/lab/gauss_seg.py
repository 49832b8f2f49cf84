#!/usr/bin/env python3
"""row segments per strip of the matrix-core Gaussian (pfx_tune gauss_mfma_segments; 0 = automatic) at several sigmas, 8K"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
for _ in range(60): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, 16.0)
for sigma in (4.0, 16.0):
    res = []
    for seg in (0, 1, 2, 3, 4):
        r.tune("gauss_mfma_segments", seg)
        for _ in range(5): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, sigma)
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(20): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, sigma)
        torch.cuda.synchronize(); r.timing_enable(False)
        res.append(f"seg {seg}: {r.timing_read('gauss_mfma')[0] / 20:.4f}")
    r.tune("gauss_mfma_segments", 0)
    print(f"{w}x{h} sigma={sigma}  " + "  ".join(res))
