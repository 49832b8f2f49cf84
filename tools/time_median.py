#!/usr/bin/env python3
"""tools/time_median.py — median filter timings at 8K (HIP events on the launch stream)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
import sys as _s
mins = [int(a) for a in _s.argv[1:]] or [3]
for bits_min in mins:
    r.tune("median_bits_min", bits_min)
    for rad in (1, 2, 3, 4, 5, 6, 7, 8):
        for _ in range(3): r.median_dev(src.data_ptr(), dst.data_ptr(), w, h, rad)
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(10): r.median_dev(src.data_ptr(), dst.data_ptr(), w, h, rad)
        torch.cuda.synchronize(); r.timing_enable(False)
        ms = r.timing_read("median")[0] / 10
        print(f"bits_min={bits_min} median r={rad}: {ms:.3f} ms  {8*w*h/ms/1e6:.0f} GB/s")
