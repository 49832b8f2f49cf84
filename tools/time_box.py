#!/usr/bin/env python3
"""tools/time_box.py — box blur timings at 8K per radius (HIP events on the launch stream); under rocprofv3 the H / V kernels separate."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
two = int(sys.argv[1]) if len(sys.argv) > 1 else 0
r.tune("box_two_pass", two)
for kv in sys.argv[2:]:
    k, v = kv.split("="); r.tune(k, int(v))
for rad in (1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 8.0, 9.0, 16.0, 24.0, 48.0, 100.0):
    for _ in range(3): r.box_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, rad)
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(10): r.box_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, rad)
    torch.cuda.synchronize(); r.timing_enable(False)
    ms = r.timing_read("box_blur")[0] / 10
    print(f"box two_pass={two} r={rad}: {ms:.3f} ms  {8*w*h/ms/1e6:.0f} GB/s")
