#!/usr/bin/env python3
"""tools/time_brush.py — brush stroke kernel time (pfx_brush_line on a host image: the "brush_stamps" timer covers the kernel, not the
upload / download): a 4K preview layer, round brush, default spacing (0.01 x size: one stamp per pixel of travel at size 100)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0)
w, h = 3840, 2160
target = np.zeros((h, w, 4), np.uint8)
for size, length in ((10.0, 2000), (100.0, 2000), (400.0, 1500)):
    b = GpuRenderer.make_brush(size, 0.75, True, color=(0.2, 0.4, 0.9, 1.0))
    p0, p1 = (200.0, 300.0), (200.0 + length, 300.0 + length * 0.4)
    for _ in range(2): r.brush_line(target, b, p0, p1)
    r.timing_reset(); r.timing_enable(True)
    for _ in range(5): r.brush_line(target, b, p0, p1)
    r.timing_enable(False)
    ms, cnt = r.timing_read("brush_stamps")
    stamps = int(np.hypot(p1[0] - p0[0], p1[1] - p0[1]) / max(size * 0.01, 1.0)) + 1
    box = (abs(p1[0] - p0[0]) + size + 2) * (abs(p1[1] - p0[1]) + size + 2)
    print(f"size {size:g}: {ms / max(cnt, 1):.3f} ms per stroke (~{stamps} stamps, bounding box {box / 1e6:.2f} Mpx, {stamps / (ms / max(cnt, 1)) / 1e3:.0f} k stamps/ms)")
