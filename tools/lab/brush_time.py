#!/usr/bin/env python3
"""tools/lab/brush_time.py — the brush stamp loop on a device-resident 8K preview layer: ms per stroke (brush_stamps timer = the kernel; wall = the whole call incl. the
host prologue and the stamp upload), stamps and stamped pixels per second, for strokes of different shapes"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
tgt = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
def line(p0, p1):   # draw_line_no_dirty's dense 1-px stepping (brush_render.rs:762-835)
    n = int(max(abs(p1[0] - p0[0]), abs(p1[1] - p0[1]))) + 1
    t = np.linspace(0.0, 1.0, n, dtype=np.float32)
    return np.stack([p0[0] + (p1[0] - p0[0]) * t, p0[1] + (p1[1] - p0[1]) * t], axis=1)
rng = np.random.default_rng(3)
def scribble(n, cx, cy, ext):
    t = np.linspace(0, 40 * np.pi, n, dtype=np.float32)
    return np.stack([cx + ext * np.cos(t * 0.37) * np.sin(t * 0.11 + 1.0), cy + ext * np.sin(t * 0.53) * np.cos(t * 0.07)], axis=1).astype(np.float32)
cases = [
    ("mouse segment: 60 stamps, size 50", r.make_brush(50.0, 0.75, True, (0.8, 0.2, 0.1, 1.0)), line((3000, 2000), (3059, 2010))),
    ("mouse segment: 60 stamps, size 300", r.make_brush(300.0, 0.75, True, (0.8, 0.2, 0.1, 1.0)), line((3000, 2000), (3059, 2010))),
    ("long diagonal: 6501 stamps, size 100", r.make_brush(100.0, 0.75, True, (0.1, 0.2, 0.9, 1.0)), line((500, 500), (7000, 3800))),
    ("scribble: 4000 stamps, size 120, 2.4K x 2.4K area", r.make_brush(120.0, 0.5, True, (0.1, 0.7, 0.2, 0.8)), scribble(4000, 3800, 2100, 1200)),
    ("dodge: long diagonal, size 100", r.make_brush(100.0, 0.75, True, (1, 1, 1, 1.0), mode=1), line((500, 500), (7000, 3800))),
    ("eraser scribble: 4000 stamps, size 120", r.make_brush(120.0, 0.5, True, (0, 0, 0, 1.0), is_eraser=True), scribble(4000, 3800, 2100, 1200)),
]
for name, b, pts in cases:
    def run():
        r.brush_stamps_dev(tgt.data_ptr(), w, h, b, pts)
    for _ in range(2): run()
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5 * 1e3
    r.timing_enable(False)
    ms = r.timing_read("brush_stamps")[0] / 5
    rad = b.size / 2
    print(f"{name:52s} kernel {ms:8.4f} ms  call {wall:8.3f} ms  {len(pts) / ms / 1e3:8.1f} Mstamps/s  {len(pts) * np.pi * rad * rad / ms / 1e6:8.1f} G stamped px/s", flush=True)

# the reference's serial loop (oracle restatement, one thread — the reference stamps on the UI thread) on the diagonal stroke, 4K crop for time
if "--cpu" in sys.argv:
    from tests import oracle_lib as O
    b = O.make_brush(size=100.0, hardness=0.75, anti_aliased=True, color=(0.1, 0.2, 0.9, 1.0))
    pts = line((500, 500), (7000, 3800))
    t = np.zeros((h, w, 4), np.uint8)
    t0 = time.perf_counter(); O.brush_line(t, b, (500.0, 500.0), (7000.0, 3800.0)); dt = time.perf_counter() - t0
    print(f"CPU oracle (serial stamp loop), diagonal stroke of {len(pts)} stamps, size 100 at 8K: {dt * 1e3:.1f} ms")
