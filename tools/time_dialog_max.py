#!/usr/bin/env python3
"""tools/time_dialog_max.py — effects at the upper ends of the reference dialogs' sliders (src/ui/dialogs/effects/*.rs), 4K: do any of them fall
off a cliff?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 3840, 2160
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
s, d = src.data_ptr(), dst.data_ptr()
cases = [
    ("median r=8", "median", lambda: r.median_dev(s, d, w, h, 8)),
    ("reduce_noise r=8", "reduce_noise", lambda: r.reduce_noise_dev(s, d, w, h, 50.0, 8)),
    ("oil r=10 levels=64", "oil_painting", lambda: r.oil_painting_dev(s, d, w, h, 10, 64)),
    ("outline width=8", "outline", lambda: r.outline_dev(s, d, w, h, 8, (0, 0, 255, 255))),
    ("box blur r=50", "box_blur", lambda: r.box_blur_dev(s, d, w, h, 50.0)),
    ("zoom blur 0.3 / 64 samples", "zoom_blur", lambda: r.zoom_blur_dev(s, d, w, h, 0.5, 0.5, 1.0, 64)),
    ("crystallize cell=2", "crystallize", lambda: r.crystallize_dev(s, d, w, h, 2.0, 42)),
    ("crystallize cell=100", "crystallize", lambda: r.crystallize_dev(s, d, w, h, 100.0, 42)),
    ("perlin 8 octaves", "add_noise", lambda: r.add_noise_dev(s, d, w, h, 50.0, "perlin", False, 42, 5.0, 8)),
]
for name in ("bokeh_blur_dev", "motion_blur_dev"):
    if hasattr(r, name):
        if name == "bokeh_blur_dev":
            cases.append(("bokeh r=10", "bokeh_blur", lambda: r.bokeh_blur_dev(s, d, w, h, 10.0)))
            cases.append(("bokeh r=40", "bokeh_blur", lambda: r.bokeh_blur_dev(s, d, w, h, 40.0)))
        else:
            cases.append(("motion blur 45 deg / 100", "motion_blur", lambda: r.motion_blur_dev(s, d, w, h, 45.0, 100.0)))
for label, timer, fn in cases:
    try:
        fn(); torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(3): fn()
        torch.cuda.synchronize(); r.timing_enable(False)
        print(f"{label}: {r.timing_read(timer)[0] / 3:.3f} ms at 4K")
    except Exception as e:
        print(f"{label}: {e}")
