#!/usr/bin/env python3
"""tools/lab/sharpen_time.py — sharpen / glow kernel time at 8K (timers of the host-buffer entry points; the copies are not in them)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); r.tune(k, int(v))
w, h = 7680, 4320
img = np.random.default_rng(1).integers(0, 256, (h, w, 4), dtype=np.uint8)
for name, fn, timers in (("sharpen a=1 r=1", lambda: r.sharpen_core(img, 1.0, 1.0), ("sharpen", "gauss_fused", "gauss_h", "gauss_v", "gauss_mfma")),
                         ("sharpen a=1.5 r=3", lambda: r.sharpen_core(img, 1.5, 3.0), ("sharpen", "gauss_fused", "gauss_h", "gauss_v", "gauss_mfma")),
                         ("glow r=3 i=0.5", lambda: r.glow_core(img, 3.0, 0.5), ("glow", "gauss_fused", "gauss_h", "gauss_v", "gauss_mfma"))):
    fn(); r.timing_reset(); r.timing_enable(True)
    for _ in range(3): fn()
    r.timing_enable(False)
    t = {k: r.timing_read(k)[0] / 3 for k in timers}
    print(name, "  ".join(f"{k} {v:.3f}" for k, v in t.items() if v > 0), f"= {sum(t.values()):.3f} ms")
