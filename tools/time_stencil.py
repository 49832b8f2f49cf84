#!/usr/bin/env python3
"""tools/time_stencil.py — box blur / median kernel times at 8K (HIP events via pfx_timing), for same-box A/B of libpfx builds."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
s = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); d = torch.empty_like(s)
out = []
for name, key, fn in (("box r=3", "box_blur", lambda: r.box_blur_dev(s.data_ptr(), d.data_ptr(), w, h, 3.0)),
                      ("box r=48", "box_blur", lambda: r.box_blur_dev(s.data_ptr(), d.data_ptr(), w, h, 48.0)),
                      ("median r=1", "median", lambda: r.median_dev(s.data_ptr(), d.data_ptr(), w, h, 1)),
                      ("median r=2", "median", lambda: r.median_dev(s.data_ptr(), d.data_ptr(), w, h, 2)),
                      ("median r=4", "median", lambda: r.median_dev(s.data_ptr(), d.data_ptr(), w, h, 4))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(10): fn()
    torch.cuda.synchronize(); r.timing_enable(False)
    out.append(f"{name} {r.timing_read(key)[0] / 10:.4f}")
print(os.environ.get("PFX_LIB_PATH", "default").split("/")[-1], " | ".join(out))
