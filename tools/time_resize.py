#!/usr/bin/env python3
"""tools/time_resize.py — pfx_resize_image_dev 8K -> 4K and 4K -> 8K, fused kernel vs the two-pass path (pfx_tune resize_two_pass)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
big = torch.randint(0, 256, (4320, 7680, 4), dtype=torch.uint8, device="cuda")
small = torch.randint(0, 256, (2160, 3840, 4), dtype=torch.uint8, device="cuda")
out_s, out_b = torch.empty_like(small), torch.empty_like(big)
for two in (0, 1, 0, 1):
    r.tune("resize_two_pass", two)
    row = []
    for name, src, (w, h), dst, (nw, nh) in (("8K->4K", big, (7680, 4320), out_s, (3840, 2160)), ("4K->8K", small, (3840, 2160), out_b, (7680, 4320))):
        for filt in ("bilinear", "lanczos3"):
            for _ in range(3): r.resize_image_dev(src.data_ptr(), w, h, dst.data_ptr(), nw, nh, filt)
            torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
            for _ in range(10): r.resize_image_dev(src.data_ptr(), w, h, dst.data_ptr(), nw, nh, filt)
            torch.cuda.synchronize(); r.timing_enable(False)
            row.append(f"{name} {filt} {r.timing_read('resize')[0] / 10:.3f} ms")
    print("two-pass" if two else "fused   ", " | ".join(row))
