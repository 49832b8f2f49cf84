#!/usr/bin/env python3
"""tools/gauss_dbg.py — gauss_strip_kernel at 8K with parts switched off (development; results are wrong by design):
dbg bit 1 no output stores, 2 no horizontal pass, 4 no vertical pass, 8 no source refills."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda")
dst = torch.empty_like(src)
sig = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
for dbg in (0, 1, 2, 4, 8, 6, 9, 14, 15):
    r.tune("gauss_v_cfg", dbg << 9)
    for _ in range(3): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, sig)
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, sig); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"sigma={sig:g} dbg={dbg:2d}: {ts[len(ts)//2]:.4f} ms")
