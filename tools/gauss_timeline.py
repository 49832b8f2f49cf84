#!/usr/bin/env python3
"""tools/gauss_timeline.py — s_memtime stamps of gauss_strip_kernel's phases (development): block 0, producer wave 0 and consumer
wave 4, iterations 10..13.  Slots: 0 iteration start, 1 after conversion / before MFMAs, 2 after the refill was issued, 3 after the
MFMAs (first use of the accumulators), 4 before the barrier, 5 after the barrier."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer, _lib
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
lib = _lib.load()
w, h = 7680, 4320
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda")
dst = torch.empty_like(src)
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
lib.pfxk_gauss_set_dbg_buf(C.c_void_p(buf.data_ptr()))
for _ in range(5): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, 16.0)
r.tune("gauss_v_cfg", 16 << 9)
r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, 16.0)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(2, 4, 8)
for role, name in ((0, "producer w0"), (1, "consumer w4")):
    for it in range(4):
        row = t[role, it]
        base = t[0, 0, 0]
        print(name, "it", 10 + it, " ".join(f"{int(v - base):6d}" if v else "     -" for v in row[:8]))

# iteration period without the per-iteration flush (dbg bit 32: stamps at the start of iterations 10 and 40 only)
buf.zero_()
r.tune("gauss_v_cfg", 32 << 9)
r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, 16.0)
torch.cuda.synchronize()
t = buf.cpu().numpy()
for role, name in ((0, "producer w0"), (1, "consumer w4")):
    print(name, "mean iteration period over 30 iterations:", int(t[role * 32 + 1] - t[role * 32]) // 30, "ticks")
