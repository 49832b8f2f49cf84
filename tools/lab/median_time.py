#!/usr/bin/env python3
"""tools/lab/median_time.py [radii…] — median at 8K with the library PFX_LIB_PATH names: ms per call (the `median` timer: planes pre-pass + select),
the column-pair kernel against the single-column one (pfx_tune "median_pair"), alternating"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
g = torch.Generator(device="cuda"); g.manual_seed(5)
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda", generator=g); dst = torch.empty_like(src)
S, D = src.data_ptr(), dst.data_ptr()
def t(rad):
    for _ in range(3): r.median_dev(S, D, w, h, rad)
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(8): r.median_dev(S, D, w, h, rad)
    torch.cuda.synchronize(); r.timing_enable(False)
    return r.timing_read("median")[0] / 8
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1: r.median_dev(S, D, w, h, 2)
radii = [int(a) for a in sys.argv[1:]] or [3, 4, 5, 7]
for pair in (1, 0, 1, 0):   # column-pair kernel / single-column kernel, alternating on the same box
    r.tune("median_pair", pair)
    out = []
    for rad in radii:
        ms = min(t(rad), t(rad), t(rad))
        out.append(f"r={rad}:{ms:.4f}")
        chk = int(dst.view(torch.int32).sum(dtype=torch.int64).item()) & 0xffffffff
        out[-1] += f"[{chk:08x}]"
    print(os.path.basename(os.environ.get("PFX_LIB_PATH", "libpfx.so")), f"pair={pair}", "  ".join(out), flush=True)
