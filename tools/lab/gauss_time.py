#!/usr/bin/env python3
"""tools/lab/gauss_time.py [sigmas…] — the matrix-core Gaussian at 8K with the library PFX_LIB_PATH names: ms per call (gauss_mfma timer), three rounds"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
g = torch.Generator(device="cuda"); g.manual_seed(5)
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda", generator=g); dst = torch.empty_like(src)
tmp = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
S, D, T = src.data_ptr(), dst.data_ptr(), tmp.data_ptr()
def t(sig):
    for _ in range(3): r.gaussian_blur_dev(S, D, w, h, sig, T)
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(10): r.gaussian_blur_dev(S, D, w, h, sig, T)
    torch.cuda.synchronize(); r.timing_enable(False)
    return r.timing_read("gauss_mfma")[0] / 10
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2: r.gaussian_blur_dev(S, D, w, h, 4.0, T)
sig = [float(a) for a in sys.argv[1:]] or [4.0, 10.0, 16.0, 24.0]
out = []
for s in sig:
    ms = min(t(s), t(s), t(s))
    chk = int(dst.view(torch.int32).sum(dtype=torch.int64).item()) & 0xffffffff
    out.append(f"sigma={s:g}:{ms:.4f}[{chk:08x}]")
print(os.path.basename(os.environ.get("PFX_LIB_PATH", "libpfx.so")), "  ".join(out), flush=True)
