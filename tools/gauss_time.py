#!/usr/bin/env python3
"""tools/gauss_time.py — Gaussian blur kernel time at 8K for a few sigmas (HIP events on the launch stream, median of --reps) and the
byte sum of each result, so that two builds of libpfx (PFX_LIB_PATH) can be compared on ONE box: tools/ab_gauss_libs.sh."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--sigmas", default="2,4,8,16")
a = ap.parse_args()
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
g = torch.Generator(device="cuda"); g.manual_seed(7)
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda", generator=g)
dst = torch.empty_like(src)
out = []
for s in [float(x) for x in a.sigmas.split(",")]:
    for _ in range(5): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, s)
    ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, s); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    out.append(f"s={s:g}: {ts[len(ts)//2]:.4f} ms sum={int(dst.to(torch.int64).sum().item())}")
print(os.environ.get("PFX_LIB_PATH", "default"), " | ".join(out))
