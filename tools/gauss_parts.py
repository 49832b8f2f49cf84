#!/usr/bin/env python3
"""tools/gauss_parts.py — the matrix-core Gaussian with 2 / 1 f16 pieces per weight and per horizontal result (pfx_tune "gauss_parts" = 22, 12, 11):
kernel time at 8K (HIP events, median) and, against the bit-exact VALU mode of the same library (which the GPU tests pin to the oracle), the
fraction of channels that differ and the largest difference — on uniform noise (the bench's worst case for rounding) and on a smooth image."""
import argparse, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--sigmas", default="4,16,24")
ap.add_argument("--modes", default="22,12,11")
a = ap.parse_args()
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
g = torch.Generator(device="cuda"); g.manual_seed(7)
noise = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda", generator=g)
yy, xx = torch.meshgrid(torch.arange(h, device="cuda"), torch.arange(w, device="cuda"), indexing="ij")
smooth = torch.stack([(xx * 255 // (w - 1)), (yy * 255 // (h - 1)), ((xx + yy) * 255 // (w + h - 2)), 255 - (xx * 255 // (w - 1))], dim=-1).to(torch.uint8).contiguous()
# a photograph-like field: smooth ramps + mild noise
photo = (smooth.to(torch.int16) + torch.randint(-12, 13, (h, w, 4), device="cuda", generator=g, dtype=torch.int16)).clamp(0, 255).to(torch.uint8).contiguous()
dst = torch.empty_like(noise); ref = torch.empty_like(noise)
for s in [float(x) for x in a.sigmas.split(",")]:
    refs = {}
    for name, img in (("noise", noise), ("smooth", smooth), ("photo", photo)):
        r.set_exact(True); r.gaussian_blur_dev(img.data_ptr(), ref.data_ptr(), w, h, s); r.set_exact(False)
        refs[name] = ref.clone()
    for mode in [int(m) for m in a.modes.split(",")]:
        r.tune("gauss_parts", mode)
        for _ in range(5): r.gaussian_blur_dev(noise.data_ptr(), dst.data_ptr(), w, h, s)
        ts = []
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r.gaussian_blur_dev(noise.data_ptr(), dst.data_ptr(), w, h, s); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        row = {"sigma": s, "parts": mode, "ms_median": round(ts[len(ts) // 2], 4), "ms_min": round(ts[0], 4)}
        for name, img in (("noise", noise), ("smooth", smooth), ("photo", photo)):
            r.gaussian_blur_dev(img.data_ptr(), dst.data_ptr(), w, h, s)
            d = (dst.to(torch.int16) - refs[name].to(torch.int16)).abs()
            row[name] = {"max_diff": int(d.max().item()), "frac_off": float((d != 0).float().mean().item())}
        print(json.dumps(row), flush=True)
r.tune("gauss_parts", 22)
