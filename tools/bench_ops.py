#!/usr/bin/env python3
"""tools/bench_ops.py — per-kernel timings of the rest of the path at BASELINE.json's config sizes (device-resident
data, HIP events on the launch stream via pfx_timing).  Not the headline bench (that is bench.py); this is the
evidence table in DESIGN.md §4 / profiles/rNN_ops.json.

    python tools/bench_ops.py [--out profiles/r01_ops.json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK = 8000.0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch

    from paintfe_amd import GpuRenderer
    from tests import inputs as I

    dev = torch.device("cuda", 0)
    r = GpuRenderer(0)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0001)
    rows = []

    def timed(name, timer_names, fn, px, alg_bytes_per_px, note=""):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        r.timing_reset()
        r.timing_enable(True)
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        r.timing_enable(False)
        ms = sum(r.timing_read(t)[0] for t in timer_names) / args.reps
        gbs = alg_bytes_per_px * px / (ms * 1e-3) / 1e9
        rows.append({"op": name, "ms": round(ms, 4), "Mpx_s": round(px / ms / 1e3, 1), "alg_bytes_px": alg_bytes_per_px,
                     "achieved_GBs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK, 4), "note": note})
        print(rows[-1], flush=True)

    # ---------------- 8K (config 2: Gaussian sigma=16 + HSL; plus the rest of the bank)
    w, h = 7680, 4320
    px = w * h
    src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev, generator=g)
    dst = torch.empty_like(src)
    tmp = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
    mask = (torch.rand((h, w), device=dev, generator=g) < 0.7).to(torch.uint8) * 255
    s, d, t, m = src.data_ptr(), dst.data_ptr(), tmp.data_ptr(), mask.data_ptr()

    timed("gaussian sigma=16 (fma)", ["gauss_h", "gauss_v"], lambda: r.gaussian_blur_dev(s, d, w, h, 16.0, t), px, 8, "776 MAC/px: VALU-bound")
    r.set_exact(True)
    timed("gaussian sigma=16 (exact, no FMA)", ["gauss_h", "gauss_v"], lambda: r.gaussian_blur_dev(s, d, w, h, 16.0, t), px, 8, "bit-exact mode")
    r.set_exact(False)
    timed("gaussian sigma=4", ["gauss_h", "gauss_v"], lambda: r.gaussian_blur_dev(s, d, w, h, 4.0, t), px, 8)
    timed("hsl(30,-20,10)", ["adjust"], lambda: r.adjust_dev(s, d, w, h, "hsl", [30.0, -20.0, 10.0]), px, 8)
    timed("hsl masked + FROM_FLAT", ["adjust"], lambda: r.adjust_dev(s, d, w, h, "hsl", [30.0, -20.0, 10.0], mask_ptr=m, sparse=1), px, 9)
    timed("invert", ["adjust"], lambda: r.adjust_dev(s, d, w, h, "invert"), px, 8)
    timed("brightness_contrast", ["adjust"], lambda: r.adjust_dev(s, d, w, h, "brightness_contrast", [30.0, 20.0]), px, 8)
    lut = np.tile(np.arange(255, -1, -1, dtype=np.uint8), (4, 1))
    timed("levels/curves LUT apply", ["adjust"], lambda: r.adjust_dev(s, d, w, h, "lut_rgba", lut=lut), px, 8)
    timed("vibrance", ["adjust"], lambda: r.adjust_dev(s, d, w, h, "vibrance", [50.0]), px, 8)
    timed("box blur r=3", ["box_blur"], lambda: r.box_blur_dev(s, d, w, h, 3.0), px, 8, "two passes, u8 intermediate (+8 B/px)")
    timed("box blur r=48", ["box_blur"], lambda: r.box_blur_dev(s, d, w, h, 48.0), px, 8)
    timed("median r=1", ["median"], lambda: r.median_dev(s, d, w, h, 1), px, 8)
    timed("median r=2", ["median"], lambda: r.median_dev(s, d, w, h, 2), px, 8)
    timed("median r=7", ["median"], lambda: r.median_dev(s, d, w, h, 7), px, 8, "225-element windows")
    del tmp, mask
    # ---------------- compositor, one blend mode at a time: 8 layers of the same mode over an opaque background (8K)
    from paintfe_amd import BLEND_MODES
    nl = 9
    layers = [src] + [torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev, generator=g) for _ in range(nl - 1)]
    layers[0][..., 3] = 255
    ptrs = [t_.data_ptr() for t_ in layers]
    for m, name in enumerate(BLEND_MODES):
        info = [(0, 1.0, True, 0)] + [(k, 0.8, True, m) for k in range(1, nl)]
        timed(f"flatten 9 layers, mode {m} {name}", ["flatten"], lambda: r.flatten_dev(ptrs, info, w, h, d), px, 4 * nl + 4,
              f"{(nl - 1)} blends/px")
    del layers, src, dst

    # ---------------- 16K (config 4: mesh warp 6x6 Catmull-Rom + liquify displacement)
    w, h = 15360, 8640
    px = w * h
    src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev, generator=g)
    dst = torch.empty_like(src)
    orig, deformed = I.jittered_mesh(6, 6, w, h)
    s, d = src.data_ptr(), dst.data_ptr()
    timed("mesh warp 6x6 fused (16K)", ["warp_mesh"], lambda: r.warp_mesh_catmull_rom_dev(s, orig, deformed, 6, 6, w, h, d), px, 8,
          "field never materialised")
    disp = torch.empty((h, w, 2), dtype=torch.float32, device=dev)
    # smooth liquify-like field: sum of a few Gaussian pushes, built with torch (input data, not the product path)
    yy = torch.arange(h, device=dev, dtype=torch.float32)[:, None]
    xx = torch.arange(w, device=dev, dtype=torch.float32)[None, :]
    fx = torch.zeros((h, w), device=dev)
    fy = torch.zeros((h, w), device=dev)
    rng = np.random.default_rng(0x5EED0004)
    for _ in range(16):
        cx, cy = rng.random() * w, rng.random() * h
        ddx, ddy = (rng.random() * 2 - 1) * 60, (rng.random() * 2 - 1) * 60
        wgt = torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * (400.0 / 3) ** 2)) * 0.8
        fx += ddx * wgt
        fy += ddy * wgt
    disp[..., 0] = fx
    disp[..., 1] = fy
    del fx, fy, yy, xx
    timed("liquify displacement warp (16K)", ["warp_displacement"], lambda: r.warp_displacement_dev(s, w, h, disp.data_ptr(), w, h, d), px, 16,
          "4 + 8 (field) + 4 B/px")

    if args.out:
        json.dump({"device": torch.cuda.get_device_name(0), "rows": rows}, open(args.out, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
