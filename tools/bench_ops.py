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
    ap.add_argument("--only", default="", help="run only the rows whose name contains this text")
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
        if args.only and args.only not in name:
            return
        import time
        t_warm = time.perf_counter()  # clocks ramp over the first ~30 ms of load (bench.py's PREWARM standard)
        while True:
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            if time.perf_counter() - t_warm >= 0.04:
                break
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

    for _ in range(40):  # bring the GPU to its steady clocks before the first measured row
        r.gaussian_blur_dev(s, d, w, h, 16.0, t)
    torch.cuda.synchronize()
    timed("gaussian sigma=16 (matrix cores)", ["gauss_mfma"], lambda: r.gaussian_blur_dev(s, d, w, h, 16.0, t), px, 8, "fused H+V strip walk on v_mfma_f32_32x32x16_f16")
    r.set_exact(True)
    timed("gaussian sigma=16 (exact, no FMA)", ["gauss_h", "gauss_v"], lambda: r.gaussian_blur_dev(s, d, w, h, 16.0, t), px, 8, "bit-exact mode")
    timed("gaussian sigma=1 (exact: fused)", ["gauss_fused"], lambda: r.gaussian_blur_dev(s, d, w, h, 1.0, t), px, 8, "bit-exact mode, both passes in one kernel (radii 1 .. 16): what sharpen / glow / drop shadow run")
    timed("gaussian sigma=4 (exact: fused)", ["gauss_fused"], lambda: r.gaussian_blur_dev(s, d, w, h, 4.0, t), px, 8, "bit-exact mode, radius 12 (fused up to 16)")
    r.set_exact(False)
    timed("gaussian sigma=4", ["gauss_mfma"], lambda: r.gaussian_blur_dev(s, d, w, h, 4.0, t), px, 8)
    timed("hsl(30,-20,10)", ["adjust"], lambda: r.adjust_dev(s, d, w, h, "hsl", [30.0, -20.0, 10.0]), px, 8)
    timed("hsl masked + FROM_FLAT", ["adjust"], lambda: r.adjust_dev(s, d, w, h, "hsl", [30.0, -20.0, 10.0], mask_ptr=m, sparse=1), px, 9)
    timed("TiledImage round trip (from_rgba_image -> to_rgba_image)", ["tiled_roundtrip"], lambda: r.tiled_roundtrip_dev(s, d, w, h), px, 8,
          "A0: chunks of 64 x 64 whose alpha is all zero are dropped; one pass")
    # chains (round 6, pfx_chain_dev): several ops, one pass where the kernels allow; every row's result equals the single-op calls one after the other
    three = [("rhai", "exposure", (0.5,)), ("rhai", "sepia"), ("rhai", "invert")]
    timed("chain: exposure -> sepia -> invert (a script's three calls, ONE pass)", ["chain"], lambda: r.chain_dev(s, d, w, h, three), px, 8,
          "pfx_chain_dev; as three launches the same ops cost 3 x a streaming pass")

    def three_launches():
        r.chain_dev(s, d, w, h, three[:1]); r.chain_dev(d, d, w, h, three[1:2]); r.chain_dev(d, d, w, h, three[2:])
    timed("the same three ops as three launches", ["chain"], three_launches, px, 24, "8 B/px each")
    g_hsl = [("gaussian", 16.0), ("adjust", "hsl", (30.0, -20.0, 10.0))]
    timed("chain: gaussian sigma=16 -> hsl (config 2)", ["gauss_mfma_chain", "gauss_mfma", "chain"], lambda: r.chain_dev(s, d, w, h, g_hsl), px, 16,
          "pfx_chain_dev: the Gaussian, then HSL in place (HSL is heavy: not fused by default)")
    r.tune("chain_fuse_heavy", 1)
    timed("gaussian sigma=16 -> hsl with HSL in the matrix-core Gaussian's store (chain_fuse_heavy, not the default)", ["gauss_mfma_chain", "gauss_mfma", "chain"], lambda: r.chain_dev(s, d, w, h, g_hsl), px, 16,
          "the default runs this pair as two launches: HSL is 143 instructions per pixel and costs in the store what it costs as its own pass")
    r.tune("chain_fuse_heavy", 0)
    g_light = [("gaussian", 16.0), ("adjust", "brightness_contrast", (20.0, 10.0)), ("adjust", "invert")]
    timed("chain: gaussian sigma=16 -> brightness/contrast -> invert (ONE launch)", ["gauss_mfma_chain", "gauss_mfma", "chain"], lambda: r.chain_dev(s, d, w, h, g_light), px, 24,
          "light ops in the Gaussian's store; 24 B/px = the three-launch form's algorithmic bytes")
    r.tune("chain_mfma", 0)
    timed("the same, gaussian sigma=16 then brightness/contrast -> invert, as two launches", ["gauss_mfma_chain", "gauss_mfma", "chain"], lambda: r.chain_dev(s, d, w, h, g_light), px, 24)
    timed("the same, gaussian sigma=16 then hsl, as two launches", ["gauss_mfma_chain", "gauss_mfma", "chain"], lambda: r.chain_dev(s, d, w, h, g_hsl), px, 16)
    r.tune("chain_mfma", 1)
    g4_hsl = [("gaussian", 4.0), ("adjust", "hsl", (30.0, -20.0, 10.0))]
    r.set_exact(True)
    timed("chain: exact gaussian sigma=4 -> hsl (config 5's first two ops)", ["gauss_fused_chain", "gauss_fused", "chain"], lambda: r.chain_dev(s, d, w, h, g4_hsl), px, 16,
          "bit-exact fused Gaussian, then HSL in place")
    g4_light = [("gaussian", 4.0), ("adjust", "exposure", (0.5,)), ("adjust", "invert")]
    timed("chain: exact gaussian sigma=4 -> exposure -> invert (ONE launch)", ["gauss_fused_chain", "gauss_fused", "chain"], lambda: r.chain_dev(s, d, w, h, g4_light), px, 24,
          "light ops in the bit-exact fused Gaussian's store")
    r.set_exact(False)
    timed("invert", ["adjust"], lambda: r.adjust_dev(s, d, w, h, "invert"), px, 8)
    timed("brightness_contrast", ["adjust"], lambda: r.adjust_dev(s, d, w, h, "brightness_contrast", [30.0, 20.0]), px, 8)
    lut = np.tile(np.arange(255, -1, -1, dtype=np.uint8), (4, 1))
    timed("levels/curves LUT apply", ["adjust"], lambda: r.adjust_dev(s, d, w, h, "lut_rgba", lut=lut), px, 8)
    timed("vibrance", ["adjust"], lambda: r.adjust_dev(s, d, w, h, "vibrance", [50.0]), px, 8)
    timed("box blur r=3", ["box_blur"], lambda: r.box_blur_dev(s, d, w, h, 3.0), px, 8, "fused column-strip walk (radii 1 .. 60)")
    timed("box blur r=9", ["box_blur"], lambda: r.box_blur_dev(s, d, w, h, 9.0), px, 8, "fused column-strip walk, u8 intermediate in an LDS ring")
    timed("box blur r=48", ["box_blur"], lambda: r.box_blur_dev(s, d, w, h, 48.0), px, 8, "fused column-strip walk")
    timed("median r=1", ["median"], lambda: r.median_dev(s, d, w, h, 1), px, 8)
    timed("median r=2", ["median"], lambda: r.median_dev(s, d, w, h, 2), px, 8, "5x5 network, sorted columns shared across lanes (wave shifts)")
    timed("median r=3", ["median"], lambda: r.median_dev(s, d, w, h, 3), px, 8, "bit-plane radix select (k_median_bits.hip), incl. the planes pre-pass")
    timed("median r=4", ["median"], lambda: r.median_dev(s, d, w, h, 4), px, 8, "bit-plane radix select")
    timed("median r=5", ["median"], lambda: r.median_dev(s, d, w, h, 5), px, 8, "bit-plane radix select")
    timed("median r=7", ["median"], lambda: r.median_dev(s, d, w, h, 7), px, 8, "225-element windows, bit-plane radix select")
    # ---------------- the rest of the effect bank (k_effects2.hip), script VM, resamplers
    timed("pixelate block=8", ["pixelate"], lambda: r.pixelate_dev(s, d, w, h, 8), px, 8, "A6: nearest sample at the block centre (reads 1 / 64 of the pixels)")
    timed("vignette", ["vignette"], lambda: r.vignette_dev(s, d, w, h, 0.8, 0.5), px, 8)
    timed("add_noise gaussian mono", ["add_noise"], lambda: r.add_noise_dev(s, d, w, h, 30.0, "gaussian", True, 42, 1.0, 1), px, 8, "f64 ln + cos per pixel")
    timed("add_noise perlin 3 octaves", ["add_noise"], lambda: r.add_noise_dev(s, d, w, h, 50.0, "perlin", False, 42, 5.0, 3), px, 8)
    timed("reduce_noise r=2", ["reduce_noise"], lambda: r.reduce_noise_dev(s, d, w, h, 10.0, 2), px, 8, "25 exp per pixel (glibc expf algorithm in f64)")
    timed("halftone", ["halftone"], lambda: r.halftone_dev(s, d, w, h, 4.0, 45.0, "circle"), px, 8)
    timed("ink (sobel)", ["ink"], lambda: r.ink_dev(s, d, w, h, 1.0, 0.5), px, 8)
    timed("oil_painting r=3 levels=20", ["oil_painting"], lambda: r.oil_painting_dev(s, d, w, h, 3, 20), px, 8, "per-lane LDS histogram sliding down a column")
    timed("crystallize cell=16", ["crystallize"], lambda: r.crystallize_dev(s, d, w, h, 16.0, 42), px, 8, "wave-scanned cell sums -> LDS table -> u64 atomics, + assign")
    timed("bulge", ["bulge"], lambda: r.bulge_dev(s, d, w, h, 0.5), px, 8)
    timed("twist 45", ["twist"], lambda: r.twist_dev(s, d, w, h, 45.0), px, 8, "f64 sin + cos per pixel")
    timed("zoom_blur 16 samples", ["zoom_blur"], lambda: r.zoom_blur_dev(s, d, w, h, 0.5, 0.5, 0.3, 16), px, 8)
    timed("outline width=2", ["outline"], lambda: r.outline_dev(s, d, w, h, 2, (0, 0, 255, 255)), px, 8, "7x7 window search on a bit plane of alpha != 0")
    timed("sharpen amount=1 radius=1", ["sharpen", "gauss_h", "gauss_v"], lambda: r.sharpen_dev(s, d, w, h, 1.0, 1.0), px, 8, "bit-exact Gaussian + combine in one kernel (k_gauss_exact.hip)")
    timed("glow radius=3 intensity=0.5", ["glow", "gauss_h", "gauss_v"], lambda: r.glow_dev(s, d, w, h, 3.0, 0.5), px, 8, "bit-exact Gaussian + combine in one kernel")
    timed("drop shadow blur=3", ["shadow_alpha", "gauss_plane", "gauss_mfma", "gauss_fused", "gauss_h", "gauss_v", "shadow_composite"], lambda: r.shadow_dev(s, d, w, h, 5, 5, 3.0, False, (0, 0, 0, 255), 0.8), px, 8)
    half = torch.empty((h // 2, w // 2, 4), dtype=torch.uint8, device=dev)
    timed("resize 8K -> 4K bilinear", ["resize"], lambda: r.resize_image_dev(s, w, h, half.data_ptr(), w // 2, h // 2, "bilinear"), px, 5, "4 B read + 1 B/px (quarter-size) written")
    timed("resize 8K -> 4K lanczos3", ["resize"], lambda: r.resize_image_dev(s, w, h, half.data_ptr(), w // 2, h // 2, "lanczos3"), px, 5)
    del half
    src_h = src.cpu().numpy()
    import time
    r.execute_script_sync("map_channels(|r, g, b, a| [255 - r, g / 2, (b * 3 + a) / 4, a]);", src_h)   # warm: a process's first launch of k_script's code object loads it (~1 ms)
    r.timing_reset()
    r.timing_enable(True)
    t0 = time.perf_counter()
    r.execute_script_sync("map_channels(|r, g, b, a| [255 - r, g / 2, (b * 3 + a) / 4, a]);", src_h)
    wall = (time.perf_counter() - t0) * 1e3
    r.timing_enable(False)
    vm_ms = r.timing_read("script_vm")[0]
    rows.append({"op": "script: map_channels closure at 8K (bytecode VM kernel)", "ms": round(vm_ms, 4), "Mpx_s": round(px / vm_ms / 1e3, 1), "alg_bytes_px": 8,
                 "achieved_GBs": round(8 * px / (vm_ms * 1e-3) / 1e9, 1), "hbm_frac": round(8 * px / (vm_ms * 1e-3) / 1e9 / HBM_PEAK, 4),
                 "note": f"pfx_script_execute wall clock incl. H2D/D2H of 133 MB each way and closure compile: {wall:.0f} ms"})
    print(rows[-1], flush=True)
    del src_h
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

    # ---------------- 4K per-image pipeline of config 5 / S4: Gaussian sigma=4 -> HSL -> 4-layer flatten (image + 3 overlays)
    w, h = 3840, 2160
    px = w * h
    img = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev, generator=g)
    overlays = [torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev, generator=g) for _ in range(3)]
    a = torch.empty_like(img)
    b = torch.empty_like(img)
    tmp4 = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
    info4 = [(0, 1.0, True, 0), (1, 0.8, True, 1), (2, 0.8, True, 2), (3, 0.8, True, 8)]  # Normal, Multiply, Screen, Overlay

    def pipeline():
        r.gaussian_blur_dev(img.data_ptr(), a.data_ptr(), w, h, 4.0, tmp4.data_ptr())
        r.adjust_dev(a.data_ptr(), b.data_ptr(), w, h, "hsl", [30.0, -20.0, 10.0])
        r.flatten_dev([b.data_ptr()] + [o.data_ptr() for o in overlays], info4, w, h, a.data_ptr())
    timed("config-5 per-image pipeline at 4K (blur s=4 + HSL + 4-layer flatten)", ["gauss_mfma", "adjust", "flatten"], pipeline, px, 36,
          "8 + 8 + 20 algorithmic B/px; images/s = 1000 / ms")
    for name in ("gauss_mfma", "adjust", "flatten"):
        print("   ", name, round(r.timing_read(name)[0] / args.reps, 4), "ms", flush=True)
    del img, overlays, a, b, tmp4

    # ---------------- brush stamp loop on a device-resident 8K preview layer (A13): px = stamped pixel visits (stamps x pi r^2)
    w, h = 7680, 4320
    tgt = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)

    def line(p0, p1):   # draw_line_no_dirty's dense 1-px stepping (brush_render.rs:762-835)
        n = int(max(abs(p1[0] - p0[0]), abs(p1[1] - p0[1]))) + 1
        t = np.linspace(0.0, 1.0, n, dtype=np.float32)
        return np.stack([p0[0] + (p1[0] - p0[0]) * t, p0[1] + (p1[1] - p0[1]) * t], axis=1)
    tt = np.linspace(0, 40 * np.pi, 4000, dtype=np.float32)
    scribble = np.stack([3800 + 1200 * np.cos(tt * 0.37) * np.sin(tt * 0.11 + 1.0), 2100 + 1200 * np.sin(tt * 0.53) * np.cos(tt * 0.07)], axis=1).astype(np.float32)
    for name, b, pts in (("brush: mouse segment, 60 stamps, size 50", r.make_brush(50.0, 0.75, True, (0.8, 0.2, 0.1, 1.0)), line((3000, 2000), (3059, 2010))),
                         ("brush: diagonal stroke, 6501 stamps, size 100", r.make_brush(100.0, 0.75, True, (0.1, 0.2, 0.9, 1.0)), line((500, 500), (7000, 3800))),
                         ("brush: scribble, 4000 stamps, size 120", r.make_brush(120.0, 0.5, True, (0.1, 0.7, 0.2, 0.8)), scribble),
                         ("brush: dodge, diagonal stroke, size 100", r.make_brush(100.0, 0.75, True, (1, 1, 1, 1.0), mode=1), line((500, 500), (7000, 3800)))):
        timed(name, ["brush_stamps"], lambda: r.brush_stamps_dev(tgt.data_ptr(), w, h, b, pts), int(len(pts) * np.pi * (b.size / 2) ** 2), 0,
              "kernel only (the call adds the host prologue and the stamp upload, ~0.03-0.06 ms); Mpx_s = stamped pixel visits (a pixel stays in its lane's register across the stamps: no HBM figure); strokes of > 64 stamps are dealt to 64 x 64 chunks")
    del tgt

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
    timed("mesh displacement field 6x6 (16K, B3: generate_displacement_from_mesh)", ["mesh_displacement"],
          lambda: r.mesh_displacement_dev(orig, deformed, 6, 6, w, h, disp.data_ptr()), px, 8, "writes the 8 B/px f32 field the fused warp never materialises")
    disp[..., 0] = fx
    disp[..., 1] = fy
    del fx, fy, yy, xx
    rngd = np.random.default_rng(7)
    dabs = [(int(m), float(rngd.uniform(2000, w - 2000)), float(rngd.uniform(1500, h - 1500)), float(rngd.uniform(-20, 20)), float(rngd.uniform(-20, 20)), 400.0, 0.6)
            for m in (0, 1, 2, 3, 4, 0, 0, 0) * 4]
    fld = torch.zeros((h, w, 2), dtype=torch.float32, device=dev)
    timed("liquify dabs into the field: 32 dabs, radius 400 (16K)", ["displacement_brush"], lambda: r.displacement_brushes_dev(fld.data_ptr(), w, h, dabs),
          int(32 * np.pi * 400 * 400), 0, "DisplacementField::apply_* on the device-resident field (N4): Mpx_s = dab pixel visits")
    del fld
    timed("liquify displacement warp (16K)", ["warp_displacement"], lambda: r.warp_displacement_dev(s, w, h, disp.data_ptr(), w, h, d), px, 16,
          "4 + 8 (field) + 4 B/px")

    if args.out:
        json.dump({"device": torch.cuda.get_device_name(0), "rows": rows}, open(args.out, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
