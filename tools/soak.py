#!/usr/bin/env python3
"""tools/soak.py [--seconds S] [--seed N] — randomized parity soak on the GPU against the oracle (test infrastructure, like tests/): random layer stacks
(sizes, depths, modes, opacities, alpha structure: noise, opaque / transparent runs and blocks, reset layers at random depths) through pfx_composite — which
picks the class-sorting, streaming or general compositor by itself —, random-sigma Gaussians in the default (<= 1 LSB) and exact (bit-exact) modes, random-radius box blurs and medians (bit-exact), random
displacement and mesh warps (bit-exact), random brush strokes of every stamp kind (bit-exact).  Prints one JSON line; exits 1 on the first mismatch with the case's seed."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer
from tests import inputs as I
from tests import oracle_lib as O

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
r = GpuRenderer(0)
t_end = time.time() + a.seconds
counts = {"stacks": 0, "gauss": 0, "warps": 0, "mesh": 0, "box": 0, "sharpen_glow": 0, "median": 0, "brush": 0, "chains": 0}
case = a.seed * 1000003


def alpha_plane(rng, w, h):
    kind = rng.integers(0, 6)
    if kind == 0: return rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    if kind == 1: return np.full((h, w), 255, np.uint8)
    if kind == 2:  # S2-like mixture
        sel = rng.integers(0, 4, size=(h, w)); v = rng.integers(1, 255, size=(h, w), dtype=np.uint8)
        return np.where(sel == 0, 0, np.where(sel == 1, 255, v)).astype(np.uint8)
    if kind == 3:  # opaque with transparent blocks
        p = np.full((h, w), 255, np.uint8)
        for _ in range(int(rng.integers(1, 6))):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w)); p[y:y + int(rng.integers(1, h + 1)), x:x + int(rng.integers(1, w + 1))] = int(rng.integers(0, 2)) * int(rng.integers(0, 256))
        return p
    if kind == 4:  # runs along rows (units of the compositor are 192 consecutive pixels)
        flat = np.repeat(rng.choice(np.array([0, 255, 128], np.uint8), size=(w * h) // 37 + 2), 37)[: w * h]
        return flat.reshape(h, w)
    return (rng.random((h, w)) < rng.random()).astype(np.uint8) * 255


while time.time() < t_end:
    case += 1
    rng = np.random.default_rng(case)
    what = rng.integers(0, 15)
    try:
        if what < 6:
            w, h = int(rng.integers(1, 700)), int(rng.integers(1, 120))
            if rng.random() < 0.3: w, h = int(rng.integers(180, 400)) * int(rng.integers(1, 4)), int(rng.integers(1, 40))
            n = int(rng.choice([1, 2, 5, 9, 15, 16, 17, 24, 32, 40]))
            stack = rng.integers(0, 256, size=(n, h, w, 4), dtype=np.uint8)
            modes = rng.integers(0, 25, size=n).astype(np.uint8)
            opac = np.where(rng.random(n) < 0.5, 1.0, rng.random(n) * 0.98 + 0.01).astype(np.float32)
            for k in range(n):
                stack[k, ..., 3] = alpha_plane(rng, w, h)
                if rng.random() < 0.25: modes[k] = 14 if rng.random() < 0.5 else 0      # reset candidates: Overwrite / Normal
                if rng.random() < 0.15: modes[k] = 13                                    # Xor lowers alpha
            for k in range(n): r.ensure_layer_texture(k, stack[k], generation=case)
            got = r.composite(w, h, [(k, float(opac[k]), True, int(modes[k])) for k in range(n)])
            ref = O.flatten_stack(stack, modes, opac)
            if not np.array_equal(got, ref): raise AssertionError(f"flatten {w}x{h}x{n}: {(got != ref).any(axis=-1).sum()} pixels differ")
            counts["stacks"] += 1
        elif what < 8:
            w, h = int(rng.integers(1, 500)), int(rng.integers(1, 300))
            sigma = float(rng.choice([0.3, 0.7, 1.5, 3.0, 5.3, 8.0, 12.0, 16.0, 21.0, 26.6, 31.0])) * float(0.9 + 0.2 * rng.random())
            img = I.random_rgba(w, h, case) if rng.random() < 0.6 else I.create_test_gradient(w, h)
            ref = O.gaussian_blur(img, sigma)
            d = np.abs(r.blur_rgba(img, sigma).astype(np.int16) - ref.astype(np.int16)).max()
            if d > 1: raise AssertionError(f"gaussian {w}x{h} sigma {sigma}: max diff {d}")
            r.set_exact(True)
            try:
                if not np.array_equal(r.blur_rgba(img, sigma), ref): raise AssertionError(f"exact gaussian {w}x{h} sigma {sigma}")
            finally:
                r.set_exact(False)
            counts["gauss"] += 1
        elif what == 11:   # sharpen / glow in the DEFAULT context mode (the library runs the bit-exact Gaussian inside them; small radii: Gaussian + combine in one kernel)
            w, h = int(rng.integers(1, 700)), int(rng.integers(1, 300))
            img = I.random_rgba(w, h, case) if rng.random() < 0.7 else I.create_test_gradient(w, h)
            mask = None if rng.random() < 0.6 else ((rng.random((h, w)) < 0.5).astype(np.uint8) * 255)
            radius = float(rng.choice([0.2, 0.5, 1.0, 1.7, 3.0, 4.9, 5.4, 8.0]))
            if rng.random() < 0.5:
                amount = float(rng.choice([-0.5, 0.3, 1.0, 2.5]))
                if not np.array_equal(r.sharpen_core(img, amount, radius, mask), O.sharpen(img, amount, radius, mask)): raise AssertionError(f"sharpen {w}x{h} amount {amount} radius {radius}")
            else:
                inten = float(rng.choice([0.2, 0.5, 1.0, 1.7]))
                if not np.array_equal(r.glow_core(img, radius, inten, mask), O.glow(img, radius, inten, mask)): raise AssertionError(f"glow {w}x{h} radius {radius} intensity {inten}")
            counts["sharpen_glow"] += 1
        elif what == 13:   # brush strokes: short (bounding-box launch) and long (dealt to 64 x 64 chunks), every stamp kind, off-canvas stamps, selections
            w, h = int(rng.integers(1, 500)), int(rng.integers(1, 300))
            n = int(rng.choice([1, 7, 40, 64, 65, 150, 500]))
            t = np.linspace(0, float(rng.uniform(2, 30)), n)
            pts = np.stack([w * (0.5 + 0.7 * np.cos(t * 0.7) * np.sin(t * 0.13 + 0.4)), h * (0.5 + 0.7 * np.sin(t * 0.9))], axis=1).astype(np.float32)
            if rng.random() < 0.5: pts += rng.uniform(-25, 25, size=pts.shape).astype(np.float32)
            kind = int(rng.integers(0, 5))
            brush = dict(size=float(rng.choice([0.8, 3.0, 9.5, 23.0, 60.0, 131.0])), hardness=float(rng.random()), anti_aliased=bool(rng.random() < 0.7),
                         color=(float(rng.random()), float(rng.random()), float(rng.random()), float(rng.uniform(0.05, 1.0))), flow=float(rng.uniform(0.05, 1.0)),
                         is_eraser=kind == 1, mode=[0, 0, 1, 2, 3][kind])
            target = I.random_rgba(w, h, case) if kind >= 2 else np.zeros((h, w, 4), np.uint8)
            if kind == 1: target[:, : w // 2, 3] = int(rng.integers(0, 256))
            sel = None if rng.random() < 0.6 else ((rng.random((h, w)) < 0.7).astype(np.uint8) * 255)
            dyn = None if rng.random() < 0.6 else dict(stamp_counter=int(rng.integers(0, 1 << 30)), scatter=float(rng.choice([0.0, 0.5])), hue_jitter=float(rng.choice([0.0, 0.6])),
                                                        brightness_jitter=float(rng.choice([0.0, 0.4])))
            got = r.brush_stamps(target, r.make_brush(**brush), pts, sel, dyn)
            ref = target.copy(); ob = O.make_brush(**brush)
            for (x, y) in pts: O.brush_stamp(ref, ob, float(x), float(y), sel, dyn)
            if not np.array_equal(got, ref): raise AssertionError(f"brush {w}x{h} n {n} kind {kind} size {brush['size']} aa {brush['anti_aliased']} sel {sel is not None} dyn {dyn}")
            counts["brush"] += 1
        elif what == 12:   # median: sorted-column networks (r <= 2), the bit-plane select on column pairs (3 .. 7) and single columns (8), the value search beyond; ties, masks
            w, h = int(rng.integers(1, 420)), int(rng.integers(1, 140))
            radius = int(rng.choice([1, 2, 3, 3, 4, 4, 5, 6, 7, 8, 9, 12]))
            img = I.random_rgba(w, h, case)
            if rng.random() < 0.5: img = (img // int(rng.choice([16, 64, 100]))) * int(rng.choice([16, 64, 100]))   # heavy ties
            if rng.random() < 0.3: img[:, : max(1, w // 3), int(rng.integers(0, 4))] = int(rng.choice([0, 255]))
            mask = None if rng.random() < 0.6 else ((rng.random((h, w)) < 0.5).astype(np.uint8) * 255)
            if not np.array_equal(r.median_core(img, radius, mask), O.median(img, radius, mask)): raise AssertionError(f"median {w}x{h} radius {radius} mask {mask is not None}")
            counts["median"] += 1
        elif what == 10:   # box blur: the fused tile (r <= 4), the fused strip walk (r <= 60) and the two-pass kernels, with and without a selection
            w, h = int(rng.integers(1, 900)), int(rng.integers(1, 400))
            radius = float(rng.choice([0.6, 1.0, 2.0, 4.0, 4.5, 5.0, 7.0, 9.0, 13.0, 24.0, 37.0, 48.0, 60.0, 61.0, 90.0])) - (0.3 if rng.random() < 0.3 else 0.0)
            img = I.random_rgba(w, h, case)
            mask = None if rng.random() < 0.6 else ((rng.random((h, w)) < 0.5).astype(np.uint8) * 255)
            if not np.array_equal(r.box_blur_core(img, radius, mask), O.box_blur(img, radius, mask)): raise AssertionError(f"box blur {w}x{h} radius {radius} mask {mask is not None}")
            counts["box"] += 1
        elif what == 14:   # chains (round 6): random runs of pointwise ops of both flavours, optionally behind a Gaussian / box blur, in the bit-exact Gaussian mode, as one
            # pfx_chain_dev call against the oracle op by op (fused stores of both Gaussians with light and — on request — heavy ops, long runs cut into launches, tables)
            w, h = int(rng.integers(1, 500)), int(rng.integers(1, 260))
            img = I.random_rgba(w, h, case)
            PW = [("adjust", "hsl", (float(rng.uniform(-180, 180)), float(rng.uniform(-100, 100)), float(rng.uniform(-50, 50)))), ("adjust", "invert"),
                  ("adjust", "brightness_contrast", (float(rng.uniform(-60, 60)), float(rng.uniform(-60, 60)))), ("adjust", "exposure", (float(rng.uniform(-2, 2)),)),
                  ("adjust", "vibrance", (float(rng.uniform(-100, 100)),)), ("adjust", "sepia"), ("adjust", "posterize", (float(rng.integers(2, 9)),)),
                  ("adjust", "threshold", (float(rng.uniform(0, 255)),)), ("adjust", "desaturate"), ("adjust", "invert_alpha"),
                  ("adjust", "lut_rgba", (), rng.integers(0, 256, 1024, dtype=np.uint8)), ("adjust", "gradient_map", (), rng.integers(0, 256, 1024, dtype=np.uint8)),
                  ("rhai", "invert"), ("rhai", "sepia_strength", (float(rng.random()),)), ("rhai", "brightness_contrast", (float(rng.uniform(-40, 40)), float(rng.uniform(-40, 40)))),
                  ("rhai", "hsl", (float(rng.uniform(-90, 90)), float(rng.uniform(-50, 50)), float(rng.uniform(-20, 20)))), ("rhai", "exposure", (float(rng.uniform(-1, 1)),)),
                  ("rhai", "levels", (float(rng.uniform(0, 60)), float(rng.uniform(180, 255)), float(rng.uniform(0.5, 2.0))))]
            ops = []
            if rng.random() < 0.6: ops.append(("gaussian", float(rng.choice([0.4, 1.0, 2.5, 4.0, 5.33, 7.0, 12.0, 16.0]))) if rng.random() < 0.75 else ("box", float(rng.choice([0.3, 1.0, 3.0, 9.0]))))
            for _ in range(int(rng.integers(0, 11))): ops.append(PW[int(rng.integers(0, len(PW)))])
            if rng.random() < 0.2 and ops: ops.insert(int(rng.integers(0, len(ops) + 1)), ("box", 2.0))
            ref = img
            for o in ops:
                ref = O.gaussian_blur(ref, o[1]) if o[0] == "gaussian" else O.box_blur(ref, o[1]) if o[0] == "box" else \
                    O.adjust(ref, o[1], o[2] if len(o) > 2 else (), lut=o[3] if len(o) > 3 else None) if o[0] == "adjust" else O.rhai_adjust(ref, o[1], o[2] if len(o) > 2 else ())
            da, db = r.dev_alloc(img.nbytes), r.dev_alloc(img.nbytes)
            r.set_exact(True); r.tune("chain_fuse_heavy", int(rng.random() < 0.5))
            try:
                r.dev_upload(da, img); r.chain_dev(da, db, w, h, ops); r.synchronize()
                got = r.dev_download(db, img.shape)
            finally:
                r.set_exact(False); r.tune("chain_fuse_heavy", 0); r.dev_free(da); r.dev_free(db)
            if not np.array_equal(got, ref): raise AssertionError(f"chain {w}x{h}: {[o[:2] for o in ops]}")
            counts["chains"] += 1
        elif what == 8:
            w, h = int(rng.integers(1, 600)), int(rng.integers(1, 200))
            sw, sh = (w, h) if rng.random() < 0.6 else (int(rng.integers(1, 600)), int(rng.integers(1, 200)))
            img = I.random_rgba(sw, sh, case)
            amp = float(rng.choice([0.5, 3.0, 20.0, 400.0]))
            disp = ((rng.random((h, w, 2)) - 0.5) * amp).astype(np.float32)
            if not np.array_equal(r.warp_displacement(img, disp), O.warp_displacement(img, disp)): raise AssertionError(f"warp {sw}x{sh} -> {w}x{h} amp {amp}")
            counts["warps"] += 1
        else:
            w, h = int(rng.integers(2, 500)), int(rng.integers(2, 200))
            cols, rows = int(rng.integers(1, 9)), int(rng.integers(1, 9))
            img = I.random_rgba(w, h, case)
            orig, deformed = I.jittered_mesh(cols, rows, w, h, seed=case)
            if rng.random() < 0.3: deformed = (deformed + (rng.random(deformed.shape) - 0.5) * np.array([w, h]) * 0.8).astype(np.float32)
            if not np.array_equal(r.warp_mesh_catmull_rom(img, orig, deformed, cols, rows), O.warp_mesh_catmull_rom(img, orig, deformed, cols, rows)):
                raise AssertionError(f"mesh {w}x{h} grid {cols}x{rows}")
            counts["mesh"] += 1
    except AssertionError as e:
        print(json.dumps({"ok": False, "case_seed": case, "error": str(e), **counts})); sys.exit(1)
print(json.dumps({"ok": True, "seconds": a.seconds, "first_case": a.seed * 1000003 + 1, "last_case": case, **counts}))
