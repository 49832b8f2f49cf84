#!/usr/bin/env python3
"""bench.py — headline benchmark: Mpixels/sec on "8K x 32-layer flatten + Gaussian sigma=16" (BASELINE.json).

One "step" = one pass of the hot path over one synthetic 8K document that is already resident in HBM:
    flatten 32 RGBA8 layers (all 25 blend modes, S2 of SURVEY.md §8d)  ->  Gaussian blur sigma=16 of the result.
Both results are materialised (140 algorithmic bytes per output pixel).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU (one process per GPU, torch.distributed over RCCL): the headline line is ONE 8K document cut into bands of whole
chunk rows: every rank flattens its band, receives ceil(3 sigma) rows of the flattened neighbours' bands (RCCL
send/recv over xGMI — the one exchange the path needs) and blurs; the result stays sharded in bands, like the TiledImage chunks it
is made of — "scaling": "strong", value = document pixels / max-over-ranks wall time.  The same run also times (a) that pipeline
followed by an all-gather of the blurred bands into EVERY rank ("band_gathered_result": (N - 1) / N of the frame into each GPU per
step, bounded by the xGMI links, not by the kernels) and (b) the collective-free mode (one independent document per GPU, the
reference's CLI file loop, src/cli.rs:159: "doc_mode").  --shard doc makes (b) the headline instead.

PyTorch is plumbing only (device buffers, the synthetic generator, torch.distributed).  The product path is
libpfx.so through its C ABI; the CPU oracle is used here only (a) to check one crop of the GPU result and
(b) as the timed `cpu_baseline` on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
PATTERN_CEILING_GBS = 5900.0   # measured: what 33 lock-step layer streams sustain without arithmetic (profiles/r05_nstream_read.txt)
W8K, H8K, NLAYERS, SIGMA = 7680, 4320, 32, 16.0


def synth_stack(torch, device, w, h, n, seed):
    """S2 generator on the device: uniform RGB; alpha 25% = 0, 25% = 255, 50% uniform 1..254; layer 0 opaque;
    mode k -> k mod 25; opacity 1.0 for even k, 0.25 + 0.75*u for odd k."""
    stack = torch.empty((n, h, w, 4), dtype=torch.uint8, device=device)
    for k in range(n):
        stack[k] = synth_layer(torch, device, w, h, k, seed)
    modes, opac = synth_params(n, seed)
    return stack, modes, opac


def synth_layer(torch, device, w, h, k, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed + k)
    px = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=device, generator=g)
    sel = torch.randint(0, 4, (h, w), dtype=torch.uint8, device=device, generator=g)
    a = torch.randint(1, 255, (h, w), dtype=torch.uint8, device=device, generator=g)
    a = torch.where(sel == 0, torch.zeros_like(a), torch.where(sel == 1, torch.full_like(a, 255), a))
    px[..., 3] = 255 if k == 0 else a
    return px


def synth_params(n, seed):
    modes = np.zeros(n, np.uint8)
    opac = np.ones(n, np.float32)
    rng = np.random.default_rng(seed)
    for k in range(n):
        modes[k] = k % 25
        if k % 2 == 1:
            opac[k] = np.float32(0.25) + np.float32(0.75) * np.float32(rng.random())
    return modes, opac


def pmc_profile(w: int, h: int, n: int):
    """Per-launch PMC figures of the timed kernels from the committed rocprofv3 passes (profiles/rNN_pmc.json, written by
    tools/prof.sh + tools/prof_summary.py: FETCH_SIZE and WRITE_SIZE in separate --pmc runs, FETCH_SIZE doubled as the gfx950 guide
    prescribes; SQ_INSTS_VALU; GRBM_GUI_ACTIVE for the clock).  PMC collection cannot run inside this process, so the figures are
    only reported for the exact configuration they were measured on (8K x 32 layers); otherwise null."""
    if (w, h, n) != (W8K, H8K, NLAYERS):
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        d["_file"] = os.path.relpath(files[-1], ROOT)
        return d
    except Exception:
        return None


def clock_power_sample(torch, step_fn, dev_index, seconds=0.4):
    """Sustained core clock and socket power WHILE the timed workload runs: the compositor runs at the board's power limit, so the
    clock it sustains is part of the bound statement.  `step_fn` is repeated back to back for `seconds` after the timed region while a thread
    reads amdsmi's gpu_metrics (current_gfxclk, current_socket_power; tools/lab/clock_probe.py shows the raw series); the first 40 % of the window
    (clock still settling) is dropped.  Returns None where amdsmi is not importable or reports nothing."""
    try:
        import threading
        import amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        h0 = hs[dev_index] if dev_index < len(hs) else hs[0]
        cap = None
        try:
            cap = amdsmi.amdsmi_get_power_cap_info(h0).get("power_cap")
            cap = round(cap / 1e6, 1) if isinstance(cap, (int, float)) and cap > 1e5 else cap
        except Exception:
            pass
        try:
            clk_max = amdsmi.amdsmi_get_clock_info(h0, amdsmi.AmdSmiClkType.GFX).get("max_clk")
        except Exception:
            clk_max = None
        samples, stop = [], [False]

        def sampler():
            while not stop[0]:
                try:
                    m = amdsmi.amdsmi_get_gpu_metrics_info(h0)
                    samples.append((time.perf_counter(), m.get("current_gfxclk"), m.get("current_socket_power")))
                except Exception:
                    pass
                time.sleep(0.002)

        th = threading.Thread(target=sampler, daemon=True)
        th.start()
        t0 = time.perf_counter()
        n_steps = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                step_fn()
            torch.cuda.synchronize()
            n_steps += 20
        t1 = time.perf_counter()
        stop[0] = True
        th.join()
        use = [(c, p) for (t, c, p) in samples if t0 + 0.4 * (t1 - t0) <= t <= t1 and isinstance(c, (int, float)) and isinstance(p, (int, float))]
        if not use:
            return None
        return {"clock_ghz_sustained": round(sum(c for c, _ in use) / len(use) / 1e3, 3), "clock_ghz_min": round(min(c for c, _ in use) / 1e3, 3),
                "clock_ghz_max_spec": round(clk_max / 1e3, 3) if isinstance(clk_max, (int, float)) else None,
                "socket_power_w": round(sum(p for _, p in use) / len(use), 1), "power_cap_w": cap, "samples": len(use),
                "how": f"amdsmi gpu_metrics every 2 ms over {n_steps} back-to-back steps after the timed region ({(t1 - t0) / n_steps * 1e3:.4f} ms/step there)"}
    except Exception:
        return None


def nan_to_none(x):
    """a run whose band pipeline failed carries NaN times: the line must stay valid JSON"""
    if isinstance(x, float) and x != x:
        return None
    if isinstance(x, dict):
        return {k: nan_to_none(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [nan_to_none(v) for v in x]
    return x


def usable_cores() -> int:
    """threads the CPU baseline may really use: scheduler affinity, capped by the cgroup CPU quota if there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // period))
            break
        except Exception:
            continue
    return n


PCIE_GEN5_X16_GBS = 63.0  # guides/MI355X_MICROARCH.md: host link, spec, per direction
BATCH_MAX_LSB, BATCH_FRAC = 8, 4e-3  # parity gate of the S4 pipeline in the default Gaussian mode (derivation: tests/test_gpu_batch.py)
PREWARM = 25  # untimed steps in front of the --warmup steps: the clock needs ~30 ms of load to settle (20 steps are 33 ms)


def s4_inputs(w, h, n_pool=4):
    """S4 generator: a pool of uniform-random images and three overlays with S2's alpha distribution"""
    rng = np.random.default_rng(0x5EED0004)
    pool = [rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8) for _ in range(n_pool)]
    overlays = []
    for k in range(3):
        o = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        sel = rng.integers(0, 4, size=(h, w))
        o[..., 3] = np.where(sel == 0, 0, np.where(sel == 1, 255, o[..., 3]))
        overlays.append(o)
    return pool, overlays, [1, 2, 8]


def s4_check(O, src, overlays, modes, got, exact):
    """oracle pipeline on one image: (ok, {max_abs_diff, channels_differing})"""
    blur = O.gaussian_blur(src, 4.0)
    hsl = O.adjust(blur, "hsl", (30.0, -20.0, 10.0))
    ref = O.flatten_stack(np.stack([hsl] + overlays), np.array([0] + modes, np.uint8), np.ones(4, np.float32))
    d = np.abs(ref.astype(np.int16) - got.astype(np.int16))
    dmax, frac = int(d.max()), float((d > 0).mean())
    ok = (dmax == 0) if exact else (dmax <= BATCH_MAX_LSB and frac <= BATCH_FRAC)
    return ok, {"max_abs_diff": dmax, "channels_differing": round(frac, 7), "gate": "bit-exact" if exact else f"max <= {BATCH_MAX_LSB}, fraction <= {BATCH_FRAC}"}


def other_configs(args, torch, r, device, O, flat, blurred):
    """The other BASELINE.json configurations on the same box, after the headline: each with the mean HIP-event kernel time, its
    algorithmic HBM bytes, the fraction of the 8 TB/s roofline and an oracle check.  Returns (dict, failed_checks)."""
    from tests import inputs as I
    w, h = W8K, H8K
    px = w * h
    out, failed = {}, []

    def kernel_ms(names, fn, iters, warm=5):
        # the same standard as the headline's PREWARM: the clock needs ~30 ms of load to settle after the idle gap of the previous configuration's oracle
        # check (five launches of a 0.2 ms pipeline read 15-20 % slow: tools/lab/config2_inplace.py)
        t_warm = time.perf_counter()
        while True:
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            if time.perf_counter() - t_warm >= 0.04:
                break
        r.timing_reset(); r.timing_enable(True)
        for _ in range(iters):
            fn()
        torch.cuda.synchronize(); r.timing_enable(False)
        return {n: r.timing_read(n)[0] / max(r.timing_read(n)[1], 1) for n in names if r.timing_read(n)[1]}

    def entry(ms, alg_bytes, **kw):
        return {"ms": round(ms, 4), "alg_bytes": int(alg_bytes), "GBs": round(alg_bytes / ms / 1e6, 1), "frac": round(alg_bytes / ms / 1e6 / HBM_PEAK_GBS, 4), **kw}

    # ---- config 2: 8K Gaussian sigma=16 + HSL(30, -20, 10) on the flattened frame; 8 + 8 algorithmic bytes per pixel ----
    hsl = torch.empty_like(flat)
    hp = (30.0, -20.0, 10.0)
    rad = int(np.ceil(np.float32(SIGMA) * np.float32(3.0)))
    y0, x0, wh, ww = 1000, 2000, 256, 512
    win = flat[y0 - rad:y0 + wh + rad, x0 - rad:x0 + ww + rad].contiguous().cpu().numpy()
    ref_blur = O.gaussian_blur(win, SIGMA)[rad:-rad, rad:-rad]

    def c2():
        r.gaussian_blur_dev(flat.data_ptr(), blurred.data_ptr(), w, h, SIGMA)
        r.adjust_dev(blurred.data_ptr(), hsl.data_ptr(), w, h, "hsl", hp)
    k = kernel_ms(("gauss_mfma", "adjust"), c2, 20)
    got_blur = blurred[y0:y0 + wh, x0:x0 + ww].contiguous().cpu().numpy()
    got_hsl = hsl[y0:y0 + wh, x0:x0 + ww].contiguous().cpu().numpy()
    dmax = int(np.abs(ref_blur.astype(np.int16) - got_blur.astype(np.int16)).max())
    hsl_ok = bool(np.array_equal(O.adjust(got_blur, "hsl", hp), got_hsl))
    ms2 = k.get("gauss_mfma", 0.0) + k.get("adjust", 0.0)
    # the same two ops through pfx_chain_dev (what a script's `apply_gaussian_blur(16.0); apply_hsl(..);` or the batch pipeline runs): light pointwise ops ride in the
    # Gaussian's store (one launch), HSL — 143 instructions per pixel — runs as a chain launch in place behind it; quoted against the same 16 B/px either way
    fused = torch.empty_like(flat)
    chain = [("gaussian", SIGMA), ("adjust", "hsl", hp)]
    kc = kernel_ms(("gauss_mfma_chain", "gauss_mfma", "chain"), lambda: r.chain_dev(flat.data_ptr(), fused.data_ptr(), w, h, chain), 20)
    same = bool(torch.equal(fused, hsl))   # whole frame: the chain is defined as the two calls one after the other
    ms2c = sum(kc.values())
    out["config2_gaussian16_hsl_8k"] = entry(ms2c, 16 * px, kernel_ms={n: round(v, 4) for n, v in kc.items()}, launches=len(kc),
                                             gaussian_mode="matrix cores: one f16 per tap, horizontal result as two f16, f32 accumulate (+-1 LSB class)",
                                             what="pfx_chain_dev(Gaussian sigma=16 -> HSL): " + ("one launch, HSL in the Gaussian's store" if len(kc) == 1 else
                                                  "the Gaussian, then HSL in place as a chain launch (HSL is 143 instructions per pixel: in the Gaussian's store it costs what "
                                                  "it costs as its own pass, so pfx_chain_dev does not fuse it; profiles/r06_tuning.md)"),
                                             check={"gaussian_window_max_diff_vs_oracle": dmax, "hsl_window_bitexact_on_the_gpu_blur": hsl_ok,
                                                    "whole_frame_identical_to_the_two_launch_form": same})
    out["config2_two_launches"] = entry(ms2, 16 * px, kernel_ms={n: round(v, 4) for n, v in k.items()}, what="pfx_gaussian_blur_dev then pfx_adjust_dev (round 5's config-2 line)")
    del fused
    if dmax > 1: failed.append("config2_gaussian")
    if not hsl_ok: failed.append("config2_hsl")
    if not same: failed.append("config2_chain_differs_from_the_two_launch_form")
    # the bit-exact Gaussian mode beside it (f32 VALU passes, intermediate in HBM: 8 + 32 bytes per pixel of traffic)
    r.set_exact(True)
    try:
        k = kernel_ms(("gauss_h", "gauss_v"), lambda: r.gaussian_blur_dev(flat.data_ptr(), blurred.data_ptr(), w, h, SIGMA), 10, warm=3)
        got = blurred[y0:y0 + wh, x0:x0 + ww].contiguous().cpu().numpy()
    finally:
        r.set_exact(False)
    ex_ok = bool(np.array_equal(ref_blur, got))
    out["config2_gaussian16_exact_mode_8k"] = entry(k.get("gauss_h", 0.0) + k.get("gauss_v", 0.0), 8 * px, kernel_ms={n: round(v, 4) for n, v in k.items()},
                                                    check={"window_bitexact_vs_oracle": ex_ok})
    if not ex_ok: failed.append("config2_exact_gaussian")
    del hsl

    # ---- config 4: 16K (15360 x 8640) fused 6x6 Catmull-Rom mesh warp (8 B/px) and Liquify displacement resample (16 B/px) ----
    W16, H16 = 15360, 8640
    g = torch.Generator(device=device); g.manual_seed(0x5EED0004)
    img = torch.randint(0, 256, (H16, W16, 4), dtype=torch.uint8, device=device, generator=g)
    dst = torch.empty_like(img)
    orig, deformed = I.jittered_mesh(6, 6, W16, H16)
    k = kernel_ms(("warp_mesh",), lambda: r.warp_mesh_catmull_rom_dev(img.data_ptr(), orig, deformed, 6, 6, W16, H16, dst.data_ptr()), 10, warm=3)
    disp = torch.zeros((H16, W16, 2), dtype=torch.float32, device=device)
    rng = np.random.default_rng(7)
    dabs = [(int(rng.integers(0, 5)), float(rng.uniform(0, W16)), float(rng.uniform(0, H16)), float(rng.uniform(-40, 40)), float(rng.uniform(-40, 40)),
             float(rng.uniform(200, 1500)), float(rng.uniform(0.2, 1.0))) for _ in range(48)]
    r.displacement_brushes_dev(disp.data_ptr(), W16, H16, dabs)
    k2 = kernel_ms(("warp_displacement",), lambda: r.warp_displacement_dev(img.data_ptr(), W16, H16, disp.data_ptr(), W16, H16, dst.data_ptr()), 10, warm=3)
    # oracle check on a 2048 x 1152 document (the same kernels; the whole-frame 16K comparison is tests/test_gpu_fullsize.py)
    cw, ch = 2048, 1152
    small = img[:ch, :cw].contiguous()
    sdst = torch.empty_like(small)
    so, sd = I.jittered_mesh(6, 6, cw, ch)
    r.warp_mesh_catmull_rom_dev(small.data_ptr(), so, sd, 6, 6, cw, ch, sdst.data_ptr())
    torch.cuda.synchronize()
    mesh_ok = bool(np.array_equal(O.warp_mesh_catmull_rom(small.cpu().numpy(), so, sd, 6, 6), sdst.cpu().numpy()))
    sdisp = torch.zeros((ch, cw, 2), dtype=torch.float32, device=device)
    r.displacement_brushes_dev(sdisp.data_ptr(), cw, ch, [(d[0], d[1] * cw / W16, d[2] * ch / H16, d[3], d[4], d[5] * cw / W16, d[6]) for d in dabs])
    r.warp_displacement_dev(small.data_ptr(), cw, ch, sdisp.data_ptr(), cw, ch, sdst.data_ptr())
    torch.cuda.synchronize()
    liq_ok = bool(np.array_equal(O.warp_displacement(small.cpu().numpy(), sdisp.cpu().numpy()), sdst.cpu().numpy()))
    out["config4_mesh_warp_16k"] = entry(k.get("warp_mesh", 0.0), 8 * W16 * H16, check={"2048x1152_whole_frame_bitexact_vs_oracle": mesh_ok})
    out["config4_liquify_16k"] = entry(k2.get("warp_displacement", 0.0), 16 * W16 * H16, check={"2048x1152_whole_frame_bitexact_vs_oracle": liq_ok})
    if not mesh_ok: failed.append("config4_mesh_warp")
    if not liq_ok: failed.append("config4_liquify")
    del img, dst, disp, small, sdst, sdisp
    torch.cuda.empty_cache()

    # ---- config 5, a 256-image slice on this GPU (0.18 s; 64 images were 45 ms, a sixth of it pipeline fill and drain: 1209 images/s against 1410 for the
    # whole 1024-image configuration, `--config batch4k`; the link gives 1450 with both directions busy, tools/lab/pcie_duplex.py): 3840 x 2160 images streamed over PCIe (Gaussian 4 -> HSL -> flatten under 3 overlays) ----
    from paintfe_amd.batch import run_batch
    w4, h4 = 3840, 2160
    pool, overlays, modes = s4_inputs(w4, h4)
    dev_index = device.index or 0
    run_batch([dev_index], 8, pool, overlays, modes, sigma=4.0, slots=args.slots)
    # the timed stream runs the pipeline's default: the bit-exact Gaussian (pfx_batch_params.out_of_contract_fast_gaussian = 0) — the blur feeds HSL, which amplifies a +-1 LSB
    # input, and the stream is PCIe-bound either way; the opt-in fast mode (f16 taps) is timed beside it and gated by its error bound
    res = run_batch([dev_index], 256, pool, overlays, modes, sigma=4.0, slots=args.slots, keep=[0])
    ok, chk = s4_check(O, pool[0], overlays, modes, res["kept"][0], exact=True)
    res_x = run_batch([dev_index], 256, pool, overlays, modes, sigma=4.0, slots=args.slots, keep=[1], fast=True)
    ok_x, chk_x = s4_check(O, pool[1], overlays, modes, res_x["kept"][1], exact=False)
    gbs = res["images_per_s"] * w4 * h4 * 4 / 1e9
    out["config5_batch_4k_slice"] = {"images": 256, "images_per_s": round(res["images_per_s"], 1), "mpixels_per_s": round(res["images_per_s"] * w4 * h4 / 1e6, 1),
                                     "resident_kernel_ms_per_image": round(res["kernel_ms_per_image"], 4), "alg_bytes_per_image": 36 * w4 * h4,
                                     "bound": "pcie", "GBs_each_direction": round(gbs, 2), "frac": round(gbs / PCIE_GEN5_X16_GBS, 3),
                                     "frac_of": f"PCIe Gen5 x16, {PCIE_GEN5_X16_GBS:g} GB/s per direction",
                                     "gaussian": "bit-exact f32 (the pipeline's default)",
                                     "fast_gaussian_variant": {"images_per_s": round(res_x["images_per_s"], 1),
                                                               "resident_kernel_ms_per_image": round(res_x["kernel_ms_per_image"], 4),
                                                               "what": "pfx_batch_params.out_of_contract_fast_gaussian = 1: f16 taps on the matrix cores, +-1 LSB before HSL"},
                                     "check": {"image0_vs_oracle": chk, "fast_gaussian_image1_vs_oracle": chk_x}}
    if not ok: failed.append("config5_image0")
    if not ok_x: failed.append("config5_fast_image1")

    # ---- config 1 (BASELINE configs[0], the reference's own CPU-runnable case): the batch tool on a 1024 x 1024 PNG with `apply_blur(4.0);` —
    # process start, context creation, PNG decode, script, PNG encode; wall clock of the whole process (src/cli.rs:105-215) ----
    try:
        import subprocess, tempfile
        from PIL import Image
        exe = os.path.join(ROOT, "paintfe_amd", "pfx")
        with tempfile.TemporaryDirectory() as td:
            src_png, out_png, script = os.path.join(td, "in.png"), os.path.join(td, "out.png"), os.path.join(td, "blur.rhai")
            img1 = I.create_test_gradient(1024, 1024)
            Image.fromarray(img1, "RGBA").save(src_png)
            open(script, "w").write("apply_blur(4.0);\n")
            walls = []
            for _ in range(3):
                t0 = time.perf_counter()
                rc = subprocess.run([exe, "-i", src_png, "-s", script, "-o", out_png], capture_output=True, timeout=120).returncode
                walls.append((time.perf_counter() - t0) * 1e3)
            got1 = np.asarray(Image.open(out_png).convert("RGBA"))
        d1 = int(np.abs(got1.astype(np.int16) - O.gaussian_blur(img1, 4.0).astype(np.int16)).max())
        out["config1_cli_png_1024_apply_blur4"] = {"wall_ms_per_process": [round(v, 1) for v in walls], "exit_code": rc, "mpixels_per_s_wall": round(1024 * 1024 / min(walls) / 1e3, 1),
                                                   "what": "paintfe_amd/pfx -i in.png -s blur.rhai -o out.png: process start + HIP context + PNG decode + script + PNG encode",
                                                   "check": {"max_diff_vs_oracle_gaussian": d1}}
        if rc != 0 or d1 > 1: failed.append("config1_cli")
    except Exception as e:  # the tool or PIL missing is a failure of this entry, not of the bench line
        out["config1_cli_png_1024_apply_blur4"] = {"error": str(e)}
        failed.append("config1_cli")
    return out, failed


def run_batch4k(args, torch, dist, rank, world, dev_index, device) -> int:
    """BASELINE config 5 / SURVEY 8d S4: `--images` independent 3840x2160 images, per image Gaussian sigma=4 -> HSL(30,-20,10) ->
    flatten under 3 overlays (Multiply / Screen / Overlay).  Sharded by image: rank r streams images r, r+world, ... through its GPU
    with pfx_batch_pipeline (pinned host memory; an upload, a kernel and a download stream over `--slots` buffer sets: H2D, kernels
    and D2H of different images overlap).
    36 algorithmic HBM bytes per pixel; the pace is set by PCIe (33 MB per image each way), which the line reports."""
    from paintfe_amd.batch import run_batch
    w, h = (3840, 2160) if (args.width, args.height) == (W8K, H8K) else (args.width, args.height)
    n_total = args.images
    mine = len(range(rank, n_total, world))
    pool, overlays, modes = s4_inputs(w, h)  # S1 images; the batch cycles a pool of 4
    run_batch([dev_index], min(8, max(mine, 1)), pool, overlays, modes, sigma=4.0, slots=args.slots)  # warm-up: clocks, allocator
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep = [0] if rank == 0 else []
    res = run_batch([dev_index], mine, pool, overlays, modes, sigma=4.0, slots=args.slots, keep=keep)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    # timed region = first enqueue .. last result back on the host (pfx_batch_stats.seconds): context creation, allocations and the
    # one-time overlay upload inside the call are set-up, like the warm-up steps of the headline mode
    elapsed = res["seconds"]
    if world > 1:
        from paintfe_amd.sharding import max_over_ranks
        elapsed = max_over_ranks(elapsed, device=device)
    img_s = n_total / elapsed
    bytes_img = w * h * 4
    out = {"metric": "images/sec: batch of 4K images, Gaussian sigma=4 + HSL + 4-layer flatten, streamed over PCIe", "value": round(img_s, 1),
           "unit": "images/s", "n_gpus": world, "steps": n_total, "warmup": 8, "ms_per_step": round(elapsed / n_total * 1e3, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"S4: {n_total} x {w}x{h} RGBA8, per image Gaussian sigma=4 -> HSL -> flatten under 3 overlays",
                      "images": n_total, "width": w, "height": h, "slots_per_gpu": args.slots,
                      "sharding": "by image (image i -> rank i mod N), no data-path collective"},
           "mpixels_per_s": round(img_s * w * h / 1e6, 1),
           "pcie": {"h2d_GBs_per_gpu": round(img_s * bytes_img / world / 1e9, 2), "d2h_GBs_per_gpu": round(img_s * bytes_img / world / 1e9, 2),
                    "peak_GBs_per_direction": PCIE_GEN5_X16_GBS, "frac_of_peak": round(img_s * bytes_img / world / 1e9 / PCIE_GEN5_X16_GBS, 3)},
           "wall_s_including_setup": round(wall, 3),
           "resident_kernel_ms_per_image": round(res["kernel_ms_per_image"], 4),
           "resident_kernel_fraction": round(res["kernel_ms_per_image"] * 1e-3 * (n_total / world) / elapsed, 3),
           "roofline": {"bound": "pcie", "achieved": round(img_s * bytes_img / world / 1e9, 2), "peak": PCIE_GEN5_X16_GBS, "unit": "GB/s",
                        "frac": round(img_s * bytes_img / world / 1e9 / PCIE_GEN5_X16_GBS, 3), "traffic": None,
                        "hbm_alg_bytes_per_px": 36, "hbm_GBs_at_kernel_time": round(36 * w * h / max(res["kernel_ms_per_image"], 1e-9) / 1e6, 1)}}
    failed = []
    if rank == 0:
        from tests import oracle_lib as O
        ok, chk = s4_check(O, pool[0], overlays, modes, res["kept"][0], exact=True)   # the pipeline's default is the bit-exact Gaussian
        out["check"] = {"image0_vs_oracle": chk}
        if not ok:
            failed.append("image0_vs_oracle")
            out["value"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 1 if failed else 0


def c_abi_group_leg(args, torch, devices, w, h, n, info, steps, warmup):
    """The multi-GPU path a Rust host would bind — ONE process driving every device through pfx_group_flatten_blur (include/pfx.h, csrc/pfx_group.cpp;
    reference seam: src/gpu/renderer.rs:533, one caller, one process) — timed beside the torch.distributed BandPipeline headline: the same document (the
    headline's seeds), sharded result and all-gathered result, under the PEER (hipMemcpyPeerAsync over xGMI) and RCCL (ncclSend / ncclRecv, ncclBroadcast)
    transports, with per-member phase clocks of one extra call (flatten / halo wait / blur / gather).  Layers reach the members through the host once,
    outside every timed region."""
    from paintfe_amd.group import GpuGroup
    from paintfe_amd import _lib as L
    out = {"members": len(devices), "devices": list(devices), "steps": steps,
           "what": "pfx_group_flatten_blur: one process, one context per device, bands of whole chunk rows, halo rows pulled from the neighbours' flattened bands"}
    g = GpuGroup(devices)
    try:
        g.set_document(w, h, n)
        g.set_exact(args.exact)
        dev0 = torch.device("cuda", devices[0])
        for k in range(n):
            g.upload_layer(k, synth_layer(torch, dev0, w, h, k, 0x5EED0002).cpu().numpy())
        torch.cuda.empty_cache()
        first = None
        transports = (("peer", GpuGroup.PEER), ("rccl", GpuGroup.RCCL)) if len(set(devices)) > 1 else (("peer", GpuGroup.PEER),)   # one device: nothing for RCCL to move
        for tname, tr in transports:
            try:
                g.set_transport(tr)
            except L.PfxError as e:
                out[tname] = {"error": str(e)[:300]}
                continue
            res = {}
            for gather in (False, True):
                key = "gathered" if gather else "sharded"
                try:
                    g.set_phase_timing(False)
                    for _ in range(warmup):
                        g.flatten_blur(info, args.sigma, gather)
                    g.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        g.flatten_blur(info, args.sigma, gather)
                    g.synchronize()
                    el = time.perf_counter() - t0
                    g.set_phase_timing(True)
                    g.flatten_blur(info, args.sigma, gather)
                    g.synchronize()
                    ph = [g.phase_ms(k) for k in range(len(devices))]
                    g.set_phase_timing(False)
                    res[key] = {"ms_per_step": round(el / steps * 1e3, 4), "value": round(w * h * steps / el / 1e6, 1), "unit": "Mpixels/s",
                                "phase_ms_per_member": [{k: round(v, 4) for k, v in p.items()} for p in ph]}
                    img = g.download()
                    if first is None:
                        first = img
                    else:
                        res[key]["identical_to_first_variant"] = bool(np.array_equal(first, img))
                except L.PfxError as e:
                    res[key] = {"error": str(e)[:300]}
            out[tname] = res
        out["_image"] = first
    finally:
        g.close()
    return out


def live_traffic(args, kernel_hint="flatten"):
    """HBM bytes per launch of the dominant kernel from rocprofv3's FETCH_SIZE / WRITE_SIZE counters, collected NOW on this box: two short child runs of this
    script (`--pmc-child`: a few launches of the compositor on the headline stack) under `rocprofv3 --kernel-trace --pmc <one counter>` — separate passes,
    FETCH_SIZE doubled, as guides/MI355X_MICROARCH.md prescribes for gfx950.  PMC collection needs rocprofv3 around the process, which is why it cannot be
    this process; the children run after every timed leg.  Returns (dict, None) or (None, reason); never raises, never outlives its timeout."""
    import csv, glob, shutil, signal, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process already runs under a profiler (no nested rocprofv3)"
    tmp = tempfile.mkdtemp(prefix="pfx_pmc_", dir="/tmp")
    res = {}
    try:
        for counter, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0), ("SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32", 1.0)):   # KB; FETCH_SIZE x 2 (gfx950 correction)
            cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "--pmc", *counter.split(), "-d", os.path.join(tmp, counter.split()[0]), "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "4", "--width", str(args.width), "--height", str(args.height),
                   "--layers", str(args.layers)]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=150)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)   # the exact process group this call started
                except Exception:
                    pass
                return None, f"{counter} pass timed out"
            for c in counter.split():
                vals = []
                for f in glob.glob(os.path.join(tmp, counter.split()[0], "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if kernel_hint in row.get("Kernel_Name", "") and row.get("Counter_Name") == c:
                            vals.append(float(row["Counter_Value"]))
                if not vals:
                    if c.startswith("SQ_"):   # the instruction counts are an extra: the traffic stands without them
                        continue
                    return None, f"{c} pass returned no rows for a {kernel_hint} kernel (exit code {p.returncode})"
                res[c] = (sum(vals) / len(vals) * scale, len(vals))
        d = {"hbm_bytes": int(round(res["FETCH_SIZE"][0] + res["WRITE_SIZE"][0], -5)), "fetch_bytes": int(res["FETCH_SIZE"][0]),
             "write_bytes": int(res["WRITE_SIZE"][0]), "launches_counted": res["FETCH_SIZE"][1],
             "how": "rocprofv3 --kernel-trace --pmc <counters> around short child runs of this script on this box (FETCH_SIZE, WRITE_SIZE and the SQ instruction counts "
                    "in separate passes; FETCH_SIZE x 2: gfx950)"}
        if "SQ_INSTS_VALU" in res:
            d["valu_wave_insts"] = int(res["SQ_INSTS_VALU"][0])
            if "SQ_INSTS_VALU_TRANS_F32" in res:
                d["valu_trans_wave_insts"] = int(res["SQ_INSTS_VALU_TRANS_F32"][0])
        return d, None
    except Exception as e:  # noqa: BLE001 — diagnostics only
        return None, f"{type(e).__name__}: {e}"[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=W8K)
    ap.add_argument("--height", type=int, default=H8K)
    ap.add_argument("--layers", type=int, default=NLAYERS)
    ap.add_argument("--sigma", type=float, default=SIGMA)
    ap.add_argument("--shard", choices=["doc", "band"], default="band",
                    help="N>1: 'band' (default) = ONE document cut into row bands, RCCL halo exchange before the blur, result left "
                         "sharded (strong scaling; the all-gathered variant is timed beside it); 'doc' = one document per GPU, no collective (weak scaling)")
    ap.add_argument("--no-gather", action="store_true", help="band mode: skip the all-gathered variant (band_gathered_result)")
    ap.add_argument("--no-group", action="store_true", help="skip the C-ABI single-process leg (c_abi_group: pfx_group_flatten_blur over every device)")
    ap.add_argument("--group-steps", type=int, default=10, help="timed steps of each c_abi_group variant")
    ap.add_argument("--band-split-edges", action="store_true",
                    help="band mode (development / tests): unpipelined steps that flatten the band's edge chunk rows first and the interior under the halo exchange")
    ap.add_argument("--config", choices=["headline", "batch4k"], default="headline",
                    help="'batch4k' = BASELINE config 5 (S4): a batch of 3840x2160 images, per image Gaussian sigma=4 -> HSL -> 4-layer "
                         "flatten, streamed over PCIe with pinned double-buffering and sharded by image across the GPUs")
    ap.add_argument("--images", type=int, default=1024, help="batch4k: images in the batch (whole job)")
    ap.add_argument("--slots", type=int, default=3, help="batch4k: buffer sets in the per-GPU upload / kernels / download pipeline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="skip the live HBM-traffic pass (two short rocprofv3 child runs, FETCH_SIZE and WRITE_SIZE, after everything else); roofline.traffic then "
                         "comes from the committed counter pass (profiles/rNN_pmc.json) and is marked static")
    ap.add_argument("--pmc-child", type=int, default=0, help=argparse.SUPPRESS)   # internal: N launches of the dominant kernel on the headline stack, nothing printed
    ap.add_argument("--headline-only", action="store_true", help="skip the other BASELINE configurations (development runs)")
    ap.add_argument("--tune", action="append", default=[], help="key=value kernel tuning knob (development)")
    ap.add_argument("--exact", action="store_true", help="Gaussian without FMA contraction (bit-exact with the CPU path)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # PFX_BENCH_BACKEND=gloo is a plumbing self-test only: it lets N ranks share the one GPU of a development box
    # (RCCL refuses two ranks per device).  The real multi-GPU run is one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("PFX_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # a collective that never completes must become an error well inside the driver's patience (the default watchdog waits 10 minutes)
        tmo = datetime.timedelta(seconds=int(os.environ.get("PFX_BENCH_COLLECTIVE_TIMEOUT_S", "120")))
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index), timeout=tmo)
        else:
            dist.init_process_group(backend=backend, timeout=tmo)
    host_pg = None
    if world > 1 and backend == "nccl":
        # a host-side group: ranks that wait for rank 0's single-process leg (c_abi_group) must not spin in an RCCL barrier kernel on the GPUs that leg uses
        host_pg = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=600))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if args.gpus != world:
        # the driver's contract: `--gpus N` is launched as N ranks.  A job that silently ran on fewer ranks (or one that was started
        # without torch.distributed.run) must not report an N-GPU number
        raise SystemExit(f"bench.py --gpus {args.gpus}: the job has WORLD_SIZE={world} ranks (launch with python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {args.gpus} ... for N > 1)")
    if world > 1:
        joined = dist.get_world_size()
        if joined != world:
            raise SystemExit(f"bench.py: {joined} ranks joined the process group, expected {world}")
        if backend == "nccl" and torch.cuda.device_count() < world:
            raise SystemExit(f"bench.py --gpus {world}: only {torch.cuda.device_count()} devices are visible (one rank per GPU over RCCL)")

    if args.config == "batch4k":
        return run_batch4k(args, torch, dist, rank, world, dev_index, device)

    from paintfe_amd import GpuRenderer
    r = GpuRenderer(dev_index)
    r.set_exact(args.exact)
    for kv in args.tune:
        k, v = kv.split('=')
        r.tune(k, int(v))
    r.set_stream(torch.cuda.current_stream().cuda_stream)  # HIP events + kernels on the stream torch synchronises

    w, h, n = args.width, args.height, args.layers
    band_mode = args.shard == "band" and world > 1
    radius = int(np.ceil(np.float32(args.sigma) * np.float32(3.0)))
    modes, opac = synth_params(n, 0x5EED0002)
    info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
    if args.pmc_child:
        # the counted launches of live_traffic(): the headline stack, the dominant kernel, nothing else on the device
        stack, _, _ = synth_stack(torch, device, w, h, n, seed=0x5EED0002)
        out_c = torch.empty((h, w, 4), dtype=torch.uint8, device=device)
        ptrs_c = [stack[k].data_ptr() for k in range(n)]
        for _ in range(args.pmc_child):
            r.flatten_dev(ptrs_c, info, w, h, out_c.data_ptr())
        torch.cuda.synchronize()
        return 0

    state = {}

    def bracket():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn):
        for _ in range(PREWARM + args.warmup):
            step_fn()
        bracket()
        r.timing_reset()
        r.timing_enable(True)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]  # on the stream the kernels run on (set_stream above)
        t0 = time.perf_counter()
        marks[0].record()
        for k in range(args.steps):
            step_fn()
            marks[k + 1].record()
        bracket()
        el = time.perf_counter() - t0
        r.timing_enable(False)
        per = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
        state["step_ms"] = {"min": round(per[0], 4), "median": round(per[len(per) // 2], 4), "max": round(per[-1], 4)}
        if world > 1:
            from paintfe_amd.sharding import max_over_ranks
            mine = torch.tensor([el / args.steps * 1e3], dtype=torch.float64, device=None if backend == "gloo" else device)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            state["per_rank_ms_per_step"] = [round(float(t.item()), 4) for t in every]
            el = max_over_ranks(el, device=device)  # the step time of the job is the slowest rank's
        kern = {}
        for name in ("flatten", "gauss_mfma", "gauss_h", "gauss_v", "gauss_fused"):
            ms, cnt = r.timing_read(name)
            if cnt:
                kern[name] = (ms / cnt, cnt)
        return el, kern

    doc_mode = None
    if band_mode:
        # ONE document for the whole job, cut into bands of whole chunk rows (pfx_band_rows): every rank generates the same layers
        # (same seeds) and keeps only its rows.  Buffers are allocated once: [top halo | band | bottom halo].
        from paintfe_amd import sharding as S
        y0, y1 = S.band_rows(h, world, rank)
        hh = y1 - y0
        stack = torch.empty((n, max(hh, 1), w, 4), dtype=torch.uint8, device=device)
        for k in range(n):
            stack[k, :hh] = synth_layer(torch, device, w, h, k, 0x5EED0002)[y0:y1]
        pipe = S.BandPipeline(r, w, h, radius, args.sigma, device, gather=not args.no_gather, split_edges=args.band_split_edges,
                              pipelined=not args.band_split_edges)
        ptrs = [stack[k].data_ptr() for k in range(n)]

        def step():
            if os.environ.get("PFX_BENCH_TEST_FAIL_BAND"):  # test hook: what an RCCL error in the band pipeline looks like to this script
                raise RuntimeError("injected band pipeline failure (PFX_BENCH_TEST_FAIL_BAND)")
            state["result"] = pipe.step(ptrs, info)

        # The band pipeline is the only part of this run with collectives in its timed region.  HEADLINE: the document stays sharded in bands
        # (TiledImage chunk rows on the GPU that computed them) — the halo rows are the one exchange the path needs; an all-gather of the blurred
        # frame into EVERY rank is timed right after it and reported beside it (`band_gathered_result`): it moves (N - 1) / N of the frame into
        # each GPU, which xGMI's inbound links bound at ~4.5x for 8 GPUs whatever the kernels do.  If the headline pipeline fails
        # (an RCCL error, a watchdog timeout) the line still carries what needs no collective — `doc_mode` below — so that a partial scaling
        # curve survives (VERDICT r03 #5b): the headline value is then null and the error is on the line.  A failure of the gathered variant
        # alone leaves the headline standing and is reported as `band_gathered_error`.
        band_gathered = None
        try:
            pipe.gather = False
            elapsed, kern = timed(step)   # pipelined: step k's halo rows travel under step k + 1's flatten; K flattens + K blurs inside the timed region
            pipe.finish()                 # the last step's blur
            flat_view = pipe.flat_band()
            state["own_band"] = pipe.last_result.clone()
            state["headline_step_ms"], state["headline_per_rank"] = state.get("step_ms"), state.get("per_rank_ms_per_step")
        except Exception as e:  # noqa: BLE001 — reported on the line, the process still exits non-zero
            state["band_error"] = f"{type(e).__name__}: {e}"[:500]
            elapsed, kern, flat_view = float("nan"), {}, None
        if not args.no_gather and not state.get("band_error"):
            try:
                pipe.gather = True
                g_el, _ = timed(step)
                pipe.finish()
                band_gathered = {"value": round(w * h * args.steps / g_el / 1e6, 1), "unit": "Mpixels/s", "scaling": "strong",
                                 "ms_per_step": round(g_el / args.steps * 1e3, 4),
                                 "sharding": "the same band pipeline + an all-gather of the blurred bands into every rank (asynchronous, double-buffered: "
                                             "it overlaps the next step's flatten)"}
                state["gathered_ok"] = True
            except Exception as e:  # noqa: BLE001
                state["gather_error"] = f"{type(e).__name__}: {e}"[:500]
        # per-rank phase clocks of one unpipelined step (flatten / halo exchange / filter / gather): where a rank's time goes on real links
        if not state.get("band_error"):
            try:
                pipe.gather = not args.no_gather and not state.get("gather_error")
                ph = pipe.phase_probe(ptrs, info)
                mine = torch.tensor([ph["flatten"], ph["halo_exchange"], ph["filter"], ph["gather"]], dtype=torch.float64, device=None if backend == "gloo" else device)
                every = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(every, mine)
                state["phase_ms_per_rank"] = [dict(zip(("flatten", "halo_exchange", "filter", "gather"), [round(float(v), 4) for v in t.tolist()])) for t in every]
            except Exception as e:  # noqa: BLE001 — diagnostics only
                state["phase_probe_error"] = f"{type(e).__name__}: {e}"[:300]
        # the collective-free mode in the same run (one independent 8K document per rank), reported beside the headline
        del stack
        torch.cuda.empty_cache()
        dstack, _, _ = synth_stack(torch, device, w, h, n, seed=0x5EED0002 + 1000 * rank)
        dflat = torch.empty((h, w, 4), dtype=torch.uint8, device=device)
        dblur = torch.empty((h, w, 4), dtype=torch.uint8, device=device)
        dptrs = [dstack[k].data_ptr() for k in range(n)]

        def doc_step():
            r.flatten_dev(dptrs, info, w, h, dflat.data_ptr())
            r.gaussian_blur_dev(dflat.data_ptr(), dblur.data_ptr(), w, h, args.sigma)

        d_el, _ = timed(doc_step)
        doc_mode = {"value": round(w * h * args.steps * world / d_el / 1e6, 1), "unit": "Mpixels/s", "scaling": "weak",
                    "ms_per_step": round(d_el / args.steps * 1e3, 4), "sharding": "one independent document per GPU, no collective"}
        del dstack
    else:
        stack, modes, opac = synth_stack(torch, device, w, h, n, seed=0x5EED0002 + 1000 * rank)
        info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
        hh, y0, y1 = h, 0, h
        flat = torch.empty((h, w, 4), dtype=torch.uint8, device=device)
        blurred = torch.empty((h, w, 4), dtype=torch.uint8, device=device)
        ptrs = [stack[k].data_ptr() for k in range(n)]

        def step():
            r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
            r.gaussian_blur_dev(flat.data_ptr(), blurred.data_ptr(), w, h, args.sigma)

        elapsed, kern = timed(step)
        flat_view = flat
        if world == 1:
            state["clock_power"] = clock_power_sample(torch, step, dev_index)
        if world == 1 and not args.exact:
            # the strict-f32 pipeline beside the default one: the same step with the Gaussian in the bit-exact mode (f32 taps, one rounding per
            # operation, no matrix cores), timed the same way into its own result buffer and checked whole-frame at tolerance 0 where the default leg is checked
            blurred_x = torch.empty((h, w, 4), dtype=torch.uint8, device=device)

            def step_exact():
                r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
                r.gaussian_blur_dev(flat.data_ptr(), blurred_x.data_ptr(), w, h, args.sigma)

            keep = state.get("step_ms")
            r.set_exact(True)
            try:
                x_el, x_kern = timed(step_exact)
            finally:
                r.set_exact(False)
            state["exact_leg"] = {"elapsed": x_el, "kern": x_kern, "step_ms": state.get("step_ms"), "image": blurred_x}
            state["step_ms"] = keep

    px_per_step = w * h
    docs = 1 if (band_mode or world == 1) else world         # band mode: the whole job is one document per step
    value = px_per_step * args.steps * docs / elapsed / 1e6  # whole-job Mpx/s
    px_per_launch = w * hh                                   # pixels one flatten launch on this rank covers

    # roofline of the dominant kernel: the compositor carries 132 of the pipeline's 140 algorithmic bytes per pixel.
    # achieved = algorithmic bytes of one launch / mean HIP-event duration of the launch on the launch stream.
    dominant = "flatten"
    d_ms, d_cnt = kern.get(dominant, (0.0, 0))
    # band mode flattens a band in up to three launches (edge chunk rows first, then the interior): the band's bytes go against the SUM of its launches
    per_step = max(1, round(d_cnt / args.steps)) if d_cnt else 1
    alg_bytes = (4 * n + 4) * px_per_launch
    achieved = alg_bytes / (d_ms * per_step * 1e-3) / 1e9 if d_ms > 0 else 0.0
    pipeline_bytes = (4 * n + 4 + 8) * px_per_step
    pmc = pmc_profile(w, h, n)
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                # PMC traffic of one whole-frame launch (committed counter pass); a rank's band launch of an N > 1 run has no counter pass of its own
                "traffic": (pmc or {}).get("flatten", {}).get("hbm_bytes") if world == 1 else None,
                "kernel_ms": {k: round(v[0], 4) for k, v in kern.items()},
                **({"flatten_launches_per_step": per_step} if per_step > 1 else {}),
                "pipeline_achieved_GBs": round(pipeline_bytes * args.steps * docs / elapsed / 1e9, 1),
                "pipeline_frac": round(pipeline_bytes * args.steps * docs / elapsed / 1e9 / HBM_PEAK_GBS, 4)}
    # what the memory system delivers to this access pattern with no arithmetic at all (33 separately allocated layers walked in lock step, raw loads, an XOR per
    # register: tools/lab/nstream_read.hip, profiles/r05_nstream_read.txt): a measured constant of the chip, beside the data-sheet peak `frac` is quoted against
    if dominant == "flatten" and achieved > 0:
        roofline["pattern_ceiling"] = {"GBs": PATTERN_CEILING_GBS, "frac_of_it": round(achieved / PATTERN_CEILING_GBS, 4), "static": True,
                                       "what": "N-stream lock-step read, >= 4 loads in flight per wave; 4.7-4.8 TB/s with the two a typed-load register budget allows",
                                       "profile": "profiles/r05_nstream_read.txt"}
    # what actually limits each kernel: the contract's `frac` stays against HBM; the issue-slot view is beside it
    if pmc and d_ms > 0 and world == 1:
        fl = pmc.get("flatten", {})
        if fl.get("valu_wave_insts"):
            # one wave64 VALU instruction occupies a SIMD's issue port for 2 cycles; 1024 SIMDs; clock = the PMC run's measured one
            clk = fl.get("clock_ghz", 2.0)
            roofline["valu_frac"] = round(fl["valu_wave_insts"] * 2 / (1024 * clk * 1e9 * d_ms * 1e-3), 3)
            roofline["valu_insts_per_layer_px"] = round(fl["valu_wave_insts"] * 64 / (n * px_per_launch), 1)
            if fl.get("valu_issue_cycles_inmix"):
                # the profiled instruction mix at its IN-MIX issue costs (every plain VALU instruction 2 SIMD cycles with >= 2 waves resident, v_rcp / v_sqrt
                # 7.85: tools/lab/valu_mix.hip, profiles/r04_valu_rates.txt) against THIS run's kernel duration at the profiled clock.  Measured costs applied
                # to counted instructions — still arithmetic on counters, not a busy counter; round 3's 2 / 4 / 8 class prices (isolated chains) over-priced it.
                roofline["valu_issue_frac_inmix_costs"] = round(fl["valu_issue_cycles_inmix"] / (1024 * clk * 1e9 * d_ms * 1e-3), 3)
            # against what the chip's VALU sustains: a pure v_fma_f32 loop (tools/ubench_valu, profiles/rNN_valu_peak.json); a quarter-rate
            # (transcendental) instruction takes four plain instructions' worth of the pipe
            try:
                import glob
                vp = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_peak.json")))[-1]))
                trans = (fl.get("valu_class_mix") or {}).get("trans_f32", vp["flatten"]["transcendental_wave_insts"])   # this profile's own count
                slots = fl["valu_wave_insts"] + 3 * trans
                roofline["valu_frac_of_sustained_fma_rate"] = round(slots * 64 / (d_ms * 1e-3) / (vp["fma_sustained_T_lane_ops_s"] * 1e12), 3)
                roofline["fma_sustained_T_lane_ops_s"] = vp["fma_sustained_T_lane_ops_s"]
            except Exception:
                pass
        roofline["per_kernel_bound"] = pmc.get("bounds")
        # traffic / valu_* / per_kernel_bound come from a committed counter pass, not from this run (PMC collection needs rocprofv3 around
        # the process): say so, and which build the pass profiled
        roofline["static"] = {"fields": ["traffic", "valu_frac", "valu_issue_frac_inmix_costs", "valu_insts_per_layer_px", "valu_frac_of_sustained_fma_rate", "per_kernel_bound"],
                              "profile": pmc.get("_file"), "profiled_commit": pmc.get("commit"), "static": True}

    if state.get("clock_power"):
        roofline.update({k: state["clock_power"][k] for k in ("clock_ghz_sustained", "socket_power_w", "power_cap_w")})
        roofline["clock_power"] = state["clock_power"]
    out = {"metric": "Mpixels/sec: 8K 32-layer flatten + Gaussian sigma=16; HBM GB/s vs peak", "value": round(value, 1),
           "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "strong" if band_mode else "weak",
           "vs_baseline": None, "dtype": "f32" if args.exact else "f32 (Gaussian leg: f16 taps x u8, f32 accumulate)", "data": "synthetic",
           "config": {"workload": f"{w}x{h} RGBA8 x {n} layers (25 blend modes cycling, S2) flatten -> Gaussian sigma={args.sigma:g}",
                      "width": w, "height": h, "layers": n, "sigma": args.sigma,
                      "gaussian_mode": "exact (f32, no FMA)" if args.exact else "matrix cores: one f16 per tap, horizontal result as two f16, f32 accumulate (+-1 LSB class)",
                      "sharding": ("ONE document in chunk-row bands: RCCL send/recv of %d halo rows before the blur; the blurred result stays sharded in "
                                   "bands%s%s" % (radius, "" if args.band_split_edges else "; back-to-back steps pipelined (a step's halo rows travel under the next step's flatten)",
                                                "" if args.no_gather else " (the all-gathered variant of the same run: band_gathered_result)")) if band_mode else
                                  ("one independent document per GPU, no collective" if world > 1 else "single GPU")},
           "step_ms_hip_events": state.get("headline_step_ms", state.get("step_ms")),   # the headline's, not a secondary mode's
           "prewarm": PREWARM,   # untimed steps in front of the --warmup steps of every timed leg (the clock needs ~30 ms of load to settle)
           "roofline": roofline}
    if state.get("exact_leg"):
        xl = state["exact_leg"]
        x_bytes = pipeline_bytes * args.steps / xl["elapsed"] / 1e9
        out["value_exact_f32"] = round(px_per_step * args.steps / xl["elapsed"] / 1e6, 1)
        out["ms_per_step_exact"] = round(xl["elapsed"] / args.steps * 1e3, 4)
        out["pipeline_frac_exact"] = round(x_bytes / HBM_PEAK_GBS, 4)
        out["exact_f32_leg"] = {"what": "the same step with the Gaussian in the bit-exact mode (f32 taps and sums, one rounding per operation; dtype f32 throughout)",
                                "kernel_ms": {k: round(v[0], 4) for k, v in xl["kern"].items()}, "step_ms_hip_events": xl["step_ms"],
                                "pipeline_achieved_GBs": round(x_bytes, 1)}
    if world > 1:
        out["ranks"] = {"world_size": dist.get_world_size(), "backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend,
                        "devices_visible": torch.cuda.device_count(),
                        "per_rank_ms_per_step": state.get("headline_per_rank", state.get("per_rank_ms_per_step")),
                        **({"phase_ms_per_rank": state["phase_ms_per_rank"]} if state.get("phase_ms_per_rank") else {}),
                        **({"phase_probe_error": state["phase_probe_error"]} if state.get("phase_probe_error") else {})}
    if doc_mode:
        out["doc_mode"] = doc_mode
    if band_mode and band_gathered:
        out["band_gathered_result"] = band_gathered
    if state.get("gather_error"):
        out["band_gathered_error"] = state["gather_error"]
    failed = []
    if state.get("band_error"):
        out["value"] = None
        out["band_pipeline_error"] = state["band_error"]
        failed.append("band_pipeline")

    if rank == 0:
        # correctness of the TIMED result against the oracle: a crop of the flatten (per-pixel, so a crop of the full-size flatten
        # equals the flatten of the cropped stack); the band-mode blur is checked by every rank below
        from tests import oracle_lib as O
        ch, cw = 256, 512
        cy, cx = min(1000, max(hh - ch, 0)), min(2000, max(w - cw, 0))
        if not band_mode and hh >= cy + ch and w >= cx + cw:
            crop_stack = stack[:, cy:cy + ch, cx:cx + cw, :].contiguous().cpu().numpy()
            ref = O.flatten_stack(crop_stack, modes, opac)
            got = flat_view[cy:cy + ch, cx:cx + cw, :].contiguous().cpu().numpy()
            ok = bool(np.array_equal(ref, got))
            out["check"] = {"flatten_crop_bitexact": ok}
            if not ok:
                out["check"]["mismatching_px"] = int((ref != got).any(-1).sum())
                failed.append("flatten_crop_bitexact")
        if band_mode and hh > 0 and not state.get("band_error"):
            # rank 0 rebuilds a window around its band from the shared seeds and runs the single-process oracle pipeline on it
            lo, hi = max(y0 - 2 * radius, 0), min(y1 + 2 * radius, h)
            if (hi - lo) * w <= (1 << 25):
                full = torch.stack([synth_layer(torch, device, w, h, k, 0x5EED0002)[lo:hi] for k in range(n)]).cpu().numpy()
                ref_blur = O.gaussian_blur(O.flatten_stack(full, modes, opac), args.sigma)
                # rows of the window whose +-radius neighbourhood is inside the window (or clamps at the true image edge)
                a0 = 0 if lo == 0 else radius
                a1 = (hi - lo) if hi == h else (hi - lo) - radius
                tol = 0 if args.exact else 1
                # ALWAYS: the rank's own band as the timed headline pipeline (pipelined steps, deferred halo wait) left it — its edge rows depend on the halo exchange
                got_rows, ref_rows = state["own_band"].contiguous().cpu().numpy(), ref_blur[y0 - lo:y1 - lo]
                dmax = int(np.abs(ref_rows.astype(np.int16) - got_rows.astype(np.int16)).max())
                out.setdefault("check", {})["band_blur_checked_rows"] = "own band of the headline (pipelined) run"
                out["check"]["band_blur_max_diff_vs_oracle"] = dmax
                if dmax > tol:
                    failed.append("band_blur_max_diff_vs_oracle")
                if state.get("gathered_ok"):    # and the window around the band, neighbours' rows included, from the gathered variant's frame
                    got_rows, ref_rows = pipe.assemble()[lo + a0:lo + a1].contiguous().cpu().numpy(), ref_blur[a0:a1]
                    dmax = int(np.abs(ref_rows.astype(np.int16) - got_rows.astype(np.int16)).max())
                    out["check"]["gathered_window_max_diff_vs_oracle"] = dmax
                    if dmax > tol:
                        failed.append("gathered_window_max_diff_vs_oracle")
                state["ref_window"] = (lo + a0, lo + a1, ref_blur[a0:a1])

        if not args.no_cpu_baseline and world == 1:
            # bounded sample of the same workload: the whole frame of the same stack while that stays a few seconds on the
            # host cores (the 8K x 32 default: ~2 s on 16 cores), otherwise a 3840x2160 window of it
            whole = n * w * h * 4 <= (6 << 30)
            sh, sw = (h, w) if whole else (min(2160, h), min(3840, w))
            sample = stack[:, :sh, :sw, :].contiguous().cpu().numpy()
            cores = usable_cores()
            O.flatten_stack(sample[:, :64, :256], modes, opac, threads=cores)  # thread pool up before the clock starts
            t1 = time.perf_counter()
            f = O.flatten_stack(sample, modes, opac, threads=cores)
            t2 = time.perf_counter()
            O.gaussian_blur(f, args.sigma, threads=cores)
            t3 = time.perf_counter()
            if whole and not band_mode:  # the baseline's own output doubles as a whole-frame parity check of the timed result
                ok = bool(np.array_equal(f, flat_view.cpu().numpy()))
                out.setdefault("check", {})["flatten_whole_frame_bitexact"] = ok
                if not ok:
                    failed.append("flatten_whole_frame_bitexact")
                gb = O.gaussian_blur(f, args.sigma, threads=cores)
                dmax = int(np.abs(gb.astype(np.int16) - blurred.cpu().numpy().astype(np.int16)).max())
                out["check"]["gaussian_whole_frame_max_diff"] = dmax
                out["check"]["gaussian_channels_off_by_one"] = round(float((gb != blurred.cpu().numpy()).mean()), 6)
                if dmax > (0 if args.exact else 1):
                    failed.append("gaussian_whole_frame_max_diff")
                if state.get("exact_leg"):
                    x_ok = bool(np.array_equal(gb, state["exact_leg"]["image"].cpu().numpy()))
                    out["exact_f32_leg"]["gaussian_whole_frame_bitexact"] = x_ok
                    if not x_ok:
                        out["value_exact_f32"] = None
                        failed.append("exact_f32_leg_gaussian_whole_frame_bitexact")
            # the reference-faithful variant beside the fair one: rayon collects the chunks and ONE thread writes them back
            # (canvas_state.rs:686-695)
            O.set_serial_writeback(True)
            t4 = time.perf_counter()
            O.flatten_stack(sample, modes, opac, threads=cores)
            t5 = time.perf_counter()
            O.set_serial_writeback(False)
            out["cpu_baseline"] = {"value": round(sw * sh / (t3 - t1) / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port",
                                   "sample": f"{'whole ' + str(sw) + 'x' + str(sh) + ' frame' if whole else str(sw) + 'x' + str(sh) + ' window'} "
                                             f"of the same {n}-layer stack, flatten {t2 - t1:.2f}s + gaussian {t3 - t2:.2f}s, "
                                             f"OpenMP restatement of PaintFE's rayon CPU path (oracle/): chunk-parallel compositing with the "
                                             f"write-back parallel too (serial in the reference), row-parallel Gaussian passes",
                                   "faithful": {"value": round(sw * sh / ((t5 - t4) + (t3 - t2)) / 1e6, 2), "unit": "Mpixels/s", "cores": cores,
                                                "what": f"same sample with the reference's collect + single-threaded put_pixel write-back "
                                                        f"(canvas_state.rs:686-695): flatten {t5 - t4:.2f}s + gaussian {t3 - t2:.2f}s"}}
        if state.get("exact_leg") and "gaussian_whole_frame_bitexact" not in out.get("exact_f32_leg", {}):
            # no whole-frame baseline in this run: a full-width window of the strict-f32 result against the oracle's Gaussian of the same flatten rows
            lo_w, hi_w = max(0, 1000 - radius), min(h, 1256 + radius)
            ref_w = O.gaussian_blur(flat_view[lo_w:hi_w].contiguous().cpu().numpy(), args.sigma)
            a0 = 0 if lo_w == 0 else radius
            a1 = (hi_w - lo_w) if hi_w == h else (hi_w - lo_w) - radius
            x_ok = bool(np.array_equal(ref_w[a0:a1], state["exact_leg"]["image"][lo_w + a0:lo_w + a1].contiguous().cpu().numpy()))
            out["exact_f32_leg"]["gaussian_window_bitexact"] = x_ok
            if not x_ok:
                out["value_exact_f32"] = None
                failed.append("exact_f32_leg_gaussian_window_bitexact")
        state.pop("exact_leg", None)
        if not args.no_group and (world == 1 or band_mode):
            # the C-ABI multi-GPU path (one process, every device), beside the torch.distributed headline; the other ranks wait on the host (host_pg)
            devices = list(range(world)) if (world > 1 and backend == "nccl") else [dev_index] * world
            try:
                gl = c_abi_group_leg(args, torch, devices, w, h, n, info, args.group_steps, PREWARM)
                img = gl.pop("_image", None)
                chk = {}
                if img is not None and world == 1 and not band_mode:
                    chk["identical_to_headline_result"] = bool(np.array_equal(img, blurred.cpu().numpy()))
                    if not chk["identical_to_headline_result"]:
                        failed.append("c_abi_group_identical_to_headline_result")
                if img is not None and state.get("ref_window"):
                    r0, r1, ref_rows = state["ref_window"]
                    chk["window_max_diff_vs_oracle"] = int(np.abs(ref_rows.astype(np.int16) - img[r0:r1].astype(np.int16)).max())
                    if chk["window_max_diff_vs_oracle"] > (0 if args.exact else 1):
                        failed.append("c_abi_group_window_max_diff_vs_oracle")
                for tname in ("peer", "rccl"):
                    for key, v in (gl.get(tname) or {}).items():
                        if isinstance(v, dict) and v.get("identical_to_first_variant") is False:
                            failed.append(f"c_abi_group_{tname}_{key}_differs")
                gl["check"] = chk
                out["c_abi_group"] = gl
                del img
            except Exception as e:  # noqa: BLE001 — a secondary leg: reported, never fatal to the headline
                out["c_abi_group"] = {"error": f"{type(e).__name__}: {e}"[:400]}
        if world == 1 and not band_mode and not args.headline_only and (w, h) == (W8K, H8K):
            del stack
            torch.cuda.empty_cache()
            cfgs, cfg_failed = other_configs(args, torch, r, device, O, flat, blurred)
            out["configs"] = cfgs
            failed += cfg_failed
        # LAST, with nothing else on the device: the dominant kernel's HBM traffic from this box's counters (the other counter fields stay static: they need
        # SQ passes whose collection slows the kernels down)
        if world == 1 and not band_mode and not args.no_live_pmc and dominant == "flatten" and not args.tune:
            torch.cuda.synchronize()
            live, why = live_traffic(args)
            rf = out.get("roofline") or {}
            if live:
                rf["traffic"] = live["hbm_bytes"]
                rf["traffic_live"] = live
                now_live = ["traffic"]
                d_ms_l = (rf.get("kernel_ms") or {}).get("flatten", 0.0)
                if live.get("valu_wave_insts") and d_ms_l > 0:
                    # the instruction count is a property of the launch (the same stack, the same kernel): this box's count against this run's duration, at the
                    # clock this run sampled while the workload ran (the committed pass's clock otherwise)
                    clk = rf.get("clock_ghz_sustained") or 2.0
                    rf["valu_frac"] = round(live["valu_wave_insts"] * 2 / (1024 * clk * 1e9 * d_ms_l * 1e-3), 3)
                    rf["valu_insts_per_layer_px"] = round(live["valu_wave_insts"] * 64 / (args.layers * args.width * args.height), 1)
                    now_live += ["valu_frac", "valu_insts_per_layer_px"]
                    if rf.get("fma_sustained_T_lane_ops_s"):
                        slots = live["valu_wave_insts"] + 3 * live.get("valu_trans_wave_insts", 0)
                        rf["valu_frac_of_sustained_fma_rate"] = round(slots * 64 / (d_ms_l * 1e-3) / (rf["fma_sustained_T_lane_ops_s"] * 1e12), 3)
                        now_live.append("valu_frac_of_sustained_fma_rate")
                if isinstance(rf.get("static"), dict):
                    rf["static"]["fields"] = [f for f in rf["static"]["fields"] if f not in now_live]
            else:
                rf["traffic_live"] = {"error": why}
        if failed:  # a wrong-but-fast kernel must not be scored
            out["value"] = None
            out["failed_checks"] = failed
        try:  # whatever native libraries left in the C library's stdout buffer (RCCL prints a version banner) goes out BEFORE the one JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(nan_to_none(out)), flush=True)

    if world > 1:
        if host_pg is not None:
            torch.cuda.synchronize()
            dist.barrier(group=host_pg)   # on the host: rank 0's single-process leg has the GPUs to itself
        dist.barrier()
        dist.destroy_process_group()
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
