#!/usr/bin/env python3
"""bench.py — headline benchmark: Mpixels/sec on "8K x 32-layer flatten + Gaussian sigma=16" (BASELINE.json).

One "step" = one pass of the hot path over one synthetic 8K document that is already resident in HBM:
    flatten 32 RGBA8 layers (all 25 blend modes, S2 of SURVEY.md §8d)  ->  Gaussian blur sigma=16 of the result.
Both results are materialised (140 algorithmic bytes per output pixel).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU: one process per GPU; documents are independent units (the reference's CLI batch loops over files,
src/cli.rs:159), so rank r processes its own 8K document with no data-path collective ("scaling": "weak").
`value` = pixels all ranks produced / max-over-ranks wall time between two barrier+synchronize brackets.

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


def pmc_traffic(kernel: str, w: int, h: int, n: int):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/rNN_traffic.json, produced by
    tools/prof.sh: FETCH_SIZE and WRITE_SIZE in separate --pmc runs, FETCH_SIZE doubled as the gfx950 guide
    prescribes).  PMC collection cannot run inside this process, so the figure is only reported for the exact
    configuration it was measured on (8K x 32 layers); otherwise null."""
    if (w, h, n) != (W8K, H8K, NLAYERS):
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))[kernel]["hbm_bytes"]
    except Exception:
        return None


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


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=W8K)
    ap.add_argument("--height", type=int, default=H8K)
    ap.add_argument("--layers", type=int, default=NLAYERS)
    ap.add_argument("--sigma", type=float, default=SIGMA)
    ap.add_argument("--shard", choices=["doc", "band"], default="doc",
                    help="N>1: 'doc' = one document per GPU, no collective (weak scaling, default); 'band' = ONE document cut "
                         "into row bands with an RCCL halo exchange before the blur (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    from paintfe_amd import GpuRenderer
    r = GpuRenderer(dev_index)
    r.set_exact(args.exact)
    for kv in args.tune:
        k, v = kv.split('=')
        r.tune(k, int(v))
    r.set_stream(torch.cuda.current_stream().cuda_stream)  # HIP events + kernels on the stream torch synchronises

    w, h, n = args.width, args.height, args.layers
    band_mode = args.shard == "band" and world > 1
    state = {}
    if band_mode:
        # ONE document for the whole job, cut into bands of whole chunk rows (paintfe_amd/sharding.py): every rank
        # generates the same layers (same seeds) and keeps only its rows
        from paintfe_amd import sharding as S
        y0, y1 = S.band_rows(h, world, rank)
        hh = y1 - y0
        modes, opac = synth_params(n, 0x5EED0002)
        stack = torch.empty((n, hh, w, 4), dtype=torch.uint8, device=device)
        for k in range(n):
            stack[k] = synth_layer(torch, device, w, h, k, 0x5EED0002)[y0:y1]
    else:
        stack, modes, opac = synth_stack(torch, device, w, h, n, seed=0x5EED0002 + 1000 * rank)
        hh = h
    flat = torch.empty((max(hh, 1), w, 4), dtype=torch.uint8, device=device)
    info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
    ptrs = [stack[k].data_ptr() for k in range(n)]
    radius = int(np.ceil(np.float32(args.sigma) * np.float32(3.0)))
    pad_rows = hh + (2 * radius if band_mode else 0)
    blurred = torch.empty((max(pad_rows, 1), w, 4), dtype=torch.uint8, device=device)
    tmp = torch.empty((max(pad_rows, 1), w, 4), dtype=torch.float32, device=device)  # f32 horizontal-pass intermediate

    def step():
        if hh > 0:
            r.flatten_dev(ptrs, info, w, hh, flat.data_ptr())
        if band_mode:
            # the only exchange of the path: `radius` rows of the flattened u8 band from each neighbour (RCCL send/recv
            # on the current stream, so it is ordered after the flatten and before the blur without host syncs)
            padded, top, bottom = S.exchange_halo(flat[:hh], h, radius)
            if hh > 0:
                r.gaussian_blur_dev(padded.data_ptr(), blurred.data_ptr(), w, int(padded.shape[0]), args.sigma, tmp.data_ptr())
                state["result"] = blurred[top:top + hh]
        else:
            r.gaussian_blur_dev(flat.data_ptr(), blurred.data_ptr(), w, h, args.sigma, tmp.data_ptr())

    def bracket():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    bracket()
    r.timing_reset()
    r.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    bracket()
    elapsed = time.perf_counter() - t0
    r.timing_enable(False)
    if world > 1:
        from paintfe_amd.sharding import max_over_ranks
        elapsed = max_over_ranks(elapsed, device=device)  # the step time of the job is the slowest rank's

    px_per_step = w * h
    docs = 1 if band_mode else world                        # band mode: the whole job is one document per step
    value = px_per_step * args.steps * docs / elapsed / 1e6  # whole-job Mpx/s
    px_per_launch = w * hh                                   # pixels one flatten launch on this rank covers

    # per-kernel launch durations from HIP events recorded on the launch stream during the timed region
    kern = {}
    for name in ("flatten", "gauss_mfma", "gauss_h", "gauss_v"):
        ms, cnt = r.timing_read(name)
        if cnt:
            kern[name] = (ms / cnt, cnt)
    alg_bytes = {"flatten": (4 * n + 4) * px_per_launch}
    dominant = "flatten"  # carries 132 of the 140 algorithmic bytes/px; named in DESIGN.md
    d_ms = kern[dominant][0]
    achieved = alg_bytes[dominant] / (d_ms * 1e-3) / 1e9 if d_ms > 0 else 0.0
    pipeline_bytes = (4 * n + 4 + 8) * px_per_step
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dominant, w, h, n),
                "kernel_ms": {k: round(v[0], 4) for k, v in kern.items()},
                "pipeline_achieved_GBs": round(pipeline_bytes * args.steps / elapsed / 1e9, 1),
                "pipeline_frac": round(pipeline_bytes * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 4)}

    out = {"metric": "Mpixels/sec: 8K 32-layer flatten + Gaussian sigma=16; HBM GB/s vs peak", "value": round(value, 1),
           "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "strong" if band_mode else "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{w}x{h} RGBA8 x {n} layers (25 blend modes cycling, S2) flatten -> Gaussian sigma={args.sigma:g}",
                      "width": w, "height": h, "layers": n, "sigma": args.sigma, "gaussian_mode": "exact" if args.exact else "fma",
                      "sharding": ("one document cut into chunk-row bands, RCCL halo exchange before the blur" if band_mode else
                                   "one document per GPU, no collective") if world > 1 else "single GPU"},
           "roofline": roofline}

    if rank == 0:
        # correctness spot check of the timed result against the oracle on a crop (flatten is per-pixel, so a crop
        # of the full-size flatten equals the flatten of the cropped stack)
        from tests import oracle_lib as O
        ch, cw = 256, 512
        cy, cx = min(1000, max(hh - ch, 0)), min(2000, max(w - cw, 0))
        crop_stack = stack[:, cy:cy + ch, cx:cx + cw, :].contiguous().cpu().numpy() if hh >= cy + ch and w >= cx + cw else None
        if crop_stack is not None:
            ref = O.flatten_stack(crop_stack, modes, opac)
            got = flat[cy:cy + ch, cx:cx + cw, :].contiguous().cpu().numpy()
            out["check"] = {"flatten_crop_bitexact": bool(np.array_equal(ref, got))}
            if not out["check"]["flatten_crop_bitexact"]:
                out["check"]["mismatching_px"] = int((ref != got).any(-1).sum())

        if band_mode and w * h <= (1 << 22) and hh > 0:
            # small documents only: rank 0 rebuilds the whole document and checks ITS band of the sharded result (which
            # needed rank 1's halo rows) against the single-process oracle pipeline
            full = torch.stack([synth_layer(torch, device, w, h, k, 0x5EED0002) for k in range(n)]).cpu().numpy()
            ref_blur = O.gaussian_blur(O.flatten_stack(full, modes, opac), args.sigma)[y0:y1]
            got_blur = state["result"].contiguous().cpu().numpy()
            dmax = int(np.abs(ref_blur.astype(np.int16) - got_blur.astype(np.int16)).max())
            out.setdefault("check", {})["band_blur_max_diff_vs_single_process"] = dmax

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
                out.setdefault("check", {})["flatten_whole_frame_bitexact"] = bool(np.array_equal(f, flat.cpu().numpy()))
            out["cpu_baseline"] = {"value": round(sw * sh / (t3 - t1) / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port",
                                   "sample": f"{'whole ' + str(sw) + 'x' + str(sh) + ' frame' if whole else str(sw) + 'x' + str(sh) + ' window'} "
                                             f"of the same {n}-layer stack, flatten {t2 - t1:.2f}s + gaussian {t3 - t2:.2f}s, "
                                             f"OpenMP restatement of PaintFE's rayon CPU path (oracle/): chunk-parallel compositing with the "
                                             f"write-back parallel too (serial in the reference), row-parallel Gaussian passes"}
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
