#!/usr/bin/env python3
"""tools/lab/srt_phases.py — where a wave of the class-sorting compositor spends its clock on the bench stack (8K x 32 layers, S2).
The diagnostic build (pfx_tune "dle_stats" = 4) brackets the phases of every unit and every layer's wait / blend with s_memtime; this prints the sums as
fractions of the waves' lifetimes, next to the launch time of the diagnostic and of the shipped build.  (rocprofv3 --att cannot run in this image: no
rocprof-trace-decoder library.)  Usage: python tools/lab/srt_phases.py [mode=M] [key=value ...]"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
one_mode = None
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "mode": one_mode = int(v)
    else: r.tune(k, int(v))
w, h, n = 7680, 4320, 32
dev = torch.device("cuda", 0)
stack, modes, opac = bench.synth_stack(torch, dev, w, h, n, seed=0x5EED0002)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
if one_mode is not None:
    modes = [int(modes[k]) if k in (0, 14) else one_mode for k in range(n)]
ptrs = [stack[k].data_ptr() for k in range(n)]
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]


def timed(reps):
    for _ in range(5): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(reps): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_enable(False)
    return r.timing_read("flatten")[0] / reps


ms_ship = timed(30)
ref = flat.clone()
r.tune("dle_stats", 4)
ms_diag = timed(10)
same = bool(torch.equal(ref, flat))
r.flatten_trace(reset=True)
r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
t = r.flatten_trace(reset=True)
r.tune("dle_stats", 0)
life = max(t["wave_clocks"], 1)
out = {"args": sys.argv[1:], "shipped_ms": round(ms_ship, 4), "diagnostic_ms": round(ms_diag, 4), "diagnostic_bitexact": same, "waves": t["waves"],
       "mean_wave_lifetime_clocks": round(life / max(t["waves"], 1)),
       "fraction_of_wave_lifetime": {k: round(t[k] / life, 4) for k in ("classify", "deal_early", "natural", "store")},
       "inside_early_passes": {"wait_for_pixels": round(t["early_wait"] / life, 4), "blend": round(t["early_blend"] / life, 4)},
       "inside_natural_passes": {"wait_for_pixels": round(t["natural_wait"] / life, 4), "blend": round(t["natural_blend"] / life, 4)}}
print(json.dumps(out))
