#!/usr/bin/env python3
"""tools/dle_stats.py — work counters and timing of the compositor's dead-layer elimination on the bench stack (8K x 32 layers, S2).
Usage: python tools/dle_stats.py [key=value ...]   (pfx_tune knobs: dle_units, dle_ring, flatten_variant; mode=M: one blend mode on layers 1-13, 15-31)"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
one_mode = None
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "mode": one_mode = int(v)   # every layer but 0 and 14 (S2's reset layers) blends with this one mode: same elimination, one mode's code
    else: r.tune(k, int(v))
w, h, n = 7680, 4320, 32
dev = torch.device("cuda", 0)
stack, modes, opac = bench.synth_stack(torch, dev, w, h, n, seed=0x5EED0002)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
if one_mode is not None:
    modes = [int(modes[k]) if k in (0, 14) else one_mode for k in range(n)]
ptrs = [stack[k].data_ptr() for k in range(n)]
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
for _ in range(20): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
r.flatten_stats(reset=True)
r.tune("dle_stats", 1)
r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
st = r.flatten_stats(reset=True)
r.tune("dle_stats", 0)
torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
for _ in range(50): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
torch.cuda.synchronize(); r.timing_enable(False)
ms = r.timing_read("flatten")[0] / 50
units = (w * h + 191) // 192
kernel0 = not any(a == "dle_kernel=1" for a in sys.argv[1:])   # class-sorting kernel: "rounds" are early GROUPS (a third of a unit each)
st["layer_units_run"] = (st["round_layers"] / 3 if kernel0 else st["round_layers"]) + st["nat_layers"]
st["layer_units_full"] = units * n
st["work_fraction"] = round(st["layer_units_run"] / st["layer_units_full"], 4)
st["round_fill"] = round(st["round_px"] / max(st["rounds"] * (64 if kernel0 else 192), 1), 4)
print(json.dumps({"args": sys.argv[1:], "flatten_ms": round(ms, 4), **st}))
