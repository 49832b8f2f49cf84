#!/usr/bin/env python3
"""tools/lab/units_ab.py — full-frame compositor (8K x 32, S2): units per wave of the long streams (pfx_tune dle_units, 0 = automatic), candidates
alternated in a fresh random order per repetition on one box; prints min / median per candidate"""
import os, sys, time, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from paintfe_amd import GpuRenderer
dev = torch.device("cuda", 0)
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, H, n = 7680, 4320, 32
rows = int(sys.argv[1]) if len(sys.argv) > 1 else H
cand = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 6, 8, 10, 12, 14, 16, 18, 22]
modes, opac = B.synth_params(n, 0x5EED0002)
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
full = torch.empty((n, H, w, 4), dtype=torch.uint8, device=dev)
for k in range(n): full[k] = B.synth_layer(torch, dev, w, H, k, 0x5EED0002)
out = torch.empty((H, w, 4), dtype=torch.uint8, device=dev)
ptrs = [full[k].data_ptr() for k in range(n)]
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2: r.flatten_dev(ptrs, info, w, rows, out.data_ptr())
torch.cuda.synchronize()
res = {u: [] for u in cand}
random.seed(1)
for rep in range(7):
    order = cand[:]; random.shuffle(order)
    for units in order:
        r.tune("dle_units", units)
        for _ in range(15): r.flatten_dev(ptrs, info, w, rows, out.data_ptr())
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(30): r.flatten_dev(ptrs, info, w, rows, out.data_ptr())
        torch.cuda.synchronize(); r.timing_enable(False)
        ms, c = r.timing_read("flatten")
        res[units].append(ms / c)
for u in cand:
    v = sorted(res[u])
    print(f"rows {rows} units {u:3d}: min {v[0]:.4f}  median {v[len(v)//2]:.4f}  max {v[-1]:.4f}", flush=True)
