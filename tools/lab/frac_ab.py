#!/usr/bin/env python3
"""tools/lab/frac_ab.py [rows] — share of the units in the long (A) and quarter-length (B) streams of the class-sorting compositor (pfx_tune dle_frac_a / dle_frac_b;
the rest runs as single units), randomised order on one box"""
import os, sys, time, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from paintfe_amd import GpuRenderer
dev = torch.device("cuda", 0)
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, H, n = 7680, 4320, 32
rows = int(sys.argv[1]) if len(sys.argv) > 1 else H
cand = [(75, 20), (60, 30), (85, 12), (90, 8), (95, 4), (97, 3), (100, 0), (50, 40), (85, 15)]
modes, opac = B.synth_params(n, 0x5EED0002)
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
full = torch.empty((n, H, w, 4), dtype=torch.uint8, device=dev)
for k in range(n): full[k] = B.synth_layer(torch, dev, w, H, k, 0x5EED0002)
out = torch.empty((H, w, 4), dtype=torch.uint8, device=dev)
ptrs = [full[k].data_ptr() for k in range(n)]
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2: r.flatten_dev(ptrs, info, w, rows, out.data_ptr())
torch.cuda.synchronize()
res = {c: [] for c in cand}
random.seed(2)
for rep in range(6):
    order = cand[:]; random.shuffle(order)
    for (fa, fb) in order:
        r.tune("dle_frac_b", 0); r.tune("dle_frac_a", fa); r.tune("dle_frac_b", fb)
        for _ in range(15): r.flatten_dev(ptrs, info, w, rows, out.data_ptr())
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(30): r.flatten_dev(ptrs, info, w, rows, out.data_ptr())
        torch.cuda.synchronize(); r.timing_enable(False)
        ms, c = r.timing_read("flatten")
        res[(fa, fb)].append(ms / c)
for c in cand:
    v = sorted(res[c])
    print(f"rows {rows} fracA/fracB {c[0]:3d}/{c[1]:2d}: min {v[0]:.4f}  median {v[len(v)//2]:.4f}  max {v[-1]:.4f}", flush=True)
