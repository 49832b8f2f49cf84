#!/usr/bin/env python3
"""tools/lab/tune_ab.py <settings> ... — full-frame compositor (8K x 32, S2) under pfx_tune settings ("key=v,key=v" per candidate; "-" = defaults), candidates in a fresh random
order per repetition on one box; every candidate resets the keys the others touch to the given defaults first (DEFAULTS below)"""
import os, sys, time, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from paintfe_amd import GpuRenderer
DEFAULTS = {"dle_cfg": 0, "dle_s1": -1, "dle_s2": -1, "dle_units": 0}
dev = torch.device("cuda", 0)
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, H, n = 7680, 4320, 32
cands = sys.argv[1:] or ["-"]
modes, opac = B.synth_params(n, 0x5EED0002)
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
full = torch.empty((n, H, w, 4), dtype=torch.uint8, device=dev)
for k in range(n): full[k] = B.synth_layer(torch, dev, w, H, k, 0x5EED0002)
out = torch.empty((H, w, 4), dtype=torch.uint8, device=dev)
ptrs = [full[k].data_ptr() for k in range(n)]
def apply(c):
    for k, v in DEFAULTS.items(): r.tune(k, v)
    if c != "-":
        for kv in c.split(","):
            k, v = kv.split("="); r.tune(k, int(v))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2: r.flatten_dev(ptrs, info, w, H, out.data_ptr())
torch.cuda.synchronize()
res = {c: [] for c in cands}
random.seed(4)
for rep in range(6):
    order = cands[:]; random.shuffle(order)
    for c in order:
        apply(c)
        for _ in range(15): r.flatten_dev(ptrs, info, w, H, out.data_ptr())
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(30): r.flatten_dev(ptrs, info, w, H, out.data_ptr())
        torch.cuda.synchronize(); r.timing_enable(False)
        ms, cnt = r.timing_read("flatten")
        res[c].append(ms / cnt)
for c in cands:
    v = sorted(res[c]); print(f"{c:28s} min {v[0]:.4f}  median {v[len(v)//2]:.4f}  max {v[-1]:.4f}", flush=True)
