#!/usr/bin/env python3
"""tools/lab/band_units_sweep.py — the compositor's stream length (units per wave of the first stream class, pfx_tune dle_units; 0 = automatic) on the
band shapes of 2 / 4 / 8 GPUs (8K x 32 layers, S2), for the small-launch rule in pfxk_flatten"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from paintfe_amd import GpuRenderer
dev = torch.device("cuda", 0)
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, H, n = 7680, 4320, 32
modes, opac = B.synth_params(n, 0x5EED0002)
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
full = torch.empty((n, H, w, 4), dtype=torch.uint8, device=dev)
for k in range(n): full[k] = B.synth_layer(torch, dev, w, H, k, 0x5EED0002)
out = torch.empty((H, w, 4), dtype=torch.uint8, device=dev)
import time
for rows in (64, 256, 576, 1088, 2176, 4320):
    ptrs = [full[k].data_ptr() for k in range(n)]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.08:   # the clock needs tens of ms of load to settle
        r.flatten_dev(ptrs, info, w, rows, out.data_ptr())
    torch.cuda.synchronize()
    cand = (0, 1, 2, 3, 4, 6, 8, 12, 16, 20, 24, 32)
    best = {u: 1e9 for u in cand}
    for rep in range(3):
        for units in cand:
            r.tune("dle_units", units)
            for _ in range(5): r.flatten_dev(ptrs, info, w, rows, out.data_ptr())
            torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
            for _ in range(20): r.flatten_dev(ptrs, info, w, rows, out.data_ptr())
            torch.cuda.synchronize(); r.timing_enable(False)
            ms, c = r.timing_read("flatten")
            best[units] = min(best[units], ms / c)
    print(f"rows {rows:5d} ({rows * w // 192} units): " + "  ".join(f"{u}:{best[u]:.4f}" for u in cand), flush=True)
