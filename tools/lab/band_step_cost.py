#!/usr/bin/env python3
"""tools/lab/band_step_cost.py — what ONE rank of an N-GPU band pipeline costs per step WITHOUT its collectives, on one GPU:
flatten of a band of 8K rows / N (edge chunk rows first, as BandPipeline.step does) + Gaussian on band + 2 x 48 halo rows.
Prints the host enqueue time per step and the device time per step (HIP events), for the DESIGN 6 prediction."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from paintfe_amd import GpuRenderer

dev = torch.device("cuda", 0)
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, H, n, sigma, radius = 7680, 4320, 32, 16.0, 48
modes, opac = B.synth_params(n, 0x5EED0002)
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
for N in (1, 2, 4, 8):
    rows = 64 * ((H // 64 + N - 1) // N) if N > 1 else H
    rows = min(rows, H)
    stack = torch.empty((n, rows, w, 4), dtype=torch.uint8, device=dev)
    for k in range(n):
        stack[k] = B.synth_layer(torch, dev, w, H, k, 0x5EED0002)[:rows]
    top = bottom = radius if N > 1 else 0
    padded = torch.zeros((top + rows + bottom, w, 4), dtype=torch.uint8, device=dev)
    blurred = torch.zeros((top + rows + bottom + 1, w, 4), dtype=torch.uint8, device=dev)
    ptrs = [stack[k].data_ptr() for k in range(n)]
    rb = w * 4
    base = padded[top:].data_ptr()
    edge = 64

    def fl(r0, r1):
        r.flatten_dev([p + r0 * rb for p in ptrs], info, w, r1 - r0, base + r0 * rb)

    def step(split):
        if split and rows > 2 * edge:
            fl(0, edge); fl(rows - edge, rows); fl(edge, rows - edge)
        else:
            fl(0, rows)
        r.gaussian_blur_dev(padded.data_ptr(), blurred.data_ptr(), w, top + rows + bottom, sigma, first_row=64 if N > 1 else 0)

    for split in ((False,) if N == 1 else (True, False)):
        for _ in range(30): step(split)
        torch.cuda.synchronize()
        K = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(K): step(split)
        t1 = time.perf_counter(); e1.record()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r.timing_reset(); r.timing_enable(True)
        for _ in range(20): step(split)
        torch.cuda.synchronize(); r.timing_enable(False)
        km = {k: r.timing_read(k) for k in ("flatten", "gauss_mfma")}
        ks = ", ".join(f"{k} {ms / 20:.4f} ms/step in {int(c / 20)} launch(es)" for k, (ms, c) in km.items() if c)
        print(f"N={N} band {rows} rows split={split}: host enqueue {1e3*(t1-t0)/K:.4f} ms/step, device {e0.elapsed_time(e1)/K:.4f} ms/step, wall {1e3*(t2-t0)/K:.4f}; {ks}", flush=True)
    del stack, padded, blurred
    torch.cuda.empty_cache()
