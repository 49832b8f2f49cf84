#!/usr/bin/env python3
"""tools/lab/event_overhead.py — what the HIP events inside bench.py's timed region cost: 200 steps (flatten + Gaussian, 8K x 32 layers) timed by the host around
a synchronize, with (a) nothing else on the stream, (b) the library's per-kernel timers (two events per kernel), (c) also one mark event per step (bench.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h, n = 7680, 4320, 32
dev = torch.device("cuda", 0)
stack, modes, opac = bench.synth_stack(torch, dev, w, h, n, seed=0x5EED0002)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); blurred = torch.empty_like(flat)
ptrs = [stack[k].data_ptr() for k in range(n)]
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
def step():
    r.flatten_dev(ptrs, info, w, h, flat.data_ptr()); r.gaussian_blur_dev(flat.data_ptr(), blurred.data_ptr(), w, h, 16.0)
for _ in range(60): step()
K = 200
for rnd in range(2):
    for mode in ("plain", "kernel timers", "kernel timers + step marks"):
        r.timing_reset(); r.timing_enable(mode != "plain")
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K):
            step()
            if mode.endswith("marks"): marks[k].record()
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        r.timing_enable(False)
        print(f"{mode:28s} {el / K * 1e3:.4f} ms per step", flush=True)
