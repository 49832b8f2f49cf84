#!/usr/bin/env python3
"""wall time per C-ABI call against the kernels' own time at interactive sizes (1080p): what the host side of a call costs"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
w, h, n = 1920, 1080, 9
stack, modes, opac = bench.synth_stack(torch, dev, w, h, n, seed=1)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); out = torch.empty_like(flat)
ptrs = [stack[k].data_ptr() for k in range(n)]
info = [(k, 0.6 if k else 1.0, True, 0) for k in range(n)]
def wall(fn, timer, reps=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    r.timing_enable(False)
    return dt, sum(r.timing_read(t)[0] for t in timer) / reps
for name, fn, timer in (("flatten 9 layers", lambda: r.flatten_dev(ptrs, info, w, h, flat.data_ptr()), ["flatten"]),
                        ("gaussian sigma 4", lambda: r.gaussian_blur_dev(flat.data_ptr(), out.data_ptr(), w, h, 4.0), ["gauss_mfma"]),
                        ("hsl", lambda: r.adjust_dev(flat.data_ptr(), out.data_ptr(), w, h, "hsl", [30.0, -20.0, 10.0]), ["adjust"]),
                        ("median r=3", lambda: r.median_dev(flat.data_ptr(), out.data_ptr(), w, h, 3), ["median"]),
                        ("box blur r=9", lambda: r.box_blur_dev(flat.data_ptr(), out.data_ptr(), w, h, 9.0), ["box_blur"])):
    a, k = wall(fn, timer)
    print(f"{name:18s} wall per call {a:.4f} ms   kernels {k:.4f} ms (timing on)")
    r.timing_enable(False)
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): fn()
    torch.cuda.synchronize()
    print(f"{'':18s} wall per call {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms   (timing off)")
