#!/usr/bin/env python3
"""tools/lab/dle_depth_ab.py — S2 stacks of 15 .. 32 layers at 8K: the class-sorting kernel (dead-layer elimination on: flatten_variant 0, dle_min_layers lowered to 8) against
the streaming kernel (flatten_variant 8), alternated — where should the switch (dle_min_layers, 16) sit?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from paintfe_amd import GpuRenderer
dev = torch.device("cuda", 0)
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
r.tune("dle_min_layers", 8)
w, H, N = 7680, 4320, 32
full = torch.empty((N, H, w, 4), dtype=torch.uint8, device=dev)
for k in range(N): full[k] = B.synth_layer(torch, dev, w, H, k, 0x5EED0002)
out = torch.empty((H, w, 4), dtype=torch.uint8, device=dev)
modes, opac = B.synth_params(N, 0x5EED0002)
def t(n):
    info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
    ptrs = [full[k].data_ptr() for k in range(n)]
    for _ in range(8): r.flatten_dev(ptrs, info, w, H, out.data_ptr())
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(20): r.flatten_dev(ptrs, info, w, H, out.data_ptr())
    torch.cuda.synchronize(); r.timing_enable(False)
    return r.timing_read("flatten")[0] / 20
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2: t(32)
for n in (15, 16, 18, 20, 24, 26, 28, 32):
    best = {0: 1e9, 8: 1e9}
    for rep in range(3):
        for v in ((0, 8) if rep % 2 == 0 else (8, 0)):
            r.tune("flatten_variant", v); best[v] = min(best[v], t(n))
    r.tune("flatten_variant", 0)
    print(f"{n} layers: class-sorting {best[0]:.4f}  streaming {best[8]:.4f}  ratio {best[0] / best[8]:.3f}", flush=True)
