#!/usr/bin/env python3
"""tools/lab/shallow_modes_ab.py — streaming compositor shapes on 9- and 12-layer 8K stacks, every blend mode (S2 data and opacities): flatten_variant 0 (shipped) against the candidates"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
dev = torch.device("cuda", 0)
cands = [int(a) for a in sys.argv[1:]] or [0, 13]
stack, _, opac = bench.synth_stack(torch, dev, w, h, 12, seed=0x5EED0002)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
def t(n, info):
    ptrs = [stack[k].data_ptr() for k in range(n)]
    for _ in range(6): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(20): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_enable(False)
    return r.timing_read("flatten")[0] / 20
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1: t(9, [(k, 1.0, True, 1) for k in range(9)])
tot = {(n, v): 0.0 for n in (9, 12) for v in cands}
for n in (9, 12):
    for mode in range(25):
        if mode in (0, 14) and n == 12: pass
        info = [(k, float(opac[k]), True, 0 if k == 0 else mode) for k in range(n)]
        # mode 0 / 14 at opacity 1 would create reset candidates; those stacks are below dle_min_layers anyway
        best = {v: 1e9 for v in cands}
        for rep in range(2):
            for v in (cands if rep == 0 else cands[::-1]):
                r.tune("flatten_variant", v); best[v] = min(best[v], t(n, info))
        for v in cands: tot[(n, v)] += best[v]
        print(f"{n} layers mode {mode:2d}: " + "  ".join(f"v{v}:{best[v]:.4f}" for v in cands), flush=True)
r.tune("flatten_variant", 0)
for n in (9, 12): print(f"SUM {n} layers: " + "  ".join(f"v{v}:{tot[(n, v)]:.4f}" for v in cands))
