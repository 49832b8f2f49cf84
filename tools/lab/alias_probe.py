#!/usr/bin/env python3
"""tools/lab/alias_probe.py — how much of the compositor's time is memory latency?  The S2 layer table (modes, opacities) is run over
(a) 32 distinct 8K layers (HBM stream) and (b) 32 descriptors that all point at layer 1's buffer, so that every load after a tile's
first one hits L1 / L2: the arithmetic per lane is the same branch-free code, only the load latency changes.
Usage: python tools/lab/alias_probe.py [key=value ...]"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
for kv in sys.argv[1:]:
    k, v = kv.split("="); r.tune(k, int(v))
w, h, n = 7680, 4320, 32
dev = torch.device("cuda", 0)
stack, modes, opac = bench.synth_stack(torch, dev, w, h, n, seed=0x5EED0002)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
def run(ptrs):
    for _ in range(20): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(50): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_enable(False)
    return round(r.timing_read("flatten")[0] / 50, 4)
out = {"args": sys.argv[1:]}
out["distinct_layers_ms"] = run([stack[k].data_ptr() for k in range(n)])
out["aliased_layers_ms"] = run([stack[0].data_ptr()] + [stack[1].data_ptr()] * (n - 1))
print(json.dumps(out))
