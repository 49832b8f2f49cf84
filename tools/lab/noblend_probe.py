#!/usr/bin/env python3
"""tools/lab/noblend_probe.py — the elimination kernel's load stream without its arithmetic (pfx_tune dle_stats=2: every blend replaced by
12 adds; results are garbage): what the memory side alone costs for the S2 access pattern (rounds are gathers).  [key=value ...]"""
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
ptrs = [stack[k].data_ptr() for k in range(n)]
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
out = {"args": sys.argv[1:]}
for name, flag in (("full", 0), ("noblend", 2)):
    r.tune("dle_stats", flag)
    for _ in range(10): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(30): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_enable(False)
    out[name + "_ms"] = round(r.timing_read("flatten")[0] / 30, 4)
r.tune("dle_stats", 0)
print(json.dumps(out))
