#!/usr/bin/env python3
"""tools/lab/stream_floor.py — the memory-side floor of the 32-layer compositor: the S2 pixel data with every layer's mode set to the
cheapest blend functions, elimination off, so that the kernel does little more than stream 4.38 GB through the typed-load path.
Usage: python tools/lab/stream_floor.py [key=value ...]"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
r.tune("flatten_variant", 8)
for kv in sys.argv[1:]:
    k, v = kv.split("="); r.tune(k, int(v))
w, h, n = 7680, 4320, 32
dev = torch.device("cuda", 0)
stack, modes, opac = bench.synth_stack(torch, dev, w, h, n, seed=0x5EED0002)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
ptrs = [stack[k].data_ptr() for k in range(n)]
out = {}
for name, mode, op in (("overwrite", 14, 1.0), ("lighten", 11, 1.0), ("normal60", 0, 0.6), ("s2", -1, 0)):
    info = [(k, float(opac[k]) if mode < 0 else (1.0 if k == 0 else op), True, int(modes[k]) if mode < 0 else (0 if k == 0 else mode)) for k in range(n)]
    for _ in range(10): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(30): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_enable(False)
    ms = r.timing_read("flatten")[0] / 30
    out[name] = {"ms": round(ms, 4), "TBs": round((4 * n + 4) * w * h / ms / 1e9, 3)}
# plain torch reads of the same bytes for comparison: sum over the stack viewed as int32
s32 = stack.view(torch.int32)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): s32.sum()
e0.record()
for _ in range(10): s32.sum()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
out["torch_sum_read"] = {"ms": round(ms, 4), "TBs": round(4 * n * w * h / ms / 1e9, 3)}
print(json.dumps(out))
