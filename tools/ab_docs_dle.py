#!/usr/bin/env python3
"""tools/ab_docs_dle.py — does the dead-layer elimination pay on stacks other than S2?  9-layer 8K stacks (S2 pixel data: alpha 25 % 0, 25 % 255,
50 % in between; opacity 1.0 on even layers) with every layer in one mode, elimination on (default) against off (flatten_variant=8), plus two
document-like stacks: an opaque photo in the middle of the stack, and nine Normal layers at 60 %."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h, n = 7680, 4320, 9
dev = torch.device("cuda", 0)
stack, _, opac = bench.synth_stack(torch, dev, w, h, n, seed=0x5EED0002)
photo = stack.clone(); photo[4, ..., 3] = 255
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
def run(st, info):
    ptrs = [st[k].data_ptr() for k in range(n)]
    out = []
    for v in (0, 8):
        r.tune("flatten_variant", v)
        for _ in range(10): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(30): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
        torch.cuda.synchronize(); r.timing_enable(False)
        out.append(round(r.timing_read("flatten")[0] / 30, 4))
    r.tune("flatten_variant", 0)
    return out
for kv in sys.argv[1:]:
    k, v = kv.split("="); r.tune(k, int(v))
for _ in range(150): r.flatten_dev([stack[k].data_ptr() for k in range(n)], [(k, 1.0, True, 1) for k in range(n)], w, h, flat.data_ptr())  # clocks settle
torch.cuda.synchronize()
res = {}
for mode, name in ((0, "normal"), (1, "multiply"), (14, "overwrite")):
    res[f"all {name}, S2 opacities"] = run(stack, [(k, float(opac[k]), True, 0 if k == 0 else mode) for k in range(n)])
res["opaque photo at layer 4 (Normal 100 %), S2 modes"] = run(photo, [(k, 1.0 if k == 4 else float(opac[k]), True, 0 if k in (0, 4) else (k % 25)) for k in range(n)])
res["nine Normal layers at 60 %"] = run(stack, [(k, 1.0 if k == 0 else 0.6, True, 0) for k in range(n)])
for k, v in res.items():
    print(f"{k:55s} elimination {v[0]:.4f} ms   off {v[1]:.4f} ms   ratio {v[0] / v[1]:.3f}")
