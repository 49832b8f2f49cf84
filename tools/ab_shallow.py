#!/usr/bin/env python3
"""tools/ab_shallow.py — the streaming compositor on shallow stacks (most documents): 8K, 9 and 4 layers, launch shapes / register shapes
(pfx_tune flatten_variant: 0 shipped, 1-5 pixels per lane x register sets, +10 grid-stride instead of one tile per wave)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
dev = torch.device("cuda", 0)
NL = int(os.environ.get('AB_LAYERS', '9'))
stack, _, opac = bench.synth_stack(torch, dev, w, h, NL, seed=0x5EED0002)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
variants = [int(a) for a in sys.argv[1:]] or [0, 10, 1, 11, 2, 12, 3, 13, 4, 14]
def run(n, info):
    ptrs = [stack[k].data_ptr() for k in range(n)]
    out = {}
    for v in variants:
        r.tune("flatten_variant", v)
        for _ in range(10): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(30): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
        torch.cuda.synchronize(); r.timing_enable(False)
        out[v] = round(r.timing_read("flatten")[0] / 30, 4)
    r.tune("flatten_variant", 0)
    return out
for _ in range(150): r.flatten_dev([stack[k].data_ptr() for k in range(9)], [(k, 1.0, True, 1) for k in range(9)], w, h, flat.data_ptr())
torch.cuda.synchronize()
for n in ((9, 7, 6, 5, 4, 3, 2) if NL == 9 else (NL, 24, 16, 12)):
    for mode, name in ((1, "multiply"),):
        res = run(n, [(k, 1.0 if k == 0 else 0.6, True, 0 if k == 0 else mode) for k in range(n)])
        if os.environ.get("AB_NOBG"):  # no opaque background: translucent accumulators, the general (division) path on every layer
            ptrs = [stack[k].data_ptr() for k in range(1, n)]
            def run2(info):
                out = {}
                for v in variants:
                    r.tune("flatten_variant", v)
                    for _ in range(10): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
                    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
                    for _ in range(30): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
                    torch.cuda.synchronize(); r.timing_enable(False)
                    out[v] = round(r.timing_read("flatten")[0] / 30, 4)
                r.tune("flatten_variant", 0)
                return out
            res = run2([(k, 0.6, True, mode) for k in range(n - 1)])
            name = name + " no-bg"
        print(f"{n} layers {name:9s} " + "  ".join(f"v{v}: {t:.4f}" for v, t in res.items()) + f"   (copy-rate floor {(4 * n + 4) * w * h / 5.3e12 * 1e3:.3f} ms)")
