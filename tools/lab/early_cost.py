#!/usr/bin/env python3
"""tools/lab/early_cost.py — what the early groups' one-pixel-per-lane passes cost the class-sorting compositor on the bench stack (8K x 32 layers, S2).
Layer 14 (Overwrite) is S2's reset layer: where its alpha is 0 (25 % of the pixels) a pixel is EARLY and runs layers [0, 14) in a 64-lane group of its own.
  A  S2 as generated                                  : classification + 14 one-pixel-per-lane steps + 18 three-pixel steps per unit
  B  layer 14 without holes (alpha 0 -> 255)          : classification + 18 three-pixel steps (no early pixel anywhere)
  C  layer 14 all holes (alpha = 0 everywhere)        : classification + 32 three-pixel steps (every pixel early: the unit is not split)
A - B = the early passes; (C - B) / 3 = what 14 layers on a third of the lanes' pixels would cost at the natural pass's rate."""
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


def timed(reps=40):
    for _ in range(8): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(reps): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_enable(False)
    return r.timing_read("flatten")[0] / reps


a14 = stack[14, ..., 3].clone()
A = timed()
stack[14, ..., 3] = torch.where(a14 == 0, torch.full_like(a14, 255), a14)
B = timed()
stack[14, ..., 3] = 0
C = timed()
stack[14, ..., 3] = a14
A2 = timed()
print(json.dumps({"args": sys.argv[1:], "A_s2_ms": round(A, 4), "A_again_ms": round(A2, 4), "B_no_early_ms": round(B, 4), "C_unsplit_32_layers_ms": round(C, 4),
                  "early_passes_ms": round((A + A2) / 2 - B, 4), "same_work_at_natural_rate_ms": round((C - B) / 3, 4),
                  "per_natural_step_us_per_launch": round((C - B) / 14 * 1e3, 2)}))
