#!/usr/bin/env python3
"""tools/lab/icache_ab.py — does cycling through 25 blend modes cost the class-sorting compositor anything beyond the modes' own arithmetic?
(VERDICT r04 #1a: 266 KB of kernel code against a 64 KB instruction cache.)

S2's pixel data and opacities throughout; layers 0 and 14 keep S2's modes (Normal / Overwrite: the reset layers, so dead-layer elimination does the
same work in every run).  Run A: every other layer blends with ONE mode M, for each M.  Run B: S2's own cycle (mode k mod 25).  If the instruction
stream's footprint were free, B = sum over layers of w_l * T(mode_l) / sum w_l, with w_l the fraction of pixels that run layer l (0.25 below layer 14 —
the pixels layer 14 does not reset — and 1 above).  The gap between that prediction and B's measured time is what mode cycling costs."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from paintfe_amd import GpuRenderer

r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h, n = 7680, 4320, 32
dev = torch.device("cuda", 0)
stack, modes, opac = bench.synth_stack(torch, dev, w, h, n, seed=0x5EED0002)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
ptrs = [stack[k].data_ptr() for k in range(n)]


def run(mode_of, reps=40, opacity_of=lambda k: float(opac[k])):
    info = [(k, opacity_of(k), True, int(mode_of(k))) for k in range(n)]
    for _ in range(8): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(reps): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_enable(False)
    return r.timing_read("flatten")[0] / reps


s2 = lambda k: int(modes[k])
T = {}
mix = [run(s2)]
for M in range(25):
    if M == 14: continue   # Overwrite on every layer is a stack of reset layers: another workload (S2 holds it once, at layer 14, which every run keeps)
    # Normal at 100 % is a reset layer too: S2's only other Normal layer (25) sits at an odd index = opacity < 1, so the all-Normal run does the same
    op = (lambda k: float(opac[k]) if (k % 2 == 1 or k in (0, 14)) else 0.75) if M == 0 else (lambda k: float(opac[k]))
    T[M] = run(lambda k: s2(k) if k in (0, 14) else M, opacity_of=op)
    if M % 8 == 7: mix.append(run(s2))
mix.append(run(s2))
wl = {k: (0.25 if k < 14 else 1.0) for k in range(n) if k not in (0, 14)}
pred = sum(wl[k] * T[s2(k)] for k in wl) / sum(wl.values())
print(json.dumps({"single_mode_ms": {str(m): round(t, 4) for m, t in T.items()}, "s2_mix_ms": [round(t, 4) for t in mix],
                  "s2_predicted_from_single_modes_ms": round(pred, 4), "mean_single_ms": round(sum(T.values()) / len(T), 4)}))
