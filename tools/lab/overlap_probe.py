#!/usr/bin/env python3
"""tools/lab/overlap_probe.py — does an HBM-bound stream overlap with the compositor's issue-bound natural passes when both run at once?
Stream 1: flatten of S2 with layer 14's holes filled (no early passes: 18 three-pixel steps per unit, 0.71 ms, 3.5 TB/s).  Stream 2: `invert` over 16K images (pure
streaming, 8 B/px) sized to move what the early passes move (1.9 GB).  Alone, alone, together: together ~ max => a memory-bound early phase COULD hide under the natural
passes if it ran in waves of its own; together ~ sum => it could not."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from paintfe_amd import GpuRenderer
dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
r1, r2 = GpuRenderer(0), GpuRenderer(0)
r1.set_stream(s1.cuda_stream); r2.set_stream(s2.cuda_stream)
w, h, n = 7680, 4320, 32
stack, modes, opac = bench.synth_stack(torch, dev, w, h, n, seed=0x5EED0002)
a14 = stack[14, ..., 3]
stack[14, ..., 3] = torch.where(a14 == 0, torch.full_like(a14, 255), a14)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
ptrs = [stack[k].data_ptr() for k in range(n)]
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
W2, H2 = 15360, 8640
big = torch.randint(0, 256, (H2, W2, 4), dtype=torch.uint8, device=dev); big2 = torch.empty_like(big)
reps_inv = int(sys.argv[1]) if len(sys.argv) > 1 else 2     # 2 x 1.06 GB


def A():
    r1.flatten_dev(ptrs, info, w, h, flat.data_ptr())


def B():
    for _ in range(reps_inv): r2.adjust_dev(big.data_ptr(), big2.data_ptr(), W2, H2, "invert")


def timed(fs, iters=20):
    for f in fs: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    s1.wait_event(e0); s2.wait_event(e0)
    for _ in range(iters):
        for f in fs: f()
    d1, d2 = torch.cuda.Event(), torch.cuda.Event()
    d1.record(s1); d2.record(s2)
    torch.cuda.current_stream().wait_event(d1); torch.cuda.current_stream().wait_event(d2)
    e1.record(torch.cuda.current_stream()); e1.synchronize()
    return e0.elapsed_time(e1) / iters


for _ in range(3): A(); B()
tA, tB, tAB = timed([A]), timed([B]), timed([A, B])
print(json.dumps({"flatten_natural_only_ms": round(tA, 4), "streaming_invert_ms": round(tB, 4), "invert_GB": round(reps_inv * W2 * H2 * 8 / 1e9, 2), "together_ms": round(tAB, 4),
                  "sum_ms": round(tA + tB, 4), "max_ms": round(max(tA, tB), 4)}))
