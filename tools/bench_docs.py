#!/usr/bin/env python3
"""tools/bench_docs.py — compositor throughput on document shapes other than the S2 stress stack (8K, 9 layers):
'photo' = opaque layers, Normal at 100 % (wave-uniform opaque fast path: the top pixel is the result);
'graded' = opaque background + opaque Multiply / Screen / Overlay layers at 100 % (opaque accumulator, opaque layer);
'soft' = opaque background + layers with smooth alpha and opacity 0.6 (opaque accumulator, general layer)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h, n = 7680, 4320, 9
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
stack = torch.randint(0, 256, (n, h, w, 4), dtype=torch.uint8, device=dev, generator=g)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
ptrs = [stack[k].data_ptr() for k in range(n)]
def run(info, label):
    for _ in range(10): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(30): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_enable(False)
    ms = r.timing_read("flatten")[0] / 30
    print(f"{label:8s} {ms:.4f} ms  {(4 * n + 4) * w * h / ms / 1e9:.2f} TB/s")
soft_alpha = stack[:, :, :, 3].clone()
stack[:, :, :, 3] = 255
run([(k, 1.0, True, 0) for k in range(n)], "photo")
run([(k, 1.0, True, [0, 1, 2, 8][k % 4] if k else 0) for k in range(n)], "graded")
stack[1:, :, :, 3] = torch.clamp(soft_alpha[1:], 1, 254)
run([(k, 1.0 if k == 0 else 0.6, True, [0, 1, 2, 8][k % 4] if k else 0) for k in range(n)], "soft")
# the general kernel: a live layer mask, and an adjustment layer (chunk existence becomes observable: the chunk-flag pre-pass runs)
stack[:, :, :, 3] = 255
mask = torch.randint(0, 256, (h, w), dtype=torch.uint8, device=dev, generator=g)
def run_general(info, label, mask_ptrs=None):
    for _ in range(5): r.flatten_dev(ptrs, info, w, h, flat.data_ptr(), mask_ptrs)
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(20): r.flatten_dev(ptrs, info, w, h, flat.data_ptr(), mask_ptrs)
    torch.cuda.synchronize(); r.timing_enable(False)
    ms = r.timing_read("flatten")[0] / 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): r.flatten_dev(ptrs, info, w, h, flat.data_ptr(), mask_ptrs)
    e1.record(); e1.synchronize()
    print(f"{label:8s} {ms:.4f} ms kernel, {e0.elapsed_time(e1) / 20:.4f} ms per call (with the pre-pass)  {(4 * n + 4) * w * h / ms / 1e9:.2f} TB/s")
run_general([(k, 1.0, True, [0, 1, 2, 8][k % 4] if k else 0) for k in range(n)], "masked", [0, 0, 0, mask.data_ptr()] + [0] * (n - 4))
run_general([(k, 1.0, True, [0, 1, 2, 8][k % 4] if k else 0) if k != 4 else (k, 0.7, True, 0, 1, [0.3]) for k in range(n)], "adjusted")
