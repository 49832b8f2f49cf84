#!/usr/bin/env python3
"""tools/lab/pcie_duplex.py — what the host link gives one GPU: H2D alone, D2H alone, both at once (33 MB pieces, pinned memory,
two streams).  The ceiling for the batch pipeline (config 5)."""
import time, torch
n = 3840 * 2160 * 4
reps = 60
h_in = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
h_out = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
d_in = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
d_out = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(h2d, d2h):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(reps):
        if h2d:
            with torch.cuda.stream(s1): d_in[k % 4].copy_(h_in[k % 4], non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2): h_out[k % 4].copy_(d_out[k % 4], non_blocking=True)
    torch.cuda.synchronize(); return time.perf_counter() - t0
for name, a, b in (("h2d only", 1, 0), ("d2h only", 0, 1), ("both", 1, 1)):
    run(a, b); t = run(a, b)
    print(f"{name}: {reps * n / t / 1e9:.1f} GB/s per direction ({reps / t:.0f} images/s)")
