#!/usr/bin/env python3
"""tools/ab_flatten.py — per-mode 9-layer flatten at 8K for the library named by PFX_LIB_PATH (A/B timing of two builds)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h, n = 7680, 4320, 9
dev = torch.device("cuda", 0)
stack, _, opac = bench.synth_stack(torch, dev, w, h, n, seed=0x5EED0002)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
ptrs = [stack[k].data_ptr() for k in range(n)]
info0 = [(k, float(opac[k]), True, k % 25) for k in range(n)]
for _ in range(40): r.flatten_dev(ptrs, info0, w, h, flat.data_ptr())  # clock ramp-up before the first timed mode
torch.cuda.synchronize()
out = []
for mode in [int(a) for a in sys.argv[1:]] or [0, 1, 3, 8, 14, 16]:
    info = [(k, float(opac[k]), True, 0 if k == 0 else mode) for k in range(n)]
    for _ in range(5): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(30): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
    torch.cuda.synchronize(); r.timing_enable(False)
    out.append(f"m{mode}:{r.timing_read('flatten')[0] / 30:.4f}")
print(os.path.basename(os.environ.get("PFX_LIB_PATH", "libpfx.so")), " ".join(out))
