#!/usr/bin/env python3
"""tools/lab/overlap_step.py — what-if: bench.py's step with the Gaussian of document k on a second stream, overlapping the flatten of
document k + 1 (two flat / blurred buffers).  A Gaussian workgroup needs 113 KB of LDS and 2 x 200 VGPRs per SIMD, a flatten wave 80
VGPRs: they can share a CU only at one flatten wave per SIMD, so the question is what the dispatcher makes of it."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from paintfe_amd import GpuRenderer
dev = torch.device("cuda", 0)
r = GpuRenderer(0)
w, h, n, sigma = 7680, 4320, 32, 16.0
stack, modes, opac = B.synth_stack(torch, dev, w, h, n, seed=0x5EED0002)
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
ptrs = [stack[k].data_ptr() for k in range(n)]
flat = [torch.empty((h, w, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
blur = [torch.empty((h, w, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
main = torch.cuda.current_stream()
def serial(steps):
    r.set_stream(main.cuda_stream)
    for k in range(steps):
        r.flatten_dev(ptrs, info, w, h, flat[0].data_ptr()); r.gaussian_blur_dev(flat[0].data_ptr(), blur[0].data_ptr(), w, h, sigma)
for prio in (0, -1):
    s_f, s_g = torch.cuda.Stream(), torch.cuda.Stream(priority=prio)
    ev_f = [torch.cuda.Event() for _ in range(2)]; ev_g = [torch.cuda.Event() for _ in range(2)]
    def overlapped(steps):
        for k in range(steps):
            b = k & 1
            s_f.wait_event(ev_g[b])
            r.set_stream(s_f.cuda_stream); r.flatten_dev(ptrs, info, w, h, flat[b].data_ptr()); ev_f[b].record(s_f)
            s_g.wait_event(ev_f[b])
            r.set_stream(s_g.cuda_stream); r.gaussian_blur_dev(flat[b].data_ptr(), blur[b].data_ptr(), w, h, sigma); ev_g[b].record(s_g)
    for name, fn in (("serial", serial), (f"overlapped (gaussian stream priority {prio})", overlapped), ("serial", serial)):
        fn(5); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(40); torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / 40 * 1e3:.4f} ms/step")
ref_f, ref_b = flat[0].clone(), blur[0].clone()
overlapped(4); torch.cuda.synchronize()
print("results equal:", bool((flat[1] == ref_f).all() and (blur[1] == ref_b).all()))
