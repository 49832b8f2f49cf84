#!/usr/bin/env python3
"""tools/time_mesh.py — mesh warp / liquify timings at 16K (BASELINE config 4), HIP events on the launch stream."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer
from tests import inputs as I
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
for kv in sys.argv[1:]:
    k, v = kv.split("="); r.tune(k, int(v))
w, h = 15360, 8640
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
orig, deformed = I.jittered_mesh(6, 6, w, h)
disp = torch.empty((h, w, 2), dtype=torch.float32, device="cuda")
r.mesh_displacement_dev(orig, deformed, 6, 6, w, h, disp.data_ptr())   # a smooth field (the mesh's own) for the liquify gather
cases = (("fused mesh warp 6x6", lambda: r.warp_mesh_catmull_rom_dev(src.data_ptr(), orig, deformed, 6, 6, w, h, dst.data_ptr()), "warp_mesh", 8),
         ("mesh displacement field", lambda: r.mesh_displacement_dev(orig, deformed, 6, 6, w, h, disp.data_ptr()), "mesh_displacement", 8),
         ("liquify displacement warp", lambda: r.warp_displacement_dev(src.data_ptr(), w, h, disp.data_ptr(), w, h, dst.data_ptr()), "warp_displacement", 16))
for name, fn, key, bpp in cases:
    for _ in range(3): fn()
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(10): fn()
    torch.cuda.synchronize(); r.timing_enable(False)
    ms = r.timing_read(key)[0] / 10
    print(f"{name}: {ms:.3f} ms, {w*h/ms/1e3:.0f} Mpx/s, {bpp*w*h/ms/1e6:.0f} GB/s algorithmic")
