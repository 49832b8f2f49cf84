#!/usr/bin/env python3
"""tools/lab/mesh_dbg.py — where the LDS-gather mesh warp's time goes (16K): whole kernel, staging only, rows only, and how many wave-rows / pixels leave the window"""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paintfe_amd import GpuRenderer, _lib
from tests import inputs as I
lib = _lib.load()
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 15360, 8640
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
orig, deformed = I.jittered_mesh(6, 6, w, h)
def run(tag):
    fn = lambda: r.warp_mesh_catmull_rom_dev(src.data_ptr(), orig, deformed, 6, 6, w, h, dst.data_ptr())
    for _ in range(3): fn()
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(10): fn()
    torch.cuda.synchronize(); r.timing_enable(False)
    return r.timing_read("warp_mesh")[0] / 10
for sw in (80, 96, 112, 128):
    r.tune("warp_tile", sw)
    out = {}
    for dbg, name in ((0, "all"), (2, "staging_only"), (1, "rows_only")):
        r.tune("warp_dbg", dbg); out[name] = round(run(name), 4)
    r.tune("warp_dbg", 0)
    st = (C.c_ulonglong * 2)(); lib.pfxk_warp_fallback_stats(st, 1)
    r.warp_mesh_catmull_rom_dev(src.data_ptr(), orig, deformed, 6, 6, w, h, dst.data_ptr()); torch.cuda.synchronize()
    lib.pfxk_warp_fallback_stats(st, 1)
    out["fallback_wave_rows_frac"] = round(st[0] / (w * h / 64), 4); out["fallback_px_frac"] = round(st[1] / (w * h), 5)
    print(sw, out, flush=True)
