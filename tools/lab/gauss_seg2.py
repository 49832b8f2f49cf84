import os, sys, torch
sys.path.insert(0, os.getcwd())
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
for s in (4.0, 10.0, 16.0, 24.0):
    out = []
    for seg in (0, 1, 2, 3, 4):
        r.tune("gauss_mfma_segments", seg)
        for _ in range(5): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, s)
        ts = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, s); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        ts.sort(); out.append(f"seg{seg} {ts[len(ts)//2]:.4f}")
    print(f"sigma {s}: " + " | ".join(out), flush=True)
