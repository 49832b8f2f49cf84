import os, sys, torch
sys.path.insert(0, "/root/repo")
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
flat = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); blurred = torch.empty_like(flat); hsl = torch.empty_like(flat)
hp = (30.0, -20.0, 10.0)
def t(fn, names):
    for _ in range(5): fn()
    torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
    for _ in range(20): fn()
    torch.cuda.synchronize(); r.timing_enable(False)
    return {n: round(r.timing_read(n)[0] / 20, 4) for n in names}
def oop(): r.gaussian_blur_dev(flat.data_ptr(), blurred.data_ptr(), w, h, 16.0); r.adjust_dev(blurred.data_ptr(), hsl.data_ptr(), w, h, "hsl", hp)
def inp(): r.gaussian_blur_dev(flat.data_ptr(), blurred.data_ptr(), w, h, 16.0); r.adjust_dev(blurred.data_ptr(), blurred.data_ptr(), w, h, "hsl", hp)
for k in range(3):
    print("out of place", t(oop, ("gauss_mfma", "adjust")), "in place", t(inp, ("gauss_mfma", "adjust")), flush=True)
