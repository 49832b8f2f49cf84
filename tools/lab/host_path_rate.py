"""tools/lab/host_path_rate.py — wall clock of the host-buffer entry points at 8K (pageable numpy buffers in and out): what a drop-in `blur_rgba(&[u8]) -> Vec<u8>` caller sees"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from paintfe_amd import GpuRenderer
r = GpuRenderer(0)
img = np.random.default_rng(1).integers(0, 256, size=(4320, 7680, 4), dtype=np.uint8)
mb = img.nbytes / 1e6
pin_in, pin_out = r.host_alloc(img.shape), r.host_alloc(img.shape)
pin_in[...] = img
for name, f in (("invert_rgba", lambda: r.invert_rgba(img)), ("invert_rgba, page-locked buffers", lambda: r.invert_rgba(pin_in, out=pin_out)), ("blur_rgba sigma 4", lambda: r.blur_rgba(img, 4.0)),
                ("dev_upload + dev_download", None)):
    if f is None:
        d = r.dev_alloc(img.nbytes); out = np.empty_like(img)
        def f():
            r.dev_upload(d, img); return r.dev_download(d, img.shape)
    f(); f()
    t = time.perf_counter(); n = 5
    for _ in range(n): f()
    dt = (time.perf_counter() - t) / n
    print(f"{name:34s} {dt * 1e3:7.1f} ms per call = {2 * mb / dt / 1e3:5.1f} GB/s over both directions ({mb:.0f} MB each way)")
