"""tools/lab/cli_phases.py <op> — where a fresh process's first call spends its time (210-460 ms per CLI process).
Prints dlopen / pfx_ctx_create / first call / second call in ms for one op; the first call includes loading that op's code object."""
import ctypes as C
import sys
import time

import numpy as np

op = sys.argv[1] if len(sys.argv) > 1 else "blur"
t0 = time.perf_counter()
lib = C.CDLL("/root/repo/paintfe_amd/libpfx.so")
t1 = time.perf_counter()
ctx = C.c_void_p()
assert lib.pfx_ctx_create(0, C.byref(ctx)) == 0
t2 = time.perf_counter()
img = np.zeros((1024, 1024, 4), np.uint8)
out = np.zeros_like(img)
p, q = img.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)


def call():
    if op == "blur":
        return lib.pfx_blur_rgba(ctx, p, q, 1024, 1024, C.c_float(4.0))
    if op == "invert":
        return lib.pfx_invert_rgba(ctx, p, q, 1024, 1024)
    if op == "median":
        return lib.pfx_median_rgba(ctx, p, q, 1024, 1024, 1)
    if op == "alloc":   # no kernel: device allocation + copy only
        d = C.c_void_p()
        lib.pfx_dev_alloc(ctx, C.c_size_t(img.nbytes), C.byref(d))
        lib.pfx_dev_upload(ctx, d, p, C.c_size_t(img.nbytes))
        return lib.pfx_dev_free(ctx, d)
    raise SystemExit("op?")


assert call() == 0
t3 = time.perf_counter()
assert call() == 0
t4 = time.perf_counter()
print("%-7s dlopen %6.1f  ctx_create %6.1f  first call %6.1f  second call %6.2f ms" % (op, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
