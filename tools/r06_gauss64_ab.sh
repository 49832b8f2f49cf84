#!/bin/bash
# tools/r06_gauss64_ab.sh ["lib ..."] — VERDICT r05 #3: the matrix-core Gaussian on 64-column strips (twelve waves per workgroup) against 32-column strips, ONE box:
# bit identity, then alternating timings at 8K for sigma 11 / 13.7 / 16 (100 launches each, five rounds), per library build
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r06_gauss64; mkdir -p $OUT
LIBS="${1:-libpfx.so}"
for lib in $LIBS; do
PFX_LIB_PATH=$ROOT/paintfe_amd/$lib python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "64_column or gaussian" 2>&1 | tail -1 | tee -a $OUT/parity.txt
done
for rnd in 1 2 3; do
for lib in $LIBS; do
PFX_LIB_PATH=$ROOT/paintfe_amd/$lib python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
for _ in range(300): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, 16.0)
torch.cuda.synchronize()
lib = os.path.basename(os.environ.get("PFX_LIB_PATH", "libpfx.so"))
for sigma in (2.0, 4.0, 5.3, 6.0, 8.0, 10.0, 16.0):
    for cols in (0, 7):
        r.tune("gauss_cols64", cols)
        for _ in range(50): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, sigma)
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(100): r.gaussian_blur_dev(src.data_ptr(), dst.data_ptr(), w, h, sigma)
        torch.cuda.synchronize(); r.timing_enable(False)
        print(f"{lib:22s} sigma {sigma:5.1f} {'64-column' if cols else '32-column'} strips: {r.timing_read('gauss_mfma')[0] / 100:.4f} ms")
PY
done
done
