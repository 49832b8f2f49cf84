#!/bin/bash
# tools/r06_median_ab.sh — VERDICT r05 #4: 5x5 median with the sorted columns shared across lanes (median_xlane2_kernel) against the per-lane network, ONE box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r06_median; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "median" 2>&1 | tail -2 | tee $OUT/parity.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
w, h = 7680, 4320
src = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
for _ in range(100): r.median_dev(src.data_ptr(), dst.data_ptr(), w, h, 2)
torch.cuda.synchronize()
for rnd in range(4):
    for xl in (0, 1, 2):
        r.tune("median_xlane", xl)
        for _ in range(20): r.median_dev(src.data_ptr(), dst.data_ptr(), w, h, 2)
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(50): r.median_dev(src.data_ptr(), dst.data_ptr(), w, h, 2)
        torch.cuda.synchronize(); r.timing_enable(False)
        print(f"round {rnd} median r=2 {('per-lane network        ', 'cross-lane, 1 row / lane', 'cross-lane, 2 rows / lane')[xl]}: {r.timing_read('median')[0] / 50:.4f} ms")
    for xl in (1, 5):
        r.tune("median_xlane", xl)
        for _ in range(20): r.median_dev(src.data_ptr(), dst.data_ptr(), w, h, 3)
        torch.cuda.synchronize(); r.timing_reset(); r.timing_enable(True)
        for _ in range(50): r.median_dev(src.data_ptr(), dst.data_ptr(), w, h, 3)
        torch.cuda.synchronize(); r.timing_enable(False)
        print(f"round {rnd} median r=3 {'cross-lane network' if xl == 5 else 'bit-plane select  '}: {r.timing_read('median')[0] / 50:.4f} ms")
    r.tune("median_xlane", 1)
PY
