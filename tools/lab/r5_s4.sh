#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r5s4; mkdir -p $O
cd $ROOT
bash tools/lab/ab_flat_libs.sh "libpfx_e0.so libpfx_e2.so libpfx.so libpfx_e6.so libpfx_e8.so libpfx_e4w7.so" 3 > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_dle.py -x -q 2>&1 | tail -3
PFX_LIB_PATH=$ROOT/paintfe_amd/libpfx_e8.so timeout 900 python -m pytest tests/test_gpu_dle.py -x -q 2>&1 | tail -3
python tools/lab/early_cost.py 2>/dev/null | tail -1
