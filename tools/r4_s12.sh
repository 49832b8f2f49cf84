#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4s12; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_dle.py -x -q -m gpu 2>&1 | tail -1
for r in 1 2 3; do for lib in libpfx libpfx_enb2 libpfx_enb3; do
    echo -n "$lib: "; PFX_LIB_PATH=$GRAFT_REPO_ROOT/paintfe_amd/$lib.so timeout 120 python tools/dle_stats.py 2>&1 | grep -v amdgpu.ids | cut -c1-40
done; done | tee $OUT/ab.txt
bash tools/ab_libs.sh paintfe_amd/libpfx.so paintfe_amd/libpfx_enb2.so 2 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_bench.txt
