#!/bin/bash
# A/B: dealt units stored in lane order (libpfx) against dealt order (libpfx_nonat); Gaussian piece counts; dealt-order tests
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4s6; mkdir -p $OUT; rm -f $OUT/ab.txt
timeout 600 python -m pytest tests/test_gpu_dle.py tests/test_gpu_parity.py -x -q -m gpu -k "dle or gauss or flatten or composite" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.txt
for r in 1 2 3; do
for lib in libpfx libpfx_nonat; do
    echo -n "$lib: " >> $OUT/ab.txt
    PFX_LIB_PATH=$GRAFT_REPO_ROOT/paintfe_amd/$lib.so timeout 120 python tools/dle_stats.py 2>&1 | grep -v amdgpu.ids | cut -c1-60 >> $OUT/ab.txt
done
done
cat $OUT/ab.txt
bash tools/ab_libs.sh paintfe_amd/libpfx.so paintfe_amd/libpfx_nonat.so 2 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_bench.txt
