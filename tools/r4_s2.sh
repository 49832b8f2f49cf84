#!/bin/bash
# round-4 GPU session: parity of the class-sorting compositor, A/B against the round-3 kernel
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4s2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dle.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/tests_dle.txt 2>&1; echo "dle tests rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/tests_dle.txt
for args in "dle_kernel=1" "dle_kernel=0" "dle_kernel=0 dle_s1=0" "dle_kernel=0 dle_s2=1" "dle_kernel=0 dle_s2=3" "dle_kernel=0 dle_s2=4" "dle_kernel=0 dle_s1=2 dle_s2=2" "dle_kernel=0 dle_cfg=1" "dle_kernel=1" "dle_kernel=0"; do
  timeout 120 python tools/dle_stats.py $args 2>&1 | grep -v amdgpu.ids >> $OUT/ab.txt
done
cat $OUT/ab.txt
