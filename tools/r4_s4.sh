#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4s4; mkdir -p $OUT; rm -f $OUT/ab.txt
timeout 900 python -m pytest tests/test_gpu_dle.py -x -q -m gpu > $OUT/tests_dle.txt 2>&1; echo "dle tests rc=$?" | tee $OUT/summary.txt
tail -2 $OUT/tests_dle.txt
for r in 1 2; do
  for args in "dle_kernel=1" "dle_kernel=0" "dle_kernel=0 dle_s1=0" "dle_kernel=0 dle_s2=1" "dle_kernel=0 dle_s2=3" "dle_kernel=0 dle_s2=4" "dle_kernel=0 dle_cfg=1"; do
    echo -n "$args: " >> $OUT/ab.txt
    timeout 120 python tools/dle_stats.py $args 2>&1 | grep -v amdgpu.ids | cut -c1-80 >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
