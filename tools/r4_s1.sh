#!/bin/bash
# round-4 GPU session 1: parity of the class-queue compositor, A/B against the round-3 kernel, clock / power probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s1
OUT=gpurun_out/r4s1
timeout 900 python -m pytest tests/test_gpu_dle.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/tests_dle.txt 2>&1; echo "dle tests rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/tests_dle.txt
for args in "dle_kernel=1" "dle_kernel=0" "dle_kernel=0 dle_s2=0" "dle_kernel=0 dle_s1=1 dle_s2=4" "dle_kernel=0 dle_s1=1 dle_s2=6" "dle_kernel=0 dle_s1=1 dle_s2=8" "dle_kernel=0 dle_s1=3 dle_s2=4" "dle_kernel=0 dle_cfg=1" "dle_kernel=1" "dle_kernel=0"; do
  timeout 300 python tools/dle_stats.py $args >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
timeout 300 python tools/lab/clock_probe.py > $OUT/clock.txt 2>&1; tail -30 $OUT/clock.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json
