#!/bin/bash
# tools/r06_chain_ab.sh — chains on ONE box: parity, then the chain rows of the operation table three times (config 2 as one launch against two, a script's
# three inline effects as one pass against three, the bit-exact Gaussian with HSL in its store)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r06_chain; mkdir -p $OUT
python -m pytest tests/test_gpu_chain.py tests/test_gpu_script_lang.py tests/test_gpu_batch.py -q -m gpu -x 2>&1 | tail -2 | tee $OUT/parity.txt
for r in 1 2 3; do
  for only in "chain" "the same"; do
    python tools/bench_ops.py --only "$only" 2>&1 | grep "'op'" | grep -v "map_channels" | python -c "
import sys, ast
for l in sys.stdin:
    d = ast.literal_eval(l.strip()); print('   %-84s %.4f ms' % (d['op'][:84], d['ms']))" | tee -a $OUT/ab.txt
  done
done
python tools/time_median.py 2 3 2>&1 | grep -v amdgpu | tee $OUT/median_bits_min.txt
