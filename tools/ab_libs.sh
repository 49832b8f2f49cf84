#!/bin/bash
# tools/ab_libs.sh <libA.so> <libB.so> [rounds] — bench.py's step alternating two builds of libpfx on ONE box (box-to-box spread is 5-8 %)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for r in $(seq 1 ${3:-2}); do
  for lib in $1 $2; do
    PFX_LIB_PATH=$ROOT/$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
