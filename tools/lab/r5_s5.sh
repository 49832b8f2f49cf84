#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for rep in 1 2; do
for a in "dle_s2=1" "dle_s2=2" "dle_s2=3" "dle_s2=4" "dle_s2=6" "dle_s1=2" "dle_s1=3" "dle_s1=2 dle_s2=2" "dle_units=4" "dle_units=12"; do
  python tools/dle_stats.py $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a', d['flatten_ms'])"
done; done
timeout 900 python -m pytest tests/test_gpu_group.py -x -q 2>&1 | grep -E "passed|failed"
