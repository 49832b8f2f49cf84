#!/bin/bash
# tools/ab_gauss.sh "<gauss_v_cfg values>" — bench.py's step with Gaussian launch variants (development A/B on one box):
# low byte = strip-segment multiplier (0 = shipped), 256 = tile kernel
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for v in $1; do
  echo "== gauss_v_cfg=$v"
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --tune gauss_v_cfg=$v 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done
