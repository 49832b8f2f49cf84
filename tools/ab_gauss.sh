#!/bin/bash
# tools/ab_gauss.sh "<dbg masks>" — bench.py's Gaussian with phases of gauss_mfma_kernel disabled (development only; results are wrong)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for v in $1; do
  echo "== gauss dbg=$v"
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --tune gauss_v_cfg=$((v*256)) 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done
