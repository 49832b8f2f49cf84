#!/bin/bash
# tools/ab_variants.sh "<v1> <v2> ..." — bench.py's step with each flatten_variant (development A/B on one box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for v in $1; do
  echo "== flatten_variant=$v"
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --tune flatten_variant=$v 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('check'))"
done
