#!/bin/bash
# tools/lab/ab_flat_libs.sh "<libA> <libB> ..." [rounds] [dle_stats args] — the compositor alone (tools/dle_stats.py: 8K x 32 layers, S2) alternating builds of libpfx on ONE box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for r in $(seq 1 ${2:-3}); do
  for lib in $1; do
    PFX_LIB_PATH=$ROOT/paintfe_amd/$lib python tools/dle_stats.py ${3:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', d['flatten_ms'])"
  done
done
