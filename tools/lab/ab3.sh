#!/bin/bash
# tools/lab/ab3.sh <libs...> — bench.py's headline step alternating several builds of libpfx on ONE box, 3 rounds
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for r in 1 2 3; do for lib in "$@"; do
  PFX_LIB_PATH=$ROOT/$lib python bench.py --no-cpu-baseline --headline-only --steps 40 --warmup 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['kernel_ms'], d['check'])"
done; done
