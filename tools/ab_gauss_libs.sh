#!/bin/bash
# tools/ab_gauss_libs.sh "<libA.so> <libB.so> ..." [rounds] — tools/gauss_time.py alternating builds of libpfx on ONE box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for r in $(seq 1 ${2:-2}); do
  for lib in $1; do PFX_LIB_PATH=$ROOT/$lib python tools/gauss_time.py; done
done
