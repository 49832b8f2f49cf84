#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4s11; mkdir -p $OUT
timeout 600 python -m pytest tests -x -q -m gpu -k "warp or mesh or fullsize or liquify or displacement" 2>&1 | tail -2
for r in 1 2 3; do
for lib in libpfx libpfx_nopin; do
  echo "== $lib: "; PFX_LIB_PATH=$GRAFT_REPO_ROOT/paintfe_amd/$lib.so timeout 120 python tools/time_mesh.py 2>&1 | grep "fused\|liquify"
done
done 2>&1 | tee $OUT/ab_pin.txt
