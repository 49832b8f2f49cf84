#!/bin/bash
# mesh warp: straight-line tap loads, rows per batch 8 / 4 / 2; parity tests first
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4s7; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu -k "warp or mesh or liquify or displacement or fullsize" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.txt
for r in 1 2; do
for lib in libpfx libpfx_yr4 libpfx_yr2; do
  echo "== $lib"; PFX_LIB_PATH=$GRAFT_REPO_ROOT/paintfe_amd/$lib.so timeout 120 python tools/time_mesh.py 2>&1 | grep -v amdgpu.ids
done
done 2>&1 | tee $OUT/ab.txt
