#!/bin/bash
# LDS-gather warp samplers: parity tests, then window sizes against the global-gather kernels on one box
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4s8; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu -k "warp or mesh or liquify or displacement or fullsize" > $OUT/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.txt
for sw in 0 80 96 112 128 96 0; do
  echo "== warp_tile=$sw"; timeout 120 python tools/time_mesh.py warp_tile=$sw 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/ab.txt
