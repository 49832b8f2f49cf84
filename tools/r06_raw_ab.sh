#!/bin/bash
# tools/r06_raw_ab.sh — VERDICT r05 #1: the compositor's early pass on raw dword loads (srt_early_raw<N>, PFX_EARLY_RAW=N) against the shipped typed-load pass, ONE box:
# parity of every variant (tests/test_gpu_dle.py), three alternations of the compositor alone (8K x 32, S2), two PMC passes of shipped + variants.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r06_raw; mkdir -p $OUT
LIBS="${1:-libpfx.so libpfx_raw4.so libpfx_raw6.so libpfx_raw8.so}"
for lib in $LIBS; do
  echo "== parity $lib" | tee -a $OUT/parity.txt
  PFX_LIB_PATH=$ROOT/paintfe_amd/$lib timeout 900 python -m pytest tests/test_gpu_dle.py -q -m gpu -x 2>&1 | tail -2 | tee -a $OUT/parity.txt
done
bash tools/lab/ab_flat_libs.sh "$LIBS" 3 2>&1 | tee $OUT/ab.txt
for lib in $LIBS; do
  bash tools/pmc_quick.sh r06_${lib%.so} $lib 2>&1 | grep -v "^$" | tee -a $OUT/pmc.txt
done
