#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4s5; mkdir -p $OUT; rm -f $OUT/ab.txt
for r in 1 2; do
for lib in libpfx libpfx_oneswitch; do
  for args in "dle_kernel=0 dle_s1=0" "dle_kernel=0" "dle_kernel=1"; do
    echo -n "$lib $args: " >> $OUT/ab.txt
    PFX_LIB_PATH=$GRAFT_REPO_ROOT/paintfe_amd/$lib.so timeout 120 python tools/dle_stats.py $args 2>&1 | grep -v amdgpu.ids | cut -c1-60 >> $OUT/ab.txt
  done
done
done
cat $OUT/ab.txt
bash tools/pmc_quick.sh onesw libpfx_oneswitch.so dle_kernel=0 2>&1 | grep "INSTS_VALU\|INSTS_SALU\|GRBM\|ICACHE_MISSES "
