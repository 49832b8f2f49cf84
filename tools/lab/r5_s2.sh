#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r5s2; mkdir -p $O
cd $ROOT
bash tools/lab/ab_flat_libs.sh "libpfx_old.so libpfx_ts7.so libpfx_tsu.so libpfx_tsu8.so" 3 > $O/ab.txt 2>&1
for l in ts7 tsu; do
  PFX_LIB_PATH=$ROOT/paintfe_amd/libpfx_$l.so timeout 900 python -m pytest tests/test_gpu_dle.py -x -q 2>&1 | tail -3 > $O/dle_$l.txt
done
PFX_LIB_PATH=$ROOT/paintfe_amd/libpfx_old.so python tools/lab/srt_phases.py > $O/phases_old.txt 2>&1
PFX_LIB_PATH=$ROOT/paintfe_amd/libpfx_ts7.so python tools/lab/srt_phases.py > $O/phases_ts7.txt 2>&1
PFX_LIB_PATH=$ROOT/paintfe_amd/libpfx_old.so python tools/lab/srt_phases.py mode=1 > $O/phases_old_m1.txt 2>&1
PFX_LIB_PATH=$ROOT/paintfe_amd/libpfx_ts7.so bash tools/pmc_quick.sh ts7 - > $O/pmc_ts7.txt 2>&1
PFX_LIB_PATH=$ROOT/paintfe_amd/libpfx_tsu.so bash tools/pmc_quick.sh tsu - > $O/pmc_tsu.txt 2>&1
cat $O/ab.txt; tail -2 $O/dle_*.txt; tail -1 $O/phases_*.txt; grep -E "INSTS_VALU|INSTS_SALU|INSTS_BRANCH|WAIT_INST|ICACHE_MISSES " $O/pmc_ts7.txt $O/pmc_tsu.txt
