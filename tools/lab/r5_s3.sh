#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r5s3; mkdir -p $O
cd $ROOT
for l in old tsu8; do PFX_LIB_PATH=$ROOT/paintfe_amd/libpfx_$l.so python tools/lab/early_cost.py 2>/dev/null | tail -1 | sed "s/^/$l /" >> $O/early.txt; done
cat $O/early.txt
