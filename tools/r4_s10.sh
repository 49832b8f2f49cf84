#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4s10; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_dle.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab_libs.sh paintfe_amd/libpfx.so paintfe_amd/libpfx_typedst.so 2 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_bench.txt
bash tools/ab_libs.sh paintfe_amd/libpfx.so paintfe_amd/libpfx_rawnonat.so 2 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_bench.txt
cd /tmp && export TMPDIR=/tmp
for lib in libpfx libpfx_typedst libpfx_rawnonat; do
PFX_LIB_PATH=$GRAFT_REPO_ROOT/paintfe_amd/$lib.so timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/w_$lib -o p -- python $GRAFT_REPO_ROOT/tools/dle_stats.py > /dev/null 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$OUT/w_$lib/**/*counter_collection.csv", recursive=True):
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "flatten" in r["Kernel_Name"] and r["Counter_Name"]=="WRITE_SIZE"]
    print("$lib WRITE_SIZE KB per launch", sum(v[-20:])/20)
PY
done
