#!/bin/bash
# tools/pmc_gauss.sh <tag> — SQ / LDS counters of the Gaussian kernels in bench.py's step
set -u
TAG=${1:-g}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_gauss_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU -d $OUT/pmc_sq2 -o bench -- $BENCH > $OUT/pmc_sq2.log 2>&1
python $ROOT/tools/prof_summary.py $OUT 2>&1 | grep -E "gauss|^#" > $OUT/summary.txt
cat $OUT/summary.txt; tail -2 $OUT/pmc_sq2.log
