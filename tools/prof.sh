#!/bin/bash
# tools/prof.sh <tag> — rocprofv3 passes for bench.py on the GPU box (run through gpurun).
# Pass 0: kernel trace + stats.  Passes 1-3: PMC counters, each in its own run (no trace domains besides kernel-trace).
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --headline-only --no-group --no-live-pmc"
# the trace pass runs bench.py's default step / warm-up counts (the clocks ramp over the first launches: a 7-launch run reads 10 % slow)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py --no-cpu-baseline --headline-only --no-group --no-live-pmc > $OUT/trace.log 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_sq2 -o bench -- $BENCH > $OUT/pmc_sq2.log 2>&1
# instruction classes: FMA / MUL / ADD f32 issue in 2 cycles per wave64, everything else in 4 (min / max / trunc / cvt / compare / select), transcendentals
# in 8 (tools/lab/valu_tput.hip -> profiles/r03_valu_rates.txt): the weighted sum is what the VALU is really busy with
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH -d $OUT/pmc_mix -o bench -- $BENCH > $OUT/pmc_mix.log 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -40
python $ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
