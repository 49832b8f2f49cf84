#!/bin/bash
# tools/pmc_flat.sh <tag> "<variants>" — SQ / TA counters of the compositor for each flatten_variant (one PMC pass per group)
set -u
TAG=${1:-fl}; VARS=${2:-0}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  OUT=$ROOT/gpurun_out/pmc_${TAG}_v$v
  mkdir -p $OUT
  BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --tune flatten_variant=$v"
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq2 -o bench -- $BENCH > $OUT/pmc_sq2.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum -d $OUT/pmc_ta -o bench -- $BENCH > $OUT/pmc_ta.log 2>&1
  python $ROOT/tools/prof_summary.py $OUT 2>&1 | grep -E "flatten|^#" > $OUT/summary.txt
  echo "=== variant $v"; cat $OUT/summary.txt
done
