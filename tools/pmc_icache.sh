#!/bin/bash
# tools/pmc_icache.sh <tag> — instruction-cache / issue-stall counters for bench.py's kernels (one PMC pass each).
set -u
TAG=${1:-ic}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/pmc_ic -o bench -- $BENCH > $OUT/pmc_ic.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC -d $OUT/pmc_ic2 -o bench -- $BENCH > $OUT/pmc_ic2.log 2>&1
python $ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
tail -3 $OUT/pmc_ic.log $OUT/pmc_ic2.log
