#!/bin/bash
# tools/pmc_flat2.sh — what the compositor's VALU slots are spent on: quarter-rate (transcendental) instructions, conversions, occupancy,
# outstanding vector-memory instructions and TA back-pressure (one PMC pass), plus the chip's sustained FMA rate (tools/ubench_valu).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_flat2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_CYCLES -d $OUT/p1 -o bench -- $BENCH > $OUT/p1.log 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "flatten_stream" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()): print(k, sum(v) / len(v), len(v))
PY
$ROOT/tools/ubench_valu | grep -i "fma"
