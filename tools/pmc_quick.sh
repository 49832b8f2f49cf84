#!/bin/bash
# tools/pmc_quick.sh <tag> <lib or -> [key=value ...] — two short PMC passes (SQ instruction counts, instruction cache) of the compositor on the bench stack
set -u
TAG=$1; LIB=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ "$LIB" != "-" ] && export PFX_LIB_PATH=$ROOT/paintfe_amd/$LIB
RUN="python $ROOT/tools/dle_stats.py $*"
pass() { n=$1; shift; timeout 90 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$n -o p -- $RUN > $OUT/$n.log 2>&1 || echo "pass $n failed/timeout"; }
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pass b SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES
python - <<PY
import csv, glob, collections
for grp in ("a", "b"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % grp, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "flatten" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for c, v in sorted(acc.items()):
            v = v[len(v) // 2:]
            print(f"$TAG {c:34s} {sum(v) / len(v):.6g}")
PY
