#!/bin/bash
# tools/pmc_flow.sh <tag> [key=value ...] — SQ / SQC / TCP / TA counters of the compositor on the bench stack (8K x 32 layers, S2) with the given
# pfx_tune knobs; one rocprofv3 --pmc pass per counter group (no trace domains mixed in), per-launch means printed and kept.
set -u
TAG=${1:-flow}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="python $ROOT/tools/dle_stats.py $*"
pass() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$n -o p -- $RUN > $OUT/$n.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS
pass sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL
pass tcp TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python - <<PY > $OUT/summary.txt
import csv, glob, collections
for grp in ("sq1", "sq2", "sq3", "tcp", "fetch", "write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % grp, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "flatten" in k:
                acc[(k.split("(")[0][-60:], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            v = v[len(v) // 2:]   # the second half of the launches: warm
            print(f"{k:60s} {c:34s} {sum(v) / len(v):.6g} (n={len(v)})")
PY
echo "== $TAG $*"; cat $OUT/summary.txt
