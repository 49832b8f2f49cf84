#!/bin/bash
# tools/pmc_cls.sh <tag> [key=value ...] — instruction and memory-path counters of the compositor on the bench stack (8K x 32 layers, S2) for one
# kernel configuration; one rocprofv3 --pmc pass per counter group (no trace domains mixed in), per-launch means of the warm launches.
set -u
TAG=${1:-cls}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="python $ROOT/tools/dle_stats.py $*"
pass() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$n -o p -- $RUN > $OUT/$n.log 2>&1; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass mix SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum TA_BUFFER_COALESCED_WRITE_CYCLES_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python - <<PY > $OUT/summary.txt
import csv, glob, collections
for grp in ("sq1", "sq2", "mix", "ta", "tcp", "tcc", "fetch", "write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % grp, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "flatten" in k:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for c, v in sorted(acc.items()):
            v = v[len(v) // 2:]   # the second half of the launches: warm
            print(f"{c:40s} {sum(v) / len(v):.6g} (n={len(v)})")
PY
echo "== $TAG $*"; cat $OUT/summary.txt
