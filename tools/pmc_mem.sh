#!/bin/bash
# tools/pmc_mem.sh <tag> [key=value ...] — memory-side counters of the compositor (L1 miss latency, TLB, L2 hit rate, HBM request latency)
set -u
TAG=${1:-mem}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="python $ROOT/tools/dle_stats.py $*"
pass() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$n -o p -- $RUN > $OUT/$n.log 2>&1; }
pass tcpA TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass tccA TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
python - <<PY > $OUT/summary.txt
import csv, glob, collections
for grp in ("tcpA", "tccA"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % grp, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "flatten" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for c, v in sorted(acc.items()):
            v = v[len(v) // 2:]
            print(f"{c:34s} {sum(v) / len(v):.6g} (n={len(v)})")
PY
echo "== $TAG $*"; cat $OUT/summary.txt
