#!/bin/bash
# tools/pmc_median.sh [bits_min] — issue / instruction-cache / LDS counters of the median kernels at 8K (one PMC pass each, kernel trace only)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_median
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/time_median.py ${1:-4}"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/p1 -o med -- $CMD > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY -d $OUT/p2 -o med -- $CMD > $OUT/p2.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections
for p in ("p1", "p2"):
    f = glob.glob(sys.argv[1] + "/" + p + "/**/*counter_collection.csv", recursive=True)
    if not f: print(p, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen = set()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (k, row["Dispatch_Id"])
        if key not in seen: seen.add(key); n[k] += 1
    for k in acc:
        if "median" not in k: continue
        print(p, k, "launches", n[k])
        for c, v in sorted(acc[k].items()): print(f"    {c:28s} {v / n[k]:.4g}")
PY
