#!/bin/bash
# tools/prof_median.sh — per-kernel times of the median paths at 8K (rocprofv3 kernel trace of tools/time_median.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_median
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o med -- python $ROOT/tools/time_median.py "$@" > $OUT/trace.log 2>&1
tail -16 $OUT/trace.log
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    print(f"{row['Name'][:90]:90s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e6:8.4f} ms")
PY
