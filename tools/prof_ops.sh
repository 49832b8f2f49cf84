#!/bin/bash
# tools/prof_ops.sh <text> — per-kernel average times of the tools/bench_ops.py rows whose name contains <text> (rocprofv3 kernel trace)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_ops
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ops -- python $ROOT/tools/bench_ops.py --only "$1" > $OUT/trace.log 2>&1
grep "'op'" $OUT/trace.log | cut -c1-160
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if float(row['AverageNs']) > 5000: print(f"{row['Name'][:100]:100s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e6:8.4f} ms")
PY
