#!/bin/bash
# tools/prof_box.sh — per-kernel times of the box blur at 8K (rocprofv3 kernel trace of tools/time_box.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_box
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o box -- python $ROOT/tools/time_box.py > $OUT/trace.log 2>&1
grep "^box" $OUT/trace.log
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "box" in r["Kernel_Name"]]
# consecutive groups of 13 launches per radius
seq = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-40:]
    seq.append((name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
i = 0
out = collections.OrderedDict()
grp = 0
last = None
import itertools
# print the average per (group of identical launch pattern)
cur = []
for name, ms in seq: cur.append((name, ms))
names = [n for n, _ in cur]
# group by runs of the repeating pattern: take the unique kernel names in order and chunk
pos = 0
while pos < len(cur):
    n0 = cur[pos][0]
    if "fused" in n0: per = 1
    else: per = 2
    chunk = cur[pos:pos + 13 * per]
    acc = collections.defaultdict(list)
    for n, ms in chunk: acc[n].append(ms)
    print(" | ".join(f"{n}: {sum(v)/len(v):.4f} ms" for n, v in acc.items()))
    pos += 13 * per
PY
