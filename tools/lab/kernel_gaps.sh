#!/bin/bash
# tools/lab/kernel_gaps.sh — idle time between the back-to-back kernels of the headline step (rocprofv3 kernel trace of bench.py's timed steps)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/gaps; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o b -- python $ROOT/bench.py --no-cpu-baseline --headline-only --no-group --no-live-pmc > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed region: the longest run of alternating flatten / gauss_strip64 launches
seq = [(s, e, n) for s, e, n in rows if "flatten_srt" in n or "gauss_strip64" in n]
gaps_fg, gaps_gf = [], []
for (s0, e0, n0), (s1, e1, n1) in zip(seq, seq[1:]):
    g = s1 - e0
    if g > 200000: continue   # a host-side pause between phases of the bench
    (gaps_fg if "flatten" in n0 else gaps_gf).append(g)
import statistics as st
for name, g in (("flatten -> gaussian", gaps_fg), ("gaussian -> flatten", gaps_gf)):
    g = g[len(g) // 3:]
    print(f"{name}: n={len(g)} median gap {st.median(g) / 1000:.2f} us, mean {st.mean(g) / 1000:.2f} us, min {min(g) / 1000:.2f}, max {max(g) / 1000:.2f}")
PY
