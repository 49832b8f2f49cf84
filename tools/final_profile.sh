#!/bin/bash
# tools/final_profile.sh <commit> — the judged artefacts of one build from ONE box: driver-style bench line, rocprofv3 kernel stats + PMC passes, operation table
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
bash tools/prof.sh final > $OUT/prof.log 2>&1
python tools/pmc_json.py gpurun_out/prof_final $OUT/pmc.json $1 > $OUT/pmc_json.log 2>&1; echo "pmc_json rc=$?"
cp gpurun_out/prof_final/summary.txt $OUT/rocprof_summary.txt
f=$(find gpurun_out/prof_final/trace -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv
timeout 500 python tools/bench_ops.py > $OUT/ops.txt 2>&1; echo "ops rc=$?"
rm -rf gpurun_out/prof_final/trace gpurun_out/prof_final/pmc_*
tail -c 600 $OUT/bench.json | head -c 300; echo; head -5 $OUT/rocprof_summary.txt
