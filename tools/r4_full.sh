#!/bin/bash
# full GPU gate + default bench on one box
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4full; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/tests.txt 2>&1; echo "gpu tests rc=$?" | tee $OUT/summary.txt
tail -3 $OUT/tests.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"].get("kernel_ms"), d["roofline"]["frac"], d.get("check"), d.get("step_ms_hip_events"))
PY
