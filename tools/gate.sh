#!/bin/bash
# tools/gate.sh — the full GPU gate (pytest -m gpu) + one default bench line, as the driver runs them at round end
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/gate; mkdir -p $O
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.txt
tail -5 $O/gpu_tests.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 3000 $O/bench.json
