#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r5bench; mkdir -p $O
cd $ROOT
timeout 900 python bench.py --headline-only --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_n1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["dtype"], d.get("failed_checks"))
print(json.dumps(d.get("c_abi_group"))[:1500])
PY
timeout 1200 python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_group.py -x -q 2>&1 | tail -4
