#!/bin/bash
# tools/lab/ab_ops_libs.sh <libA> <libB> — tools/bench_ops.py with two builds of libpfx on ONE box, per-operation ms side by side (A B A B)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=$ROOT/gpurun_out/ab_ops; mkdir -p $O
for rep in 1 2; do for lib in $1 $2; do PFX_LIB_PATH=$ROOT/paintfe_amd/$lib python tools/bench_ops.py ${3:+--only "$3"} --out $O/${lib}_$rep.json > /dev/null 2>&1; done; done
python - <<PY
import json
A=[json.load(open("$O/$1_%d.json"%k)) for k in (1,2)]; B=[json.load(open("$O/$2_%d.json"%k)) for k in (1,2)]
rows=lambda d: {r["op"]: r["ms"] for r in (d["rows"] if isinstance(d,dict) else d)}
a=[rows(x) for x in A]; b=[rows(x) for x in B]
for op in a[0]:
    am=min(a[0][op],a[1][op]); bm=min(b[0].get(op,9e9),b[1].get(op,9e9))
    flag = " <<<" if bm < 0.96*am else (" >>> worse" if bm > 1.04*am else "")
    print(f"{op[:60]:60s} {am:8.4f} {bm:8.4f} {bm/am:6.3f}{flag}")
PY
