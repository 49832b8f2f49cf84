#!/bin/bash
# tools/pmc_ops_lds.sh — LDS bank-conflict share per kernel over the whole operation table (tools/bench_ops.py), one PMC pass
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_ops_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES -d $OUT/p -o ops -- python $ROOT/tools/bench_ops.py > $OUT/log.txt 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"][:70]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (k, row["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k] += 1
rows = []
for k, c in acc.items():
    act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    if act <= 0: continue
    rows.append((c.get("SQ_LDS_BANK_CONFLICT", 0.0) / act, act / n[k], c.get("SQ_INSTS_LDS", 0) / n[k], c.get("SQ_INSTS_VALU", 0) / n[k], n[k], k))
for r in sorted(rows, reverse=True)[:40]:
    print(f"conflict/active {r[0]:.2f}  lds_active/launch {r[1]:.3g}  lds_insts {r[2]:.3g}  valu {r[3]:.3g}  launches {r[4]}  {r[5]}")
PY
