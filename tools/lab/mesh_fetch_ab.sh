#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do
  O=$ROOT/gpurun_out/mesh_pmc_$x; rm -rf $O; mkdir -p $O
  timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O -o p -- python $ROOT/tools/time_mesh.py mesh_xcd=$x > $O/log.txt 2>&1
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "mesh_roll" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("mesh_xcd=$x", {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
done
