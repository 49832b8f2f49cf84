#!/bin/bash
# tools/lab/pmc_vm.sh — instruction counts of the script VM kernel (k_script.hip: vm_kernel) for bench_ops' map_channels closure at 8K
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_vm; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/run_vm.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from paintfe_amd import GpuRenderer
r = GpuRenderer(0)
img = np.random.default_rng(1).integers(0, 256, size=(4320, 7680, 4), dtype=np.uint8)
for _ in range(3): r.execute_script_sync("map_channels(|r, g, b, a| [255 - r, g / 2, (b * 3 + a) / 4, a]);", img)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/p -o v -- python /tmp/run_vm.py > $OUT/p.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "vm_kernel" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
w = sum(acc["SQ_WAVES"]) / max(len(acc["SQ_WAVES"]), 1)
for c, v in sorted(acc.items()): print("%-20s %.4g per launch, %.1f per wave" % (c, sum(v) / len(v), sum(v) / len(v) / w))
PY
rm -rf $OUT/p
