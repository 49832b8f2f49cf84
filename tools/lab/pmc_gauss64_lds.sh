#!/bin/bash
# tools/lab/pmc_gauss64_lds.sh [lib] — LDS counters of gauss_strip64_kernel<8> at 8K, sigma 16 (what tools/lds_bank_sim.py models)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
[ -n "${1:-}" ] && export PFX_LIB_PATH=$ROOT/paintfe_amd/$1
OUT=$ROOT/gpurun_out/pmc_g64_${1:-default}; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/run_g64.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from paintfe_amd import GpuRenderer
r = GpuRenderer(0)
w, h = 7680, 4320
img = np.random.default_rng(1).integers(0, 256, size=(h, w, 4), dtype=np.uint8)
a, b = r.dev_alloc(img.nbytes), r.dev_alloc(img.nbytes)
r.dev_upload(a, img)
for _ in range(20): r.gaussian_blur_dev(a, b, w, h, 16.0)
r.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/p -o g -- python /tmp/run_g64.py > $OUT/p.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "strip64" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in sorted(acc.items()): print("${1:-libpfx.so} %-24s %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
