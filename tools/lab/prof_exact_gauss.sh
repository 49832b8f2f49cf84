#!/bin/bash
# tools/lab/prof_exact_gauss.sh — counters of the bit-exact fused Gaussian (k_gauss_exact.hip) at 8K: glow radius 3 (kernel radius 9), sharpen radius 1 (3), sigma 4 (12)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_gexact; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/run_gexact.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from paintfe_amd import GpuRenderer
r = GpuRenderer(0)
w, h = 7680, 4320
img = np.random.default_rng(1).integers(0, 256, size=(h, w, 4), dtype=np.uint8)
a, b = r.dev_alloc(img.nbytes), r.dev_alloc(img.nbytes)
r.dev_upload(a, img)
r.set_exact(True)
for _ in range(12):
    r.glow_dev(a, b, w, h, 3.0, 0.5)
    r.sharpen_dev(a, b, w, h, 1.0, 1.0)
    r.gaussian_blur_dev(a, b, w, h, 4.0)
r.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o g -- python /tmp/run_gexact.py > $OUT/trace.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o g -- python /tmp/run_gexact.py > $OUT/pmc_sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD -d $OUT/pmc_sq2 -o g -- python /tmp/run_gexact.py > $OUT/pmc_sq2.log 2>&1
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_gexact"
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gauss" in row["Name"]: print(row["Name"][:70], row["Calls"], row["AverageNs"])
for d in ("pmc_sq", "pmc_sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "gauss" in row["Kernel_Name"]:
                agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        print(k)
        for c, v in sorted(cs.items()): print("   %-26s %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
