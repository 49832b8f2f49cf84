#!/bin/bash
# tools/lab/pmc_median.sh [radii…] — SQ wave-state counters of the median kernels (tools/lab/median_time.py): where a resident wave's cycles go
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_median; rm -rf $OUT; mkdir -p $OUT
timeout 250 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/a -o p -- python $R/tools/lab/median_time.py "$@" > $OUT/a.log 2>&1 || echo "pass failed"
python - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].replace("(anonymous namespace)::",""); k=re.sub(r"^void ","",k); k=re.sub(r"\(.*","",k)
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("kernel                                 waves   cyc(M)  wave_cyc/(SIMD cyc)  VALU insts  active_valu/wave_cyc  wait_inst/wave_cyc  wait_any/wave_cyc  active_any/wave_cyc  active_valu per VALU inst (quad-cycles)")
for k,c in acc.items():
    if "median" not in k: continue
    m=lambda n: (sum(c[n])/len(c[n])) if c.get(n) else 0.0
    cyc=m("GRBM_GUI_ACTIVE")/8; wc=m("SQ_WAVE_CYCLES")
    print(f"{k[:38]:38s} {m('SQ_WAVES'):7.0f} {cyc/1e6:7.3f} {wc*4/(1024*cyc):8.2f} {m('SQ_INSTS_VALU'):12.4g} {m('SQ_ACTIVE_INST_VALU')/wc:8.3f} {m('SQ_WAIT_INST_ANY')/wc:8.3f} {m('SQ_WAIT_ANY')/wc:8.3f} {m('SQ_ACTIVE_INST_ANY')/wc:8.3f} {m('SQ_ACTIVE_INST_VALU')/max(m('SQ_INSTS_VALU'),1):8.3f}   busy {m('SQ_BUSY_CYCLES'):.4g}")
PY
