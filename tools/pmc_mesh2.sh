#!/bin/bash
# texture-path / L1 counters of the fused mesh warp (16K): one short pass per group, each under its own timeout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_mesh2; mkdir -p $OUT
[ -n "${1:-}" ] && export PFX_LIB_PATH=$R/paintfe_amd/$1
timeout 60 rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA_[A-Z_]+|TD_[A-Z_]+|TCP_[A-Z_]+)\b" | sort -u > $OUT/avail.txt; wc -l $OUT/avail.txt
pass() { n=$1; shift; timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$n -o m -- python $R/tools/time_mesh.py > $OUT/$n.log 2>&1 || echo "pass $n failed/timeout"; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD
pass b SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
pass c TA_BUSY_avr TA_BUSY_max TD_BUSY_avr TCP_PENDING_STALL_CYCLES_sum
pass d TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
python - <<PY
import csv,glob,collections
for d in "abcd":
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"].replace("(anonymous namespace)::","")
            if "mesh_" in k or "warp_disp" in k:
                acc[(k.split("(")[0][-40:],row["Counter_Name"])].append(float(row["Counter_Value"]))
        for k,v in sorted(acc.items()): print(d,k[0],k[1],"%.6g"%(sum(v)/len(v)),len(v))
PY
