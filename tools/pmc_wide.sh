#!/bin/bash
# texture-path / LDS counters of tools/lab/wide_load's variants (the refutation wants TA_BUSY, TD_BUSY, TCP_PENDING_STALL_CYCLES, LDS conflicts)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_wide; rm -rf $OUT; mkdir -p $OUT
pass() { n=$1; shift; timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$n -o p -- $R/tools/lab/wide_load 8 > $OUT/$n.log 2>&1 || echo "pass $n failed/timeout"; }
pass a GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pass b TD_TD_BUSY_sum TA_TA_BUSY_sum SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum
python - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in "ab":
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d, recursive=True):
        for row in csv.DictReader(open(f)):
            m=re.search(r"k<(\d+), (\d+), (\d+), (\d+)", row["Kernel_Name"])
            if m: acc[tuple(int(x) for x in m.groups())][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("mode F depth wpb | cycles(M)  TA_busy  TA_max  TD_busy  TCP_pending/cyc/CU  VALU  loads(M)  cache_acc/load  LDS_busy  LDS_confl")
for k in sorted(acc, key=lambda t:(t[1],t[0],t[2])):
    c=acc[k]; m=lambda n: (sum(c[n][-6:])/len(c[n][-6:])) if c.get(n) else 0.0
    cyc=m("GRBM_GUI_ACTIVE")/8
    if not cyc: continue
    print(f"{k[0]} {k[1]:3d} {k[2]} {k[3]} | {cyc/1e6:6.3f} {m('TA_BUSY_avr')/cyc:7.2f} {m('TA_BUSY_max')/cyc:7.2f} {m('TD_TD_BUSY_sum')/(256*cyc):7.2f} {m('TCP_PENDING_STALL_CYCLES_sum')/(256*cyc):10.2f} {m('SQ_INSTS_VALU')*2/(1024*cyc):8.2f} {m('SQ_INSTS_VMEM_RD')/1e6:8.2f} {m('TCP_TOTAL_CACHE_ACCESSES_sum')/max(m('SQ_INSTS_VMEM_RD'),1):10.1f} {m('SQ_LDS_IDX_ACTIVE')/(256*cyc):9.2f} {m('SQ_LDS_BANK_CONFLICT')/max(m('SQ_LDS_IDX_ACTIVE'),1):9.2f}")
PY
