#!/bin/bash
# tools/pmc_ops.sh [text] — unit occupancy of every kernel of tools/bench_ops.py (rows containing <text>): VALU issue, LDS, texture-address unit, waits
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_ops; rm -rf $OUT; mkdir -p $OUT
RUN="python $R/tools/bench_ops.py --reps 3"; [ -n "${1:-}" ] && RUN="$RUN --only $1"
pass() { n=$1; shift; timeout 280 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$n -o p -- $RUN > $OUT/$n.log 2>&1 || echo "pass $n failed/timeout"; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD
pass b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_WR
pass c FETCH_SIZE
pass d WRITE_SIZE
python - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for d in "abcd":
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d, recursive=True):
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"].replace("(anonymous namespace)::",""); k=re.sub(r"^void ","",k); k=re.sub(r"\(.*","",k)
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv"%d, recursive=True):
        if d!="a": continue
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"].replace("(anonymous namespace)::",""); k=re.sub(r"^void ","",k); k=re.sub(r"\(.*","",k)
            dur[k].append(float(row["End_Timestamp"])-float(row["Start_Timestamp"]))
print(f"{'kernel':58s} {'ms':>7s} {'GHz':>5s} {'VALU':>5s} {'SALU':>5s} {'LDS':>5s} {'confl':>5s} {'TA':>5s} {'waves':>5s} {'rdGB':>6s} {'wrGB':>6s}")
for k,c in sorted(acc.items(), key=lambda kv: -sum(dur.get(kv[0],[0]))):
    m=lambda n: (sum(c[n])/len(c[n])) if c.get(n) else 0.0
    cyc=m("GRBM_GUI_ACTIVE")/8
    if cyc<20000: continue
    ms=(sum(dur[k])/len(dur[k])/1e6) if dur.get(k) else 0
    # GRBM_GUI_ACTIVE / duration is a clock only when the launch is long against the counter's start / stop latency: a derived clock above the 2.4 GHz
    # spec means the launch was too short for it, and every "busy" fraction built on that clock with it
    if ms and cyc/(ms*1e6) > 2.45:
        print(f"{k[:58]:58s} {ms:7.3f}  launch too short for the busy fractions (derived clock {cyc/(ms*1e6):.2f} GHz > 2.4 spec); rdGB {m('FETCH_SIZE')*2048/1e9:.2f} wrGB {m('WRITE_SIZE')*1024/1e9:.2f}")
        continue
    print(f"{k[:58]:58s} {ms:7.3f} {cyc/(ms*1e6) if ms else 0:5.2f} {m('SQ_INSTS_VALU')*2/(1024*cyc):5.2f} {m('SQ_INSTS_SALU')/(256*cyc):5.2f} {m('SQ_LDS_IDX_ACTIVE')/(256*cyc):5.2f} {m('SQ_LDS_BANK_CONFLICT')/max(m('SQ_LDS_IDX_ACTIVE'),1):5.2f} {m('TA_BUSY_avr')/cyc:5.2f} {m('SQ_WAVE_CYCLES')*4/(1024*cyc):5.1f} {m('FETCH_SIZE')*2048/1e9:6.2f} {m('WRITE_SIZE')*1024/1e9:6.2f}")
PY
