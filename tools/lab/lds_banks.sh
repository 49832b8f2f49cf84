#!/bin/bash
# tools/lab/lds_banks.sh — run on the GPU box: timings + SQ_LDS counters of tools/lab/_bin/lds_banks, printed as profiles/rNN_lds_banks.txt wants them
R=$GRAFT_REPO_ROOT; B=$R/tools/lab/_bin/lds_banks
$B > /tmp/lds_t.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d /tmp/lds_banks -o m -- $B > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections,json
acc=collections.OrderedDict()
for f in glob.glob("/tmp/lds_banks/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc.setdefault((row["Dispatch_Id"],row["Kernel_Name"]),{})[row["Counter_Name"]]=float(row["Counter_Value"])
rows=[json.loads(l) for l in open("/tmp/lds_t.txt")]
keys=sorted(acc, key=lambda k:int(k[0]))
print("# tools/lab/lds_banks.hip on MI355X: LDS cycles per 64-lane instruction (SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS), the part of them that is bank conflict")
print("# (SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS) and the wall-clock cycles per instruction with 4 waves per SIMD issuing nothing else; second launch of each pattern")
for j,r in enumerate(rows):
    v=acc[keys[2*j+1]]; n=max(v.get("SQ_INSTS_LDS",1),1)
    print("%-48s active %5.2f  conflict %5.2f  wall %5.2f"%(r["pattern"], v.get("SQ_LDS_IDX_ACTIVE",0)/n, v.get("SQ_LDS_BANK_CONFLICT",0)/n, r["cycles_per_wave_instruction"]))
PY
