cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/pm1 -o m -- python $R/tools/time_mesh.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS -d $R/gpurun_out/pm2 -o m -- python $R/tools/time_mesh.py > /dev/null 2>&1
python - <<PY
import csv,glob,collections
for d in ("pm1","pm2"):  # a third pass with the TCP / TCC counters did not come back within its 300 s on this pool
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "mesh_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k,v in acc.items(): print(d,k,sum(v)/len(v),len(v))
PY
