mkdir -p gpurun_out/pmc3dw
cd /tmp && export TMPDIR=/tmp
ROOT=$OLDPWD
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout 150 rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/pmc3dw/$N -o pmc -- python3 $ROOT/tools/roll_time.py 3 16384 1000 6 f64 malloc > $ROOT/gpurun_out/pmc3dw/$N.log 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc3dw/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rollout3d" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in acc.items():
    print(k)
    for c,x in sorted(v.items()): print("   %-28s n=%d avg=%.4g" % (c, len(x), sum(x)/len(x)))
PY
