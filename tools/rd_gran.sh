#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel of tools/rd_gran (run on the GPU box from the repo root): gpurun_out/rd_gran/{run.txt,fetch.txt,write.txt}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/rd_gran; mkdir -p $OUT
$ROOT/tools/rd_gran > $OUT/run.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- $ROOT/tools/rd_gran > $OUT/$C.log 2>&1 || echo "$C pass failed"
done
cd $ROOT
python3 - <<'PY' > $OUT/counters.txt
import csv, glob, collections
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.OrderedDict()
    for f in glob.glob("gpurun_out/rd_gran/%s/**/*counter_collection.csv" % C, recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != C: continue
            acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print("%-10s %-60s launches %2d  last %.1f (counter units as reported)" % (C, k[:60], len(v), v[-1]))
PY
cat $OUT/run.txt $OUT/counters.txt
