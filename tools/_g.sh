mkdir -p gpurun_out; rm -f gpurun_out/r3_ab11.txt
for rep in 1 2 3; do
for l in a b c; do
  echo "lib=$l (a: all window loads nontemporal; b: first pass plain, reload nontemporal; c: all plain)" >> gpurun_out/r3_ab11.txt
  SNAC_HIP_LIB=$PWD/ab/libsnac_$l.so timeout -k 10 120 python tools/step_time.py 3 524288 200 f64 >> gpurun_out/r3_ab11.txt 2>&1
done
done
grep -v amdgpu gpurun_out/r3_ab11.txt | grep -v "^lib" | awk '{print (NR-1)%3, $0}' | sort -n | cut -c1-100
cd /tmp && export TMPDIR=/tmp
for l in a b c; do
  SNAC_HIP_LIB=$OLDPWD/ab/libsnac_$l.so timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OLDPWD/gpurun_out/pmcnt_$l -o pmc -- python3 $OLDPWD/tools/step_time.py 3 524288 20 f64 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("$OLDPWD/gpurun_out/pmcnt_$l/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "k_step3d" in r["Kernel_Name"]]
print("lib $l FETCH_SIZE avg KiB", sum(v)/max(len(v),1), len(v))
PY
done
