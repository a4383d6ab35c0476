mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_rollout3dw.py -x -q > gpurun_out/r3_t6.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t6.log
tail -4 gpurun_out/r3_t6.log
rm -f gpurun_out/r3_ab6.txt
for rep in 1 2; do
for w in 0 8 16; do
  SNAC_3D_WIDE=$w timeout -k 10 120 python tools/roll_time.py 3 16384 1000 24 f64 vmm >> gpurun_out/r3_ab6.txt 2>&1
done
done
SNAC_3D_WIDE=0 timeout -k 10 120 python tools/roll_time.py 3 65536 250 24 f64 vmm >> gpurun_out/r3_ab6.txt 2>&1
SNAC_3D_WIDE=8 timeout -k 10 120 python tools/roll_time.py 3 65536 250 24 f64 vmm >> gpurun_out/r3_ab6.txt 2>&1
grep -v amdgpu gpurun_out/r3_ab6.txt
