mkdir -p gpurun_out; rm -f gpurun_out/r3_ab9.txt
for rep in 1 2 3; do
for l in a b; do
  echo "lib=$l (a: plain record loads in k_step2d; b: nontemporal)" >> gpurun_out/r3_ab9.txt
  SNAC_HIP_LIB=$PWD/ab/libsnac_$l.so timeout -k 10 120 python tools/step_time.py 2 524288 200 f64 >> gpurun_out/r3_ab9.txt 2>&1
done
done
SNAC_HIP_LIB=$PWD/ab/libsnac_b.so timeout -k 10 120 python tools/step_time.py 3 524288 200 f64 >> gpurun_out/r3_ab9.txt 2>&1
SNAC_HIP_LIB=$PWD/ab/libsnac_b.so timeout -k 10 200 python -m pytest tests/test_gpu_step_tile.py -x -q 2>&1 | tail -2 >> gpurun_out/r3_ab9.txt
grep -v amdgpu gpurun_out/r3_ab9.txt | cut -c1-110
