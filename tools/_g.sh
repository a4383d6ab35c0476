mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_step_tile.py tests/test_gpu_parity.py tests/test_gpu_facade.py tests/test_gpu_trajmem.py -x -q > gpurun_out/r3_t9.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t9.log
tail -3 gpurun_out/r3_t9.log
bash tools/profile.sh r03 > gpurun_out/r03_profile.log 2>&1
tail -1 gpurun_out/r03_profile.log | cut -c1-220
PROG=tools/prof_step.py bash tools/profile.sh r03_step > gpurun_out/r03_step_profile.log 2>&1
find gpurun_out/prof_r03_step -name "*.csv" ! -name "*kernel_stats.csv" -delete
grep "k_step" gpurun_out/prof_r03_step/summary.txt | grep "n=50 avg=.*us" | cut -c1-120
