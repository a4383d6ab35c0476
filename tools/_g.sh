mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_plan_generators.py -m gpu -x -q > gpurun_out/r3_t8.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t8.log
tail -12 gpurun_out/r3_t8.log
