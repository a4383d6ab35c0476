mkdir -p gpurun_out
timeout -k 10 120 tools/vmm_stale 200 64 > gpurun_out/r03_vmm_stale.txt 2>&1; echo "rc=$?" >> gpurun_out/r03_vmm_stale.txt
cat gpurun_out/r03_vmm_stale.txt
timeout -k 10 1000 python -m pytest tests/test_hindsight_2d_dynamic.py tests/test_gpu_fullsize.py tests/test_gpu_robust.py tests/test_gpu_bench.py tests/test_gpu_parity.py tests/test_gpu_facade.py tests/test_gpu_property.py -m gpu -x -q > gpurun_out/r3_t5.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t5.log
tail -12 gpurun_out/r3_t5.log
timeout -k 10 120 python tools/step_overhead.py > gpurun_out/r03_step_overhead.txt 2>&1
cat gpurun_out/r03_step_overhead.txt | grep -v amdgpu
timeout -k 10 600 python bench.py > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; echo "bench rc=$?"
tail -3 gpurun_out/r3_bench1.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench1.json").read().strip().splitlines()[-1])
print("value %.4e ms_per_step %.4f kernel_ms %.4f frac %.4f traffic_source %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["traffic_source"]))
print(json.dumps(d["extra"], indent=1)[:3000])
print(d["cpu_baseline"]["cpu_model"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["value"], d["ranks_devices"], d["placement"])
PY
