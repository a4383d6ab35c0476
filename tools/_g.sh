mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_mcts.py tests/test_gpu_property.py -x -q > gpurun_out/r3_t7.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t7.log
tail -6 gpurun_out/r3_t7.log
rm -f gpurun_out/r3_ab10.txt
for rep in 1 2; do
for e in 32 64; do
  echo "SNAC_T2D_E=$e" >> gpurun_out/r3_ab10.txt
  SNAC_T2D_E=$e timeout -k 10 200 python - >> gpurun_out/r3_ab10.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, "tools")
import bench_configs as b
ms, m = b.transition_time(2, parents=(1 << 17))
print("transition 2D dynamic: %d edges %8.3f ms  %.3e edges/s" % (m, ms, m / ms * 1e3))
PY
done
done
grep -v amdgpu gpurun_out/r3_ab10.txt
