mkdir -p gpurun_out
timeout -k 10 120 tools/vmm_stale 300 64 > gpurun_out/r03_vmm_stale.txt 2>&1; echo "rc=$?" >> gpurun_out/r03_vmm_stale.txt
cat gpurun_out/r03_vmm_stale.txt
SNAC_TRAJ_DEBUG=1 timeout -k 10 300 python -m pytest tests/test_gpu_trajmem.py -x -q > gpurun_out/r3_t4.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t4.log
grep -v "probe round" gpurun_out/r3_t4.log | tail -15
timeout -k 10 120 python - > gpurun_out/r3_traj16.txt 2>&1 <<'PY'
import time, torch, os
os.environ["SNAC_TRAJ_DEBUG"]="1"
from snac_amd import trajmem
for i in range(4):
    t0=time.perf_counter()
    t=trajmem.traj_empty((600,65536,51), torch.float64, "cuda")
    print("16 GB block %d: %.2f s, layout %s, free %.1f GB" % (i, time.perf_counter()-t0, trajmem.layout_of(t), torch.cuda.mem_get_info()[0]/1e9), flush=True)
    del t
PY
grep -v "probe round" gpurun_out/r3_traj16.txt | tail
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_trajmem.py > gpurun_out/r3_full.log 2>&1; echo "rc=$?" >> gpurun_out/r3_full.log
tail -8 gpurun_out/r3_full.log
