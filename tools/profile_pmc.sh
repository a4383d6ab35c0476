#!/bin/bash
# Extra PMC passes for one bench workload (run on the GPU box from the repo root):
#   bash tools/profile_pmc.sh <tag> <bench.py args...>
# Stall-side SQ counters that tools/profile.sh does not collect: where the waves of the rollout kernel wait.
# (TA_* / TCC_* counter passes abort rocprofv3 on this pool and then hang: SQ counters only, every pass under `timeout`.)
TAG=${1:-x}
shift
ARGS="$@"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_IFETCH" \
         "SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o pmc -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu $ARGS > $OUT/p$i.log 2>&1
done
cd $ROOT
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -i "rollout" $OUT/summary.txt | awk '{print $(NF-2), $(NF-1), $NF}'
