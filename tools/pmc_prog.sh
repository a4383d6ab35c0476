#!/bin/bash
# SQ counter passes (+ FETCH_SIZE / WRITE_SIZE) for any python program (run on the GPU box from the repo root):
#   bash tools/pmc_prog.sh <tag> tools/roll_time.py 2 65536 60 6 f64 vmm      (environment variables pass through)
# Output: gpurun_out/pmc_<tag>/summary.txt (per kernel and launch, tools/summarize_prof.py).  SQ / FETCH / WRITE counters only (TA_* /
# TCC_* passes abort rocprofv3 on this pool); every pass under `timeout`, the program itself right behind `--`.
TAG=${1:-x}
shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
PROG=$ROOT/$1
shift
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_IFETCH" \
         "SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o pmc -- python3 $PROG "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $PROG "$@" > $OUT/trace.log 2>&1
cd $ROOT
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -v "^$" $OUT/summary.txt | grep -i "k_rollout2d\|k_step\|k_trans" | awk '{print $(NF-3), $(NF-2), $(NF-1), $NF}' | head -80
