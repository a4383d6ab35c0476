#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace stats + separate PMC passes.
#   bash tools/profile.sh <tag> [bench.py args...]          profiles bench.py (--steps 5 --warmup 1 --no-cpu + args)
#   PROG=tools/prof_step.py bash tools/profile.sh <tag>     profiles another python program
# Outputs land in gpurun_out/prof_<tag>/ ; summaries are copied to profiles/ by hand.  Every pass runs under `timeout`
# (a counter pass that aborts would otherwise sit until gpurun's own limit).  SQ / FETCH / WRITE counters only: TA_* and
# TCC_* passes abort rocprofv3 on this pool.
# SNAC_BENCH_TILED=0: bench.py's extra passes into the tile-major layout use the same kernel symbol and would mix their (shorter)
# launches into the per-kernel average; the profiled launches are the headline's own (candidates, pre-roll, warm-up, timed).
export SNAC_BENCH_TILED=0
export SNAC_BENCH_EXTRAS=0   # the secondary configurations of the bench line are measured un-profiled; here only the headline's launches count
TAG=${1:-r1}
shift
ARGS="$@"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
if [ -n "$PROG" ]; then CMD="python3 $ROOT/$PROG"; CMD2="$CMD"; else CMD="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu $ARGS"; CMD2="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu $ARGS"; fi
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- $CMD2 > $OUT/pmc_$N.log 2>&1
done
cd $ROOT
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
