#!/bin/bash
# run on the GPU box via gpurun: kernel-trace profile of bench.py, summaries land in gpurun_out/
set -x
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 0 $2 > $OUT/bench.log 2>&1
tail -2 $OUT/bench.log
find $OUT -name "*kernel_stats*.csv" | head -3
f=$(find $OUT -name "*kernel_stats*.csv" | head -1)
head -25 "$f"
# keep only the small summaries (the per-dispatch trace is large)
find $OUT -name "*kernel_trace*.csv" -size +20M -delete
