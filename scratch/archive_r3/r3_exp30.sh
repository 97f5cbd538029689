#!/bin/bash
# kernel trace of the forced world-1 data-parallel step under the final stream layout (which queue does each stream run on?)
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 6 --warmup 3 --preheat 0.3 --force-sync"
REPO=$PWD; OUT=$REPO/gpurun_out/prof_fs2; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o t -- python3 $REPO/bench.py $F > $OUT/trace.log 2>&1
cd $REPO; python3 scratch/trace_timeline.py $OUT/trace/t_results.db -2 > gpurun_out/r3_exp30_timeline.txt 2>&1; rm -rf $OUT/trace
