#!/bin/bash
# kernel timeline around the step boundary (where is the GPU idle?): usage bash scratch/trace_boundary.sh [bench args]
REPO=$PWD; OUT=$REPO/gpurun_out/trace_b; mkdir -p $OUT
B="--no-cpu-baseline --no-parity --no-torch-baseline --no-mae --no-c5 --preheat 0.3 --no-kernel-stats"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace -o t -- python3 $REPO/bench.py --steps 6 --warmup 3 $B "$@" > $OUT/trace.log 2>&1
cd $REPO
python3 scratch/trace_timeline.py $OUT/trace/t_results.db -2 0,1.4 > $OUT/head.txt 2>&1
python3 scratch/trace_timeline.py $OUT/trace/t_results.db -2 9.6,13 | sed -n '/---- gaps/,$p' > $OUT/tail.txt 2>&1
rm -rf $OUT/trace
