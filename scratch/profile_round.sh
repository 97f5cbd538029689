#!/bin/bash
# usage (on the GPU box, from the repo root): bash scratch/profile_round.sh <tag> [bench args...]
# kernel-trace stats + two PMC passes (FETCH_SIZE / WRITE_SIZE separately, as MI355X_MICROARCH.md prescribes) of bench.py
TAG=$1; shift
REPO=$PWD; OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
grep "^{\"metric\"" $OUT/stats.log > $OUT/bench.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-stats "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w --output-format csv -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-stats "$@" > $OUT/write.log 2>&1
ls $OUT/stats $OUT/fetch $OUT/write
