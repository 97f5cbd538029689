#!/bin/bash
# usage: bash scratch/trace.sh <outdir> <tag> [ENV=..]... -- [bench args]   kernel-trace + stats + per-stream timeline of one step
O=$PWD/$1; TAG=$2; shift 2; mkdir -p $O
ENVS=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do ENVS+=("$1"); shift; done; shift
R=$PWD
for e in "${ENVS[@]}"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace_$TAG -o t -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-stats --no-parity --no-torch-baseline --no-mae "$@" > $O/trace_$TAG.log 2>&1
cd $R
python3 scratch/trace_timeline.py $O/trace_$TAG/t_results.db -2 > $O/timeline_$TAG.txt 2>&1
head -24 $O/timeline_$TAG.txt
