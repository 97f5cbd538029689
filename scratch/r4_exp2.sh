#!/bin/bash
# round 4, experiment 2: MAE at the metric's own batch (bs = 64/GPU: encoder M = 3 200, decoder M = 12 608) -- is the two-chain forward
# split still right when a chain holds 1 600 rows?  + a timeline of that step
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="--workload mae --batch 64 --steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats"
for rep in 1 2; do
for split in 2 1; do
  PM_SPLIT_FWD=$split timeout -k 10 200 python bench.py $B > gpurun_out/r4_exp2_split${split}_${rep}.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r4_exp2_split${split}_${rep}.json')); print('split $split rep $rep:', d['value'], 'img/s', d['ms_per_step'], 'ms, host enqueue', d['host_enqueue_ms_per_step'])"
done; done | tee gpurun_out/r4_exp2_mae64_split.txt
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $REPO/gpurun_out/prof_r4_mae64 -o t -- python3 $REPO/bench.py --workload mae --batch 64 --steps 6 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --preheat 0.3 > $REPO/gpurun_out/r4_exp2_trace.log 2>&1
cd $REPO
python3 scratch/trace_timeline.py gpurun_out/prof_r4_mae64/t_results.db -2 > gpurun_out/r4_exp2_mae64_timeline.txt 2>&1
rm -rf gpurun_out/prof_r4_mae64
head -30 gpurun_out/r4_exp2_mae64_timeline.txt
