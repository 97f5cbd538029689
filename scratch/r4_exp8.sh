#!/bin/bash
# round 4, experiment 8: hipGraph replay of the whole step at the small batch (MAE bs = 64/GPU: 576 kernels in 10 ms, host enqueue 5.5-7.4 ms)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-c5 --no-mae"
for rep in 1 2; do
for g in off on; do
  for wl in "mae 64" "mae 256" "cls 64"; do
    set -- $wl
    timeout -k 10 300 python bench.py --workload $1 --batch $2 --graph $g $B > gpurun_out/r4_exp8_tmp.json 2> gpurun_out/r4_exp8_tmp.err || { tail -5 gpurun_out/r4_exp8_tmp.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/r4_exp8_tmp.json')); print('graph $g rep $rep $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms, host enqueue', d['host_enqueue_ms_per_step'], 'loss', d['config']['final_loss'])"
  done
done; done | tee gpurun_out/r4_exp8_graph.txt
