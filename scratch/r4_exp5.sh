#!/bin/bash
# round 4, experiment 5: the few-tiles rule with a cap on M (4 096: small chains only; 7 000: + half-batch decoder / cls chains; 16 384: + full-batch decoder dgrads)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-c5 --no-mae"
for rep in 1 2; do
for cap in 0 4096 7000 16384; do
  for wl in "mae 64" "cls 64"; do
    set -- $wl
    [ "$1" = cls ] && [ $cap = 7000 -o $cap = 16384 ] && continue
    ft=128; [ $cap = 0 ] && ft=0
    PM_FEW_TILES=$ft PM_FEW_TILES_MAXM=$cap timeout -k 10 200 python bench.py --workload $1 --batch $2 $B > gpurun_out/r4_exp5_tmp.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r4_exp5_tmp.json')); print('cap $cap rep $rep $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done; done | tee gpurun_out/r4_exp5_few_tiles_cap.txt
