#!/bin/bash
# round 4, experiment 9: the few-tiles rule's narrow-N clause (N <= 512 ignores the cap on M) -- MAE bs = 64 / 256 and cls, same box
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-c5 --no-mae"
for rep in 1 2; do
for nn in 0 512; do
  for wl in "mae 64" "mae 256" "cls 64"; do
    set -- $wl
    PM_FEW_TILES_NARROW_N=$nn timeout -k 10 200 python bench.py --workload $1 --batch $2 $B > gpurun_out/r4_exp9_tmp.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r4_exp9_tmp.json')); print('narrow_n $nn rep $rep $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done; done | tee gpurun_out/r4_exp9_few_tiles_narrow.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_parity_large.py -q -m gpu -x 2>&1 | tail -3
