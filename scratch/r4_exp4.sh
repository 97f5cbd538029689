#!/bin/bash
# round 4, experiment 4: the few-tiles dispatch rule (128 x 128 kernel below 128 tiles of 256 x 256) in the step: cls bs=64, MAE bs=256, MAE bs=64,
# same box, PM_FEW_TILES=0 (round-3 dispatch) against the default, two rounds; then the op tests that cover both kernels
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-c5 --no-mae"
for rep in 1 2; do
for ft in 0 128; do
  for wl in "cls 64" "mae 256" "mae 64"; do
    set -- $wl
    PM_FEW_TILES=$ft timeout -k 10 200 python bench.py --workload $1 --batch $2 $B > gpurun_out/r4_exp4_tmp.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r4_exp4_tmp.json')); print('few_tiles $ft rep $rep $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done; done | tee gpurun_out/r4_exp4_few_tiles.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -m gpu -x 2>&1 | tail -3
