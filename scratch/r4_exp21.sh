#!/bin/bash
# round 4, experiment 21: the classifier's top block on its cls rows (PM_SPARSE_TOP, default on) against the dense backward
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_models.py -q -m gpu -x -s -k "top_block or freeze or grad_accum" 2>&1 | grep -E "measured\] sparse|passed|failed|rror" | tail -14
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-mae"
for rep in 1 2 3; do
for sp in 0 1; do
  PM_SPARSE_TOP=$sp timeout -k 10 300 python bench.py $B > gpurun_out/r4_exp21_tmp.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r4_exp21_tmp.json')); c=d['config']; print('sparse_top $sp rep $rep: cls', d['value'], 'img/s', d['ms_per_step'], 'ms; head+1', c['finetune_head_plus_1_img_s'], 'head+2', c['finetune_head_plus_2_img_s'], 'probe', c['finetune_none_img_s'])"
done; done | tee gpurun_out/r4_exp21_step.txt
