#!/bin/bash
# round 4, experiment 23: the classifier's top block behind its attention on the cls rows only, FORWARD and backward (PM_SPARSE_TOP)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_models.py -q -m gpu -x -s -k "top_block" 2>&1 | grep -E "measured\] cls-row|passed|failed|rror|assert" | tail -14
timeout -k 10 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity_large.py tests/test_gpu_fp16.py tests/test_gpu_parallel.py tests/test_gpu_schedule.py -q -m gpu -x 2>&1 | tail -3
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-mae"
for rep in 1 2 3; do
for sp in 0 1; do
  PM_SPARSE_TOP=$sp timeout -k 10 300 python bench.py $B > gpurun_out/r4_exp23_tmp.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r4_exp23_tmp.json')); c=d['config']; print('cls_top $sp rep $rep: cls', d['value'], 'img/s', d['ms_per_step'], 'ms; head+1', c['finetune_head_plus_1_img_s'], 'head+2', c['finetune_head_plus_2_img_s'], 'probe', c['finetune_none_img_s'], 'eval', c['finetune_none_eval_img_s'])"
done; done | tee gpurun_out/r4_exp23_step.txt
