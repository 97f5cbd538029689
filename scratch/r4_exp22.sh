#!/bin/bash
# round 4, experiment 22: the top block's attention backward told that only the cls query carries a gradient (PM_ATTN_LIVE)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "attention" 2>&1 | tail -2
timeout -k 10 600 python -m pytest tests/test_gpu_models.py -q -m gpu -x -k "top_block or freeze or grad_accum or block_" 2>&1 | tail -2
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-mae"
for rep in 1 2 3; do
for lv in 0 1; do
  PM_ATTN_LIVE=$lv timeout -k 10 300 python bench.py $B > gpurun_out/r4_exp22_tmp.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r4_exp22_tmp.json')); c=d['config']; print('attn_live $lv rep $rep: cls', d['value'], 'img/s', d['ms_per_step'], 'ms; head+1', c['finetune_head_plus_1_img_s'], 'head+2', c['finetune_head_plus_2_img_s'])"
done; done | tee gpurun_out/r4_exp22_step.txt
