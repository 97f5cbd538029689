#!/bin/bash
# round 4, experiment 19: the fused attention backward walking several heads per workgroup (PM_ATTN_BWD_WALK=1)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
PM_ATTN_BWD_WALK=2 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "attention" 2>&1 | tail -3 | tee gpurun_out/r4_exp19_tests.txt
grep -q failed gpurun_out/r4_exp19_tests.txt && exit 1
PM_ATTN_BWD_WALK=0 timeout -k 10 100 python scratch/attn_bwd_hash.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_exp19_hash0.txt || exit 1
PM_ATTN_BWD_WALK=2 timeout -k 10 100 python scratch/attn_bwd_hash.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_exp19_hash1.txt || exit 1
cat gpurun_out/r4_exp19_hash0.txt; cmp gpurun_out/r4_exp19_hash0.txt gpurun_out/r4_exp19_hash1.txt && echo "walk == fused, bit for bit" || { echo "MISMATCH"; cat gpurun_out/r4_exp19_hash1.txt; exit 1; }
for w in 0 1; do
  echo "== PM_ATTN_BWD_WALK=$w"
  PM_ATTN_BWD_WALK=$w timeout -k 10 200 python scratch/bench_attn.py 2>&1 | grep -v amdgpu.ids | tail -12
done | tee gpurun_out/r4_exp19_standalone.txt
PM_ATTN_BWD_WALK=1 timeout -k 10 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity_large.py -q -m gpu -x -k "cls or classifier or freeze" 2>&1 | tail -2
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-c5 --no-mae"
for rep in 1 2 3; do
for w in 0 1; do
  PM_ATTN_BWD_WALK=$w timeout -k 10 200 python bench.py --workload cls --batch 64 $B > gpurun_out/r4_exp19_tmp.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r4_exp19_tmp.json')); print('walk $w rep $rep cls bs64:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
done; done | tee gpurun_out/r4_exp19_step.txt
