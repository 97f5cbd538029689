#!/bin/bash
# round 4, experiment 16b: the 128 x 128 kernel's epilogue with all sixteen loads of a wave tile in one batch (PM_EPI_HOIST=1) against per-vector chains
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for hz in 0 1; do
  echo "== PM_EPI_HOIST=$hz"
  PM_EPI_HOIST=$hz MS=1600,3200 timeout -k 10 200 python scratch/bench_gemm_smallm.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4_exp16b_standalone.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm or linear" 2>&1 | tail -2
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-c5 --no-mae"
for rep in 1 2 3; do
for hz in 0 1; do
  PM_EPI_HOIST=$hz timeout -k 10 200 python bench.py --workload mae --batch 64 $B > gpurun_out/r4_exp16_tmp.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r4_exp16_tmp.json')); print('hoist $hz rep $rep mae bs64:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
done; done | tee gpurun_out/r4_exp16b_step.txt
