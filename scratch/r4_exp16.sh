#!/bin/bash
# round 4, experiment 16: epilogue loads of the wave's whole tile issued up front (f32 residual: next row block in flight; dGELU: all
# pre-activations before the LDS staging) -- PM_EPI_HOIST=1 (default) against 0
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for hz in 0 1; do
  echo "== PM_EPI_HOIST=$hz"
  PM_EPI_HOIST=$hz MS=3200,6304,12608 timeout -k 10 200 python scratch/bench_gemm_smallm.py 2>&1 | grep -v amdgpu.ids
  PM_EPI_HOIST=$hz timeout -k 10 200 python scratch/bench_gemm_deadtime.py 2>&1 | grep -E "resid|back-to-back"
done | tee gpurun_out/r4_exp16_standalone.txt
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_schedule.py tests/test_gpu_parity_large.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/r4_exp16_tests.txt
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-c5 --no-mae"
for rep in 1 2; do
for hz in 0 1; do
  for wl in "cls 64" "mae 256" "mae 64"; do
    set -- $wl
    PM_EPI_HOIST=$hz timeout -k 10 200 python bench.py --workload $1 --batch $2 $B > gpurun_out/r4_exp16_tmp.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r4_exp16_tmp.json')); print('hoist $hz rep $rep $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
  PM_EPI_HOIST=$hz timeout -k 10 200 python scratch/bench_huge.py --model mae_vit_huge_patch14 --batch 64 2>&1 | grep "ms/step" | cut -c1-90 | sed "s|^|hoist $hz rep $rep |"
done; done | tee gpurun_out/r4_exp16_step.txt
