#!/bin/bash
# round 4, experiment 12: the residual / dGELU epilogue's operand tile pulled into the L2 at tile start by LDS-DMA into scratch (PM_EPI_WARM)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for w in 0 1; do
  echo "== PM_EPI_WARM=$w"
  PM_EPI_WARM=$w MS=3200,6304,12608 timeout -k 10 200 python scratch/bench_gemm_smallm.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4_exp12_standalone.txt
PM_EPI_WARM=1 timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -m gpu -x 2>&1 | tail -2
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-c5 --no-mae"
for rep in 1 2; do
for w in 0 1; do
  for wl in "cls 64" "mae 256" "mae 64"; do
    set -- $wl
    PM_EPI_WARM=$w timeout -k 10 200 python bench.py --workload $1 --batch $2 $B > gpurun_out/r4_exp12_tmp.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r4_exp12_tmp.json')); print('warm $w rep $rep $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done; done | tee gpurun_out/r4_exp12_step.txt
