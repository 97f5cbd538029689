#!/bin/bash
# round 4, experiment 15: Phi(x) as a degree-10 polynomial on the clamped argument in the 16-bit GELU / dGELU epilogues (default) against the
# erf form (-DPM_GELU_EXACT alt library)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ALT=$GRAFT_REPO_ROOT/ssl4polyp_amd/lib/libpolypmae_alt.so
for lib in "" "$ALT"; do
  echo "== ${lib:-default (polynomial)}"
  POLYPMAE_LIB=$lib MS=3200,6304,12608 timeout -k 10 200 python scratch/bench_gemm_smallm.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4_exp15_standalone.txt
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fp16.py tests/test_gpu_parity_large.py -q -m gpu -x -s 2>&1 | grep -E "parity|passed|failed|rror" | tee gpurun_out/r4_exp15_tests.txt | tail -30
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-c5 --no-mae"
for rep in 1 2; do
for lib in "" "$ALT"; do
  for wl in "cls 64" "mae 256" "mae 64"; do
    set -- $wl
    POLYPMAE_LIB=$lib timeout -k 10 200 python bench.py --workload $1 --batch $2 $B > gpurun_out/r4_exp15_tmp.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r4_exp15_tmp.json')); print('${lib:+exact}${lib:-poly} rep $rep $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
  POLYPMAE_LIB=$lib timeout -k 10 200 python scratch/bench_huge.py --model mae_vit_huge_patch14 --batch 64 2>&1 | grep "ms/step" | sed "s/^/${lib:+exact}${lib:-poly} rep $rep /"
done; done | tee gpurun_out/r4_exp15_step.txt
