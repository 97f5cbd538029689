#!/bin/bash
# round 4, experiment 18: the 128 x 128 kernel with three buffers and a counted wait for launches of at most one workgroup per CU
# (PM_GLDS_RING = largest such grid; 0 = off)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for r in 0 256 512; do
  echo "== PM_GLDS_RING=$r"
  PM_GLDS_RING=$r MS=1600,3200 timeout -k 10 200 python scratch/bench_gemm_smallm.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4_exp18_standalone.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_schedule.py -q -m gpu -x 2>&1 | tail -2
B="--steps 60 --warmup 10 --no-cpu-baseline --no-torch-baseline --no-parity --no-kernel-stats --no-fp16 --no-fp32 --no-c5 --no-mae"
for rep in 1 2 3; do
for r in 0 256 512; do
  for wl in "mae 64" "cls 64"; do
    set -- $wl
    PM_GLDS_RING=$r timeout -k 10 200 python bench.py --workload $1 --batch $2 $B > gpurun_out/r4_exp18_tmp.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/r4_exp18_tmp.json')); print('ring $r rep $rep $1 bs$2:', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done; done | tee gpurun_out/r4_exp18_step.txt
