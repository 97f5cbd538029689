#!/bin/bash
# round 4, experiment 14: dead time of a ring-GEMM tile (K = 64 against K = 768 / 3072) per epilogue
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 200 python scratch/bench_gemm_deadtime.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_exp14_gemm_deadtime.txt
