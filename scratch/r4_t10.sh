#!/bin/bash
# round 4, trip 10: first run of the ViT-H/14 shapes (D = 1280 LayerNorm, 80-wide heads, N = 257, padded 588-element patch) + experiment 12
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "layernorm or attention or patch_embed or mae_loss" > gpurun_out/r4_t10_ops.log 2>&1; rc=$?
tail -5 gpurun_out/r4_t10_ops.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_parity_large.py -q -m gpu -x -s -k "huge or large_factory" > gpurun_out/r4_t10_huge.log 2>&1; rc=$?
grep -E "parity|passed|failed|Error|error" gpurun_out/r4_t10_huge.log | tail -20
[ $rc -eq 0 ] || exit $rc
bash scratch/r4_exp12.sh
