#!/bin/bash
# round 4, trip 21: patch-embedding path at a padded 16-B patch size; the pre-training CLI with the ViT-H factory (2 synthetic epochs)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "patch_embed or mae_loss" 2>&1 | tail -2
timeout -k 10 600 python -m ssl4polyp_amd.main_pretrain --model mae_vit_huge_patch14 --synthetic 8 --epochs 2 --batch_size 8 --output_dir gpurun_out/r4_t21_vith_cli 2>&1 | grep -v amdgpu.ids | tail -6
ls gpurun_out/r4_t21_vith_cli 2>/dev/null | head; rm -rf gpurun_out/r4_t21_vith_cli
