#!/bin/bash
# round 4, trip 11: step time of the larger factories (ViT-L/16, ViT-H/14 MAE pre-train at bs = 64 / GPU) + a kernel trace of the ViT-H step
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout -k 10 200 python scratch/bench_huge.py --model mae_vit_huge_patch14 --batch 64 --precision bf16 || exit 1
timeout -k 10 200 python scratch/bench_huge.py --model mae_vit_huge_patch14 --batch 64 --precision fp16 || exit 1
timeout -k 10 200 python scratch/bench_huge.py --model mae_vit_huge_patch14 --batch 16 --precision fp32 --steps 5 --warmup 2 || exit 1
timeout -k 10 200 python scratch/bench_huge.py --model mae_vit_large_patch16 --batch 64 --precision bf16 || exit 1
timeout -k 10 200 python scratch/bench_huge.py --model mae_vit_base_patch16 --batch 64 --precision bf16 || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_t11_factories.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r4_vith -o vith --output-format csv -- python3 $R/scratch/bench_huge.py --model mae_vit_huge_patch14 --batch 64 --precision bf16 --steps 5 --warmup 3 > $R/gpurun_out/r4_t11_prof.log 2>&1 || exit 1
cp $R/gpurun_out/prof_r4_vith/*kernel_stats.csv $R/gpurun_out/r4_t11_vith_kernel_stats.csv
rm -rf $R/gpurun_out/prof_r4_vith
head -25 $R/gpurun_out/r4_t11_vith_kernel_stats.csv | cut -c1-200
