#!/bin/bash
# Which clock does the chip hold on the ring GEMM k-loops with v_mfma_f32_16x16x32_bf16 instead of 32x32x16?  (timing only:
# the -DPM_MFMA16_TIMING build computes garbage).  usage: bash scratch/mfma16_timing.sh   (on the GPU box)
set -e
bash scratch/build_alt.sh "-DPM_MFMA16_TIMING"
echo "== shipped (32x32x16) =="; CFGS=0 python scratch/bench_gemm6.py 2>&1 | grep -v amdgpu | tail -8
echo "== 16x16x32 timing build =="; CFGS=0 POLYPMAE_LIB=$PWD/ssl4polyp_amd/lib/libpolypmae_alt.so python scratch/bench_gemm6.py 2>&1 | grep -v amdgpu | tail -8
echo "== shipped again =="; CFGS=0 python scratch/bench_gemm6.py 2>&1 | grep -v amdgpu | tail -8
rm -rf ssl4polyp_amd/lib/libpolypmae_alt.so ssl4polyp_amd/lib/obj_libpolypmae_alt.so
