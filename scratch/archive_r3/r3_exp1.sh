#!/bin/bash
# round-3 experiment batch 1 (GPU box): grouped-wgrad tile order A/B, MFMA-shape timing on the grouped kernel, cfg sweep at the MAE shapes
echo "== wgroup, auto order =="; python scratch/bench_wgroup.py 2>&1 | grep -v amdgpu
echo "== wgroup, row order (PM_GROUP_ORDER=0) =="; PM_GROUP_ORDER=0 python scratch/bench_wgroup.py 2>&1 | grep -v amdgpu
bash scratch/build_alt.sh "-DPM_MFMA16_TIMING" > /dev/null 2>&1
echo "== wgroup, 16x16x32 timing build (results wrong by design) =="; POLYPMAE_LIB=$PWD/ssl4polyp_amd/lib/libpolypmae_alt.so python scratch/bench_wgroup.py 2>&1 | grep -v amdgpu
rm -rf ssl4polyp_amd/lib/libpolypmae_alt.so ssl4polyp_amd/lib/obj_libpolypmae_alt.so
echo "== cfg sweep MAE decoder M=50432 D=512 =="; M=50432 D=512 CFGS=0,6,8,9,10,24,25,26 python scratch/bench_gemm6.py 2>&1 | grep -v amdgpu | tail -8
echo "== cfg sweep MAE encoder M=12800 D=768 =="; M=12800 D=768 CFGS=0,6,8,9,10,24,25,26 python scratch/bench_gemm6.py 2>&1 | grep -v amdgpu | tail -8
