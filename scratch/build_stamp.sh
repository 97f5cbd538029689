#!/bin/bash
# diagnostic library with in-kernel stamps (never shipped / never loaded by the package by default)
cd "$(dirname "$0")/../ssl4polyp_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -DPM_GEMM_STAMP \
  -o ../lib/libpolypmae_stamp.so pm_gemm.hip pm_attention.hip pm_layernorm.hip pm_misc.hip pm_mae.hip
