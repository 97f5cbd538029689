// pm_gemm.hip -- MFMA GEMM with fused epilogues for every Linear / conv-as-GEMM of the ViT path and for
// their dgrad / wgrad (reference: timm Attention.qkv/proj, Mlp.fc1/fc2, PatchEmbed.proj, decoder_embed,
// decoder_pred -- models_mae.py:33,39-41,48,53-58; autograd backward engine_pretrain.py:65, tc.py:4533).
//
//   acc[m,n] = sum_k X(m,k) * W(n,k)        X = "A" matrix (M side), W = "B" matrix (N side)
//
// Tile: 128(m) x 128(n) x 128 BYTES of k (64 bf16 / 32 f32) per step, 256 threads = 4 wave64 in 2x2, each
// wave owns 64x64 = 2x2 accumulators of v_mfma_f32_32x32x16_bf16 (or 4x v_mfma_f32_32x32x2_f32 per 16-B
// fragment in f32 mode: exact f32 fma chain).  The W fragment is the MFMA's first operand, so the
// accumulator has n in registers (4 consecutive n per register quad) and m on the lane.
//
// Operand storage (either side independently):
//   k-normal  [R][K]: LDS image 128 rows x 128 B, 16-B chunk c of row r at c ^ ((r>>1)&7)  -> ds_read_b128,
//                     conflict-free for the 32x32x16 operand map (lane (r,h) reads chunk 2*kk+h of row r).
//   k-major   [K][R]: bf16: LDS image 64 k-rows x 256 B, chunk c of row k at c ^ ((k&3)<<2), fragments by
//                     ds_read_b64_tr_b16 (hardware transpose read); f32: 32 k-rows x 512 B, ds_read_b32.
//   -> forward: X k-normal, W k-normal;  dgrad: dY k-normal, W (as stored [out][in]) k-major;
//      wgrad: dY k-major (M side = out features), X k-major (N side = in features).
//
// Two kernels share the LDS images and the fragment readers:
//   gemm_glds_kernel (fast path, K % k-step == 0): global -> LDS by LDS-DMA (global_load_lds_dwordx4; the LDS
//     image is lane-linear, so the swizzle is applied to each lane's SOURCE address), double buffered, one
//     barrier per k-step, no staging registers and no ds_write; rows beyond M / N are clamped to the last
//     valid row (their results are never stored).  Optional split-K over blockIdx.y writes f32 partial slabs
//     that splitk_reduce_kernel sums in a fixed order (deterministic; used by wgrad where K = #tokens).
//     Epilogue: each wave stages its 64x64 accumulators through LDS and leaves as whole 16-B vectors along n
//     (bias / GELU / dGELU / residual applied on the way out) -> fully coalesced stores.
//   gemm_generic_kernel (any K % chunk == 0): register-staged, fully predicated loads (tiny / odd shapes).
// Roofline: MFMA-bound; algorithmic FLOPs 2*M*N*K.
#include "pm_common.h"

namespace {

constexpr int BM = 128, BN = 128, KB = 128;  // KB: bytes of k per row per step
constexpr int TILE_BYTES = 128 * KB;         // 16 KiB per side per buffer
constexpr int kThreads = 256;
constexpr int STAGE_ROW = 64 * 4 + 16;       // epilogue staging: 64 f32 per row + 16 B pad (bank spread)
constexpr int STAGE_BYTES = 64 * STAGE_ROW;  // per wave
constexpr int GLDS_LDS_BYTES = (4 * STAGE_BYTES > 4 * TILE_BYTES) ? 4 * STAGE_BYTES : 4 * TILE_BYTES;

struct GemmArgs {
  const void* X;
  const void* W;
  long ldx, ldw;
  const float* bias;
  void* C;
  long ldc;
  void* aux;
  const float* resid;
  int M, N, K;
  int epilogue;
  int c_dtype;
  int tiles_m, tiles_n;
  int split_k;       // >1: blockIdx.y = split, C = f32 slabs [split][M][ldc]
  int ksteps_split;  // k-steps per split
};

// ---- global -> register staging (generic kernel) ---------------------------------------------------
template <typename T, bool KMAJOR>
__device__ __forceinline__ void stage_load(u32x4 (&regs)[4], const T* __restrict__ base, long ld, int r0, int R,
                                           int k0, int K, int tid) {
  constexpr int EPC = 16 / sizeof(T);  // elements per chunk
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + kThreads * i;
    u32x4 v = {0u, 0u, 0u, 0u};
    if constexpr (!KMAJOR) {
      const int row = id >> 3, c = id & 7;
      const int gr = r0 + row, gk = k0 + c * EPC;
      if (gr < R && gk < K) v = *reinterpret_cast<const u32x4*>(base + (long)gr * ld + gk);
    } else {
      constexpr int CPR = 128 / EPC;  // chunks per k-row (16 bf16 / 32 f32)
      const int krow = id / CPR, c = id % CPR;
      const int gk = k0 + krow, gr = r0 + c * EPC;
      if (gk < K && gr < R) v = *reinterpret_cast<const u32x4*>(base + (long)gk * ld + gr);
    }
    regs[i] = v;
  }
}

template <typename T, bool KMAJOR>
__device__ __forceinline__ void stage_store(const u32x4 (&regs)[4], char* tile, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + kThreads * i;
    int off;
    if constexpr (!KMAJOR) {
      const int row = id >> 3, c = id & 7;
      off = row * 128 + 16 * (c ^ ((row >> 1) & 7));
    } else if constexpr (sizeof(T) == 2) {
      const int krow = id >> 4, c = id & 15;
      off = krow * 256 + 16 * (c ^ ((krow & 3) << 2));
    } else {
      off = id * 16;  // [32 k][128 r] f32, linear
    }
    *reinterpret_cast<u32x4*>(tile + off) = regs[i];
  }
}

// ---- global -> LDS by LDS-DMA (fast kernel) -------------------------------------------------------
// One side's 16-KiB tile = 16 wave-instructions of 1 KiB (lane l lands at tile + 1024*j + 16*l); wave w issues
// j = w, w+4, w+8, w+12.  The lane's SOURCE chunk is the inverse swizzle of its landing slot.
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T, bool KMAJOR>
__device__ __forceinline__ void stage_glds(char* tile, const T* __restrict__ base, long ld, int r0, int R, int k0,
                                           int wave, int lane) {
  constexpr int EPC = 16 / sizeof(T);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = wave + 4 * i;
    const T* src;
    if constexpr (!KMAJOR) {
      const int row = 8 * j + (lane >> 3), cs = lane & 7;
      const int c = cs ^ ((row >> 1) & 7);
      int gr = r0 + row;
      gr = gr < R ? gr : R - 1;  // clamp: rows beyond R feed outputs that are never stored
      src = base + (long)gr * ld + k0 + c * EPC;
    } else if constexpr (sizeof(T) == 2) {
      const int krow = 4 * j + (lane >> 4), cs = lane & 15;
      const int c = cs ^ ((krow & 3) << 2);
      int gc = r0 + c * EPC;
      gc = gc < R ? gc : R - EPC;
      src = base + (long)(k0 + krow) * ld + gc;
    } else {
      const int krow = 2 * j + (lane >> 5), c = lane & 31;
      int gc = r0 + c * EPC;
      gc = gc < R ? gc : R - EPC;
      src = base + (long)(k0 + krow) * ld + gc;
    }
    glds16(src, tile + 1024 * j);
  }
}

// ---- LDS -> MFMA operand fragment -----------------------------------------------------------
template <typename T, bool KMAJOR>
__device__ __forceinline__ Frag16 read_frag(const char* tile, int rb, int kk, int lane) {
  Frag16 f;
  if constexpr (!KMAJOR) {
    const int row = rb + (lane & 31);
    const int c = 2 * kk + (lane >> 5);
    f.u = *reinterpret_cast<const u32x4*>(tile + row * 128 + 16 * (c ^ ((row >> 1) & 7)));
  } else if constexpr (sizeof(T) == 2) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int ch = (rb >> 3) + 2 * (g & 1) + (p >> 1);
    const int sw = 16 * (ch ^ (q << 2)) + 8 * (p & 1);
    const int kbase = kk * 16 + 8 * (g >> 1) + q;
    using lds_s4 = __attribute__((address_space(3))) short4v;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + kbase * 256 + sw));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + (kbase + 4) * 256 + sw));
    f.u[0] = ((unsigned)(unsigned short)lo[0]) | (((unsigned)(unsigned short)lo[1]) << 16);
    f.u[1] = ((unsigned)(unsigned short)lo[2]) | (((unsigned)(unsigned short)lo[3]) << 16);
    f.u[2] = ((unsigned)(unsigned short)hi[0]) | (((unsigned)(unsigned short)hi[1]) << 16);
    f.u[3] = ((unsigned)(unsigned short)hi[2]) | (((unsigned)(unsigned short)hi[3]) << 16);
  } else {
    const int r = rb + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int e = 0; e < 4; ++e) f.f[e] = *reinterpret_cast<const float*>(tile + (8 * kk + 4 * h + e) * 512 + 4 * r);
  }
  return f;
}

template <typename T, bool XK, bool WK>
__device__ __forceinline__ void mma_kstep(const char* bx, const char* bw, int wm, int wn, int lane, f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    Frag16 fw[2], fx[2];
    fw[0] = read_frag<T, WK>(bw, wn * 64, kk, lane);
    fw[1] = read_frag<T, WK>(bw, wn * 64 + 32, kk, lane);
    fx[0] = read_frag<T, XK>(bx, wm * 64, kk, lane);
    fx[1] = read_frag<T, XK>(bx, wm * 64 + 32, kk, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma16B<T>(fw[i], fx[j], acc[i][j]);
  }
}

// XCD-aware tile order: the 8 XCDs take blocks round-robin; give each XCD a contiguous run of tiles so
// that neighbours (same X panel, consecutive W panels) hit the same private L2.  Bijective for any grid.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + (bid >> 3);
}

// Epilogue on 4 consecutive n of row m (v = acc + bias already applied by the caller where relevant).
template <typename T>
__device__ __forceinline__ void epilogue4(const GemmArgs& a, int epi, long off, f32x4 v) {
  if (epi == PM_EPI_RESIDUAL) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(a.resid + off);
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.C) + off) = r + v;
  } else if (epi == PM_EPI_ACCUM) {
    float* c = reinterpret_cast<float*>(a.C) + off;
    *reinterpret_cast<f32x4*>(c) = *reinterpret_cast<const f32x4*>(c) + v;
  } else {
    if (epi == PM_EPI_GELU) {
      store4<T>(reinterpret_cast<T*>(a.aux) + off, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_erf(to_f32<T>(from_f32<T>(v[e])));  // the value backward will see
    } else if (epi == PM_EPI_DGELU) {
      const f32x4 pre = load4<T>(reinterpret_cast<const T*>(a.aux) + off);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(pre[e]);
    }
    if (a.c_dtype == PM_F32)
      store4<float>(reinterpret_cast<float*>(a.C) + off, v);
    else
      store4<__bf16>(reinterpret_cast<__bf16*>(a.C) + off, v);
  }
}

// ------------------------------------------------------------------------------------------------
// fast path
// ------------------------------------------------------------------------------------------------
template <typename T, bool XK, bool WK>
__global__ __launch_bounds__(kThreads, 2) void gemm_glds_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][X tile | W tile]; reused for staging
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / a.tiles_n, tn = tile % a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  constexpr int KE = KB / sizeof(T);  // k elements per step
  const int nk_total = a.K / KE;
  const int kbeg = blockIdx.y * a.ksteps_split;
  const int kend = (kbeg + a.ksteps_split) < nk_total ? (kbeg + a.ksteps_split) : nk_total;
  const T* X = reinterpret_cast<const T*>(a.X);
  const T* W = reinterpret_cast<const T*>(a.W);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kbeg < kend) {
    stage_glds<T, XK>(smem, X, a.ldx, m0, a.M, kbeg * KE, wave, lane);
    stage_glds<T, WK>(smem + TILE_BYTES, W, a.ldw, n0, a.N, kbeg * KE, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = kbeg; t < kend; ++t) {
      const int cur = (t - kbeg) & 1;
      const char* bx = smem + cur * 2 * TILE_BYTES;
      if (t + 1 < kend) {
        char* nb = smem + (cur ^ 1) * 2 * TILE_BYTES;
        // EXP no glds
        // EXP no glds
      }
      mma_kstep<T, XK, WK>(bx, bx + TILE_BYTES, wm, wn, lane, acc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue: stage this wave's 64(m) x 64(n) accumulators through LDS, leave as 16-B vectors along n ----
  char* st = smem + wave * STAGE_BYTES;  // all waves are past the last barrier: tiles are dead
  const int h = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ml = j * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        *reinterpret_cast<f32x4*>(st + ml * STAGE_ROW + (i * 32 + 8 * g + 4 * h) * 4) = v;
      }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own LDS writes
  __builtin_amdgcn_wave_barrier();
  const bool split = a.split_k > 1;
  const int epi = split ? PM_EPI_STORE : a.epilogue;
  float* slab = split ? reinterpret_cast<float*>(a.C) + (long)blockIdx.y * a.M * a.ldc : nullptr;
  const int mw = m0 + wm * 64, nw = n0 + wn * 64;
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int ml = it * 4 + (lane >> 4), c4 = (lane & 15) * 4;
    const int m = mw + ml, n = nw + c4;
    if (m >= a.M || n >= a.N) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(st + ml * STAGE_ROW + c4 * 4);
    const long off = (long)m * a.ldc + n;
    if (split) {
      *reinterpret_cast<f32x4*>(slab + off) = v;
      continue;
    }
    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
    epilogue4<T>(a, epi, off, v);
  }
}

// out[m][n] = (accumulate ? out : 0) + sum_s slab[s][m][n], fixed order (deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                            long ldc, int M, int N, int splits, int accumulate) {
  const long nvec = (long)M * (N >> 2);
  const long slab_stride = (long)M * ldc;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    const int m = i / (N >> 2), n = (i % (N >> 2)) * 4;
    const long off = (long)m * ldc + n;
    f32x4 v = accumulate ? *reinterpret_cast<const f32x4*>(out + off) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s) v += *reinterpret_cast<const f32x4*>(slabs + s * slab_stride + off);
    *reinterpret_cast<f32x4*>(out + off) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// generic path (predicated, register staged)
// ------------------------------------------------------------------------------------------------
template <typename T, bool XK, bool WK>
__global__ __launch_bounds__(kThreads, 2) void gemm_generic_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][X tile | W tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / a.tiles_n, tn = tile % a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  constexpr int KE = KB / sizeof(T);  // k elements per step
  const int nk = (a.K + KE - 1) / KE;
  const T* X = reinterpret_cast<const T*>(a.X);
  const T* W = reinterpret_cast<const T*>(a.W);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 rx[4], rw[4];
  stage_load<T, XK>(rx, X, a.ldx, m0, a.M, 0, a.K, tid);
  stage_load<T, WK>(rw, W, a.ldw, n0, a.N, 0, a.K, tid);
  stage_store<T, XK>(rx, smem, tid);
  stage_store<T, WK>(rw, smem + TILE_BYTES, tid);
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const char* bx = smem + (t & 1) * 2 * TILE_BYTES;
    const bool more = (t + 1) < nk;
    if (more) {
      stage_load<T, XK>(rx, X, a.ldx, m0, a.M, (t + 1) * KE, a.K, tid);
      stage_load<T, WK>(rw, W, a.ldw, n0, a.N, (t + 1) * KE, a.K, tid);
    }
    mma_kstep<T, XK, WK>(bx, bx + TILE_BYTES, wm, wn, lane, acc);
    if (more) {
      char* nb = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
      stage_store<T, XK>(rx, nb, tid);
      stage_store<T, WK>(rw, nb + TILE_BYTES, tid);
    }
    __syncthreads();
  }

  // ---- epilogue straight from registers: lane = row m, register quad = 4 consecutive n ----
  const int h = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wm * 64 + j * 32 + (lane & 31);
    if (m >= a.M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + i * 32 + 8 * g + 4 * h;
        if (n >= a.N) continue;
        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
        epilogue4<T>(a, a.epilogue, (long)m * a.ldc + n, v);
      }
    }
  }
}

template <typename T>
int launch_generic(const GemmArgs& a, int xk, int wk, hipStream_t s) {
  const dim3 grid(a.tiles_m * a.tiles_n), block(kThreads);
  const size_t lds = 4 * TILE_BYTES;
  if (!xk && !wk)
    hipLaunchKernelGGL((gemm_generic_kernel<T, false, false>), grid, block, lds, s, a);
  else if (!xk && wk)
    hipLaunchKernelGGL((gemm_generic_kernel<T, false, true>), grid, block, lds, s, a);
  else if (xk && wk)
    hipLaunchKernelGGL((gemm_generic_kernel<T, true, true>), grid, block, lds, s, a);
  else
    hipLaunchKernelGGL((gemm_generic_kernel<T, true, false>), grid, block, lds, s, a);
  return pm_check_launch();
}

template <typename T>
int launch_glds(const GemmArgs& a, int xk, int wk, hipStream_t s) {
  const dim3 grid(a.tiles_m * a.tiles_n, a.split_k), block(kThreads);
  const size_t lds = GLDS_LDS_BYTES;
  if (!xk && !wk) {
    auto kern = gemm_glds_kernel<T, false, false>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  } else if (!xk && wk) {
    auto kern = gemm_glds_kernel<T, false, true>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  } else if (xk && wk) {
    auto kern = gemm_glds_kernel<T, true, true>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  } else {
    auto kern = gemm_glds_kernel<T, true, false>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  }
  return pm_check_launch();
}

}  // namespace

extern "C" int pm_gemm_ws(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor, int in_dtype,
                          const float* bias, void* C, long ldc, int c_dtype, int epilogue, void* aux,
                          const float* resid, int M, int N, int K, void* workspace, size_t ws_bytes, void* stream) {
  if (!A || !B || !C) return PM_EINVAL;
  if (M <= 0 || N <= 0 || K <= 0) return PM_ESHAPE;
  if (in_dtype != PM_BF16 && in_dtype != PM_F32) return PM_EINVAL;
  if (c_dtype != PM_BF16 && c_dtype != PM_F32) return PM_EINVAL;
  const int epc = in_dtype == PM_BF16 ? 8 : 4;
  // 16-byte global chunks: the contiguous dimension of each operand and its leading dimension must be chunk multiples
  if ((lda % epc) || (ldb % epc)) return PM_EALIGN;
  if (!a_kmajor && (K % epc)) return PM_EALIGN;
  if (a_kmajor && (M % epc)) return PM_EALIGN;
  if (!b_kmajor && (K % epc)) return PM_EALIGN;
  if (b_kmajor && (N % epc)) return PM_EALIGN;
  if ((N & 3) || (ldc & 3)) return PM_EALIGN;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15)) return PM_EALIGN;
  if (epilogue < PM_EPI_STORE || epilogue > PM_EPI_ACCUM) return PM_EINVAL;
  if ((epilogue == PM_EPI_GELU || epilogue == PM_EPI_DGELU) && !aux) return PM_EINVAL;
  if (epilogue == PM_EPI_RESIDUAL && (!resid || c_dtype != PM_F32)) return PM_EINVAL;
  if (epilogue == PM_EPI_ACCUM && c_dtype != PM_F32) return PM_EINVAL;
  GemmArgs a;
  a.X = A; a.W = B; a.ldx = lda; a.ldw = ldb; a.bias = bias; a.C = C; a.ldc = ldc; a.aux = aux; a.resid = resid;
  a.M = M; a.N = N; a.K = K; a.epilogue = epilogue; a.c_dtype = c_dtype;
  a.tiles_m = (M + BM - 1) / BM;
  a.tiles_n = (N + BN - 1) / BN;
  a.split_k = 1;
  hipStream_t s = pm_stream(stream);
  const int ke = in_dtype == PM_BF16 ? 64 : 32;
  const bool fast = (K % ke) == 0;
  if (!fast) {
    a.ksteps_split = 0;
    return in_dtype == PM_BF16 ? launch_generic<__bf16>(a, a_kmajor, b_kmajor, s) : launch_generic<float>(a, a_kmajor, b_kmajor, s);
  }
  const int nk = K / ke;
  a.ksteps_split = nk;
  // split-K: only for f32 plain-store / accumulate outputs without bias (the wgrad shapes: K = #tokens, few tiles)
  const int tiles = a.tiles_m * a.tiles_n;
  const bool splittable = workspace && !bias && c_dtype == PM_F32 && (epilogue == PM_EPI_STORE || epilogue == PM_EPI_ACCUM) &&
                          ldc == N;
  if (splittable && tiles < 256 && nk >= 16) {
    int split = 512 / tiles;                      // fill the 2-blocks-per-CU machine once
    if (split > nk / 8) split = nk / 8;           // >= 8 k-steps per split
    if (split > 16) split = 16;
    const size_t need = (size_t)split * M * N * sizeof(float);
    if (split > 1 && need <= ws_bytes) {
      a.split_k = split;
      a.ksteps_split = (nk + split - 1) / split;
      a.split_k = (nk + a.ksteps_split - 1) / a.ksteps_split;  // no empty split
      void* out = a.C;
      a.C = workspace;
      int st = in_dtype == PM_BF16 ? launch_glds<__bf16>(a, a_kmajor, b_kmajor, s) : launch_glds<float>(a, a_kmajor, b_kmajor, s);
      if (st) return st;
      const long nvec = (long)M * (N >> 2);
      int grid = (int)((nvec + 255) / 256);
      if (grid > 2048) grid = 2048;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, (const float*)workspace, (float*)out, ldc, M, N,
                         a.split_k, epilogue == PM_EPI_ACCUM ? 1 : 0);
      return pm_check_launch();
    }
  }
  return in_dtype == PM_BF16 ? launch_glds<__bf16>(a, a_kmajor, b_kmajor, s) : launch_glds<float>(a, a_kmajor, b_kmajor, s);
}

extern "C" int pm_gemm(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor, int in_dtype,
                       const float* bias, void* C, long ldc, int c_dtype, int epilogue, void* aux, const float* resid,
                       int M, int N, int K, void* stream) {
  return pm_gemm_ws(A, lda, a_kmajor, B, ldb, b_kmajor, in_dtype, bias, C, ldc, c_dtype, epilogue, aux, resid, M, N, K,
                    nullptr, 0, stream);
}
