// pm_gemm.hip -- MFMA GEMM with fused epilogues for every Linear / conv-as-GEMM of the ViT path and for
// their dgrad / wgrad (reference: timm Attention.qkv/proj, Mlp.fc1/fc2, PatchEmbed.proj, decoder_embed,
// decoder_pred -- models_mae.py:33,39-41,48,53-58; autograd backward engine_pretrain.py:65, tc.py:4533).
//
//   acc[m,n] = sum_k X(m,k) * W(n,k)        X = "A" matrix (M side), W = "B" matrix (N side)
//
// 128x128 kernels -- tile: 128(m) x 128(n) x 128 BYTES of k (64 bf16 / 32 f32) per step, 256 threads = 4 wave64 in 2x2, each
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
// Three kernels:
//   gemm_v3_kernel (below, "large-tile path"): the bf16 workhorse of the training step -- 256x256 / 192x256 / 256x128
//     block tiles, 8 waves, 4-slot LDS-DMA ring, ping-pong or software-pipelined k-loop, batched-load / 16-B-store
//     epilogues; forward, dgrad and split-K wgrad of every ViT-B shape go through it.
//   The two 128x128 kernels serve f32 mode and the small / odd shapes, and share LDS images and fragment readers:
//   gemm_glds_kernel (K % k-step == 0): global -> LDS by LDS-DMA (global_load_lds_dwordx4; the LDS
//     image is lane-linear, so the swizzle is applied to each lane's SOURCE address), double buffered, one
//     barrier per k-step, no staging registers and no ds_write; rows beyond M / N are clamped to the last
//     valid row (their results are never stored).  Optional split-K over blockIdx.y writes f32 partial slabs
//     that splitk_reduce_kernel sums in a fixed order (deterministic; used by wgrad where K = #tokens).
//     Epilogue: each wave stages its 64x64 accumulators through LDS and leaves as whole 16-B vectors along n
//     (bias / GELU / dGELU / residual applied on the way out) -> fully coalesced stores.
//   gemm_generic_kernel (any K % chunk == 0): register-staged, fully predicated loads (tiny / odd shapes).
// Roofline: MFMA-bound; algorithmic FLOPs 2*M*N*K.
#include "pm_common.h"
#include <type_traits>
#include <stdio.h>

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
  float* xsum;         // grouped wgrad: xsum[m] += sum_k X(m, k) (= the bias gradient: column sums of dY), or NULL
  int xsum_store;      // xsum[m] = ... instead of += (partial row sums of a k-slice, reduced later)
  int epi_hoist;       // gemm_glds_kernel: the sixteen epilogue loads of a wave tile in one batch (0: one load -> store chain per vector)
  int col_major;       // tile index -> (tm, tn): 0 row by row, 1 column by column (grouped wgrad: see wgrad_group_kernel), >= 2: row by
                       // row inside bands of that many tile columns (wide-N GEMMs: see gemm_dispatch)
#ifdef PM_GEMM_STAMP
  unsigned long long* stamps;  // diagnostic build only: per-wave cycle sums of the k-loop segments
#endif
};

// Diagnostic build (-DPM_GEMM_STAMP, scratch/stamp_gemm.py): s_memtime stamps around the segments of the ping-pong
// k-loop.  The shipped library never executes a stamp.
#ifdef PM_GEMM_STAMP
#define PM_STAMP(i)                               \
  do {                                            \
    __builtin_amdgcn_sched_barrier(0);            \
    stamp_t[i] = __builtin_readcyclecounter();    \
    __builtin_amdgcn_sched_barrier(0);            \
  } while (0)
#else
#define PM_STAMP(i)
#endif

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
__device__ __forceinline__ Frag16 read_frag(const char* PM_LDS_IMAGE tile, int rb, int kk, int lane) {
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

// The 32x32x16 MFMA of the ring kernels' k-loops.  Diagnostic build -DPM_MFMA16_TIMING (scratch/mfma16_timing.sh): the same
// operand registers feed TWO v_mfma_f32_16x16x32_bf16 (same FLOPs, same pipe cycles: 2 x 16 = 32) on two quarters of the
// accumulator -- the results are WRONG (the fragment maps differ), the build only answers "what clock does the chip hold on
// this k-loop with the other MFMA shape" (MI355X_MICROARCH.md, DVFS give-back item 7).  Never shipped.
template <typename E>
__device__ __forceinline__ void ring_mfma(const Frag16& a, const Frag16& b, f32x16& acc, int kk) {
#ifdef PM_MFMA16_TIMING
  f32x4 lo = {acc[8 * kk], acc[8 * kk + 1], acc[8 * kk + 2], acc[8 * kk + 3]};
  f32x4 hi = {acc[8 * kk + 4], acc[8 * kk + 5], acc[8 * kk + 6], acc[8 * kk + 7]};
  lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, lo, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, hi, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    acc[8 * kk + e] = lo[e];
    acc[8 * kk + 4 + e] = hi[e];
  }
#else
  (void)kk;
  acc = mfma16B<E>(a, b, acc);
#endif
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
      if (a.aux) store4<T>(reinterpret_cast<T*>(a.aux) + off, v);   // (no aux: nobody will run a backward through this Linear)
      v = gelu_act4<T>(round_through<T>(v));  // gelu of the value backward will see
    } else if (epi == PM_EPI_DGELU) {
      const f32x4 pre = load4<T>(reinterpret_cast<const T*>(a.aux) + off);
      v *= gelu_act_grad4<T>(pre);
    }
    if (a.c_dtype == PM_F32) {
      store4<float>(reinterpret_cast<float*>(a.C) + off, v);
    } else if constexpr (sizeof(T) == 2) {
      store4<T>(reinterpret_cast<T*>(a.C) + off, v);  // (a 16-bit C has the operands' type: checked by the dispatcher)
    } else {
      if (a.c_dtype == PM_F16) store4<_Float16>(reinterpret_cast<_Float16*>(a.C) + off, v);
      else store4<__bf16>(reinterpret_cast<__bf16*>(a.C) + off, v);
    }
  }
}

// The same epilogues on NV vectors at once: every load the epilogue needs (residual / saved pre-activation / C) is
// issued before the first store, so the loads overlap instead of forming one load -> store latency chain per vector
// (stamped on the 256x256 tile: 25.6k cycles per wave for 32 dependent chains).  off[] must be valid addresses (the
// caller clamps rows / columns beyond M / N); ok[] gates the stores.
template <typename E, int NV>
__device__ __forceinline__ void epilogue_batch(const GemmArgs& a, int epi, const long (&off)[NV], const bool (&ok)[NV],
                                               f32x4 (&v)[NV]) {
  if (epi == PM_EPI_RESIDUAL || epi == PM_EPI_ACCUM) {
    const float* src = epi == PM_EPI_RESIDUAL ? a.resid : reinterpret_cast<const float*>(a.C);
    f32x4 r[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) r[e] = *reinterpret_cast<const f32x4*>(src + off[e]);
#pragma unroll
    for (int e = 0; e < NV; ++e)
      if (ok[e]) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.C) + off[e]) = r[e] + v[e];
    return;
  }
  if (epi == PM_EPI_DGELU) {
    f32x4 pre[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) pre[e] = load4<E>(reinterpret_cast<const E*>(a.aux) + off[e]);
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] *= gelu_act_grad4<E>(pre[e]);
  } else if (epi == PM_EPI_GELU) {
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      if (a.aux && ok[e]) store4<E>(reinterpret_cast<E*>(a.aux) + off[e], v[e]);
      v[e] = gelu_act4<E>(round_through<E>(v[e]));  // what backward will see
    }
  }
  if (a.c_dtype == PM_F32) {
#pragma unroll
    for (int e = 0; e < NV; ++e)
      if (ok[e]) store4<float>(reinterpret_cast<float*>(a.C) + off[e], v[e]);
  } else {
#pragma unroll
    for (int e = 0; e < NV; ++e)
      if (ok[e]) store4<E>(reinterpret_cast<E*>(a.C) + off[e], v[e]);
  }
}

// 8 consecutive n per vector, act-typed (16-bit) C, epilogues STORE / GELU / DGELU: 16-B accesses.
template <typename E, int NV>
__device__ __forceinline__ void epilogue_batch8(const GemmArgs& a, int epi, const long (&off)[NV], const bool (&ok)[NV],
                                                f32x4 (&lo)[NV], f32x4 (&hi)[NV]) {
  if (epi == PM_EPI_DGELU) {
    f32x4 plo[NV], phi[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) load8_16<E>(reinterpret_cast<const E*>(a.aux) + off[u], plo[u], phi[u]);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      lo[u] *= gelu_act_grad4<E>(plo[u]);
      hi[u] *= gelu_act_grad4<E>(phi[u]);
    }
  } else if (epi == PM_EPI_GELU) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      if (a.aux && ok[u]) store8_16<E>(reinterpret_cast<E*>(a.aux) + off[u], lo[u], hi[u]);
      lo[u] = gelu_act4<E>(round_through<E>(lo[u]));  // gelu of the value backward will see (the rounded pre-activation)
      hi[u] = gelu_act4<E>(round_through<E>(hi[u]));
    }
  }
#pragma unroll
  for (int u = 0; u < NV; ++u)
    if (ok[u]) store8_16<E>(reinterpret_cast<E*>(a.C) + off[u], lo[u], hi[u]);
}

// exchange between lane l and lane l+32: afterwards the low half-wave holds (x of lane l, x of lane l+32) in (x, y)
// and the high half-wave holds (y of lane l-32, y of its own) -- v_permlane32_swap
__device__ __forceinline__ void swap_halves(float& x, float& y) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
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
        stage_glds<T, XK>(nb, X, a.ldx, m0, a.M, (t + 1) * KE, wave, lane);
        stage_glds<T, WK>(nb + TILE_BYTES, W, a.ldw, n0, a.N, (t + 1) * KE, wave, lane);
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
  if (!split && a.epi_hoist && (sizeof(T) == 2 || a.c_dtype == PM_F32)) {  // (f32 operands with a 16-bit C: the per-vector path below)
    // every load the epilogue needs for the wave's 64 x 64 tile (f32 residual / C, or the saved pre-activation) is issued before the
    // first store: one exposed trip to memory per tile instead of sixteen load -> store chains (the compiler cannot hoist a load over
    // a store that may alias it).  Clamped addresses for rows / columns beyond M / N, the stores are gated.
    const int c4 = (lane & 15) * 4, n = nw + c4;
    const bool nok = n < a.N;
    const long ncl = nok ? n : a.N - 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + ncl);
    long off[16];
    bool ok[16];
    f32x4 v[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int ml = it * 4 + (lane >> 4), m = mw + ml;
      ok[it] = m < a.M && nok;
      off[it] = (long)(m < a.M ? m : a.M - 1) * a.ldc + ncl;
      v[it] = *reinterpret_cast<const f32x4*>(st + ml * STAGE_ROW + c4 * 4) + bv;
    }
    epilogue_batch<T, 16>(a, epi, off, ok, v);
    return;
  }
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

// ------------------------------------------------------------------------------------------------
// large-tile path (bf16, X k-normal): BM x BN block tile, 8 waves, 4-stage LDS-DMA ring
// ------------------------------------------------------------------------------------------------
// Why: with 128x128 tiles every k-step moves 32 KiB into LDS for 2.1 MFLOP (64 FLOP/B) and has only ONE stage
// of prefetch, so a k-step can never be shorter than the global->LDS round trip.  Here a block owns 256x256
// (or 256x128) outputs = 128 FLOP/B, a k-step is 32 elements (64-B rows: 4 chunks, XOR-swizzled by (row>>2)&3),
// and 4 ring slots keep 3 stages (96 KiB / CU) in flight behind a COUNTED s_waitcnt vmcnt(N) and a raw
// s_barrier (one per k-step): the DMA of stages t+1, t+2 overlaps the MFMAs of stage t.
// Wave layout WM x WN; per-wave tile (BM/WM) x (BN/WN) = 128x64 (256^2) or 64x64 (256x128):
// 12 / 8 ds_read_b128 per 16 / 8 MFMAs per k-step.
constexpr int V3_STAGES = 4;
constexpr int V3_KE = 32;                    // bf16 elements of k per stage
constexpr int V3_STAGE_WAVE = 64 * STAGE_ROW;  // epilogue staging per wave (64 rows x 64 f32, padded)

// One 1-KiB LDS-DMA piece j of a k-normal tile (16 rows x 64 B; the lane's SOURCE chunk is the inverse swizzle of its slot).
// XOR applied to the 16-B chunk index of row `row` of a k-normal stage image (64-B rows).  S16 = false: the 32x32x16 operand
// read (lane = row l & 31, k-half l >> 5); S16 = true: the 16x16x32 operand read (lane = row l & 15, chunk l >> 4), for which
// rows 4q .. 4q+3 take XOR {0, 0, 3, 3}[q]: every 16-lane group of the ds_read_b128 then covers all 16 bank slots once.
template <bool S16> __device__ __forceinline__ int v3_swz_kn(int row) {
  const int q = (row >> 2) & 3;
  if constexpr (S16) return (q >> 1) * 3;
  else return q;
}

template <bool S16 = false>
__device__ __forceinline__ void v3_piece_kn(char* tile, const __bf16* __restrict__ base, long ld, int r0, int R, int k0,
                                            int j, int lane) {
  const int row = 16 * j + (lane >> 2), cs = lane & 3;
  const int c = cs ^ v3_swz_kn<S16>(row);
  int gr = r0 + row;
  gr = gr < R ? gr : R - 1;
  glds16(base + (long)gr * ld + k0 + c * 8, tile + 1024 * j);
}

template <int ROWS, int NW = 8, bool S16 = false>
__device__ __forceinline__ void v3_stage_kn(char* tile, const __bf16* __restrict__ base, long ld, int r0, int R, int k0,
                                            int wave, int lane) {
  constexpr int TI = ROWS * 64 / 1024;  // wave-instructions for the tile, dealt round-robin to the NW waves
#pragma unroll
  for (int i = 0; i < (TI + NW - 1) / NW; ++i) {
    const int j = wave + NW * i;
    if (j >= TI) break;  // (192-row tiles: waves 4-7 issue one instruction less)
    v3_piece_kn<S16>(tile, base, ld, r0, R, k0, j, lane);
  }
}

// One piece j of a k-major tile (COLS columns: 1024 / (2 COLS) k-rows per piece).
template <int COLS>
__device__ __forceinline__ void v3_piece_km(char* tile, const __bf16* __restrict__ base, long ld, int r0, int R, int k0,
                                            int j, int lane) {
  constexpr int RB = COLS * 2;         // bytes per k-row
  constexpr int KPI = 1024 / RB;       // k-rows per wave-instruction (2 or 4)
  constexpr int CPR = RB / 16;         // chunks per k-row
  const int krow = KPI * j + lane / CPR, cs = lane % CPR;
  const int c = cs ^ ((krow & 3) << 2);
  int gc = r0 + c * 8;
  gc = gc < R ? gc : R - 8;
  glds16(base + (long)(k0 + krow) * ld + gc, tile + 1024 * j);
}

template <int COLS, int NW = 8>
__device__ __forceinline__ void v3_stage_km(char* tile, const __bf16* __restrict__ base, long ld, int r0, int R, int k0,
                                            int wave, int lane) {
  constexpr int NI = 32 * COLS * 2 / 1024 / NW;
#pragma unroll
  for (int i = 0; i < NI; ++i) v3_piece_km<COLS>(tile, base, ld, r0, R, k0, wave + NW * i, lane);
}

__device__ __forceinline__ Frag16 v3_frag_kn(const char* tile, int rb, int kk, int lane) {
  const int row = rb + (lane & 31);
  const int c = 2 * kk + (lane >> 5);
  Frag16 f;
  f.u = *reinterpret_cast<const u32x4*>(tile + row * 64 + 16 * (c ^ ((row >> 2) & 3)));
  return f;
}

// 16 rows x 32 k (one whole 64-B row segment per 4 lanes): the operand of v_mfma_f32_16x16x32_bf16 in ONE ds_read_b128
__device__ __forceinline__ Frag16 v3_frag16_kn(const char* tile, int rb, int lane) {
  const int row = rb + (lane & 15);
  const int c = lane >> 4;
  Frag16 f;
  f.u = *reinterpret_cast<const u32x4*>(tile + row * 64 + 16 * (c ^ v3_swz_kn<true>(row)));
  return f;
}

template <int RB>
__device__ __forceinline__ Frag16 v3_frag_km(const char* PM_LDS_IMAGE tile, int rb, int kk, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int ch = (rb >> 3) + 2 * (g & 1) + (p >> 1);
  const int sw = 16 * (ch ^ (q << 2)) + 8 * (p & 1);
  const int kbase = kk * 16 + 8 * (g >> 1) + q;
  using lds_s4 = __attribute__((address_space(3))) short4v;
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + kbase * RB + sw));
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(tile + (kbase + 4) * RB + sw));
  Frag16 f;
  f.u[0] = ((unsigned)(unsigned short)lo[0]) | (((unsigned)(unsigned short)lo[1]) << 16);
  f.u[1] = ((unsigned)(unsigned short)lo[2]) | (((unsigned)(unsigned short)lo[3]) << 16);
  f.u[2] = ((unsigned)(unsigned short)hi[0]) | (((unsigned)(unsigned short)hi[1]) << 16);
  f.u[3] = ((unsigned)(unsigned short)hi[2]) | (((unsigned)(unsigned short)hi[3]) << 16);
  return f;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else static_assert(N == 0, "add the immediate");
}

// One block = one BM_ x BN_ output tile (`tile` = row-major tile index, `split_idx` = k-slice for split-K).  The body of
// gemm_v3_kernel (one GEMM per launch) and of wgrad_group_kernel (the weight gradients of a whole transformer block in
// one launch, full-K tiles).
// M16: the k-loop issues v_mfma_f32_16x16x32_bf16 (16 x 16 output tiles, the whole 32-element k-step per instruction) instead of
// 32x32x16: the same FLOPs per pipe cycle and the same LDS bytes per k-step, but the chip holds a higher clock on it
// (MI355X_MICROARCH.md, DVFS give-back item 7; measured here with the operand registers of the 32x32x16 loop fed to the other
// shape: qkv 55.5 -> 48.8 us, fc1+GELU 102 -> 96).  Built for the forward kernels whose both operands are k-normal, on the
// ping-pong loop with the LDS-staged epilogue (the accumulator layout changes: lane = m % 16, registers = 4 consecutive n).
template <int BM_, int BN_, int WM, int WN, bool WK, int STAGES, bool DIRECT, bool PP, bool XK, int NW, bool SWP,
          bool XSUM = false, bool M16 = false, typename E = __bf16>
__device__ __forceinline__ void gemm_v3_tile(GemmArgs& a, const int tile, const int split_idx) {
  static_assert(!M16 || (PP && !DIRECT && !XK && !WK && !XSUM && !SWP && NW == 8), "16x16x32 variant: k-normal ping-pong, staged epilogue");
  constexpr int TM = BM_ / WM, TN = BN_ / WN;      // per-wave tile
  constexpr int MT = TM / 32, NTL = TN / 32;       // 32x32 accumulators per wave
  constexpr int XB = BM_ * 64, WB = BN_ * 64;      // bytes per stage per side
  constexpr int SB = XB + WB;
  constexpr int G = (SB / 1024 + NW - 1) / NW;     // LDS-DMA instructions per wave per stage (max over waves)
  constexpr int D = STAGES - 1;                    // stages in flight
  static_assert(WM * WN == NW && (NW == 8 || (NW == 4 && !PP)) && (WB % (1024 * NW)) == 0 &&
                    ((XB % (1024 * NW)) == 0 || ((PP || SWP) && !XK && XB == 12288 && DIRECT)),
                "NW waves (ping-pong: 8); whole wave-instructions per wave, except the 192-row k-normal X tile of the "
                "ping-pong kernel");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
#ifdef PM_GEMM_STAMP
  const unsigned long long stamp_entry = __builtin_readcyclecounter();
  const unsigned long long stamp_entry_rt = __builtin_amdgcn_s_memrealtime();  // 100 MHz: clock = d(cycles) / d(realtime) x 100 MHz
  unsigned long long stamp_loop_end = 0, stamp_loop_end_rt = 0;
  auto stamp_finish = [&]() {
    const unsigned long long t_issued = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_acked = __builtin_readcyclecounter();
    if (a.stamps && lane == 0 && blockIdx.y == 0) {
      unsigned long long* o = a.stamps + ((long)blockIdx.x * NW + wave) * 16;
      o[8] = stamp_entry; o[9] = stamp_loop_end; o[10] = t_issued; o[11] = t_acked;
      o[12] = stamp_entry_rt; o[13] = stamp_loop_end_rt;
    }
  };
#endif
  int tm, tn;
  if (a.col_major >= 2) {  // bands of col_major tile columns, row by row inside a band (the last band may be narrower)
    const int per_band = a.tiles_m * a.col_major;
    const int band = tile / per_band, r = tile - band * per_band;
    const int left = a.tiles_n - band * a.col_major;
    const int bw = left < a.col_major ? left : a.col_major;
    tm = r / bw;
    tn = band * a.col_major + (r - tm * bw);
  } else {
    tm = a.col_major ? tile % a.tiles_m : tile / a.tiles_n;
    tn = a.col_major ? tile / a.tiles_m : tile % a.tiles_n;
  }
  const int m0 = tm * BM_, n0 = tn * BN_;
  const __bf16* X = reinterpret_cast<const __bf16*>(a.X);
  const __bf16* W = reinterpret_cast<const __bf16*>(a.W);
  int nk = a.K / V3_KE;
  if (a.split_k > 1) {  // split-K: this block owns k-steps [kbeg, kend) and writes an f32 partial slab
    const int kbeg = split_idx * a.ksteps_split;
    const int kend = (kbeg + a.ksteps_split) < nk ? (kbeg + a.ksteps_split) : nk;
    nk = kend - kbeg;
    X += XK ? (long)kbeg * V3_KE * a.ldx : (long)kbeg * V3_KE;
    W += WK ? (long)kbeg * V3_KE * a.ldw : (long)kbeg * V3_KE;
    a.C = reinterpret_cast<float*>(a.C) + (long)split_idx * a.M * a.ldc;
  }

  f32x16 acc[NTL][MT];
#pragma unroll
  for (int i = 0; i < NTL; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  constexpr int NT16 = M16 ? TN / 16 : 1, MT16 = M16 ? TM / 16 : 1;   // 16 x 16 accumulators per wave (M16 only)
  f32x4 acc16[NT16][MT16];
#pragma unroll
  for (int i = 0; i < NT16; ++i)
#pragma unroll
    for (int j = 0; j < MT16; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Row sums of the X operand beside the GEMM (grouped weight gradients: X = dY^T, so xsum = the Linear's bias gradient).
  // The 32x32x16 X fragment (lane l: row l&31, k-half l>>5) is fed as the B operand of a 16x16x32 MFMA, which reads it as
  // column l&15, k-group l>>4: rows r and r+16 of the fragment fall on the same column in k-groups {0,2} / {1,3}.  An A
  // operand that is all ones in row 0 for k-groups {0,2} and in row 1 for k-groups {1,3} (lanes 0, 32 / 17, 49) therefore
  // yields D[0][c] = sum_k X[c][k], D[1][c] = sum_k X[16+c][k]: lane c < 16 holds them in registers 0 and 1.
  // 2*MT extra 16-cycle MFMAs per k-step, only in the waves of column 0 of the tiles of column 0 (each row once).
  static_assert(!XSUM || XK, "row sums ride on the k-major X fragments");
  f32x4 xs[MT];
  Frag16 ones_sel;
  bool do_xsum = false;
  if constexpr (XSUM) {
    do_xsum = a.xsum != nullptr && tn == 0 && wn == 0;
    const bool one = lane == 0 || lane == 32 || lane == 17 || lane == 49;
    const unsigned v = one ? ones2<E>() : 0u;  // two 1.0 of the operand type
    ones_sel.u = u32x4{v, v, v, v};
#pragma unroll
    for (int j = 0; j < MT; ++j) xs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  auto issue = [&](int t) {
    char* buf = smem + (t % STAGES) * SB;
    if constexpr (XK) v3_stage_km<BM_, NW>(buf, X, a.ldx, m0, a.M, t * V3_KE, wave, lane);
    else v3_stage_kn<BM_, NW, M16>(buf, X, a.ldx, m0, a.M, t * V3_KE, wave, lane);
    if constexpr (WK) v3_stage_km<BN_, NW>(buf + XB, W, a.ldw, n0, a.N, t * V3_KE, wave, lane);
    else v3_stage_kn<BN_, NW, M16>(buf + XB, W, a.ldw, n0, a.N, t * V3_KE, wave, lane);
  };
  auto read_frags = [&](int t, Frag16 (&fw)[2][NTL], Frag16 (&fx)[2][MT]) {
    const char* bx = smem + (t % STAGES) * SB;
    const char* bw = bx + XB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < NTL; ++i) {
        if constexpr (WK) fw[kk][i] = v3_frag_km<BN_ * 2>(bw, wn * TN + 32 * i, kk, lane);
        else fw[kk][i] = v3_frag_kn(bw, wn * TN + 32 * i, kk, lane);
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        if constexpr (XK) fx[kk][j] = v3_frag_km<BM_ * 2>(bx, wm * TM + 32 * j, kk, lane);
        else fx[kk][j] = v3_frag_kn(bx, wm * TM + 32 * j, kk, lane);
      }
    }
  };
  auto mma_all = [&](const Frag16 (&fw)[2][NTL], const Frag16 (&fx)[2][MT]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          ring_mfma<E>(fw[kk][i], fx[kk][j], acc[i][j], kk);
    if constexpr (XSUM) {
      if (do_xsum) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int j = 0; j < MT; ++j)
            xs[j] = mfma16x16<E>(ones_sel, fx[kk][j], xs[j]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  if constexpr (SWP) {
    // Software-pipelined loop, one barrier per k-step.  Iteration t runs the MFMAs of stage t on fragments that were
    // read from LDS during iteration t-1, and between those MFMAs issues (a) the fragment reads of stage t+1 into the
    // other register set and (b) this wave's LDS-DMA pieces of stage t+3 into the slot of stage t-1 (every wave
    // consumed its stage-(t-1) fragments before it reached this iteration's barrier).  LDS reads and DMA issue cost
    // far less in the shadow of the wave's own MFMAs than in a phase of their own (stamped: 12 ds_read_b128 276 cyc,
    // 4 pieces 416 cyc in the ping-pong read phase), and the SIMD's two waves cover each other's remaining stalls.
    // Ring of STAGES slots: the pieces of stage t+STAGES-1 go into the slot of stage t-1; stages t+2 .. t+STAGES-2
    // may still be in flight at the barrier of iteration t (STAGES = 3: none -- the block's twin on the CU covers it).
    static_assert((STAGES == 4 || STAGES == 3) && !PP, "software-pipelined ring: 3 or 4 slots");
    constexpr int AHEAD = STAGES - 1;   // issue distance
    constexpr int FLY = STAGES - 3;     // stages allowed in flight across the barrier
    constexpr int TIX = XB / 1024, TIW = WB / 1024;         // DMA pieces per side per stage
    constexpr int GX = (TIX + NW - 1) / NW, GW = TIW / NW;  // per wave (GX: max over waves)
    constexpr int NR = 2 * (NTL + MT), NMF = 2 * NTL * MT;  // fragment reads / MFMAs per k-step
    constexpr int PER = NMF / (GX + GW);                    // one DMA piece every PER MFMAs
    static_assert(GX + GW == G && NR <= NMF && PER >= 2, "interleave plan");
    constexpr bool UNEVEN = (TIX % NW) != 0;
    const int gq = wave >> 2;
    auto wait_stages = [&](auto n) {  // all but the n youngest stages of this wave's pieces have landed
      constexpr int N = decltype(n)::value;
      if constexpr (UNEVEN) {
        if (gq == 0) wait_vmcnt<N * G>(); else wait_vmcnt<N * (G - 1)>();
      } else {
        wait_vmcnt<N * G>();
      }
    };
    auto piece = [&](int t, int p) {
      char* buf = smem + (t % STAGES) * SB;
      const int k0 = t * V3_KE;
      if (p < GX) {
        const int j = wave + NW * p;
        if (UNEVEN && j >= TIX) return;
        if constexpr (XK) v3_piece_km<BM_>(buf, X, a.ldx, m0, a.M, k0, j, lane);
        else v3_piece_kn(buf, X, a.ldx, m0, a.M, k0, j, lane);
      } else {
        const int j = wave + NW * (p - GX);
        if constexpr (WK) v3_piece_km<BN_>(buf + XB, W, a.ldw, n0, a.N, k0, j, lane);
        else v3_piece_kn(buf + XB, W, a.ldw, n0, a.N, k0, j, lane);
      }
    };
    auto read_one = [&](int t, int q, Frag16 (&fw)[2][NTL], Frag16 (&fx)[2][MT]) {
      const char* bx = smem + (t % STAGES) * SB;
      const char* bw = bx + XB;
      const int kk = q / (NTL + MT), r = q % (NTL + MT);
      if (r < NTL) {
        if constexpr (WK) fw[kk][r] = v3_frag_km<BN_ * 2>(bw, wn * TN + 32 * r, kk, lane);
        else fw[kk][r] = v3_frag_kn(bw, wn * TN + 32 * r, kk, lane);
      } else {
        if constexpr (XK) fx[kk][r - NTL] = v3_frag_km<BM_ * 2>(bx, wm * TM + 32 * (r - NTL), kk, lane);
        else fx[kk][r - NTL] = v3_frag_kn(bx, wm * TM + 32 * (r - NTL), kk, lane);
      }
    };
#ifdef PM_GEMM_STAMP
    unsigned long long stamp_t[8], stamp_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    // FULL: stages t+1 and t+3 exist (steady state, no branches in the MFMA stream)
    auto step =[&](auto full, int t, const Frag16 (&cw)[2][NTL], const Frag16 (&cx)[2][MT], Frag16 (&nw)[2][NTL],
                    Frag16 (&nx)[2][MT]) {
      constexpr bool FULL = decltype(full)::value;
      PM_STAMP(0);
      __builtin_amdgcn_sched_barrier(0);
      if (FLY > 0 && (FULL || t + 2 < nk)) wait_stages(std::integral_constant<int, FLY>{});  // stage t+1 landed
      else wait_vmcnt<0>();
      PM_STAMP(1);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      PM_STAMP(2);
      const bool rd = FULL || t + 1 < nk, ld = FULL || t + AHEAD < nk;
#pragma unroll
      for (int m = 0; m < NMF; ++m) {
        const int kk = m / (NTL * MT), i = (m / MT) % NTL, j = m % MT;
        ring_mfma<E>(cw[kk][i], cx[kk][j], acc[i][j], kk);
        __builtin_amdgcn_sched_barrier(0);
        if (m < NR && rd) read_one(t + 1, m, nw, nx);
        if ((m % PER) == 1 && m / PER < G && ld) piece(t + AHEAD, m / PER);
        __builtin_amdgcn_sched_barrier(0);
#ifdef PM_GEMM_STAMP
        if (m == NMF / 2 - 1) PM_STAMP(3);
#endif
      }
      if constexpr (XSUM) {
        if (do_xsum) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < MT; ++j)
              xs[j] = mfma16x16<E>(ones_sel, cx[kk][j], xs[j]);
        }
      }
#ifdef PM_GEMM_STAMP
      PM_STAMP(4);
#pragma unroll
      for (int i = 0; i < 4; ++i) stamp_acc[i] += stamp_t[i + 1] - stamp_t[i];
#endif
    };
#pragma unroll
    for (int t = 0; t < AHEAD; ++t)
      if (t < nk) issue(t);
    if (nk > 2) wait_stages(std::integral_constant<int, (AHEAD > 2 ? 2 : 1)>{});
    else if (nk > 1) wait_stages(std::integral_constant<int, 1>{});
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    Frag16 fwA[2][NTL], fxA[2][MT], fwB[2][NTL], fxB[2][MT];
    read_frags(0, fwA, fxA);
    __builtin_amdgcn_sched_barrier(0);
    int t = 0;
    for (; t + 1 + AHEAD < nk; t += 2) {
      step(std::true_type{}, t, fwA, fxA, fwB, fxB);
      step(std::true_type{}, t + 1, fwB, fxB, fwA, fxA);
    }
    for (; t < nk; t += 2) {
      step(std::false_type{}, t, fwA, fxA, fwB, fxB);
      if (t + 1 < nk) step(std::false_type{}, t + 1, fwB, fxB, fwA, fxA);
    }
#ifdef PM_GEMM_STAMP
    if (a.stamps && lane == 0 && blockIdx.y == 0) {
#pragma unroll
      for (int i = 0; i < 7; ++i) a.stamps[((long)blockIdx.x * NW + wave) * 16 + i] = stamp_acc[i];
      a.stamps[((long)blockIdx.x * NW + wave) * 16 + 7] = nk;
    }
#endif
  } else if constexpr (!PP) {
#pragma unroll
    for (int t = 0; t < D; ++t)
      if (t < nk) issue(t);
    for (int t = 0; t < nk; ++t) {
      // stage t must have landed (this wave's share); later stages may stay in flight
      const int ahead = nk - 1 - t;  // stages issued after t that exist
      if (ahead >= D - 1) wait_vmcnt<G * (D - 1)>();
      else if (ahead == 1) wait_vmcnt<G>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();  // everyone's share of stage t landed; everyone finished reading stage t-1
      // all fragment reads of the k-step first (their latency then overlaps the DMA issue below), then the DMA
      // refill of the slot that stage t-1 occupied, then 2*MT*NTL back-to-back MFMAs
      Frag16 fw[2][NTL], fx[2][MT];
      read_frags(t, fw, fx);
      __builtin_amdgcn_sched_barrier(0);
      if (t + D < nk) issue(t + D);
      __builtin_amdgcn_sched_barrier(0);
      mma_all(fw, fx);
    }
  } else {
    // Ping-pong: waves 0-3 and 4-7 (SIMD partners: wave w and w+4 share a SIMD) run the same two-phase loop
    //   R_t: 12 LDS fragment reads of stage t + LDS-DMA refill      M_t: 16 back-to-back MFMAs
    // one phase apart (waves 4-7 pass one extra barrier first), so on every SIMD one wave's MFMA phase runs
    // beside its partner's read / DMA-issue phase instead of both stalling on LDS latency together.
    // Ring: DP = STAGES-2 stages in flight; R_t refills the slot of stage t-2, which both halves finished reading
    // (and waited lgkmcnt(0) on) at least two barriers ago.  Stage readiness: every wave waits for its own DMA share
    // of stage t+1 before the barrier that opens its M_t -- for the lagging half that barrier is the one that opens
    // the leading half's R_{t+1}.
    // (Issuing the refill, or half of it, between the MFMAs of M_t instead was measured neutral to slightly worse:
    // the stamped read phase shrinks from 810 to 392 cycles but the MFMA phase grows from 575 to 730.)
    constexpr int DP = STAGES - 2;
    static_assert(STAGES == 4, "ping-pong ring tuned for 4 slots");
    const int gq = wave >> 2;
#pragma unroll
    for (int t = 0; t < DP; ++t)
      if (t < nk) issue(t);
    // DMA instructions per stage of THIS wave: G for whole tiles; 192-row X tiles deal 12 instructions to 8 waves,
    // so waves 4-7 (= the lagging half) issue G - 1
    constexpr bool UNEVEN = (XB % 8192) != 0;
    auto wait_one_stage_in_flight = [&]() {
      if constexpr (UNEVEN) {
        if (gq == 0) wait_vmcnt<G>(); else wait_vmcnt<G - 1>();
      } else {
        wait_vmcnt<G>();
      }
    };
    if (nk > 1) wait_one_stage_in_flight();
    else wait_vmcnt<0>();
    if (gq == 1) __builtin_amdgcn_s_barrier();
#ifdef PM_GEMM_STAMP
    unsigned long long stamp_t[8], stamp_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    for (int t = 0; t < nk; ++t) {
      PM_STAMP(0);
      __builtin_amdgcn_s_barrier();  // opens R_t
      PM_STAMP(1);
      Frag16 fw[2][NTL], fx[2][MT];
      Frag16 gw[NT16], gx[MT16];
      if constexpr (M16) {
        const char* bx = smem + (t % STAGES) * SB;
        const char* bw = bx + XB;
#pragma unroll
        for (int i = 0; i < NT16; ++i) gw[i] = v3_frag16_kn(bw, wn * TN + 16 * i, lane);
#pragma unroll
        for (int j = 0; j < MT16; ++j) gx[j] = v3_frag16_kn(bx, wm * TM + 16 * j, lane);
      } else {
        read_frags(t, fw, fx);
      }
      __builtin_amdgcn_sched_barrier(0);
      PM_STAMP(2);
      if (t + DP < nk) issue(t + DP);
      PM_STAMP(3);
      if (t + 2 < nk) wait_one_stage_in_flight();  // stage t+1 landed, stage t+2 may fly
      else wait_vmcnt<0>();
      PM_STAMP(4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      PM_STAMP(5);
      __builtin_amdgcn_s_barrier();  // opens M_t
      PM_STAMP(6);
      if constexpr (M16) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NT16; ++i)
#pragma unroll
          for (int j = 0; j < MT16; ++j)
            acc16[i][j] = mfma16x16<E>(gw[i], gx[j], acc16[i][j]);
        __builtin_amdgcn_s_setprio(0);
      } else {
        mma_all(fw, fx);
      }
#ifdef PM_GEMM_STAMP
      PM_STAMP(7);
#pragma unroll
      for (int i = 0; i < 7; ++i) stamp_acc[i] += stamp_t[i + 1] - stamp_t[i];
#endif
    }
#ifdef PM_GEMM_STAMP
    if (a.stamps && lane == 0 && blockIdx.y == 0) {
#pragma unroll
      for (int i = 0; i < 7; ++i) a.stamps[((long)blockIdx.x * NW + wave) * 16 + i] = stamp_acc[i];
      a.stamps[((long)blockIdx.x * NW + wave) * 16 + 7] = nk;
    }
#endif
    if (gq == 0) __builtin_amdgcn_s_barrier();
  }
#ifdef PM_GEMM_STAMP
  stamp_loop_end = __builtin_readcyclecounter();
  stamp_loop_end_rt = __builtin_amdgcn_s_memrealtime();
#endif
  if constexpr (XSUM) {
    if (do_xsum && lane < 16) {  // one wave per row range: plain read-modify-write, fixed order
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int m = m0 + wm * TM + 32 * j + lane;
        if (m < a.M) a.xsum[m] = (a.xsum_store ? 0.f : a.xsum[m]) + xs[j][0];
        if (m + 16 < a.M) a.xsum[m + 16] = (a.xsum_store ? 0.f : a.xsum[m + 16]) + xs[j][1];
      }
    }
  }
  const int h = lane >> 5;
  const int epi = a.epilogue;
  if constexpr (DIRECT) {
    // straight from registers (lane = row m, register quad = 4 consecutive n): no LDS, no block barrier -- the
    // stores drain while the CU's other resident block keeps the MFMA pipe busy
    if (a.c_dtype != PM_F32 && epi != PM_EPI_RESIDUAL && epi != PM_EPI_ACCUM && (a.N & 7) == 0 && (a.ldc & 7) == 0) {
      // act-typed output: lanes l and l+32 (same row m, columns 8g+4h) trade 4-vectors so that each holds 8
      // consecutive columns -> 16-B stores
      constexpr int NV8 = NTL * 2;
      long noff8[NV8];
      bool nok8[NV8];
      f32x4 b0[NV8], b1[NV8];
#pragma unroll
      for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int n = n0 + wn * TN + i * 32 + 16 * p + 8 * h;
          nok8[i * 2 + p] = n < a.N;
          noff8[i * 2 + p] = n < a.N ? n : a.N - 8;
        }
#pragma unroll
      for (int e = 0; e < NV8; ++e) {
        b0[e] = b1[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
          b0[e] = *reinterpret_cast<const f32x4*>(a.bias + noff8[e]);
          b1[e] = *reinterpret_cast<const f32x4*>(a.bias + noff8[e] + 4);
        }
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int m = m0 + wm * TM + j * 32 + (lane & 31);
        const bool mok = m < a.M;
        const long row = (long)(mok ? m : a.M - 1) * a.ldc;
#pragma unroll
        for (int i = 0; i < NTL; ++i) {
          long off[2];
          bool ok[2];
          f32x4 lo[2], hi[2];
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int e = i * 2 + p;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float x = acc[i][j][8 * p + k], y = acc[i][j][8 * p + 4 + k];
              swap_halves(x, y);
              lo[p][k] = x + b0[e][k];
              hi[p][k] = y + b1[e][k];
            }
            off[p] = row + noff8[e];
            ok[p] = mok && nok8[e];
          }
          epilogue_batch8<E, 2>(a, epi, off, ok, lo, hi);
        }
      }
#ifdef PM_GEMM_STAMP
      stamp_finish();
#endif
      return;
    }
    constexpr int NV = NTL * 4;
    long noff[NV];
    bool nok[NV];
    f32x4 bv[NV];
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * TN + i * 32 + 8 * g + 4 * h;
        nok[i * 4 + g] = n < a.N;
        noff[i * 4 + g] = n < a.N ? n : a.N - 4;
      }
    if (a.bias) {
#pragma unroll
      for (int e = 0; e < NV; ++e) bv[e] = *reinterpret_cast<const f32x4*>(a.bias + noff[e]);
    } else {
#pragma unroll
      for (int e = 0; e < NV; ++e) bv[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m0 + wm * TM + j * 32 + (lane & 31);
      const bool mok = m < a.M;
      const long row = (long)(mok ? m : a.M - 1) * a.ldc;
#pragma unroll
      for (int i = 0; i < NTL; ++i) {  // 4 vectors (one 32x32 accumulator row) per batch
        long off[4];
        bool ok[4];
        f32x4 v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int e = i * 4 + g;
          off[g] = row + noff[e];
          ok[g] = mok && nok[e];
          v[g] = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]} + bv[e];
        }
        epilogue_batch<E, 4>(a, epi, off, ok, v);
      }
    }
#ifdef PM_GEMM_STAMP
    stamp_finish();
#endif
    return;
  }
  __syncthreads();  // all ring slots dead: reuse LDS for the epilogue staging
  char* st = smem + wave * V3_STAGE_WAVE;
#pragma unroll
  for (int mh = 0; mh < MT / 2; ++mh) {    // 64 rows of the wave tile per pass
#pragma unroll
    for (int nh = 0; nh < NTL / 2; ++nh) {  // 64 columns per pass
      if constexpr (M16) {
        // 16 x 16 accumulators: lane l holds row m = l & 15, columns n = 4 (l >> 4) .. + 3 of its tile
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int ii = 0; ii < 4; ++ii)
            *reinterpret_cast<f32x4*>(st + (16 * jj + (lane & 15)) * STAGE_ROW + (16 * ii + 4 * (lane >> 4)) * 4) =
                acc16[M16 ? nh * 4 + ii : 0][M16 ? mh * 4 + jj : 0];
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ml = j * 32 + (lane & 31);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x16& t16 = acc[nh * 2 + i][mh * 2 + j];
              const f32x4 v = {t16[4 * g], t16[4 * g + 1], t16[4 * g + 2], t16[4 * g + 3]};
              *reinterpret_cast<f32x4*>(st + ml * STAGE_ROW + (i * 32 + 8 * g + 4 * h) * 4) = v;
            }
        }
      }
      __builtin_amdgcn_wave_barrier();
      const int mw = m0 + wm * TM + mh * 64, nw = n0 + wn * TN + nh * 64;
      if (a.c_dtype != PM_F32 && epi != PM_EPI_RESIDUAL && epi != PM_EPI_ACCUM && (a.N & 7) == 0 && (a.ldc & 7) == 0) {
        // act-typed output: 8 columns per lane -> 16-B stores (8 rows x 128 B per wave instruction)
        const int c8 = (lane & 7) * 8, n8 = nw + c8;
        const bool nok8 = n8 < a.N;
        const long ncl8 = nok8 ? n8 : a.N - 8;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (a.bias) {
          b0 = *reinterpret_cast<const f32x4*>(a.bias + ncl8);
          b1 = *reinterpret_cast<const f32x4*>(a.bias + ncl8 + 4);
        }
#pragma unroll
        for (int it4 = 0; it4 < 2; ++it4) {
          long off[4];
          bool ok[4];
          f32x4 lo[4], hi[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int ml = (it4 * 4 + u) * 8 + (lane >> 3);
            const int m = mw + ml;
            ok[u] = m < a.M && nok8;
            off[u] = (long)(m < a.M ? m : a.M - 1) * a.ldc + ncl8;
            lo[u] = *reinterpret_cast<const f32x4*>(st + ml * STAGE_ROW + c8 * 4) + b0;
            hi[u] = *reinterpret_cast<const f32x4*>(st + ml * STAGE_ROW + c8 * 4 + 16) + b1;
          }
          epilogue_batch8<E, 4>(a, epi, off, ok, lo, hi);
        }
        __builtin_amdgcn_wave_barrier();
        continue;
      }
      const int c4 = (lane & 15) * 4, n = nw + c4;
      const bool nok = n < a.N;
      const long ncl = nok ? n : a.N - 4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + ncl);
#pragma unroll
      for (int it4 = 0; it4 < 4; ++it4) {  // 4 rows x 4 batches of 4: loads of a batch overlap
        long off[4];
        bool ok[4];
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ml = (it4 * 4 + u) * 4 + (lane >> 4);
          const int m = mw + ml;
          ok[u] = m < a.M && nok;
          off[u] = (long)(m < a.M ? m : a.M - 1) * a.ldc + ncl;
          v[u] = *reinterpret_cast<const f32x4*>(st + ml * STAGE_ROW + c4 * 4) + bv;
        }
        epilogue_batch<E, 4>(a, epi, off, ok, v);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
#ifdef PM_GEMM_STAMP
  stamp_finish();
#endif
}

template <int BM_, int BN_, int WM, int WN, bool WK, int STAGES, int MINW, bool DIRECT, bool PP, bool XK = false, int NW = 8,
          bool SWP = false, bool M16 = false, typename E = __bf16>
__global__ __launch_bounds__(NW * 64, MINW) void gemm_v3_kernel(GemmArgs a) {
  gemm_v3_tile<BM_, BN_, WM, WN, WK, STAGES, DIRECT, PP, XK, NW, SWP, false, M16, E>(a, xcd_remap(blockIdx.x, gridDim.x), blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// grouped weight gradients: every dW of one transformer block in ONE launch, one full-K tile per workgroup
// ------------------------------------------------------------------------------------------------
// The weight gradients have a huge reduction dimension (K = number of tokens: 12 608 ... 50 432) and few output tiles
// (ViT-B: 27 + 9 + 36 + 36 tiles of 256x256).  One GEMM per launch therefore needs split-K to fill the chip: short
// k-loops, f32 slabs written and re-read, a reduce launch each -- and the prologue / epilogue paid per slice.  Grouping
// the four gradients of a block gives ~108 independent tiles at once: each workgroup runs ONE tile over the whole K
// (394+ k-steps: prologue and epilogue vanish, no slabs, no reduce, the result is deterministic by construction) and the
// launch takes ~100 CUs, which is the share the engine wants to give the weight-gradient stream beside the dgrad chain.
constexpr int kMaxGroup = 8;
struct WgradProb {
  const void* dY;     // [K][M]  (tokens x out features: the Linear's output gradient as stored)
  const void* X;      // [K][N]  (tokens x in features: the Linear's input as stored)
  float* dW;          // [M][N]
  long lddy, ldx, lddw;
  int M, N;           // out features, in features
  int tiles_n, tiles_m;
  int tile_begin;     // first global tile id of this problem
  int accumulate;     // dW += instead of dW =
  float* dbias;       // [M] += column sums of dY (the Linear's bias gradient), or NULL
  float* slab;        // split > 1: f32 partial products [split][M][N] (caller's workspace)
  float* bias_part;   // split > 1 and dbias: partial row sums [split][M]
  long vec_begin;     // split > 1: first float4 index of this problem in the reduce launch
};
struct WgradGroupArgs {
  WgradProb p[kMaxGroup];
  int n, K, total_tiles;
  int split;          // k-slices per tile (1: the tile's workgroup writes dW itself)
  int ksteps_split;
  int auto_order;     // per-problem tile order: the shorter side fastest (0: row by row, the A/B baseline)
  long total_vec, bias_begin, bias_total;  // reduce launch: float4 items, then bias rows
  int xcds;           // XCDs the work is confined to (8: all).  < 8: the grid is 8 / xcds times the work and the workgroups the
                      // hardware deals to the other XCDs (workgroup id % 8 >= xcds) leave at once -- see pm_wgrad_group
#ifdef PM_GEMM_STAMP
  unsigned long long* stamps;
#endif
};

// one (k-slice, tile) work item of a grouped launch
template <int BM_, int BN_, int WM, int WN, typename E, int NW>
__device__ __forceinline__ void group_item(const WgradGroupArgs& g, const int w) {
  {
    const int slice = w / g.total_tiles, tile = w - slice * g.total_tiles;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroup; ++i)
      if (i < g.n && tile >= g.p[i].tile_begin) pi = i;
    const WgradProb& pr = g.p[pi];
    GemmArgs a;
    a.X = pr.dY; a.W = pr.X; a.ldx = pr.lddy; a.ldw = pr.ldx; a.bias = nullptr; a.C = pr.dW; a.ldc = pr.lddw; a.aux = nullptr;
    a.resid = nullptr; a.M = pr.M; a.N = pr.N; a.K = g.K; a.epilogue = pr.accumulate ? PM_EPI_ACCUM : PM_EPI_STORE;
    a.c_dtype = PM_F32; a.tiles_m = pr.tiles_m; a.tiles_n = pr.tiles_n; a.split_k = 1; a.ksteps_split = 0;
    a.xsum = pr.dbias; a.xsum_store = 0; a.epi_hoist = 0;
    // Consecutive work items run on one XCD (xcd_remap) at the same pace, so what they share they fetch once into that XCD's
    // L2: a tile reads a dY panel [K x 256] and an X panel [K x 256] (6.4 MB each at K = 12 608).  Walk the SHORTER side of the
    // problem fastest, so that a run of ~13 tiles covers a compact rectangle: fc2's gradient is 3 x 12 tiles -- row by row a run
    // touches 2 + 12 panels, column by column 3 + 5 (PMC before: 557 MB fetched per ViT-B block against 335 MB algorithmic).
    a.col_major = g.auto_order ? (pr.tiles_m < pr.tiles_n) : 0;
    if (g.split > 1) {  // a k-slice: plain f32 partials into the slab (gemm_v3_tile offsets C by the slice), reduced afterwards
      a.C = pr.slab; a.ldc = pr.N; a.epilogue = PM_EPI_STORE; a.split_k = g.split; a.ksteps_split = g.ksteps_split;
      a.xsum = pr.dbias ? pr.bias_part + (long)slice * pr.M : nullptr; a.xsum_store = 1;
    }
#ifdef PM_GEMM_STAMP
    a.stamps = g.stamps;  // (rows of the block index: the caller hands a region of its own to the grouped launches)
#endif
    // (the software-pipelined loop needs 254 VGPRs without the row sums: with them it spills 73 and runs 1.5x slower;
    //  without them, bias gradients by separate column-sum passes, it equals this loop with the row sums inside)
    if constexpr (NW == 8)
      gemm_v3_tile<BM_, BN_, WM, WN, true, 4, true, true, true, 8, false, true, false, E>(a, tile - pr.tile_begin, slice);
    else
      gemm_v3_tile<BM_, BN_, WM, WN, true, 4, true, false, true, NW, true, true, false, E>(a, tile - pr.tile_begin, slice);
    __syncthreads();  // every wave is done with the LDS ring before the next tile's first stages are issued
  }
}

// NW = 8: the ping-pong loop (two waves per SIMD, 128 x 64 per wave).  NW = 4 (experiment, PM_GROUP_KERNEL=4): ONE wave per SIMD,
// 128 x 128 per wave on the software-pipelined loop -- 16 fragment reads per 32 MFMAs instead of 12 per 16, i.e. 2/3 of the LDS
// fragment traffic that co-limits the 8-wave loop here (both operands arrive by transpose reads), at 512 VGPRs per wave.
template <int BM_, int BN_, int WM, int WN, typename E = __bf16, int NW = 8>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void wgrad_group_kernel(WgradGroupArgs g) {
  // gridDim.x workgroups (a multiple of 8, or the whole work list) walk the (k-slice, tile) items t = blockIdx.x,
  // + gridDim.x, ...: the caller chooses how many CUs the weight-gradient stream takes from the dgrad chain beside it.
  // xcd_remap gives every XCD a contiguous run of items; within a k-slice consecutive items are neighbouring tiles
  // (same dY panel -> same private L2); with gridDim.x % 8 == 0 a workgroup stays on its XCD's run.
  const int work = g.total_tiles * g.split;
  if (g.xcds < 8) {
    // few tiles (the (proj, qkv) launch of a ViT-B block: 36): confined to g.xcds of the 8 XCDs, every XCD's run of consecutive
    // tiles is 8 / xcds times longer, i.e. shares more dY / X panels in ONE L2 instead of fetching them into several
    const int xcd = blockIdx.x & 7;
    if (xcd >= g.xcds) return;
    const int t = (blockIdx.x >> 3) * g.xcds + xcd;   // one work item per participating workgroup (grid sized for it)
    if (t >= work) return;
    const int q = work / g.xcds, r = work % g.xcds;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t / g.xcds);
    group_item<BM_, BN_, WM, WN, E, NW>(g, w);
    return;
  }
  for (int t = blockIdx.x; t < work; t += gridDim.x) {
    const int w = xcd_remap(t, work);
    group_item<BM_, BN_, WM, WN, E, NW>(g, w);   // (ends in a block barrier: the LDS ring is free for the next item)
  }
}

// dW (+)= sum over the k-slices of the slab, dbias += sum of the partial row sums: fixed order, one launch for the group.
__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(WgradGroupArgs g) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < g.bias_begin + g.bias_total; v += stride) {
    if (v < g.total_vec) {
      int pi = 0;
#pragma unroll
      for (int i = 1; i < kMaxGroup; ++i)
        if (i < g.n && v >= g.p[i].vec_begin) pi = i;
      const WgradProb& pr = g.p[pi];
      const long e = (v - pr.vec_begin) * 4;  // element index in [M][N] (N % 8 == 0)
      const long per = (long)pr.M * pr.N;
      f32x4 acc = *reinterpret_cast<const f32x4*>(pr.slab + e);
      for (int sl = 1; sl < g.split; ++sl) acc += *reinterpret_cast<const f32x4*>(pr.slab + sl * per + e);
      const long m = e / pr.N, c = e - m * pr.N;
      float* out = pr.dW + m * pr.lddw + c;
      if (pr.accumulate) acc += *reinterpret_cast<const f32x4*>(out);
      *reinterpret_cast<f32x4*>(out) = acc;
    } else if (v >= g.bias_begin) {
      long r = v - g.bias_begin;
      for (int i = 0; i < g.n; ++i) {
        const WgradProb& pr = g.p[i];
        if (!pr.dbias) continue;
        if (r < pr.M) {
          float sum = 0.f;
          for (int sl = 0; sl < g.split; ++sl) sum += pr.bias_part[(long)sl * pr.M + r];
          pr.dbias[r] += sum;
          break;
        }
        r -= pr.M;
      }
    }
  }
}

// forward GEMM on the 16x16x32 loop (both operands k-normal): 256 x 256 ping-pong, LDS-staged epilogue
template <typename E>
int launch_v3_m16(GemmArgs a, hipStream_t s) {
  a.tiles_m = (a.M + 255) / 256;
  a.tiles_n = (a.N + 255) / 256;
  constexpr int ring = 4 * 512 * 64, stage = 8 * V3_STAGE_WAVE;
  const size_t lds = ring > stage ? ring : stage;
  auto kern = gemm_v3_kernel<256, 256, 2, 4, false, 4, 2, false, true, false, 8, false, true, E>;
  PM_ALLOW_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(512), lds, s, a);
  return pm_check_launch();
}

template <typename E, int BM_, int BN_, int WM, int WN, int STAGES, int MINW, bool DIRECT, bool PP = false, int NW = 8, bool SWP = false>
int launch_v3(GemmArgs a, int wk, hipStream_t s) {
  a.tiles_m = (a.M + BM_ - 1) / BM_;
  a.tiles_n = (a.N + BN_ - 1) / BN_;
  constexpr int ring = STAGES * (BM_ + BN_) * 64;
  constexpr int stage = DIRECT ? 0 : NW * V3_STAGE_WAVE;
  const size_t lds = ring > stage ? ring : stage;
  const dim3 grid(a.tiles_m * a.tiles_n), block(NW * 64);
  if (wk) {
    auto kern = gemm_v3_kernel<BM_, BN_, WM, WN, true, STAGES, MINW, DIRECT, PP, false, NW, SWP, false, E>;
    PM_ALLOW_LDS(kern, lds);
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  } else {
    auto kern = gemm_v3_kernel<BM_, BN_, WM, WN, false, STAGES, MINW, DIRECT, PP, false, NW, SWP, false, E>;
    PM_ALLOW_LDS(kern, lds);
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  }
  return pm_check_launch();
}

// wgrad shape (both operands k-major, K = #tokens): 256x128 ping-pong ring kernel, direct epilogue, split-K slabs
template <typename E, int BM_, int BN_, int WM, int WN, bool SWP>
int launch_v3_wgrad(GemmArgs a, hipStream_t s) {
  a.tiles_m = (a.M + BM_ - 1) / BM_;
  a.tiles_n = (a.N + BN_ - 1) / BN_;
  constexpr int ring = 4 * (BM_ + BN_) * 64;
  const dim3 grid(a.tiles_m * a.tiles_n, a.split_k), block(512);
  auto kern = gemm_v3_kernel<BM_, BN_, WM, WN, true, 4, 2, true, !SWP, true, 8, SWP, false, E>;
  PM_ALLOW_LDS(kern, ring);
  hipLaunchKernelGGL(kern, grid, block, ring, s, a);
  return pm_check_launch();
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
    PM_ALLOW_LDS(kern, lds);
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  } else if (!xk && wk) {
    auto kern = gemm_glds_kernel<T, false, true>;
    PM_ALLOW_LDS(kern, lds);
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  } else if (xk && wk) {
    auto kern = gemm_glds_kernel<T, true, true>;
    PM_ALLOW_LDS(kern, lds);
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  } else {
    auto kern = gemm_glds_kernel<T, true, false>;
    PM_ALLOW_LDS(kern, lds);
    hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  }
  return pm_check_launch();
}

// the ring-kernel variants the dispatcher chooses from (cfg: see gemm_dispatch), per 16-bit operand type
template <typename E>
int launch_v3_cfg(int cfg, const GemmArgs& a, int b_kmajor, hipStream_t s) {
  switch (cfg) {
    case 6: return launch_v3<E, 256, 256, 2, 4, 4, 2, true>(a, b_kmajor, s);          // plain ring loop, register epilogue
    case 8: return launch_v3<E, 256, 256, 2, 4, 4, 2, false, true>(a, b_kmajor, s);   // ping-pong, LDS-staged epilogue
    case 40: if (!b_kmajor) return launch_v3_m16<E>(a, s);                            // the same on v_mfma_f32_16x16x32_bf16
             return launch_v3<E, 256, 256, 2, 4, 4, 2, false, true>(a, b_kmajor, s);
    case 9: return launch_v3<E, 256, 256, 2, 4, 4, 2, true, true>(a, b_kmajor, s);    // ping-pong, register epilogue
    case 10: return launch_v3<E, 192, 256, 2, 4, 4, 2, true, true>(a, b_kmajor, s);   // 192-row tiles: finer M granularity
    // 128-row tiles (experiment 6, round 4: the half-batch proj / fc2 of the forward chains hold 99 tiles of 192 rows on 256 CUs)
    case 12: return launch_v3<E, 128, 256, 2, 4, 4, 2, true, false, 8, true>(a, b_kmajor, s);
    case 13: return launch_v3<E, 128, 256, 2, 4, 4, 2, true, true>(a, b_kmajor, s);
    // software-pipelined loop (fragment reads and DMA issue between the wave's own MFMAs)
    case 24: return launch_v3<E, 256, 256, 2, 4, 4, 2, false, false, 8, true>(a, b_kmajor, s);
    case 25: return launch_v3<E, 256, 256, 2, 4, 4, 2, true, false, 8, true>(a, b_kmajor, s);
    case 26: return launch_v3<E, 192, 256, 2, 4, 4, 2, true, false, 8, true>(a, b_kmajor, s);
    // Measured and no longer instantiated (the template still admits them):
    //   two 4-wave blocks per CU   launch_v3<E, 128, 256, 2, 2, 3, 2, false, false, 4, true>   slower on every shape
    //   two 8-wave blocks per CU   launch_v3<E, 256, 128, 4, 2, 3, 4, true> / <128, 256, 2, 4, 3, 4, true>: <= 128 VGPRs, 100-150
    //                              spills, 85 instead of 128 FLOP per LDS-fill byte: 1.7x slower
    //   one wave per SIMD          launch_v3<E, 256, 256, 2, 2, 4, 1, true, false, 4, true>: 128x128 per wave, 512 VGPRs, 2/3 of the
    //                              LDS reads per FLOP: equals the 8-wave loop on long-K dgrads (decoder dfc1 1.0 PFLOP/s both),
    //                              loses 5-30 % wherever prologue / epilogue matter
    default: return launch_v3<E, 256, 256, 2, 4, 4, 2, false>(a, b_kmajor, s);
  }
}

// Split-K plan of a weight-gradient GEMM on the ring kernel (shared by the dispatcher and pm_gemm_workspace_bytes).
// 256x256 tiles (twice the MFMAs per barrier) from 2x2 tiles up: ViT-B qkv / fc1 / fc2 gradients 74 us vs 83 with
// 256x128; the MAE decoder's 512-wide gradients (K = 50 432 tokens) +2.8 % step rate; neutral for 768x768.
// Variant bits 6-7: 1 = force 256x128, 2 = force 256x256, 3 = software-pipelined 256x256 (slower: tr reads).
struct WgradPlan {
  int wv;     // tile variant (1: 256x128, 2: 256x256, 3: software-pipelined 256x256)
  int split;  // k-slices (each writes an f32 slab when > 1)
};
inline WgradPlan plan_wgrad(int M, int N, int K, int force_cfg, int wgrad_blocks, size_t ws_bytes) {
  WgradPlan p;
  p.wv = (force_cfg >> 6) & 3;
  if (p.wv == 0) p.wv = ((M + 255) / 256) * ((N + 255) / 256) >= 4 ? 2 : 1;
  const int bn3 = p.wv >= 2 ? 256 : 128;
  const int t3 = ((M + 255) / 256) * ((N + bn3 - 1) / bn3);
  const int nk3 = K / V3_KE;
  int split = wgrad_blocks / t3;
  if (split > nk3 / 16) split = nk3 / 16;
  if (split > 16) split = 16;
  if (split < 1) split = 1;
  while (split > 1 && (size_t)split * M * N * sizeof(float) > ws_bytes) --split;
  p.split = split;
  return p;
}
inline bool is16(int dtype) { return dtype == PM_BF16 || dtype == PM_F16; }
inline bool wgrad_ring_shape(int in_dtype, int a_kmajor, int b_kmajor, int M, int N, int K) {
  return is16(in_dtype) && a_kmajor && b_kmajor && (K % V3_KE) == 0 && K >= 2048 && M >= 256 && N >= 128;
}

int gemm_dispatch(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor, int in_dtype,
                  const float* bias, void* C, long ldc, int c_dtype, int epilogue, void* aux, const float* resid, int M, int N,
                  int K, void* workspace, size_t ws_bytes, const pm_gemm_opts* opts, void* stream);

}  // namespace

extern "C" size_t pm_gemm_workspace_bytes(int a_kmajor, int b_kmajor, int in_dtype, int M, int N, int K,
                                          const pm_gemm_opts* opts) {
  if (M <= 0 || N <= 0 || K <= 0 || !(a_kmajor && b_kmajor)) return 0;  // only the split-K weight gradients use scratch
  int blocks = (opts && opts->max_blocks > 0) ? opts->max_blocks : 256;
  blocks = blocks < 16 ? 16 : (blocks > 1024 ? 1024 : blocks);
  const int variant = opts ? opts->variant : 0;
  if (wgrad_ring_shape(in_dtype, a_kmajor, b_kmajor, M, N, K) && (variant & 63) != 1) {
    const WgradPlan p = plan_wgrad(M, N, K, variant, blocks, (size_t)-1);
    return p.split > 1 ? (size_t)p.split * M * N * sizeof(float) : 0;
  }
  // 128x128 split-K path: up to 16 slabs
  const int ke = is16(in_dtype) ? 64 : 32;
  if (K % ke) return 0;
  const int nk = K / ke, tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (tiles >= 256 || nk < 16) return 0;
  int split = 512 / tiles;
  if (split > nk / 8) split = nk / 8;
  if (split > 16) split = 16;
  return split > 1 ? (size_t)split * M * N * sizeof(float) : 0;
}

#ifdef PM_GEMM_STAMP
namespace { unsigned long long* g_stamps = nullptr; }
extern "C" void pm_debug_gemm_stamps(void* p) { g_stamps = reinterpret_cast<unsigned long long*>(p); }
#endif

extern "C" int pm_gemm_ex(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor, int in_dtype,
                          const float* bias, void* C, long ldc, int c_dtype, int epilogue, void* aux, const float* resid,
                          int M, int N, int K, void* workspace, size_t ws_bytes, const pm_gemm_opts* opts, void* stream) {
  return gemm_dispatch(A, lda, a_kmajor, B, ldb, b_kmajor, in_dtype, bias, C, ldc, c_dtype, epilogue, aux, resid, M, N, K,
                       workspace, ws_bytes, opts, stream);
}

extern "C" int pm_gemm_ws(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor, int in_dtype,
                          const float* bias, void* C, long ldc, int c_dtype, int epilogue, void* aux,
                          const float* resid, int M, int N, int K, void* workspace, size_t ws_bytes, void* stream) {
  return gemm_dispatch(A, lda, a_kmajor, B, ldb, b_kmajor, in_dtype, bias, C, ldc, c_dtype, epilogue, aux, resid, M, N, K,
                       workspace, ws_bytes, nullptr, stream);
}

namespace {

// tuning hook, read once: PM_CFG_CLASS="<nt_store>,<nt_gelu>,<nt_residual>,<nn_store>,<nn_dgelu>" forces a ring-kernel variant per
// (operand layout, epilogue) class for in-step A/B runs (0 = the heuristics); unset in production
const int* cfg_class_override() {
  static int v[5] = {0, 0, 0, 0, 0};
  static const bool init = [] {
    const char* e = getenv("PM_CFG_CLASS");
    if (e) sscanf(e, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]);
    return true;
  }();
  (void)init;
  return v;
}
int tall_m() {  // tuning hook, read once: row count from which a problem counts as "tall" (see gemm_dispatch)
  static const int v = [] { const char* e = getenv("PM_TALL_M"); return e && e[0] ? atoi(e) : 20000; }();
  return v;
}
int few_tiles_threshold() {  // tuning hook, read once: PM_FEW_TILES=0 keeps every M >= 1024 problem on the ring kernel (round-3 dispatch)
  static const int v = [] { const char* e = getenv("PM_FEW_TILES"); return e && e[0] ? atoi(e) : 128; }();
  return v;
}
// A/B switch, read once on the host and carried in GemmArgs: PM_EPI_HOIST=0 restores the 128 x 128 kernel's per-vector epilogue
int epi_hoist_host() {
  static const int v = [] { const char* e = getenv("PM_EPI_HOIST"); return e && e[0] ? atoi(e) : 1; }();
  return v;
}
int tile_band() {  // tuning hook, read once: forward / dgrad tiles walk bands of PM_TILE_BAND tile columns (default 6; 0: row by row)
  static const int v = [] { const char* e = getenv("PM_TILE_BAND"); return e && e[0] ? atoi(e) : 6; }();
  return v;
}
int few_tiles_narrow_n() {  // tuning hook, read once: widest N for which the few-tiles rule ignores the cap on M
  static const int v = [] { const char* e = getenv("PM_FEW_TILES_NARROW_N"); return e && e[0] ? atoi(e) : 512; }();
  return v;
}
int few_tiles_max_m() {  // tuning hook, read once: largest M the few-tiles rule applies to
  static const int v = [] { const char* e = getenv("PM_FEW_TILES_MAXM"); return e && e[0] ? atoi(e) : 4096; }();
  return v;
}
bool dgrad_pp() {  // A/B switch, read once: dgrads on the ping-pong loop
  static const bool v = [] { const char* e = getenv("PM_DGRAD_PP"); return e && e[0] == '1'; }();
  return v;
}

int gemm_dispatch(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor, int in_dtype,
                  const float* bias, void* C, long ldc, int c_dtype, int epilogue, void* aux, const float* resid, int M, int N,
                  int K, void* workspace, size_t ws_bytes, const pm_gemm_opts* opts, void* stream) {
  // per-call options (no process-wide state): kernel variant override (tuning scripts, tests) and the number of
  // workgroups a split-K weight gradient may spread over
  const int force_cfg = opts ? opts->variant : 0;
  int wgrad_blocks = (opts && opts->max_blocks > 0) ? opts->max_blocks : 256;
  if (wgrad_blocks < 16) wgrad_blocks = 16;
  if (wgrad_blocks > 1024) wgrad_blocks = 1024;
  if (!A || !B || !C) return PM_EINVAL;
  if (M <= 0 || N <= 0 || K <= 0) return PM_ESHAPE;
  if (!is16(in_dtype) && in_dtype != PM_F32) return PM_EINVAL;
  if (!is16(c_dtype) && c_dtype != PM_F32) return PM_EINVAL;
  if (is16(in_dtype) && is16(c_dtype) && c_dtype != in_dtype) return PM_EINVAL;  // a 16-bit C has the operands' type
  const int epc = is16(in_dtype) ? 8 : 4;
  // 16-byte global chunks: the contiguous dimension of each operand and its leading dimension must be chunk multiples
  if ((lda % epc) || (ldb % epc)) return PM_EALIGN;
  if (!a_kmajor && (K % epc)) return PM_EALIGN;
  if (a_kmajor && (M % epc)) return PM_EALIGN;
  if (!b_kmajor && (K % epc)) return PM_EALIGN;
  if (b_kmajor && (N % epc)) return PM_EALIGN;
  if ((N & 3) || (ldc & 3)) return PM_EALIGN;
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15)) return PM_EALIGN;
  if (epilogue < PM_EPI_STORE || epilogue > PM_EPI_ACCUM) return PM_EINVAL;
  if (epilogue == PM_EPI_DGELU && !aux) return PM_EINVAL;   // (GELU without aux: the pre-activation is not kept -- forward-only use)
  if (epilogue == PM_EPI_RESIDUAL && (!resid || c_dtype != PM_F32)) return PM_EINVAL;
  if (epilogue == PM_EPI_ACCUM && c_dtype != PM_F32) return PM_EINVAL;
  GemmArgs a;
  a.X = A; a.W = B; a.ldx = lda; a.ldw = ldb; a.bias = bias; a.C = C; a.ldc = ldc; a.aux = aux; a.resid = resid;
  a.M = M; a.N = N; a.K = K; a.epilogue = epilogue; a.c_dtype = c_dtype;
  a.tiles_m = (M + BM - 1) / BM;
  a.tiles_n = (N + BN - 1) / BN;
  a.split_k = 1;
  a.xsum = nullptr;
  a.xsum_store = 0;
  // Tile order of a wide problem (experiment 11, round 4).  xcd_remap hands each XCD a contiguous run of tiles; row by row, a run of
  // ~75 tiles of the dGELU dgrad (50 x 12 tiles) walks all 12 W panels (4.7 MB: more than the XCD's 4 MB L2, beside a 77-MB stream of
  // saved pre-activations) six times -- PMC: 205.8 MB fetched per launch against ~135 MB if every XCD read W once.  In bands of 6 tile
  // columns a run is ~12 row blocks x 6 panels: the band's W (2.4 MB) stays in the L2 while the row blocks stream through once.
  // Measured (profiles/r4_exp11_tile_bands.txt): dGELU dgrad 205.7 -> 172.7 MB fetched, fc1 + GELU 67.5 -> 57.7 MB; stand-alone fc1 + GELU
  // 99.5 -> 95.9 us; in the step +0.6 ... +1.2 % (cls), +0.7 % (MAE bs = 256), two same-box rounds each.
  a.epi_hoist = epi_hoist_host();
  a.col_major = 0;
  {
    const int band = tile_band();
    const int tn256 = (N + 255) / 256;
    if (band >= 2 && tn256 > band && !a_kmajor) a.col_major = band;
  }
#ifdef PM_GEMM_STAMP
  a.stamps = g_stamps;
#endif
  hipStream_t s = pm_stream(stream);
  const int ke = is16(in_dtype) ? 64 : 32;
  const bool fast = (K % ke) == 0;
  if (!fast) {
    a.ksteps_split = 0;
    PM_DISPATCH_ACT(in_dtype, T, return launch_generic<T>(a, a_kmajor, b_kmajor, s));
    return PM_EINVAL;
  }
  // Few tiles (round 4, scratch/bench_gemm_smallm.py, profiles/r4_exp3_gemm_smallm*.txt): a problem with fewer than ~half as many
  // 256 x 256 tiles as the chip has CUs runs ONE partial round of long workgroups on the ring kernel -- its time is a tile's
  // latency whatever M is (proj at M = 1 600 ... 6 304: 25-28 us) -- while the 128 x 128 LDS-DMA kernel spreads the same work over
  // 4x the workgroups, two per CU: M = 3 200 (MAE encoder at bs = 64/GPU) qkv 25.3 -> 21.2 us, proj 27.1 -> 18.7, fc2 57.4 -> 43.9,
  // dfc1 55.0 -> 37.4, dqkv 43.3 -> 29.3; the half-batch forward chains of the fine-tune (M = 6 304) proj 28.1 -> 26.0, fc2 58.7 ->
  // 54.0; the 512-wide MAE decoder at M = 12 608: dfc1 44.1 -> 38.5, dqkv 35.6 -> 30.7.  The crossover sits between 117 tiles
  // (128 x 128 wins) and 150 (the ring wins) on all 56 measured (shape, epilogue) points: fewer than 128 tiles -> 128 x 128.
  const long tiles256 = (long)((M + 255) / 256) * ((N + 255) / 256);
  // ... in the step the rule holds only where the WHOLE chain is small (experiment 4, profiles/r4_exp4_few_tiles.txt): sent to the
  // 128 x 128 kernel, the half-batch proj / fc2 of the fine-tune forward (M = 6 304) cost the step 4 % (MAE bs = 256: 1.6 %) although
  // they are faster alone -- two 66-KB workgroups on a CU keep the other chain's 128-KB ring workgroups off it -- while MAE at
  // bs = 64/GPU gains 4.5-6 %.  Hence the cap on M.
  // ... or narrow (N <= 512: the MAE decoder's proj / fc2 / dgrads at bs = 64/GPU, M = 6 304 / 12 608 -- 50 / 100 tiles; the same
  // experiment: +0.7 % on top; a 512-wide problem never belongs to the fine-tune's or the bs = 256 encoder's half-batch chains).
  const bool few_tiles = (force_cfg & 63) == 0 && tiles256 < few_tiles_threshold() &&
                         (M <= few_tiles_max_m() || N <= few_tiles_narrow_n());
  // large-tile ring kernel: bf16, X k-normal (forward and dgrad GEMMs), big M
  if (is16(in_dtype) && !a_kmajor && (K % V3_KE) == 0 && M >= 1024 && (force_cfg & 63) != 1 && !few_tiles) {
    // Tile / pipeline choice, tuned on the ViT-B/16 shapes at M = 12608 (scratch/bench_gemm6.py).
    // pm_debug_gemm_config(cfg) forces one of the variants below (0 = the heuristics).
    int cfg = force_cfg & 63;
    if (cfg == 0) {
      // 256x256 ping-pong everywhere; LDS-staged epilogue for the wide act-typed outputs (qkv, fc1+GELU: whole
      // 128-B row segments per store), direct register epilogue for f32 residual outputs and the dgrads
      const bool wide_act = epilogue == PM_EPI_GELU || (epilogue == PM_EPI_STORE && !b_kmajor && N >= 2048);
      cfg = wide_act ? 8 : 9;
      if (cfg == 9) {  // whole tiles per CU round: 192-row tiles when they need fewer (rounds x rows)
        const long nt = (N + 255) / 256;
        const long c256 = (((M + 255) / 256 * nt + 255) / 256) * 256, c192 = (((M + 191) / 192 * nt + 255) / 256) * 192;
        if (c192 < c256) cfg = 10;
        // f32 residual outputs of the forward (k-normal W): the software-pipelined loop wins (fc2 80 -> 70 us);
        // so it does for every dgrad (W read as stored by ds_read_b64_tr_b16) since those reads stopped waiting for the
        // whole DMA ring (PM_LDS_IMAGE): dfc1 66 -> 61 us, dqkv 51 -> 47, dfc2 101 -> 96, decoder dfc1 118 -> 107
        if ((epilogue == PM_EPI_RESIDUAL && !b_kmajor) || (b_kmajor && !dgrad_pp())) cfg = cfg == 10 ? 26 : 25;
        // round 3 (scratch/archive_r3/r3_exp1.sh, r3_exp3.sh):
        //  * the dGELU dgrad (reads the saved pre-activation, writes an act-typed [M, 4D] tensor) goes to the ping-pong loop
        //    with the LDS-staged epilogue: stand-alone 96.8 vs 95.1 us at ViT-B, 197 vs 220 us at the MAE decoder, and
        //    +0.6 % cls step rate in-step (whole 128-B row segments per store instead of 16-B pieces at a 6-KB stride)
        //  * tall problems (MAE decoder: M = 50 432 -> 197 row tiles, 1.5 rounds of the chip): the epilogue is paid 197 x 2
        //    times and the direct register epilogue's strided 16-B stores cost more than the staging pass: software-pipelined
        //    loop + staged epilogue (cfg 24) for f32-residual and dgrad outputs (proj 59 vs 66 us, fc2 145 vs 147, dqkv 80 vs
        //    83, dproj 37.5 vs 38.5), ping-pong + staged (cfg 8) for the act-typed qkv (96 vs 102)
        //  * plain dgrads (dfc1 / dproj / dqkv: N = 768): 256-row tiles with the staged epilogue (cfg 24) although the 192-row
        //    direct kernel is faster ALONE (59-62 vs 65 us): in the step the chain runs beside the grouped weight gradients,
        //    which hold 108 CUs -- 150 tiles of 256 rows + 108 = the chip, 198 tiles of 192 rows + 108 oversubscribe it and the
        //    chain's third of a round queues behind the long weight-gradient workgroups (scratch/archive_r3/r3_exp8.sh, r3_exp9.sh:
        //    +1.0 % cls in four same-box pairs, MAE +0.2 %)
        if (epilogue == PM_EPI_DGELU && !dgrad_pp()) cfg = 8;
        else if (b_kmajor && !dgrad_pp()) cfg = 24;
        else if (M >= tall_m()) cfg = (epilogue == PM_EPI_STORE && !b_kmajor) ? 8 : 24;
      }
      const int cls = !b_kmajor ? (epilogue == PM_EPI_GELU ? 1 : epilogue == PM_EPI_RESIDUAL ? 2 : 0) : (epilogue == PM_EPI_DGELU ? 4 : 3);
      if (cfg_class_override()[cls]) cfg = cfg_class_override()[cls];
    }
    PM_DISPATCH_16(in_dtype, E, return launch_v3_cfg<E>(cfg, a, b_kmajor, s));
  }
  const int nk = K / ke;
  a.ksteps_split = nk;
  // split-K: only for f32 plain-store / accumulate outputs without bias (the wgrad shapes: K = #tokens, few tiles)
  const int tiles = a.tiles_m * a.tiles_n;
  const bool splittable = workspace && !bias && c_dtype == PM_F32 && (epilogue == PM_EPI_STORE || epilogue == PM_EPI_ACCUM) &&
                          ldc == N;
  // large-K wgrad on the ping-pong ring kernel: 256x128 tiles, split so that tiles x splits ~ one block per CU
  if (splittable && wgrad_ring_shape(in_dtype, a_kmajor, b_kmajor, M, N, K) && (force_cfg & 63) != 1) {
    // 256x256 tiles (twice the MFMAs per barrier) from 2x2 tiles up: ViT-B qkv / fc1 / fc2 gradients 74 us vs 83 with
    // 256x128; the MAE decoder's 512-wide gradients (K = 50 432 tokens) +2.8 % step rate; neutral for 768x768.
    // Tuning hook bits 6-7: 1 = force 256x128, 2 = force 256x256, 3 = software-pipelined 256x256 (slower: tr reads).
    const WgradPlan plan = plan_wgrad(M, N, K, force_cfg, wgrad_blocks, ws_bytes);
    const int wv = plan.wv, nk3 = K / V3_KE, split = plan.split;
    GemmArgs w = a;
    w.ksteps_split = (nk3 + split - 1) / split;
    w.split_k = (nk3 + w.ksteps_split - 1) / w.ksteps_split;
    void* out = a.C;
    if (w.split_k > 1) {
      w.C = workspace;
      w.epilogue = PM_EPI_STORE;
    }
    int st = PM_EINVAL;
    PM_DISPATCH_16(in_dtype, E, st = wv == 1   ? launch_v3_wgrad<E, 256, 128, 4, 2, false>(w, s)
                                     : wv == 2 ? launch_v3_wgrad<E, 256, 256, 2, 4, false>(w, s)
                                               : launch_v3_wgrad<E, 256, 256, 2, 4, true>(w, s));
    if (st || w.split_k == 1) return st;
    const long nvec = (long)M * (N >> 2);
    int grid = (int)((nvec + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, (const float*)workspace, (float*)out, ldc, M, N,
                       w.split_k, epilogue == PM_EPI_ACCUM ? 1 : 0);
    return pm_check_launch();
  }
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
      int st = PM_EINVAL;
      PM_DISPATCH_ACT(in_dtype, T, st = launch_glds<T>(a, a_kmajor, b_kmajor, s));
      if (st) return st;
      const long nvec = (long)M * (N >> 2);
      int grid = (int)((nvec + 255) / 256);
      if (grid > 2048) grid = 2048;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, (const float*)workspace, (float*)out, ldc, M, N,
                         a.split_k, epilogue == PM_EPI_ACCUM ? 1 : 0);
      return pm_check_launch();
    }
  }
  PM_DISPATCH_ACT(in_dtype, T, return launch_glds<T>(a, a_kmajor, b_kmajor, s));
  return PM_EINVAL;
}

}  // namespace

extern "C" int pm_gemm_colsum(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor, int in_dtype,
                              const float* bias, void* C, long ldc, int c_dtype, int epilogue, void* aux,
                              const float* resid, float* colsum, int M, int N, int K, void* workspace, size_t ws_bytes,
                              void* stream) {
  if (!colsum) return PM_EINVAL;
  const int st = gemm_dispatch(A, lda, a_kmajor, B, ldb, b_kmajor, in_dtype, bias, C, ldc, c_dtype, epilogue, aux, resid, M, N, K,
                               nullptr, 0, nullptr, stream);
  if (st) return st;
  return pm_colsum_ws(C, ldc, c_dtype, colsum, M, N, workspace, ws_bytes, stream);
}

extern "C" int pm_gemm(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor, int in_dtype,
                       const float* bias, void* C, long ldc, int c_dtype, int epilogue, void* aux, const float* resid,
                       int M, int N, int K, void* stream) {
  return pm_gemm_ws(A, lda, a_kmajor, B, ldb, b_kmajor, in_dtype, bias, C, ldc, c_dtype, epilogue, aux, resid, M, N, K,
                    nullptr, 0, stream);
}

namespace {

// A/B switch, read once: PM_GROUP_ORDER=0 walks every problem's tiles row by row (round-2 behaviour)
int group_order_auto() {
  static const int v = [] { const char* e = getenv("PM_GROUP_ORDER"); return (e && e[0] == '0') ? 0 : 1; }();
  return v;
}

// tuning hook, read once: work items (tiles x k-slices) a group with few tiles is cut into (default 224 of the 256 CUs)
int group_split_target() {
  static const int v = [] { const char* e = getenv("PM_GROUP_SPLIT_TARGET"); return e && e[0] ? atoi(e) : 224; }();
  return v;
}

// tuning hooks, read once: a whole-K group of at most PM_GROUP_CONFINE_WORK tiles (default 0 = off) runs on PM_GROUP_CONFINE_XCDS XCDs
int group_confine_xcds() {
  static const int v = [] { const char* e = getenv("PM_GROUP_CONFINE_XCDS"); const int x = e && e[0] ? atoi(e) : 4; return x < 1 ? 1 : (x > 8 ? 8 : x); }();
  return v;
}
int group_confine_max_work() {
  static const int v = [] { const char* e = getenv("PM_GROUP_CONFINE_WORK"); return e && e[0] ? atoi(e) : 0; }();
  return v;
}

// experiment hook, read once: PM_GROUP_KERNEL=4 -> the 4-wave (one wave per SIMD, 128 x 128 per wave) software-pipelined body
int group_kernel_waves() {
  static const int v = [] { const char* e = getenv("PM_GROUP_KERNEL"); return e && e[0] ? atoi(e) : 8; }();
  return v;
}

// experiment hook, read once: PM_GROUP_FORCE_SPLIT=2|3 cuts the tiles of a LARGE group (ViT-B block: 108 tiles) into k-slices too,
// so that a launch limited to fewer workgroups than tiles (max_blocks) walks equal shares (scratch/archive_r3/r3_exp17.sh)
int group_force_split() {
  static const int v = [] { const char* e = getenv("PM_GROUP_FORCE_SPLIT"); return e && e[0] ? atoi(e) : 0; }();
  return v;
}

struct GroupPlan {
  int status;        // PM_OK or the refusal
  int bn;            // tile width
  int split;         // k-slices per tile
  int ksteps_split;
  size_t ws_bytes;   // slabs + partial row sums (0 when split == 1)
};

GroupPlan plan_group(const pm_wgrad_item* items, int n, int K, int in_dtype) {
  GroupPlan pl{PM_OK, 256, 1, 0, 0};
  if (!items || n <= 0) { pl.status = PM_EINVAL; return pl; }
  if (n > kMaxGroup || K <= 0) { pl.status = PM_ESHAPE; return pl; }
  if (!is16(in_dtype)) { pl.status = PM_ESHAPE; return pl; }               // (f32 mode keeps the per-GEMM split-K path)
  if ((K % V3_KE) != 0 || K < 2048) { pl.status = PM_ESHAPE; return pl; }  // ring kernel: whole 32-element k-steps, long reduction
  long t256 = 0;
  for (int i = 0; i < n; ++i) {
    const pm_wgrad_item& it = items[i];
    if (!it.dY || !it.X || !it.dW) { pl.status = PM_EINVAL; return pl; }
    if (it.n_out < 256 || it.n_in < 128) { pl.status = PM_ESHAPE; return pl; }
    if ((it.n_out & 7) || (it.n_in & 7) || (it.lddy & 7) || (it.ldx & 7) || (it.lddw & 3)) { pl.status = PM_EALIGN; return pl; }
    if (((uintptr_t)it.dY & 15) || ((uintptr_t)it.X & 15) || ((uintptr_t)it.dW & 15)) { pl.status = PM_EALIGN; return pl; }
    t256 += (long)((it.n_out + 255) / 256) * ((it.n_in + 255) / 256);
  }
  // >= 64 tiles of 256x256 (a ViT-B block: 108): one full-K tile per workgroup, no slabs.  Fewer (the 512-wide MAE decoder
  // block: 48 tiles, K = 50 432 tokens): each tile is cut into k-slices so that tiles x slices ~ the chip (48 x 4 = 192 work
  // items of 394 k-steps), f32 partials in the caller's workspace, ONE reduce launch for the whole group.
  const int nk = K / V3_KE;
  if (t256 < 64 || group_force_split() > 1) {
    int split = t256 < 64 ? (int)(group_split_target() / t256) : group_force_split();
    if (split > 8) split = 8;
    if (split > nk / 128) split = nk / 128;  // >= 128 k-steps per slice: prologue / epilogue stay small
    if (split > 1) {
      pl.ksteps_split = (nk + split - 1) / split;
      pl.split = (nk + pl.ksteps_split - 1) / pl.ksteps_split;  // no empty slice
    }
  }
  if (pl.split > 1) {
    for (int i = 0; i < n; ++i) {
      pl.ws_bytes += (size_t)pl.split * items[i].n_out * items[i].n_in * sizeof(float);
      if (items[i].dbias) pl.ws_bytes += (((size_t)pl.split * items[i].n_out * sizeof(float)) + 15) & ~(size_t)15;
    }
  } else {
    pl.bn = t256 >= 64 ? 256 : 128;  // without slices: 256x128 tiles double the workgroups of a small group
  }
  return pl;
}

}  // namespace

extern "C" size_t pm_wgrad_group_workspace_bytes(const pm_wgrad_item* items, int n, int K, int in_dtype) {
  const GroupPlan pl = plan_group(items, n, K, in_dtype);
  return pl.status == PM_OK ? pl.ws_bytes : 0;
}

extern "C" int pm_wgrad_group_plan(const pm_wgrad_item* items, int n, int K, int in_dtype, size_t* ws_bytes, int* tiles256,
                                   int* k_slices) {
  const GroupPlan pl = plan_group(items, n, K, in_dtype);
  if (ws_bytes) *ws_bytes = pl.status == PM_OK ? pl.ws_bytes : 0;
  if (k_slices) *k_slices = pl.status == PM_OK ? pl.split : 0;
  if (tiles256) {
    long t = 0;
    if (pl.status == PM_OK)
      for (int i = 0; i < n; ++i) t += (long)((items[i].n_out + 255) / 256) * ((items[i].n_in + 255) / 256);
    *tiles256 = (int)t;
  }
  return pl.status;
}

extern "C" int pm_wgrad_group(const pm_wgrad_item* items, int n, int K, int in_dtype, int max_blocks, void* workspace,
                              size_t ws_bytes, void* stream) {
  GroupPlan pl = plan_group(items, n, K, in_dtype);
  if (pl.status != PM_OK) return pl.status;
  if (pl.split > 1 && (!workspace || ws_bytes < pl.ws_bytes || ((uintptr_t)workspace & 15))) {
    pl.split = 1;  // no room for the slabs: whole-K tiles, narrower so that there are twice as many
    pl.bn = 128;
  }
  if (max_blocks == PM_GROUP_WHOLE_K) {  // the caller pairs this small group with another launch: whole-K 256x256 tiles, no slabs
    pl.split = 1;
    pl.bn = 256;
  }
  const int bn = pl.bn;
  WgradGroupArgs g;
  g.n = n; g.K = K; g.split = pl.split; g.ksteps_split = pl.ksteps_split;
  g.auto_order = group_order_auto();
#ifdef PM_GEMM_STAMP
  g.stamps = g_stamps ? g_stamps + 4096L * 8 * 16 : nullptr;  // rows [4096, 4608) x 8 waves of the caller's stamp buffer
#endif
  int total = 0;
  long vec = 0;
  char* ws = reinterpret_cast<char*>(workspace);
  for (int i = 0; i < n; ++i) {
    const pm_wgrad_item& it = items[i];
    WgradProb& p = g.p[i];
    p.dY = it.dY; p.X = it.X; p.dW = it.dW; p.lddy = it.lddy; p.ldx = it.ldx; p.lddw = it.lddw;
    p.M = it.n_out; p.N = it.n_in; p.accumulate = it.accumulate ? 1 : 0; p.dbias = it.dbias;
    p.tiles_n = (it.n_in + bn - 1) / bn;
    p.tiles_m = (it.n_out + 255) / 256;
    p.tile_begin = total;
    total += ((it.n_out + 255) / 256) * p.tiles_n;
    p.slab = nullptr; p.bias_part = nullptr; p.vec_begin = vec;
    if (pl.split > 1) {
      p.slab = reinterpret_cast<float*>(ws);
      ws += (size_t)pl.split * it.n_out * it.n_in * sizeof(float);
      if (it.dbias) {
        p.bias_part = reinterpret_cast<float*>(ws);
        ws += (((size_t)pl.split * it.n_out * sizeof(float)) + 15) & ~(size_t)15;
      }
      vec += (long)it.n_out * it.n_in / 4;
    }
  }
  for (int i = n; i < kMaxGroup; ++i) g.p[i] = g.p[0];
  g.total_tiles = total;
  g.total_vec = vec; g.bias_begin = vec; g.bias_total = 0;
  if (pl.split > 1)
    for (int i = 0; i < n; ++i)
      if (items[i].dbias) g.bias_total += items[i].n_out;
  const int work = total * pl.split;
  int grid = work;
  g.xcds = 8;
  if (max_blocks > 0 && max_blocks < work) {
    grid = (max_blocks / 8) * 8;  // whole XCD rounds: a workgroup keeps walking its own XCD's run of items
    if (grid < 8) grid = 8;
    if (grid > work) grid = work;
  } else if (pl.split == 1 && work <= group_confine_max_work() && group_confine_xcds() < 8) {
    // A small whole-K group (the (proj, qkv) launch of a ViT-B block: 36 tiles) on all 8 XCDs gives each L2 a run of 4-5 tiles:
    // few tiles share a dY / X panel in one L2, the panels are fetched by several (PMC: 258.8 MB against 125.7 MB algorithmic).
    // Confined to 4 XCDs the runs are 9 tiles long.  The hardware deals workgroups to the XCDs round-robin by id, so the grid is
    // 8 / xcds times the work and the workgroups that land on the other XCDs return at once.
    g.xcds = group_confine_xcds();
    grid = (work + g.xcds - 1) / g.xcds * 8;
  }
  hipStream_t s = pm_stream(stream);
  PM_DISPATCH_16(in_dtype, E, {
    if (bn == 256 && group_kernel_waves() == 4) {
      auto kern = wgrad_group_kernel<256, 256, 2, 2, E, 4>;
      constexpr int ring = 4 * (256 + 256) * 64;
      PM_ALLOW_LDS(kern, ring);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ring, s, g);
    } else if (bn == 256) {
      auto kern = wgrad_group_kernel<256, 256, 2, 4, E>;
      constexpr int ring = 4 * (256 + 256) * 64;
      PM_ALLOW_LDS(kern, ring);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), ring, s, g);
    } else {
      auto kern = wgrad_group_kernel<256, 128, 4, 2, E>;
      constexpr int ring = 4 * (256 + 128) * 64;
      PM_ALLOW_LDS(kern, ring);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), ring, s, g);
    }
  });
  int st = pm_check_launch();
  if (st || pl.split == 1) return st;
  const long items_r = g.bias_begin + g.bias_total;
  int rgrid = (int)((items_r + 255) / 256);
  if (rgrid > 2048) rgrid = 2048;
  hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3(rgrid), dim3(256), 0, s, g);
  return pm_check_launch();
}
