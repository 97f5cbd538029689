// pm_layernorm.hip -- LayerNorm forward / backward for the pre-LN ViT blocks (HBM-bound).
// Replaces nn.LayerNorm(eps=1e-6) of timm Block.norm1/norm2 and MaskedAutoencoderViT.norm/decoder_norm
// (reference models_mae.py:39-42,53-57,168,188).  One wave64 per row, the row lives in registers
// (D <= 1280, D % 4 == 0: 768 / 512 for ViT-B/16 and its MAE decoder, 1024 / 1280 for ViT-L / ViT-H; the number of f32x4 slots
// per lane is a template parameter), f32 statistics, two-pass variance.
// Algorithmic bytes per row: fwd 4D (x) + sizeof(act)*D (y); bwd sizeof(act)*D (dy) + 4D (x) + 4D (dres)
// + 4D (dx) + sizeof(act)*D (dx_act).
#include "pm_common.h"

namespace {

constexpr int kMaxD = 1280;  // 5 f32x4 slots per lane (ViT-H)

template <typename TOut, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TOut* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int M,
                                                     int D, float eps) {
  constexpr int kMaxVec = NV;  // f32x4 slots per lane for this D: D <= 256 * NV
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nvec = D >> 2;
  for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
    const float* xr = x + row * ldx;
    f32x4 v[kMaxVec];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * c);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[i][e] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
    TOut* yr = y + row * (long)D;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
        store4<TOut>(yr + 4 * c, o);
      }
    }
  }
}

template <typename TDy, typename TAct, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDy* __restrict__ dy, const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     long lddres, float* __restrict__ dx, long lddx,
                                                     TAct* __restrict__ dx_act, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dcolsum,
                                                     float* __restrict__ partials, int M, int D) {
  __shared__ float red[3][4][256 + 4];  // [vector][wave][lane*4 + e] per vec slot, reused per slot
  constexpr int kMaxVec = NV;  // f32x4 slots per lane for this D (shadows the file-scope bound): D <= 256 * NV
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nvec = D >> 2;
  f32x4 g[kMaxVec];
  f32x4 acc_g[kMaxVec], acc_b[kMaxVec], acc_c[kMaxVec];
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i) {
    const int c = lane + 64 * i;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    g[i] = (c < nvec) ? *reinterpret_cast<const f32x4*>(gamma + 4 * c) : z;
    acc_g[i] = z;
    acc_b[i] = z;
    acc_c[i] = z;
  }
  const float invD = 1.0f / (float)D;
  // Two rows per wave per iteration: their loads and the two shuffle-reduction chains are independent, which is the
  // instruction-level parallelism this latency-bound loop was missing (one wave otherwise serialises
  // load -> 2 x 6 shuffles -> store per row).
  const long stride = (long)gridDim.x * 4;
  for (long row0 = (long)blockIdx.x * 4 + wave; row0 < M; row0 += 2 * stride) {
    long rows[2] = {row0, row0 + stride};
    const bool valid[2] = {true, rows[1] < M};
    if (!valid[1]) rows[1] = row0;  // harmless duplicate loads; contributions and stores are suppressed below
    float mu[2], rs[2], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    f32x4 xh[2][kMaxVec], dv[2][kMaxVec], dr[2][kMaxVec];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      mu[u] = mean[rows[u]];
      rs[u] = rstd[rows[u]];
#pragma unroll
      for (int i = 0; i < kMaxVec; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
          xh[u][i] = *reinterpret_cast<const f32x4*>(x + rows[u] * ldx + 4 * c);
          dv[u][i] = load4<TDy>(dy + rows[u] * (long)D + 4 * c);
          if (dres) dr[u][i] = *reinterpret_cast<const f32x4*>(dres + rows[u] * lddres + 4 * c);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float live = valid[u] ? 1.0f : 0.0f;
#pragma unroll
      for (int i = 0; i < kMaxVec; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xh[u][i][e] = (xh[u][i][e] - mu[u]) * rs[u];
            const float gg = dv[u][i][e] * g[i][e];
            s1[u] += gg;
            s2[u] += gg * xh[u][i][e];
            acc_g[i][e] += live * dv[u][i][e] * xh[u][i][e];
            acc_b[i][e] += live * dv[u][i][e];
          }
        }
      }
    }
    float c1[2], c2[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      c1[u] = s1[u];
      c2[u] = s2[u];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {  // the four reductions interleaved
      c1[0] += __shfl_xor(c1[0], o, 64);
      c1[1] += __shfl_xor(c1[1], o, 64);
      c2[0] += __shfl_xor(c2[0], o, 64);
      c2[1] += __shfl_xor(c2[1], o, 64);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!valid[u]) continue;
      const float k1 = c1[u] * invD, k2 = c2[u] * invD;
#pragma unroll
      for (int i = 0; i < kMaxVec; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rs[u] * (dv[u][i][e] * g[i][e] - k1 - xh[u][i][e] * k2);
          if (dres) o += dr[u][i];
          *reinterpret_cast<f32x4*>(dx + rows[u] * lddx + 4 * c) = o;
          if (dx_act) store4<TAct>(dx_act + rows[u] * (long)D + 4 * c, o);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc_c[i][e] += o[e];
        }
      }
    }
  }
  // cross-wave reduction of the column partials, then one atomic per column per block
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i) {
    const int c = lane + 64 * i;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][wave][lane * 4 + e] = acc_g[i][e];
      red[1][wave][lane * 4 + e] = acc_b[i][e];
      red[2][wave][lane * 4 + e] = acc_c[i][e];
    }
    __syncthreads();
    if (wave < 3 && c < nvec) {
      float* dst = wave == 0 ? dgamma : (wave == 1 ? dbeta : dcolsum);
      if (dst) {
        f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          t[e] = (red[wave][0][lane * 4 + e] + red[wave][1][lane * 4 + e]) +
                 (red[wave][2][lane * 4 + e] + red[wave][3][lane * 4 + e]);
        if (partials) {  // two-stage: plain store of this block's partial row, summed by ln_bwd_reduce_kernel
          *reinterpret_cast<f32x4*>(partials + ((long)blockIdx.x * 3 + wave) * D + 4 * c) = t;
        } else {         // no workspace: one atomic per column per block (contended when the grid is large)
#pragma unroll
          for (int e = 0; e < 4; ++e) atomicAdd(dst + 4 * c + e, t[e]);
        }
      }
    }
  }
}

// dst_v[d] += sum over blocks of partials[b][v][d], fixed order (deterministic)
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ partials, float* dgamma, float* dbeta,
                                                             float* dcolsum, int nblocks, int D) {
  // 64 columns per block, 16 row groups (one per wave) each summing every 16th partial row with 4 independent
  // accumulators; the order of additions is fixed by (nblocks) only -> run-to-run deterministic.
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + lane;
  const int v = blockIdx.y;
  float* dst = v == 0 ? dgamma : (v == 1 ? dbeta : dcolsum);
  if (!dst) return;
  float s0 = 0.f;
  if (d < D) s0 = strided_sum<16>(partials + (long)v * D + d, 3L * D, rg, 16, nblocks);
  red[rg][lane] = s0;
  __syncthreads();
  if (rg == 0 && d < D) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][lane];
    dst[d] += t;
  }
}

inline int ln_grid(int M) {
  int g = (M + 3) / 4;
  return g < 1 ? 1 : (g > 1024 ? 1024 : g);
}

}  // namespace

extern "C" int pm_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, void* y, int out_dtype,
                                float* mean, float* rstd, int M, int D, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd) return PM_EINVAL;
  if (M <= 0 || D <= 0 || D > kMaxD || (D & 3) || (ldx & 3)) return PM_ESHAPE;
  const int grid = (M + 3) / 4 > 4096 ? 4096 : (M + 3) / 4;
#define PM_LN_FWD(TO, NV) \
  hipLaunchKernelGGL((ln_fwd_kernel<TO, NV>), dim3(grid), dim3(256), 0, pm_stream(stream), x, ldx, gamma, beta, (TO*)y, mean, rstd, M, D, eps)
  PM_DISPATCH_ACT(out_dtype, T, {
    if (D <= 1024) PM_LN_FWD(T, 4); else PM_LN_FWD(T, 5);
  });
#undef PM_LN_FWD
  return pm_check_launch();
}

extern "C" int pm_layernorm_bwd(const void* dy, int dy_dtype, const float* x, long ldx, const float* gamma,
                                const float* mean, const float* rstd, const float* dres, long lddres, float* dx,
                                long lddx, void* dx_act, int act_dtype, float* dgamma, float* dbeta, float* dcolsum,
                                int M, int D, void* workspace, size_t ws_bytes, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx) return PM_EINVAL;
  if (M <= 0 || D <= 0 || D > kMaxD || (D & 3) || (ldx & 3) || (lddx & 3) || (dres && (lddres & 3))) return PM_ESHAPE;
  if (dx_act && act_dtype != dy_dtype) return PM_EINVAL;
  int grid = (M + 3) / 4;
  float* partials = nullptr;
  const bool want_sums = dgamma || dbeta || dcolsum;
  if (want_sums && workspace && ws_bytes >= (size_t)64 * 3 * D * sizeof(float)) {
    int cap = (int)(ws_bytes / ((size_t)3 * D * sizeof(float)));
    if (cap > 1024) cap = 1024;
    if (grid > cap) grid = cap;
    partials = reinterpret_cast<float*>(workspace);
  } else if (grid > 256) {
    grid = 256;  // atomics fallback: keep the number of contending blocks low
  }
  if (grid < 1) grid = 1;
  hipStream_t s = pm_stream(stream);
  const int nv = (D + 255) / 256;  // f32x4 slots per lane: 3 for D = 768, 2 for 512
#define PM_LN_BWD(TD, NV)                                                                                               \
  hipLaunchKernelGGL((ln_bwd_kernel<TD, TD, NV>), dim3(grid), dim3(256), 0, s, (const TD*)dy, x, ldx, gamma, mean, rstd, \
                     dres, lddres, dx, lddx, (TD*)dx_act, dgamma, dbeta, dcolsum, partials, M, D)
  PM_DISPATCH_ACT(dy_dtype, T, {
    if (nv == 1) PM_LN_BWD(T, 1); else if (nv == 2) PM_LN_BWD(T, 2); else if (nv == 3) PM_LN_BWD(T, 3);
    else if (nv == 4) PM_LN_BWD(T, 4); else PM_LN_BWD(T, 5);
  });
#undef PM_LN_BWD
  if (partials)
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((D + 63) / 64, 3), dim3(1024), 0, s, partials, dgamma, dbeta, dcolsum, grid, D);
  return pm_check_launch();
}
