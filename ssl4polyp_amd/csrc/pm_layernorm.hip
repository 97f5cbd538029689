// pm_layernorm.hip -- LayerNorm forward / backward for the pre-LN ViT blocks (HBM-bound).
// Replaces nn.LayerNorm(eps=1e-6) of timm Block.norm1/norm2 and MaskedAutoencoderViT.norm/decoder_norm
// (reference models_mae.py:39-42,53-57,168,188).  One wave64 per row, the row lives in registers
// (D <= 1024, D % 4 == 0: 768 / 512 for ViT-B/16 and its MAE decoder), f32 statistics, two-pass variance.
// Algorithmic bytes per row: fwd 4D (x) + sizeof(act)*D (y); bwd sizeof(act)*D (dy) + 4D (x) + 4D (dres)
// + 4D (dx) + sizeof(act)*D (dx_act).
#include "pm_common.h"

namespace {

constexpr int kMaxVec = 4;  // f32x4 per lane -> D <= 4*64*4 = 1024

template <typename TOut>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TOut* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int M,
                                                     int D, float eps) {
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

template <typename TDy, typename TAct>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDy* __restrict__ dy, const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     long lddres, float* __restrict__ dx, long lddx,
                                                     TAct* __restrict__ dx_act, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dcolsum,
                                                     float* __restrict__ partials, int M, int D) {
  __shared__ float red[3][4][256 + 4];  // [vector][wave][lane*4 + e] per vec slot, reused per slot
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
  for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
    const float mu = mean[row];
    const float rs = rstd[row];
    const float* xr = x + row * ldx;
    const TDy* dyr = dy + row * (long)D;
    f32x4 xh[kMaxVec], dv[kMaxVec];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + 4 * c);
        dv[i] = load4<TDy>(dyr + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[i][e] = (xv[e] - mu) * rs;
          const float gg = dv[i][e] * g[i][e];
          s1 += gg;
          s2 += gg * xh[i][e];
          acc_g[i][e] += dv[i][e] * xh[i][e];
          acc_b[i][e] += dv[i][e];
        }
      }
    }
    const float c1 = wave_sum(s1) * invD;
    const float c2 = wave_sum(s2) * invD;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (dv[i][e] * g[i][e] - c1 - xh[i][e] * c2);
        if (dres) {
          const f32x4 r = *reinterpret_cast<const f32x4*>(dres + row * lddres + 4 * c);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += r[e];
        }
        *reinterpret_cast<f32x4*>(dx + row * lddx + 4 * c) = o;
        if (dx_act) store4<TAct>(dx_act + row * (long)D + 4 * c, o);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc_c[i][e] += o[e];
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
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (d < D) {
    const float* p = partials + (long)v * D + d;
    const long stride = 3L * D;
    int b = rg;
    for (; b + 48 < nblocks; b += 64) {
      s0 += p[(long)b * stride];
      s1 += p[(long)(b + 16) * stride];
      s2 += p[(long)(b + 32) * stride];
      s3 += p[(long)(b + 48) * stride];
    }
    for (; b < nblocks; b += 16) s0 += p[(long)b * stride];
  }
  red[rg][lane] = (s0 + s1) + (s2 + s3);
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
  if (M <= 0 || D <= 0 || D > 1024 || (D & 3) || (ldx & 3)) return PM_ESHAPE;
  const int grid = (M + 3) / 4 > 4096 ? 4096 : (M + 3) / 4;
  if (out_dtype == PM_BF16)
    hipLaunchKernelGGL(ln_fwd_kernel<__bf16>, dim3(grid), dim3(256), 0, pm_stream(stream), x, ldx, gamma, beta,
                       (__bf16*)y, mean, rstd, M, D, eps);
  else if (out_dtype == PM_F32)
    hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3(grid), dim3(256), 0, pm_stream(stream), x, ldx, gamma, beta,
                       (float*)y, mean, rstd, M, D, eps);
  else
    return PM_EINVAL;
  return pm_check_launch();
}

extern "C" int pm_layernorm_bwd(const void* dy, int dy_dtype, const float* x, long ldx, const float* gamma,
                                const float* mean, const float* rstd, const float* dres, long lddres, float* dx,
                                long lddx, void* dx_act, int act_dtype, float* dgamma, float* dbeta, float* dcolsum,
                                int M, int D, void* workspace, size_t ws_bytes, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx) return PM_EINVAL;
  if (M <= 0 || D <= 0 || D > 1024 || (D & 3) || (ldx & 3) || (lddx & 3) || (dres && (lddres & 3))) return PM_ESHAPE;
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
  if (dy_dtype == PM_BF16)
    hipLaunchKernelGGL((ln_bwd_kernel<__bf16, __bf16>), dim3(grid), dim3(256), 0, s, (const __bf16*)dy, x, ldx, gamma, mean,
                       rstd, dres, lddres, dx, lddx, (__bf16*)dx_act, dgamma, dbeta, dcolsum, partials, M, D);
  else if (dy_dtype == PM_F32)
    hipLaunchKernelGGL((ln_bwd_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)dy, x, ldx, gamma, mean,
                       rstd, dres, lddres, dx, lddx, (float*)dx_act, dgamma, dbeta, dcolsum, partials, M, D);
  else
    return PM_EINVAL;
  if (partials)
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((D + 63) / 64, 3), dim3(1024), 0, s, partials, dgamma, dbeta, dcolsum, grid, D);
  return pm_check_launch();
}
