// pm_misc.hip -- the HBM-bound kernels around the GEMMs: patch im2col (+ kept-patch gather), token
// assembly (cls / pos-embed), bias-gradient column sums, casts, fused AdamW (the classifier head: pm_head.hip).
// Reference ops replaced are named per kernel.  All are streaming kernels: 16-B vector accesses,
// one wave64 per token row where a row is the unit, grid capped and grid-strided.
#include "pm_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// column sums: out[n] += sum_m x[m][n]   (bias gradients)
// ---------------------------------------------------------------------------------------------
// `partials` != NULL: each (column strip, row split) block stores its partial row [blockIdx.y][N]; colsum_reduce_kernel
// then sums the splits in a fixed order (deterministic).  NULL: one float atomic per column per block.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, long ldx, float* __restrict__ out,
                                                     float* __restrict__ partials, int M, int N) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 256 + lane * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    for (long m = (long)blockIdx.y * 4 + wave; m < M; m += (long)gridDim.y * 4) acc += load4<T>(x + m * ldx + n);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wave][lane * 4 + e] = acc[e];
  __syncthreads();
  const int t = threadIdx.x;
  const int nn = blockIdx.x * 256 + t;
  if (nn < N) {
    const float s = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    if (partials) partials[(long)blockIdx.y * N + nn] = s;
    else atomicAdd(out + nn, s);
  }
}

__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ partials, float* __restrict__ out,
                                                             int splits, int N) {
  // 64 columns per block; the 16 waves each sum every 16th split (fixed order, 8 loads in flight), then one LDS combine
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  red[rg][lane] = n < N ? strided_sum<8>(partials + n, N, rg, 16, splits) : 0.f;
  __syncthreads();
  if (rg == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][lane];
    out[n] += t;
  }
}

// ---------------------------------------------------------------------------------------------
// im2col of the k = s = p patch-embedding conv, optionally gathering only kept patches
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ imgs, const int* __restrict__ ids_keep,
                                                     T* __restrict__ cols, long ldcols, int C, int img, int p, int keep) {
  const int b = blockIdx.x / keep, j = blockIdx.x % keep;
  const int grid = img / p;
  const int pid = ids_keep ? ids_keep[b * keep + j] : j;
  const int gy = pid / grid, gx = pid % grid;
  const float* src = imgs + (long)b * C * img * img + (long)gy * p * img + gx * p;
  T* dst = cols + (long)blockIdx.x * ldcols;
  const int PE = C * p * p;
  if ((p & 3) == 0) {  // 16-B reads of a patch row (p = 16: 64 B per row)
    const int p4 = p >> 2;
    const int nvec = C * p * p4;
    for (int v = threadIdx.x; v < nvec; v += blockDim.x) {
      const int c = v / (p * p4), rem = v % (p * p4);
      const int py = rem / p4, px = (rem % p4) * 4;
      const f32x4 val = *reinterpret_cast<const f32x4*>(src + (long)c * img * img + (long)py * img + px);
      store4<T>(dst + (c * p + py) * p + px, val);
    }
  } else {             // p = 14 (ViT-H/14): a patch row starts on an 8-B boundary only; element-wise
    for (int e = threadIdx.x; e < PE; e += blockDim.x) {
      const int c = e / (p * p), rem = e % (p * p);
      const int py = rem / p, px = rem % p;
      dst[e] = from_f32<T>(src[(long)c * img * img + (long)py * img + px]);
    }
  }
  for (int e = PE + threadIdx.x; e < ldcols; e += blockDim.x) dst[e] = from_f32<T>(0.f);  // row padding (K of the GEMM)
}

// dst[r][c] = r < rows && c < cols ? src[r][c] : 0 over rows_pad x cols_pad, f32 -> act: a weight whose reduction dimension is not
// a multiple of the GEMM's k-step (588 = 3 x 14 x 14) laid out as the zero-padded operand the matrix-core kernels take
template <typename T>
__global__ __launch_bounds__(256) void pad_cast_kernel(const float* __restrict__ src, long lds, T* __restrict__ dst, long ldd, int rows,
                                                       int cols, int rows_pad, int cols_pad) {
  const long total = (long)rows_pad * cols_pad;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = i / cols_pad, c = i % cols_pad;
    const float v = (r < rows && c < cols) ? src[(long)r * lds + c] : 0.f;
    dst[(long)r * ldd + c] = from_f32<T>(v);
  }
}

// dst[r][c] (+)= src[r][c] over rows x cols: the valid part of a gradient computed in the padded layout
__global__ __launch_bounds__(256) void unpad_add_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd,
                                                        int rows, int cols, int accumulate) {
  const long total = (long)rows * cols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = i / cols, c = i % cols;
    const float v = src[(long)r * lds + c];
    float* d = dst + (long)r * ldd + c;
    *d = accumulate ? *d + v : v;
  }
}

// ---------------------------------------------------------------------------------------------
// token assembly (cls + pos-embed) and its backward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assemble_kernel(const float* __restrict__ emb, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, const int* __restrict__ ids_keep,
                                                       float* __restrict__ x, int B, int keep, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long rows = (long)B * (keep + 1);
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const int b = row / (keep + 1), t = row % (keep + 1);
    const float* src;
    const float* pp;
    if (t == 0) {
      src = cls;
      pp = pos;
    } else {
      const int j = t - 1;
      const int pid = ids_keep ? ids_keep[b * keep + j] : j;
      src = emb + ((long)b * keep + j) * D;
      pp = pos + (long)(1 + pid) * D;
    }
    for (int c = lane * 4; c < D; c += 256) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + c);
      const f32x4 q = *reinterpret_cast<const f32x4*>(pp + c);
      *reinterpret_cast<f32x4*>(x + row * D + c) = a + q;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void assemble_bwd_kernel(const float* __restrict__ dx, const int* __restrict__ ids_keep,
                                                           T* __restrict__ demb, float* __restrict__ dpos, int B, int keep,
                                                           int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long rows = (long)B * keep;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const int b = row / keep, j = row % keep;
    const float* src = dx + ((long)b * (keep + 1) + 1 + j) * D;
    const int pid = ids_keep ? ids_keep[b * keep + j] : j;
    for (int c = lane * 4; c < D; c += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
      store4<T>(demb + row * D + c, v);
      if (dpos) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dpos + (long)(1 + pid) * D + c + e, v[e]);
      }
    }
  }
}

// dcls[d] += sum_b dx[b, 0, d]   (also dpos[0] when the positional table is learnable)
// The rows are one sample apart (600 KB): a thread that walks all B of them serially pays B dependent-latency loads at the
// very end of the backward pass, when nothing else runs (27 us at B = 64, 92 us at B = 256).  64 columns per block, the four
// waves take every fourth sample with four independent accumulators each, partials combined in wave order: fixed order,
// ~2 rounds of load latency.
__global__ __launch_bounds__(256) void cls_grad_kernel(const float* __restrict__ dx, float* __restrict__ dcls,
                                                       float* __restrict__ dpos0, int B, long sample_stride, int D) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (d < D) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = wave;
    for (; b + 12 < B; b += 16) {
      a0 += dx[(long)b * sample_stride + d];
      a1 += dx[(long)(b + 4) * sample_stride + d];
      a2 += dx[(long)(b + 8) * sample_stride + d];
      a3 += dx[(long)(b + 12) * sample_stride + d];
    }
    for (; b < B; b += 4) a0 += dx[(long)b * sample_stride + d];
    s = (a0 + a1) + (a2 + a3);
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && d < D) {
    const float t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    dcls[d] += t;
    if (dpos0) dpos0[d] += t;
  }
}

// ---------------------------------------------------------------------------------------------
// input tail: uint8 HWC frames -> normalised f32 NCHW  (torchvision ToTensor + Normalize, optional flips)
// ---------------------------------------------------------------------------------------------
// One thread = 4 consecutive output pixels of one row: 12 source bytes -> one f32x4 per channel plane.
// Arithmetic and its order are torchvision's: v = float(u8) / 255 ; (v - mean) / std, IEEE f32 division.
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ src,
                                                            const unsigned char* __restrict__ flags, float* __restrict__ dst,
                                                            int B, int H, int W, f32x4 mean, f32x4 stdv) {
  const int w4 = W >> 2;
  const long total = (long)B * H * w4;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int xq = id % w4;
    const int y = (id / w4) % H;
    const int b = id / ((long)w4 * H);
    const int f = flags ? flags[b] : 0;
    const int sy = (f & 2) ? H - 1 - y : y;
    const unsigned char* row = src + ((long)b * H + sy) * W * 3;
    float px[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x = xq * 4 + i;
      const int sx = (f & 1) ? W - 1 - x : x;
#pragma unroll
      for (int c = 0; c < 3; ++c) px[i][c] = (float)row[sx * 3 + c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __fdiv_rn(__fdiv_rn(px[i][c], 255.0f) - mean[c], stdv[c]);
      *reinterpret_cast<f32x4*>(dst + (((long)b * 3 + c) * H + y) * W + xq * 4) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// cast
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
  const long nvec = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256)
    store4<T>(dst + 4 * i, *reinterpret_cast<const f32x4*>(src + 4 * i));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(nvec << 2) + threadIdx.x] = (T)src[(nvec << 2) + threadIdx.x];
}

// out = dy * gelu_erf'(pre): the backward of timm Mlp.act as a stand-alone elementwise pass (the training step gets it from the
// dfc2 GEMM's PM_EPI_DGELU epilogue; this is for callers that compose their own blocks from the registered ops)
template <typename T>
__global__ __launch_bounds__(256) void dgelu_kernel(const T* __restrict__ dy, const T* __restrict__ pre, T* __restrict__ out, long n) {
  const long nvec = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256)
    store4<T>(out + 4 * i, load4<T>(dy + 4 * i) * gelu_act_grad4<T>(load4<T>(pre + 4 * i)));
}

// ---------------------------------------------------------------------------------------------
// fused AdamW (torch.optim.AdamW update rule) + act-typed shadow refresh
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, T* __restrict__ shadow, long n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1, float rsqrt_bc2,
                                                    float gscale) {
  const long nvec = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    f32x4 pp = *reinterpret_cast<const f32x4*>(p + 4 * i);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * i);
    f32x4 mm = *reinterpret_cast<const f32x4*>(m + 4 * i);
    f32x4 vv = *reinterpret_cast<const f32x4*>(v + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = gg[e] * gscale;
      pp[e] *= (1.0f - lr * wd);
      mm[e] = beta1 * mm[e] + (1.0f - beta1) * gr;
      vv[e] = beta2 * vv[e] + (1.0f - beta2) * gr * gr;
      const float denom = sqrtf(vv[e]) * rsqrt_bc2 + eps;
      pp[e] -= (lr / bc1) * (mm[e] / denom);
    }
    *reinterpret_cast<f32x4*>(p + 4 * i) = pp;
    *reinterpret_cast<f32x4*>(m + 4 * i) = mm;
    *reinterpret_cast<f32x4*>(v + 4 * i) = vv;
    if (shadow) store4<T>(shadow + 4 * i, pp);
  }
}


// Device-resident hyper-parameters (hipGraph-replayable optimizer step): one 16-float record per param group
//   [0] lr [1] beta1 [2] beta2 [3] eps [4] weight_decay [5] grad_scale [6] step [7] bc1 [8] rsqrt_bc2
//   [9] skip (pm_loss_scale_update: this step's gradients hold inf / nan)  [10] 1 / loss scale (0 = no scaler: treated as 1)
constexpr int kHyperStride = 16;
__global__ void adamw_tick_kernel(float* __restrict__ hyper, int n_groups) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  float* h = hyper + (long)g * kHyperStride;
  if (h[9] != 0.f) return;  // a skipped step does not count (GradScaler.step: optimizer.step() is not called)
  const float step = h[6] + 1.0f;
  h[6] = step;
  h[7] = (float)(1.0 - pow((double)h[1], (double)step));
  h[8] = (float)(1.0 / sqrt(1.0 - pow((double)h[2], (double)step)));
}

template <typename T>
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, T* __restrict__ shadow, long n,
                                                        const float* __restrict__ hyper) {
  if (hyper[9] != 0.f) return;  // non-finite gradients under loss scaling: the update is skipped, nothing is touched
  const float lr = hyper[0], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], wd = hyper[4];
  const float gscale = hyper[5] * (hyper[10] != 0.f ? hyper[10] : 1.0f);
  const float bc1 = hyper[7], rsqrt_bc2 = hyper[8];
  const long nvec = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    f32x4 pp = *reinterpret_cast<const f32x4*>(p + 4 * i);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * i);
    f32x4 mm = *reinterpret_cast<const f32x4*>(m + 4 * i);
    f32x4 vv = *reinterpret_cast<const f32x4*>(v + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = gg[e] * gscale;
      pp[e] *= (1.0f - lr * wd);
      mm[e] = beta1 * mm[e] + (1.0f - beta1) * gr;
      vv[e] = beta2 * vv[e] + (1.0f - beta2) * gr * gr;
      const float denom = sqrtf(vv[e]) * rsqrt_bc2 + eps;
      pp[e] -= (lr / bc1) * (mm[e] / denom);
    }
    *reinterpret_cast<f32x4*>(p + 4 * i) = pp;
    *reinterpret_cast<f32x4*>(m + 4 * i) = mm;
    *reinterpret_cast<f32x4*>(v + 4 * i) = vv;
    if (shadow) store4<T>(shadow + 4 * i, pp);
  }
}

// ---------------------------------------------------------------------------------------------
// gradient statistics in one pass: out[0] += sum g^2, out[1] += #NaN, out[2] += #Inf
// (replaces the per-parameter host-synchronising loops of tc.py:1437-1454 and misc.py:387-400)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_stats_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
  __shared__ float red[3][4];
  float ss = 0.f, nn = 0.f, ni = 0.f;
  const long nvec = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ss += v[e] * v[e];
      nn += (v[e] != v[e]) ? 1.f : 0.f;
      ni += (fabsf(v[e]) == INFINITY) ? 1.f : 0.f;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(nvec << 2) + threadIdx.x];
    ss += v * v;
    nn += (v != v) ? 1.f : 0.f;
    ni += (fabsf(v) == INFINITY) ? 1.f : 0.f;
  }
  ss = wave_sum(ss);
  nn = wave_sum(nn);
  ni = wave_sum(ni);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = ss;
    red[1][wave] = nn;
    red[2][wave] = ni;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    atomicAdd(out + k, (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]));
  }
}

// ---------------------------------------------------------------------------------------------
// dynamic loss scaling on the device (precision mode fp16): torch.cuda.amp.GradScaler's step() + update()
// (tc.py:4533-4546, engine_pretrain.py:65-72 through misc.py:252-282) without its host read-back of found_inf
//   state[0] scale  [1] growth tracker  [2] found_inf of the last step  [3] skipped steps  [4] steps  [5] 1 / scale used
// ---------------------------------------------------------------------------------------------
__global__ void loss_scale_update_kernel(float* __restrict__ state, const float* __restrict__ stats, float* __restrict__ hyper,
                                         int n_groups, float growth, float backoff, int interval) {
  const float scale = state[0];
  const float found = (stats[1] + stats[2]) > 0.f ? 1.f : 0.f;
  const float tracker = state[1];
  __syncthreads();
  for (int g = threadIdx.x; g < n_groups; g += blockDim.x) {
    hyper[(long)g * kHyperStride + 9] = found;
    hyper[(long)g * kHyperStride + 10] = 1.0f / scale;
  }
  if (threadIdx.x == 0) {
    state[2] = found;
    state[4] += 1.f;
    state[5] = 1.0f / scale;
    if (found != 0.f) {
      state[0] = scale * backoff;
      state[1] = 0.f;
      state[3] += 1.f;
    } else {
      const float t = tracker + 1.f;
      if (t >= (float)interval) {
        state[0] = scale * growth;
        state[1] = 0.f;
      } else {
        state[1] = t;
      }
    }
  }
}

inline int cap_grid(long work_items, int per_block, int cap) {
  long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}


// ---------------------------------------------------------------------------------------------
// Row gather / scatter for the classifier's top block (engine.py BlockStack.backward, sparse_top): with out_token "cls" the head's
// backward leaves a gradient in ONE row per sample, and the MLP branch of the last block (dfc2 + dGELU, dfc1, LayerNorm backward,
// proj dgrad and three of the four weight gradients) sees nothing but those rows until the attention backward spreads them again.
// ---------------------------------------------------------------------------------------------
// dst[r][:] = src[idx[r]][:], rows of `words` 32-bit words (source row pitch ld_words)
__global__ __launch_bounds__(256) void gather_rows_kernel(const unsigned int* __restrict__ src, long ld_words, const int* __restrict__ idx,
                                                          unsigned int* __restrict__ dst, int R, long words) {
  const long total = (long)R * words;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = i / words;
    const long w = i % words;
    dst[i] = src[(long)idx[r] * ld_words + w];
  }
}
// dst[m][:] = inv[m] >= 0 ? src[inv[m]][:] : 0 for all M rows of `vecs` 16-byte vectors
__global__ __launch_bounds__(256) void scatter_rows_zero_kernel(const u32x4* __restrict__ src, const int* __restrict__ inv,
                                                                u32x4* __restrict__ dst, int M, long vecs) {
  const long total = (long)M * vecs;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = i / vecs;
    const long v = i % vecs;
    const int r = inv[m];
    dst[i] = r >= 0 ? src[(long)r * vecs + v] : u32x4{0u, 0u, 0u, 0u};
  }
}

}  // namespace

extern "C" int pm_colsum_ws(const void* x, long ldx, int dtype, float* out, int M, int N, void* workspace, size_t ws_bytes,
                            void* stream) {
  if (!x || !out) return PM_EINVAL;
  if (M <= 0 || N <= 0 || (N & 3) || (ldx & 3)) return PM_ESHAPE;
  const int splits = cap_grid(M, 64, 128);
  const dim3 grid((N + 255) / 256, splits);
  float* partials = (workspace && ws_bytes >= (size_t)splits * N * sizeof(float)) ? reinterpret_cast<float*>(workspace) : nullptr;
  hipStream_t s = pm_stream(stream);
  PM_DISPATCH_ACT(dtype, T, hipLaunchKernelGGL(colsum_kernel<T>, grid, dim3(256), 0, s, (const T*)x, ldx, out, partials, M, N));
  if (partials) hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 63) / 64), dim3(1024), 0, s, partials, out, splits, N);
  return pm_check_launch();
}

extern "C" int pm_colsum(const void* x, long ldx, int dtype, float* out, int M, int N, void* stream) {
  return pm_colsum_ws(x, ldx, dtype, out, M, N, nullptr, 0, stream);
}

extern "C" int pm_patch_im2col(const float* imgs, const int* ids_keep, void* cols, long ldcols, int out_dtype, int B, int C,
                               int img, int p, int keep, void* stream) {
  if (!imgs || !cols) return PM_EINVAL;
  if (B <= 0 || C <= 0 || img <= 0 || p <= 0 || keep <= 0 || (img % p)) return PM_ESHAPE;
  if (ldcols < (long)C * p * p) return PM_ESHAPE;
  if ((p & 3) == 0 && (ldcols & 3)) return PM_EALIGN;  // (vector stores)
  const int L = (img / p) * (img / p);
  if (keep > L || (!ids_keep && keep != L)) return PM_ESHAPE;
  const dim3 grid(B * keep);
  PM_DISPATCH_ACT(out_dtype, T, hipLaunchKernelGGL(im2col_kernel<T>, grid, dim3(256), 0, pm_stream(stream), imgs, ids_keep, (T*)cols,
                                                   ldcols, C, img, p, keep));
  return pm_check_launch();
}

extern "C" int pm_pad_cast(const float* src, long lds, void* dst, long ldd, int dst_dtype, int rows, int cols, int rows_pad,
                           int cols_pad, void* stream) {
  if (!src || !dst) return PM_EINVAL;
  if (rows <= 0 || cols <= 0 || rows_pad < rows || cols_pad < cols || lds < cols || ldd < cols_pad) return PM_ESHAPE;
  const long total = (long)rows_pad * cols_pad;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  PM_DISPATCH_ACT(dst_dtype, T, hipLaunchKernelGGL(pad_cast_kernel<T>, dim3(grid), dim3(256), 0, pm_stream(stream), src, lds, (T*)dst,
                                                   ldd, rows, cols, rows_pad, cols_pad));
  return pm_check_launch();
}

extern "C" int pm_unpad_add(const float* src, long lds, float* dst, long ldd, int rows, int cols, int accumulate, void* stream) {
  if (!src || !dst) return PM_EINVAL;
  if (rows <= 0 || cols <= 0 || lds < cols || ldd < cols) return PM_ESHAPE;
  const long total = (long)rows * cols;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(unpad_add_kernel, dim3(grid), dim3(256), 0, pm_stream(stream), src, lds, dst, ldd, rows, cols, accumulate ? 1 : 0);
  return pm_check_launch();
}

extern "C" int pm_assemble_tokens(const float* emb, const float* cls, const float* pos, const int* ids_keep, float* x,
                                  int B, int keep, int D, void* stream) {
  if (!emb || !cls || !pos || !x) return PM_EINVAL;
  if (B <= 0 || keep <= 0 || D <= 0 || (D & 3)) return PM_ESHAPE;
  hipLaunchKernelGGL(assemble_kernel, dim3(cap_grid((long)B * (keep + 1), 4, 4096)), dim3(256), 0, pm_stream(stream), emb,
                     cls, pos, ids_keep, x, B, keep, D);
  return pm_check_launch();
}

extern "C" int pm_assemble_tokens_bwd(const float* dx, const int* ids_keep, void* demb, int act_dtype, float* dcls,
                                      float* dpos, int B, int keep, int D, void* stream) {
  if (!dx || !demb) return PM_EINVAL;
  if (B <= 0 || keep <= 0 || D <= 0 || (D & 3)) return PM_ESHAPE;
  const dim3 grid(cap_grid((long)B * keep, 4, 4096));
  PM_DISPATCH_ACT(act_dtype, T, hipLaunchKernelGGL(assemble_bwd_kernel<T>, grid, dim3(256), 0, pm_stream(stream), dx, ids_keep,
                                                   (T*)demb, dpos, B, keep, D));
  if (dcls)
    hipLaunchKernelGGL(cls_grad_kernel, dim3((D + 63) / 64), dim3(256), 0, pm_stream(stream), dx, dcls, dpos, B,
                       (long)(keep + 1) * D, D);
  return pm_check_launch();
}

extern "C" int pm_preprocess_u8(const unsigned char* src, const unsigned char* flip_flags, float* dst, int B, int H, int W,
                                float mean_r, float mean_g, float mean_b, float std_r, float std_g, float std_b,
                                void* stream) {
  if (!src || !dst) return PM_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0 || (W & 3)) return PM_ESHAPE;
  if (((uintptr_t)dst & 15)) return PM_EALIGN;
  if (!(std_r > 0.f) || !(std_g > 0.f) || !(std_b > 0.f)) return PM_EINVAL;
  const long total = (long)B * H * (W >> 2);
  const dim3 grid(cap_grid(total, 256, 8192));
  hipLaunchKernelGGL(preprocess_u8_kernel, grid, dim3(256), 0, pm_stream(stream), src, flip_flags, dst, B, H, W,
                     f32x4{mean_r, mean_g, mean_b, 0.f}, f32x4{std_r, std_g, std_b, 1.f});
  return pm_check_launch();
}

extern "C" int pm_cast(const float* src, void* dst, int dst_dtype, long n, void* stream) {
  if (!src || !dst) return PM_EINVAL;
  if (n <= 0) return PM_ESHAPE;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return PM_EALIGN;
  const dim3 grid(cap_grid(n >> 2, 256, 4096));
  PM_DISPATCH_ACT(dst_dtype, T, hipLaunchKernelGGL(cast_kernel<T>, grid, dim3(256), 0, pm_stream(stream), src, (T*)dst, n));
  return pm_check_launch();
}

extern "C" int pm_gather_rows(const void* src, long ld_bytes, const int* idx, void* dst, int R, long row_bytes, void* stream) {
  if (!src || !idx || !dst) return PM_EINVAL;
  if (R <= 0 || row_bytes <= 0 || ld_bytes < row_bytes) return PM_ESHAPE;
  if ((row_bytes & 3) || (ld_bytes & 3) || ((uintptr_t)src & 3) || ((uintptr_t)dst & 3)) return PM_EALIGN;
  const long total = (long)R * (row_bytes >> 2);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cap_grid(total, 256, 4096)), dim3(256), 0, pm_stream(stream), (const unsigned int*)src,
                     ld_bytes >> 2, idx, (unsigned int*)dst, R, row_bytes >> 2);
  return pm_check_launch();
}

extern "C" int pm_scatter_rows_zero(const void* src, const int* inv, void* dst, int M, long row_bytes, void* stream) {
  if (!src || !inv || !dst) return PM_EINVAL;
  if (M <= 0 || row_bytes <= 0) return PM_ESHAPE;
  if ((row_bytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return PM_EALIGN;
  const long total = (long)M * (row_bytes >> 4);
  hipLaunchKernelGGL(scatter_rows_zero_kernel, dim3(cap_grid(total, 256, 8192)), dim3(256), 0, pm_stream(stream), (const u32x4*)src, inv,
                     (u32x4*)dst, M, row_bytes >> 4);
  return pm_check_launch();
}

extern "C" int pm_dgelu(const void* dy, const void* pre, void* out, int dtype, long n, void* stream) {
  if (!dy || !pre || !out) return PM_EINVAL;
  if (n <= 0 || (n & 3)) return PM_ESHAPE;
  if (((uintptr_t)dy & 7) || ((uintptr_t)pre & 7) || ((uintptr_t)out & 7)) return PM_EALIGN;
  const dim3 grid(cap_grid(n >> 2, 256, 4096));
  PM_DISPATCH_ACT(dtype, T, hipLaunchKernelGGL(dgelu_kernel<T>, grid, dim3(256), 0, pm_stream(stream), (const T*)dy, (const T*)pre,
                                               (T*)out, n));
  return pm_check_launch();
}

extern "C" int pm_adamw(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, long n, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                        void* stream) {
  if (!p || !g || !m || !v) return PM_EINVAL;
  if (n <= 0 || (n & 3) || step < 1) return PM_ESHAPE;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float rsqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const dim3 grid(cap_grid(n >> 2, 256, 4096));
  if (!shadow) shadow_dtype = PM_BF16;
  PM_DISPATCH_ACT(shadow_dtype, T, hipLaunchKernelGGL(adamw_kernel<T>, grid, dim3(256), 0, pm_stream(stream), p, g, m, v, (T*)shadow, n,
                                                      lr, beta1, beta2, eps, weight_decay, (float)bc1, rsqrt_bc2, grad_scale));
  return pm_check_launch();
}


extern "C" int pm_adamw_tick(float* hyper, int n_groups, void* stream) {
  if (!hyper) return PM_EINVAL;
  if (n_groups <= 0) return PM_ESHAPE;
  hipLaunchKernelGGL(adamw_tick_kernel, dim3((n_groups + 63) / 64), dim3(64), 0, pm_stream(stream), hyper, n_groups);
  return pm_check_launch();
}

extern "C" int pm_adamw_dev(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, long n,
                            const float* hyper, void* stream) {
  if (!p || !g || !m || !v || !hyper) return PM_EINVAL;
  if (n <= 0 || (n & 3)) return PM_ESHAPE;
  const dim3 grid(cap_grid(n >> 2, 256, 4096));
  if (!shadow) shadow_dtype = PM_BF16;
  PM_DISPATCH_ACT(shadow_dtype, T, hipLaunchKernelGGL(adamw_dev_kernel<T>, grid, dim3(256), 0, pm_stream(stream), p, g, m, v, (T*)shadow,
                                                      n, hyper));
  return pm_check_launch();
}

extern "C" int pm_loss_scale_update(float* state, const float* stats, float* hyper, int n_groups, float growth_factor,
                                    float backoff_factor, int growth_interval, void* stream) {
  if (!state || !stats || !hyper) return PM_EINVAL;
  if (n_groups <= 0 || growth_interval <= 0 || !(growth_factor >= 1.f) || !(backoff_factor > 0.f && backoff_factor <= 1.f)) return PM_ESHAPE;
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, pm_stream(stream), state, stats, hyper, n_groups,
                     growth_factor, backoff_factor, growth_interval);
  return pm_check_launch();
}

extern "C" int pm_grad_stats(const float* g, long n, float* out, void* stream) {
  if (!g || !out) return PM_EINVAL;
  if (n <= 0) return PM_ESHAPE;
  if ((uintptr_t)g & 15) return PM_EALIGN;
  hipLaunchKernelGGL(grad_stats_kernel, dim3(cap_grid(n >> 2, 256, 1024)), dim3(256), 0, pm_stream(stream), g, n, out);
  return pm_check_launch();
}

extern "C" const char* pm_strerror(int status) {
  switch (status) {
    case PM_OK: return "ok";
    case PM_EINVAL: return "invalid argument (null pointer / unknown dtype or epilogue)";
    case PM_ESHAPE: return "unsupported shape";
    case PM_EARCH: return "no gfx950 device";
    case PM_ELAUNCH: return "kernel launch failed";
    case PM_EALIGN: return "pointer / leading dimension not 16-byte aligned";
    default: return "unknown status";
  }
}

// 3: pm_gemm_ex / pm_gemm_opts replace pm_tune, workspace queries, pm_vit_head_*, pm_supervised_loss_fwd, pm_scale
// 10: PM_F16 (precision mode fp16), pm_loss_scale_update, pm_dgelu
extern "C" int pm_abi_version(void) { return 14; }

extern "C" size_t pm_workspace_bytes(int kind, int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  switch (kind) {
    case PM_WS_LAYERNORM_BWD: {  // one partial row triple per block, grid = min(ceil(M/4), 1024)
      long g = (M + 3) / 4;
      if (g > 1024) g = 1024;
      if (g < 64) g = 64;
      return (size_t)g * 3 * N * sizeof(float);
    }
    case PM_WS_COLSUM: return (size_t)cap_grid(M, 64, 128) * N * sizeof(float);
    case PM_WS_GEMM_COLSUM: {  // per (row tile of 192, wave row) partial rows of the fused epilogue, or the plain column sum
      const size_t fused = (size_t)((M + 191) / 192) * 2 * N * sizeof(float);
      const size_t plain = (size_t)cap_grid(M, 64, 128) * N * sizeof(float);
      return fused > plain ? fused : plain;
    }
    case PM_WS_UNSHUFFLE_BWD: return (size_t)1024 * N * sizeof(float);  // partial rows of the mask-token gradient
    default: return 0;
  }
}
