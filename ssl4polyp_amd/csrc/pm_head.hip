// pm_head.hip -- what sits on top of the last encoder block on the fine-tune path: final LayerNorm + token selection
// (cls row, or the mean over the patch rows) + lin_head, and the supervised loss on the logits.
// Reference: ViT_from_MAE.forward / VisionTransformer_from_Any.forward (models.py:127,134-139,209,216-221)
//   x = norm(x); x = x[:, 0] if out_token == "cls" else x[:, 1:].mean(1); x = lin_head(x) if head
// and train_classification.py:3347-3374,6086-6104 (_compute_supervised_loss with BCEWithLogitsLoss(pos_weight) on
// logits[:,1]-logits[:,0] for the two-class packs, CrossEntropyLoss(weight) otherwise).
// Latency-bound: O(B*D) work.  One block per sample; every reduction has a fixed order (reproducible step).
#include "pm_common.h"

namespace {

constexpr int kHeadVec = 4;  // f32x4 per lane -> D <= 1024

// block-wide sum of one float per thread (256 threads), fixed order; every thread gets the result
__device__ __forceinline__ float block_sum256(float v, float* s4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s4[wave] = v;
  __syncthreads();
  return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// ---------------------------------------------------------------------------------------------
// pool = cls: LayerNorm of row 0 + Linear(D -> n_class)  (W == NULL: features only, head=False)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cls_head_fwd_kernel(const float* __restrict__ x, long sample_stride,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ W, const float* __restrict__ bias,
                                                          float* __restrict__ xn, float* __restrict__ mean_out,
                                                          float* __restrict__ rstd_out, float* __restrict__ logits, int D,
                                                          int n_class, float eps) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* xr = x + (long)b * sample_stride;
  const int nvec = D >> 2;
  f32x4 v[kHeadVec];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kHeadVec; ++i) {
    const int c = lane + 64 * i;
    if (c < nvec) {
      v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * c);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kHeadVec; ++i) {
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
    mean_out[b] = mean;
    rstd_out[b] = rstd;
  }
#pragma unroll
  for (int i = 0; i < kHeadVec; ++i) {
    const int c = lane + 64 * i;
    if (c < nvec) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * c);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + 4 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * g[e] + bb[e];
      *reinterpret_cast<f32x4*>(xn + (long)b * D + 4 * c) = v[i];
    }
  }
  if (!W) return;
  for (int k = 0; k < n_class; ++k) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < kHeadVec; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(W + (long)k * D + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += v[i][e] * w[e];
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) logits[(long)b * n_class + k] = acc + (bias ? bias[k] : 0.f);
  }
}

// d loss / d feat[b][d]: given directly (head=False) or dlogits . W
__device__ __forceinline__ float feat_grad(const float* __restrict__ dfeat, const float* __restrict__ dlogits,
                                           const float* __restrict__ W, int b, int d, int D, int n_class) {
  if (dfeat) return dfeat[(long)b * D + d];
  float g = 0.f;
  for (int k = 0; k < n_class; ++k) g += dlogits[(long)b * n_class + k] * W[(long)k * D + d];
  return g;
}

template <typename T>
__global__ __launch_bounds__(256) void cls_head_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ dfeat,
                                                           const float* __restrict__ x, int N,
                                                           const float* __restrict__ gamma, const float* __restrict__ W,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* __restrict__ dx, T* __restrict__ dx_act, int D,
                                                           int n_class) {
  __shared__ float s_red[2][4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long sample = (long)N * D;
  const float* xr = x + (long)b * sample;
  const float mu = mean[b], rs = rstd[b];
  // each thread owns columns tid, tid+256, ...
  float s1 = 0.f, s2 = 0.f;
  for (int d = tid; d < D; d += 256) {
    const float g = feat_grad(dfeat, dlogits, W, b, d, D, n_class);
    const float xh = (xr[d] - mu) * rs;
    const float gg = g * gamma[d];
    s1 += gg;
    s2 += gg * xh;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    s_red[0][wave] = s1;
    s_red[1][wave] = s2;
  }
  __syncthreads();
  const float c1 = ((s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3])) / (float)D;
  const float c2 = ((s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3])) / (float)D;
  for (int d = tid; d < D; d += 256) {
    const float g = feat_grad(dfeat, dlogits, W, b, d, D, n_class);
    const float xh = (xr[d] - mu) * rs;
    const float o = rs * (g * gamma[d] - c1 - xh * c2);
    dx[(long)b * sample + d] = o;
    if (dx_act) dx_act[(long)b * sample + d] = (T)o;
  }
  // zero the gradient of every non-cls token of this sample
  const long rest = sample - D;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (long i = tid; i < (rest >> 2); i += 256) {
    *reinterpret_cast<f32x4*>(dx + (long)b * sample + D + 4 * i) = z;
    if (dx_act) store4<T>(dx_act + (long)b * sample + D + 4 * i, z);
  }
}

// ---------------------------------------------------------------------------------------------
// pool = spatial: LayerNorm of rows 1..N-1, mean over them, Linear  (models.py:136-137, 218-219)
// mean_n(xhat_n * gamma + beta) = (mean_n xhat_n) * gamma + beta: the affine is applied once to the pooled xhat.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spatial_head_fwd_kernel(const float* __restrict__ x, int N,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ W, const float* __restrict__ bias,
                                                               float* __restrict__ feat, float* __restrict__ xhat_mean,
                                                               float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                               float* __restrict__ logits, int D, int n_class, float eps) {
  __shared__ __attribute__((aligned(16))) float red[4][1024];
  __shared__ float s4[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nvec = D >> 2;
  f32x4 acc[kHeadVec];
#pragma unroll
  for (int i = 0; i < kHeadVec; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int n = 1 + wave; n < N; n += 4) {  // rows in increasing order per wave: a fixed summation order
    const float* xr = x + ((long)b * N + n) * D;
    f32x4 v[kHeadVec];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kHeadVec; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * c);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kHeadVec; ++i) {
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
      mean_out[(long)b * N + n] = mean;
      rstd_out[(long)b * N + n] = rstd;
    }
#pragma unroll
    for (int i = 0; i < kHeadVec; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] += (v[i][e] - mean) * rstd;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kHeadVec; ++i) {
    const int c = lane + 64 * i;
    if (c < nvec) *reinterpret_cast<f32x4*>(&red[wave][4 * c]) = acc[i];
  }
  __syncthreads();
  const float invL = 1.0f / (float)(N - 1);
  float part[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) part[k] = 0.f;
  for (int d = tid; d < D; d += 256) {
    const float xm = ((red[0][d] + red[1][d]) + (red[2][d] + red[3][d])) * invL;
    const float f = xm * gamma[d] + beta[d];
    xhat_mean[(long)b * D + d] = xm;
    feat[(long)b * D + d] = f;
    if (W) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < n_class) part[k] += f * W[(long)k * D + d];
    }
  }
  if (!W) return;
  for (int k0 = 0; k0 < n_class; k0 += 8) {
    if (k0 > 0) {  // more than 8 classes: further passes over the stored features
#pragma unroll
      for (int k = 0; k < 8; ++k) part[k] = 0.f;
      for (int d = tid; d < D; d += 256) {
        const float f = feat[(long)b * D + d];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k0 + k < n_class) part[k] += f * W[(long)(k0 + k) * D + d];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k0 + k >= n_class) break;
      const float t = block_sum256(part[k], s4);
      if (tid == 0) logits[(long)b * n_class + k0 + k] = t + (bias ? bias[k0 + k] : 0.f);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void spatial_head_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ dfeat,
                                                               const float* __restrict__ x, int N,
                                                               const float* __restrict__ gamma, const float* __restrict__ W,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               float* __restrict__ dx, T* __restrict__ dx_act, int D,
                                                               int n_class) {
  __shared__ __attribute__((aligned(16))) float sgg[1024];  // d loss / d y_n[d] * gamma[d], identical for every patch row
  __shared__ float s4[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nvec = D >> 2;
  const float invL = 1.0f / (float)(N - 1);
  float s1 = 0.f;
  for (int d = tid; d < D; d += 256) {
    const float gg = feat_grad(dfeat, dlogits, W, b, d, D, n_class) * invL * gamma[d];
    sgg[d] = gg;
    s1 += gg;
  }
  const float c1 = block_sum256(s1, s4) / (float)D;  // (also publishes sgg)
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  if (wave == 0) {  // the cls row does not reach the pooled feature
#pragma unroll
    for (int i = 0; i < kHeadVec; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        *reinterpret_cast<f32x4*>(dx + (long)b * N * D + 4 * c) = z;
        if (dx_act) store4<T>(dx_act + (long)b * N * D + 4 * c, z);
      }
    }
  }
  for (int n = 1 + wave; n < N; n += 4) {
    const long row = (long)b * N + n;
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[kHeadVec], gg[kHeadVec];
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < kHeadVec; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        xh[i] = *reinterpret_cast<const f32x4*>(x + row * D + 4 * c);
        gg[i] = *reinterpret_cast<const f32x4*>(&sgg[4 * c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[i][e] = (xh[i][e] - mu) * rs;
          s2 += gg[i][e] * xh[i][e];
        }
      }
    }
    const float c2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < kHeadVec; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (gg[i][e] - c1 - xh[i][e] * c2);
        *reinterpret_cast<f32x4*>(dx + row * D + 4 * c) = o;
        if (dx_act) store4<T>(dx_act + row * D + 4 * c, o);
      }
    }
  }
}

// Parameter gradients of the head (lin_head weight / bias, final-norm gamma / beta): one lane per column, a fixed summation
// order instead of one float atomic per (sample, column), so the whole training step is reproducible bit for bit.
// xhat_mean != NULL (spatial pooling): the pooled xhat saved by the forward; else xhat of the cls row from x / mean / rstd.
__global__ __launch_bounds__(256) void head_pgrad_kernel(const float* __restrict__ dlogits, const float* __restrict__ dfeat,
                                                         const float* __restrict__ x, long sample_stride,
                                                         const float* __restrict__ W, const float* __restrict__ feat,
                                                         const float* __restrict__ xhat_mean, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ dW,
                                                         float* __restrict__ dbias, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, int B, int D, int n_class) {
  // 64 columns per block; the four waves take every fourth sample (the cls rows are one sample = 600 KB apart: a thread
  // that walks all B serially pays B load latencies -- 55 us at B = 64 at the very start of the backward pass), two
  // independent accumulator pairs per wave, partials combined in wave order through LDS: a fixed summation order.
  __shared__ float part[2][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + lane;
  const bool live = d < D;
  if (dgamma || dbeta) {
    float g0 = 0.f, b0 = 0.f, g1 = 0.f, b1 = 0.f;
    if (live) {
      int b = wave;
      for (; b + 4 < B; b += 8) {
        const float ga = feat_grad(dfeat, dlogits, W, b, d, D, n_class), gb = feat_grad(dfeat, dlogits, W, b + 4, d, D, n_class);
        const float xa = xhat_mean ? xhat_mean[(long)b * D + d] : (x[(long)b * sample_stride + d] - mean[b]) * rstd[b];
        const float xb = xhat_mean ? xhat_mean[(long)(b + 4) * D + d]
                                   : (x[(long)(b + 4) * sample_stride + d] - mean[b + 4]) * rstd[b + 4];
        g0 += ga * xa; b0 += ga;
        g1 += gb * xb; b1 += gb;
      }
      for (; b < B; b += 4) {
        const float ga = feat_grad(dfeat, dlogits, W, b, d, D, n_class);
        const float xa = xhat_mean ? xhat_mean[(long)b * D + d] : (x[(long)b * sample_stride + d] - mean[b]) * rstd[b];
        g0 += ga * xa; b0 += ga;
      }
    }
    part[0][wave][lane] = g0 + g1;
    part[1][wave][lane] = b0 + b1;
    __syncthreads();
    if (wave == 0 && live) {
      if (dgamma) dgamma[d] += (part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane]);
      if (dbeta) dbeta[d] += (part[1][0][lane] + part[1][1][lane]) + (part[1][2][lane] + part[1][3][lane]);
    }
  }
  if (dW) {  // per class: the waves split the samples as above, partials combined in wave order
    for (int k = 0; k < n_class; ++k) {
      float s0 = 0.f, s1 = 0.f;
      if (live) {
        int b = wave;
        for (; b + 4 < B; b += 8) {
          s0 += dlogits[(long)b * n_class + k] * feat[(long)b * D + d];
          s1 += dlogits[(long)(b + 4) * n_class + k] * feat[(long)(b + 4) * D + d];
        }
        for (; b < B; b += 4) s0 += dlogits[(long)b * n_class + k] * feat[(long)b * D + d];
      }
      __syncthreads();  // (the previous round's partials have been read)
      part[0][wave][lane] = s0 + s1;
      __syncthreads();
      if (wave == 0 && live) dW[(long)k * D + d] += (part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane]);
    }
  }
  if (dbias && blockIdx.x == 0 && threadIdx.x < n_class) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dlogits[(long)b * n_class + threadIdx.x];
    dbias[threadIdx.x] += s;
  }
}

// ---------------------------------------------------------------------------------------------
// supervised loss on the logits + its gradient, one block, fixed summation order
// ---------------------------------------------------------------------------------------------
// n_class == 2 (the reference's "binary_bce" mode, tc.py:3347-3374,6090-6102): z = l1 - l0,
//   loss = mean_b[(1 - y) z + (1 + (pw - 1) y) (log1p(exp(-|z|)) + max(-z, 0))]     (torch BCEWithLogitsLoss(pos_weight))
// otherwise (tc.py:6104): loss = sum_b w[y_b] (lse(l_b) - l_b[y_b]) / sum_b w[y_b]   (torch CrossEntropyLoss(weight))
__global__ __launch_bounds__(256) void sup_loss_kernel(const float* __restrict__ logits, const long long* __restrict__ targets,
                                                       const float* __restrict__ pos_weight,
                                                       const float* __restrict__ class_weights, float* __restrict__ loss,
                                                       float* __restrict__ dlogits, int B, int n_class) {
  __shared__ float s4[4];
  const int tid = threadIdx.x;
  float num = 0.f, den = 0.f;
  if (n_class == 2) {
    const float pw = pos_weight ? pos_weight[0] : 1.0f;
    for (int b = tid; b < B; b += 256) {
      const float z = logits[2 * b + 1] - logits[2 * b];
      const float y = (float)targets[b];
      const float lw = 1.0f + (pw - 1.0f) * y;
      num += (1.0f - y) * z + lw * (log1pf(expf(-fabsf(z))) + fmaxf(-z, 0.f));
    }
    const float tot = block_sum256(num, s4);
    const float invB = 1.0f / (float)B;
    if (tid == 0) loss[0] = tot * invB;
    for (int b = tid; b < B; b += 256) {
      const float z = logits[2 * b + 1] - logits[2 * b];
      const float y = (float)targets[b];
      const float lw = 1.0f + (pw - 1.0f) * y;
      const float sig_neg = 1.0f / (1.0f + expf(z));  // sigmoid(-z)
      const float g = ((1.0f - y) - lw * sig_neg) * invB;
      dlogits[2 * b] = -g;
      dlogits[2 * b + 1] = g;
    }
    return;
  }
  // A target outside [0, n_class) (e.g. torch's ignore_index = -100, which this path does not implement) must not index
  // l[] / class_weights[]: the index is clamped for the reads and the launch reports it the only way a kernel can without
  // a host sync -- loss and every dlogit become NaN, which the training loops stop on (torch.nn.CrossEntropyLoss raises).
  float bad = 0.f;
  for (int b = tid; b < B; b += 256) {
    const float* l = logits + (long)b * n_class;
    const long long yt = targets[b];
    const bool oob = yt < 0 || yt >= n_class;
    bad += oob ? 1.f : 0.f;
    const int y = oob ? 0 : (int)yt;
    float m = l[0];
    for (int k = 1; k < n_class; ++k) m = fmaxf(m, l[k]);
    float se = 0.f;
    for (int k = 0; k < n_class; ++k) se += expf(l[k] - m);
    const float w = class_weights ? class_weights[y] : 1.0f;
    num += w * (m + logf(se) - l[y]);
    den += w;
  }
  const float tn = block_sum256(num, s4);
  const float td = block_sum256(den, s4);
  const float poison = block_sum256(bad, s4) > 0.f ? __builtin_nanf("") : 0.f;
  if (tid == 0) loss[0] = tn / td + poison;
  const float inv = 1.0f / td + poison;
  for (int b = tid; b < B; b += 256) {
    const float* l = logits + (long)b * n_class;
    const long long yt = targets[b];
    const int y = (yt < 0 || yt >= n_class) ? 0 : (int)yt;
    float m = l[0];
    for (int k = 1; k < n_class; ++k) m = fmaxf(m, l[k]);
    float se = 0.f;
    for (int k = 0; k < n_class; ++k) se += expf(l[k] - m);
    const float w = (class_weights ? class_weights[y] : 1.0f) * inv;
    const float rse = 1.0f / se;
    for (int k = 0; k < n_class; ++k) dlogits[(long)b * n_class + k] = w * (expf(l[k] - m) * rse - (k == y ? 1.0f : 0.f));
  }
}

__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ out,
                                                    long n) {
  const float f = s[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = x[i] * f;
}

}  // namespace

extern "C" int pm_vit_head_fwd(const float* x, int N, int pool, const float* gamma, const float* beta, const float* W,
                               const float* bias, float* feat, float* xhat_mean, float* mean, float* rstd, float* logits,
                               int B, int D, int n_class, float eps, void* stream) {
  if (!x || !gamma || !beta || !feat || !mean || !rstd) return PM_EINVAL;
  if (W && !logits) return PM_EINVAL;
  if (pool != 0 && pool != 1) return PM_EINVAL;
  if (B <= 0 || N <= 0 || D <= 0 || D > 1024 || (D & 3) || (W && (n_class <= 0 || n_class > 256))) return PM_ESHAPE;
  if (pool == 1 && (N < 2 || !xhat_mean)) return N < 2 ? PM_ESHAPE : PM_EINVAL;
  if (pool == 0)
    hipLaunchKernelGGL(cls_head_fwd_kernel, dim3(B), dim3(64), 0, pm_stream(stream), x, (long)N * D, gamma, beta, W, bias, feat,
                       mean, rstd, logits, D, n_class, eps);
  else
    hipLaunchKernelGGL(spatial_head_fwd_kernel, dim3(B), dim3(256), 0, pm_stream(stream), x, N, gamma, beta, W, bias, feat,
                       xhat_mean, mean, rstd, logits, D, n_class, eps);
  return pm_check_launch();
}

extern "C" int pm_vit_head_bwd(const float* dlogits, const float* dfeat, const float* x, int N, int pool, const float* gamma,
                               const float* W, const float* feat, const float* xhat_mean, const float* mean,
                               const float* rstd, float* dx, void* dx_act, int act_dtype, float* dW, float* dbias,
                               float* dgamma, float* dbeta, int B, int D, int n_class, void* stream) {
  if (!x || !gamma || !mean || !rstd) return PM_EINVAL;
  if (!dfeat && (!dlogits || !W)) return PM_EINVAL;      // one of the two gradient sources
  if ((dW || dbias) && (!dlogits || !feat)) return PM_EINVAL;
  if (!dx && dx_act) return PM_EINVAL;
  if (pool != 0 && pool != 1) return PM_EINVAL;
  if (pool == 1 && !xhat_mean) return PM_EINVAL;
  if (B <= 0 || N <= 0 || D <= 0 || D > 1024 || (D & 3) || (!dfeat && (n_class <= 0 || n_class > 256))) return PM_ESHAPE;
  if (pool == 1 && N < 2) return PM_ESHAPE;
  hipStream_t s = pm_stream(stream);
  if (dx) {  // NULL: frozen backbone, nothing below the head needs a gradient
#define PM_HEAD_BWD(KERN, T)                                                                                            \
  hipLaunchKernelGGL(KERN<T>, dim3(B), dim3(256), 0, s, dlogits, dfeat, x, N, gamma, W, mean, rstd, dx, (T*)dx_act, D, n_class)
    PM_DISPATCH_ACT(act_dtype, T, {
      if (pool == 0) PM_HEAD_BWD(cls_head_bwd_kernel, T); else PM_HEAD_BWD(spatial_head_bwd_kernel, T);
    });
#undef PM_HEAD_BWD
  }
  if (dW || dbias || dgamma || dbeta)
    hipLaunchKernelGGL(head_pgrad_kernel, dim3((D + 63) / 64), dim3(256), 0, s, dlogits, dfeat, x, (long)N * D, W, feat,
                       pool == 1 ? xhat_mean : nullptr, mean, rstd, dW, dbias, dgamma, dbeta, B, D, n_class);
  return pm_check_launch();
}

// the round-1 entry points: cls row + lin_head
extern "C" int pm_cls_head_fwd(const float* x, int N, const float* gamma, const float* beta, const float* W,
                               const float* bias, float* xn, float* mean, float* rstd, float* logits, int B, int D,
                               int n_class, float eps, void* stream) {
  if (!W) return PM_EINVAL;
  return pm_vit_head_fwd(x, N, 0, gamma, beta, W, bias, xn, nullptr, mean, rstd, logits, B, D, n_class, eps, stream);
}

extern "C" int pm_cls_head_bwd(const float* dlogits, const float* x, int N, const float* gamma, const float* W,
                               const float* xn, const float* mean, const float* rstd, float* dx, void* dx_act,
                               int act_dtype, float* dW, float* dbias, float* dgamma, float* dbeta, int B, int D,
                               int n_class, void* stream) {
  if (!dlogits || !W || !xn) return PM_EINVAL;
  return pm_vit_head_bwd(dlogits, nullptr, x, N, 0, gamma, W, xn, nullptr, mean, rstd, dx, dx_act, act_dtype, dW, dbias, dgamma,
                         dbeta, B, D, n_class, stream);
}

extern "C" int pm_supervised_loss_fwd(const float* logits, const long long* targets, const float* pos_weight,
                                      const float* class_weights, float* loss, float* dlogits, int B, int n_class,
                                      void* stream) {
  if (!logits || !targets || !loss || !dlogits) return PM_EINVAL;
  if (B <= 0 || n_class < 2) return PM_ESHAPE;
  hipLaunchKernelGGL(sup_loss_kernel, dim3(1), dim3(256), 0, pm_stream(stream), logits, targets, pos_weight, class_weights, loss,
                     dlogits, B, n_class);
  return pm_check_launch();
}

extern "C" int pm_scale(const float* x, const float* scale, float* out, long n, void* stream) {
  if (!x || !scale || !out) return PM_EINVAL;
  if (n <= 0) return PM_ESHAPE;
  long g = (n + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(scale_kernel, dim3((int)g), dim3(256), 0, pm_stream(stream), x, scale, out, n);
  return pm_check_launch();
}
