// pm_mae.hip -- MAE-specific HBM-bound kernels: random masking (argsort of noise), decoder un-shuffle
// with mask-token fill, and the fused patchify + masked-MSE pixel-reconstruction loss.
// Reference: src/ssl4polyp/models/mae/models_mae.py:123-148 (random_masking), :177-183 (forward_decoder
// pre-blocks), :95-107 + :198-214 (patchify / forward_loss).
#include "pm_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// masking: stable ascending argsort of noise[b, :] by rank counting (L <= 1024; L = 196 for ViT-B/16)
//   ids_restore[i] = rank(i);  ids_shuffle[rank(i)] = i;  mask[i] = rank(i) >= len_keep
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void masking_kernel(const float* __restrict__ noise, int* __restrict__ ids_shuffle,
                                                      int* __restrict__ ids_restore, float* __restrict__ mask, int L,
                                                      int len_keep) {
  __shared__ float s[1024];
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < L; i += 256) s[i] = noise[(long)b * L + i];
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += 256) {
    const float v = s[i];
    int rank = 0;
    for (int j = 0; j < L; ++j) {
      const float w = s[j];
      rank += (w < v) || (w == v && j < i);
    }
    ids_restore[(long)b * L + i] = rank;
    ids_shuffle[(long)b * L + rank] = i;
    mask[(long)b * L + i] = rank >= len_keep ? 1.0f : 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------
// masking noise: counter-based U[0, 1) (Philox4x32-10, Salmon et al. SC'11), one 128-bit block per 4 values
//   key = (seed_lo, seed_hi), counter = (element index / 4, stream, 0, 0); value = (x >> 8) * 2^-24: 24 random bits, like torch.rand
// The reference draws torch.rand(N, L, device=x.device) (models_mae.py:132): a device generator whose stream no other device
// or library version reproduces.  A counter-based generator makes the noise a pure function of (seed, stream, index): the same
// on every run, rank layout and launch geometry, and it needs no ATen launch.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__global__ __launch_bounds__(256) void noise_kernel(float* __restrict__ out, long n, unsigned long long seed, unsigned stream_id) {
  const long nq = (n + 3) >> 2;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
    unsigned c[4] = {(unsigned)q, (unsigned)(q >> 32), stream_id, 0u};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * q + e < n) out[4 * q + e] = (float)(c[e] >> 8) * 0x1p-24f;
  }
}

// ---------------------------------------------------------------------------------------------
// decoder input: out[b,0] = emb[b,0] + dpos[0];
//                out[b,1+i] = (r = ids_restore[b,i]) < keep ? emb[b,1+r] : mask_token;  + dpos[1+i]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unshuffle_kernel(const float* __restrict__ emb, const float* __restrict__ mask_token,
                                                        const float* __restrict__ dpos, const int* __restrict__ ids_restore,
                                                        float* __restrict__ out, int B, int L, int keep, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long rows = (long)B * (L + 1);
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const int b = row / (L + 1), t = row % (L + 1);
    const float* src;
    if (t == 0) {
      src = emb + (long)b * (keep + 1) * D;
    } else {
      const int r = ids_restore[(long)b * L + t - 1];
      src = r < keep ? emb + ((long)b * (keep + 1) + 1 + r) * D : mask_token;
    }
    for (int c = lane * 4; c < D; c += 256) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + c);
      const f32x4 p = *reinterpret_cast<const f32x4*>(dpos + (long)t * D + c);
      *reinterpret_cast<f32x4*>(out + row * D + c) = a + p;
    }
  }
}

// demb[b,0] = dout[b,0]; demb[b,1+j] = dout[b, 1 + ids_shuffle[b,j]]  (j < keep)
template <typename T>
__global__ __launch_bounds__(256) void unshuffle_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ ids_shuffle,
                                                            T* __restrict__ demb, int B, int L, int keep, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long rows = (long)B * (keep + 1);
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const int b = row / (keep + 1), t = row % (keep + 1);
    const int srow = t == 0 ? 0 : 1 + ids_shuffle[(long)b * L + t - 1];
    const float* src = dout + ((long)b * (L + 1) + srow) * D;
    for (int c = lane * 4; c < D; c += 256) store4<T>(demb + row * D + c, *reinterpret_cast<const f32x4*>(src + c));
  }
}

// dmask_token[d] += sum over masked positions (j >= keep) of dout[b, 1 + ids_shuffle[b,j], d]
// `partials` != NULL: block row blockIdx.y stores its sums there and mask_token_reduce_kernel adds the rows in a fixed
// order (deterministic); NULL: one float atomic per column per block.
__global__ __launch_bounds__(256) void mask_token_grad_kernel(const float* __restrict__ dout, const int* __restrict__ ids_shuffle,
                                                              float* __restrict__ dmask_token, float* __restrict__ partials,
                                                              int B, int L, int keep, int D) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int nm = L - keep;
  const long total = (long)B * nm;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < D) {
    for (long i = (long)blockIdx.y * 4 + wave; i < total; i += (long)gridDim.y * 4) {
      const int b = i / nm, j = keep + i % nm;
      const int srow = 1 + ids_shuffle[(long)b * L + j];
      acc += *reinterpret_cast<const f32x4*>(dout + ((long)b * (L + 1) + srow) * D + c);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wave][lane * 4 + e] = acc[e];
  __syncthreads();
  const int t = threadIdx.x, cc = blockIdx.x * 256 + t;
  if (cc < D) {
    const float v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    if (partials) partials[(long)blockIdx.y * D + cc] = v;
    else atomicAdd(dmask_token + cc, v);
  }
}

__global__ __launch_bounds__(1024) void mask_token_reduce_kernel(const float* __restrict__ partials, float* __restrict__ out,
                                                                 int rows, int D) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  red[rg][lane] = n < D ? strided_sum<8>(partials + n, D, rg, 16, rows) : 0.f;
  __syncthreads();
  if (rg == 0 && n < D) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][lane];
    out[n] += t;
  }
}

// ---------------------------------------------------------------------------------------------
// loss: one wave per patch.  Element e of a patch = (py*p + px)*C + c  (models_mae.py:105 'nhwpqc').
// ---------------------------------------------------------------------------------------------
constexpr int kLossVec = 4;  // patch elements <= 1024 (768 for 16x16x3)

struct PatchTarget {
  f32x4 t[kLossVec];
};

__device__ __forceinline__ PatchTarget load_target(const float* __restrict__ imgs, int b, int l, int C, int img, int p,
                                                   int norm_pix, int lane) {
  const int grid = img / p, gy = l / grid, gx = l % grid;
  const int PE = p * p * C, nvec = PE >> 2;
  const float* src = imgs + (long)b * C * img * img + (long)gy * p * img + gx * p;
  PatchTarget o;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kLossVec; ++i) {
    const int v = lane + 64 * i;
    if (v < nvec) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = 4 * v + e;
        const int c = idx % C, pix = idx / C, py = pix / p, px = pix % p;
        o.t[i][e] = src[(long)c * img * img + (long)py * img + px];
        s += o.t[i][e];
      }
    }
  }
  if (norm_pix) {
    const float mean = wave_sum(s) / (float)PE;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kLossVec; ++i) {
      const int v = lane + 64 * i;
      if (v < nvec) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = o.t[i][e] - mean;
          q += d * d;
        }
      }
    }
    const float var = wave_sum(q) / (float)(PE - 1);  // torch.var default: unbiased (models_mae.py:208)
    const float inv = 1.0f / sqrtf(var + 1.0e-6f);
#pragma unroll
    for (int i = 0; i < kLossVec; ++i) {
      const int v = lane + 64 * i;
      if (v < nvec) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o.t[i][e] = (o.t[i][e] - mean) * inv;
      }
    }
  }
  return o;
}

__global__ __launch_bounds__(256) void loss_fwd_kernel(const float* __restrict__ imgs, const float* __restrict__ pred, long ldp,
                                                       int has_cls, float* __restrict__ patch_loss, int B, int L, int C,
                                                       int img, int p, int norm_pix) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int PE = p * p * C, nvec = PE >> 2;
  const long total = (long)B * L;
  for (long pi = (long)blockIdx.x * 4 + wave; pi < total; pi += (long)gridDim.x * 4) {
    const int b = pi / L, l = pi % L;
    const PatchTarget tg = load_target(imgs, b, l, C, img, p, norm_pix, lane);
    const float* pr = pred + ((long)b * (L + has_cls) + has_cls + l) * ldp;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kLossVec; ++i) {
      const int v = lane + 64 * i;
      if (v < nvec) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(pr + 4 * v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = pv[e] - tg.t[i][e];
          s += d * d;
        }
      }
    }
    s = wave_sum(s);
    if (lane == 0) patch_loss[pi] = s / (float)PE;
  }
}

// deterministic single-block reduction: sums = {sum(loss*mask), sum(mask)}, loss = ratio
__global__ __launch_bounds__(1024) void loss_finish_kernel(const float* __restrict__ patch_loss, const float* __restrict__ mask,
                                                           long n, float* __restrict__ sums, float* __restrict__ loss) {
  __shared__ float r0[16], r1[16];
  float a = 0.f, m = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) {
    const float mk = mask[i];
    a += patch_loss[i] * mk;
    m += mk;
  }
  a = wave_sum(a);
  m = wave_sum(m);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    r0[wave] = a;
    r1[wave] = m;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float sa = 0.f, sm = 0.f;
    for (int w = 0; w < 16; ++w) {
      sa += r0[w];
      sm += r1[w];
    }
    sums[0] = sa;
    sums[1] = sm;
    loss[0] = sa / sm;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ imgs, const float* __restrict__ pred, long ldp,
                                                       int has_cls, const float* __restrict__ mask,
                                                       const float* __restrict__ sums, const float* __restrict__ dloss,
                                                       T* __restrict__ dpred, long lddp, int B, int L, int C, int img, int p,
                                                       int norm_pix) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int PE = p * p * C, nvec = PE >> 2, nvec_ld = (int)(lddp >> 2);
  const long rows = (long)B * (L + has_cls);
  const float g0 = dloss[0] / sums[1] * 2.0f / (float)PE;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const int b = row / (L + has_cls), t = row % (L + has_cls);
    T* dr = dpred + row * lddp;
    const int l = t - has_cls;
    const float mk = l >= 0 ? mask[(long)b * L + l] : 0.f;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int v = nvec + lane; v < nvec_ld; v += 64) store4<T>(dr + 4 * v, z);  // row padding (lddp > PE: the GEMMs' padded K)
    if (mk == 0.f) {  // kept patch or cls row: no gradient
      for (int v = lane; v < nvec; v += 64) store4<T>(dr + 4 * v, z);
      continue;
    }
    const PatchTarget tg = load_target(imgs, b, l, C, img, p, norm_pix, lane);
    const float* pr = pred + row * ldp;
    const float g = g0 * mk;
#pragma unroll
    for (int i = 0; i < kLossVec; ++i) {
      const int v = lane + 64 * i;
      if (v < nvec) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(pr + 4 * v);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = g * (pv[e] - tg.t[i][e]);
        store4<T>(dr + 4 * v, o);
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

}  // namespace

extern "C" int pm_mae_noise(float* noise, long n, unsigned long long seed, unsigned int stream_id, void* stream) {
  if (!noise) return PM_EINVAL;
  if (n <= 0) return PM_ESHAPE;
  long g = ((n + 3) / 4 + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(noise_kernel, dim3((int)g), dim3(256), 0, pm_stream(stream), noise, n, seed, stream_id);
  return pm_check_launch();
}

extern "C" int pm_mae_masking(const float* noise, int* ids_shuffle, int* ids_restore, float* mask, int B, int L,
                              int len_keep, void* stream) {
  if (!noise || !ids_shuffle || !ids_restore || !mask) return PM_EINVAL;
  if (B <= 0 || L <= 0 || L > 1024 || len_keep < 0 || len_keep > L) return PM_ESHAPE;
  hipLaunchKernelGGL(masking_kernel, dim3(B), dim3(256), 0, pm_stream(stream), noise, ids_shuffle, ids_restore, mask, L, len_keep);
  return pm_check_launch();
}

extern "C" int pm_mae_unshuffle(const float* emb, const float* mask_token, const float* dpos, const int* ids_restore,
                                float* out, int B, int L, int keep, int D, void* stream) {
  if (!emb || !mask_token || !dpos || !ids_restore || !out) return PM_EINVAL;
  if (B <= 0 || L <= 0 || keep <= 0 || keep > L || D <= 0 || (D & 3)) return PM_ESHAPE;
  hipLaunchKernelGGL(unshuffle_kernel, dim3(cap_grid((long)B * (L + 1), 4, 4096)), dim3(256), 0, pm_stream(stream), emb,
                     mask_token, dpos, ids_restore, out, B, L, keep, D);
  return pm_check_launch();
}

extern "C" int pm_mae_unshuffle_bwd(const float* dout, const int* ids_shuffle, void* demb, int act_dtype,
                                    float* dmask_token, int B, int L, int keep, int D, void* workspace, size_t ws_bytes,
                                    void* stream) {
  if (!dout || !ids_shuffle || !demb) return PM_EINVAL;
  if (B <= 0 || L <= 0 || keep <= 0 || keep > L || D <= 0 || (D & 3)) return PM_ESHAPE;
  const dim3 grid(cap_grid((long)B * (keep + 1), 4, 4096));
  PM_DISPATCH_ACT(act_dtype, T, hipLaunchKernelGGL(unshuffle_bwd_kernel<T>, grid, dim3(256), 0, pm_stream(stream), dout, ids_shuffle,
                                                   (T*)demb, B, L, keep, D));
  if (dmask_token && keep < L) {
    // one block row per 64 masked positions, up to 1024 rows: each wave then walks ~16 (index, row) load pairs instead of 73
    // (70 -> ~20 us at B = 256, in the serial stretch between the decoder's and the encoder's backward)
    const int rows = cap_grid((long)B * (L - keep), 64, 1024);
    float* partials = (workspace && ws_bytes >= (size_t)rows * D * sizeof(float)) ? reinterpret_cast<float*>(workspace) : nullptr;
    hipLaunchKernelGGL(mask_token_grad_kernel, dim3((D + 255) / 256, rows), dim3(256), 0, pm_stream(stream), dout, ids_shuffle,
                       dmask_token, partials, B, L, keep, D);
    if (partials)
      hipLaunchKernelGGL(mask_token_reduce_kernel, dim3((D + 63) / 64), dim3(1024), 0, pm_stream(stream), partials, dmask_token,
                         rows, D);
  }
  return pm_check_launch();
}

extern "C" int pm_mae_loss_fwd(const float* imgs, const float* pred, long ldp, int has_cls_row, float* patch_loss, int B,
                               int C, int img, int p, int norm_pix, void* stream) {
  if (!imgs || !pred || !patch_loss) return PM_EINVAL;
  if (B <= 0 || C <= 0 || img <= 0 || p <= 0 || (img % p) || ((p * p * C) & 3) || p * p * C > 1024 || (ldp & 3)) return PM_ESHAPE;
  const int L = (img / p) * (img / p);
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(cap_grid((long)B * L, 4, 8192)), dim3(256), 0, pm_stream(stream), imgs, pred, ldp,
                     has_cls_row ? 1 : 0, patch_loss, B, L, C, img, p, norm_pix);
  return pm_check_launch();
}

extern "C" int pm_mae_loss_finish(const float* patch_loss, const float* mask, long n, float* sums, float* loss,
                                  void* stream) {
  if (!patch_loss || !mask || !sums || !loss) return PM_EINVAL;
  if (n <= 0) return PM_ESHAPE;
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1024), 0, pm_stream(stream), patch_loss, mask, n, sums, loss);
  return pm_check_launch();
}

extern "C" int pm_mae_loss_bwd(const float* imgs, const float* pred, long ldp, int has_cls_row, const float* mask,
                               const float* sums, const float* dloss, void* dpred, long lddp, int act_dtype, int B, int C,
                               int img, int p, int norm_pix, void* stream) {
  if (!imgs || !pred || !mask || !sums || !dloss || !dpred) return PM_EINVAL;
  if (B <= 0 || C <= 0 || img <= 0 || p <= 0 || (img % p) || ((p * p * C) & 3) || p * p * C > 1024 || (ldp & 3)) return PM_ESHAPE;
  if (lddp < (long)p * p * C || (lddp & 3)) return PM_ESHAPE;
  const int L = (img / p) * (img / p);
  const int hc = has_cls_row ? 1 : 0;
  const dim3 grid(cap_grid((long)B * (L + hc), 4, 8192));
  PM_DISPATCH_ACT(act_dtype, T, hipLaunchKernelGGL(loss_bwd_kernel<T>, grid, dim3(256), 0, pm_stream(stream), imgs, pred, ldp, hc,
                                                   mask, sums, dloss, (T*)dpred, lddp, B, L, C, img, p, norm_pix));
  return pm_check_launch();
}
