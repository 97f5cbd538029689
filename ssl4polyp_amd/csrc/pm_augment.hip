// pm_augment.hip -- the train-time augmentation of the reference's input pipeline on the device, on uint8 HWC frames
// (classification/data/transforms.py:234-246: Resize, ColorJitter, GaussianBlur((25,25)), flips, RandomRotation(180), then
// ToTensor + Normalize; the reference runs them per image in PIL / torchvision 0.10 on DataLoader workers).
// Each kernel restates the arithmetic of the library routine the reference ends up in -- Pillow's Resample.c (8bpc), Blend.c,
// Convert.c (rgb2l, rgb2hsv, hsv2rgb), Geometry.c (affine_fixed) and torchvision's tensor gaussian_blur -- integer for integer
// and float for float, so that the results equal oracle/augment_ref.py bit for bit (and through it Pillow 12.2, which pins the
// oracle in the build container; the blur is the one unpinned stage).  All kernels are HBM-bound byte movers: one pass over a
// 9.6 MB uint8 batch each (the blur two passes with an f32 intermediate), no LDS, coalesced along the 672-byte pixel rows.
// Random parameters are drawn on the host (data.py) and arrive as small per-sample arrays.
#include "pm_common.h"

// The library routines restated here are plain C compiled WITHOUT fused multiply-add (x86-64 baseline), and numpy does not fuse
// either: every `a * b + c` below must round twice.  hipcc contracts by default (-ffp-contract=fast, which ignores the pragma
// below; HIP's __fmul_rn / __fadd_rn are plain operators that contract like any other), so this translation unit is compiled
// with -ffp-contract=off (__graft_entry__.SOURCE_FLAGS) -- the pragma documents the requirement for other build recipes.
#pragma clang fp contract(off)

namespace {

inline int aug_grid(long work, int cap = 8192) {
  long g = (work + 255) / 256;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ---------------------------------------------------------------------------------------------
// Resize: Pillow ImagingResampleHorizontal_8bpc / Vertical_8bpc (fixed-point taps built on the host as precompute_coeffs does)
// ---------------------------------------------------------------------------------------------
constexpr int kPrecisionBits = 32 - 8 - 2;

// out[b][y][xx][c] = clip8((1 << 21) + sum_t src[b][y][xmin + t][c] * k[xx][t]) >> 22)      (axis = 1: along rows instead)
__global__ __launch_bounds__(256) void resample_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                       const int* __restrict__ bounds, const int* __restrict__ taps, int ksize,
                                                       int B, int Hs, int Ws, int Ho, int Wo, int vertical) {
  const long total = (long)B * Ho * Wo;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int xo = id % Wo;
    const int yo = (id / Wo) % Ho;
    const int b = id / ((long)Wo * Ho);
    const int o = vertical ? yo : xo;
    const int lo = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = taps + (long)o * ksize;
    int ss0 = 1 << (kPrecisionBits - 1), ss1 = ss0, ss2 = ss0;
    const unsigned char* p = vertical ? src + (((long)b * Hs + lo) * Ws + xo) * 3 : src + (((long)b * Hs + yo) * Ws + lo) * 3;
    const long step = vertical ? (long)Ws * 3 : 3;
    for (int t = 0; t < n; ++t) {
      const int kk = k[t];
      ss0 += p[0] * kk;
      ss1 += p[1] * kk;
      ss2 += p[2] * kk;
      p += step;
    }
    unsigned char* q = dst + (((long)b * Ho + yo) * Wo + xo) * 3;
    ss0 >>= kPrecisionBits; ss1 >>= kPrecisionBits; ss2 >>= kPrecisionBits;
    q[0] = (unsigned char)(ss0 < 0 ? 0 : (ss0 > 255 ? 255 : ss0));
    q[1] = (unsigned char)(ss1 < 0 ? 0 : (ss1 > 255 ? 255 : ss1));
    q[2] = (unsigned char)(ss2 < 0 ? 0 : (ss2 > 255 ? 255 : ss2));
  }
}

// Per-sample taps on the device: Resample.c precompute_coeffs + normalize_coeffs_8bpc in the C source's double arithmetic, one
// thread per (sample, output index).  RandomResizedCrop gives every sample its own crop size, i.e. its own scale and taps; the
// two tables of a batch (B x 224 x <= 17 taps) are built here instead of 2 x B Python loops on the host.
__device__ __forceinline__ double resample_filter(double x, int bicubic) {
  if (x < 0.0) x = -x;
  if (!bicubic) return x < 1.0 ? 1.0 - x : 0.0;
  const double a = -0.5;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// box[b] = (top, left, h, w); axis 0: in_size = w (horizontal taps), axis 1: in_size = h
__global__ __launch_bounds__(256) void resample_coeffs_kernel(const int* __restrict__ box, int axis, int out_size, int bicubic,
                                                              int ksize_max, int* __restrict__ bounds, int* __restrict__ taps,
                                                              int B) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= B * out_size) return;
  const int b = id / out_size, xx = id % out_size;
  const int in_size = box[4 * b + (axis ? 2 : 3)];
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = (bicubic ? 2.0 : 1.0) * filterscale;
  const double center = (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > ksize_max) xmax = ksize_max;  // (cannot happen when the caller sized ksize_max from the largest crop)
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += resample_filter((x + xmin - center + 0.5) * ss, bicubic);
  int* k = taps + (long)id * ksize_max;
  for (int x = 0; x < ksize_max; ++x) {
    int v = 0;
    if (x < xmax) {
      double w = resample_filter((x + xmin - center + 0.5) * ss, bicubic);
      if (ww != 0.0) w /= ww;
      v = w < 0 ? (int)(-0.5 + w * (double)(1 << kPrecisionBits)) : (int)(0.5 + w * (double)(1 << kPrecisionBits));
    }
    k[x] = v;
  }
  bounds[2 * id] = xmin;
  bounds[2 * id + 1] = xmax;
}

// one pass of the cropped resize with per-sample tables: horizontal (rows top .. top + h of src -> tmp rows 0 .. h) or vertical
// (tmp rows -> dst)
__global__ __launch_bounds__(256) void resample_crop_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                            const int* __restrict__ box, const int* __restrict__ bounds,
                                                            const int* __restrict__ taps, int ksize, int B, int Hs, int Ws,
                                                            int Hd, int Wd, int out_size, int vertical) {
  // horizontal: src [B][Hs][Ws], dst = tmp [B][Hd = Hs][Wd = out]; vertical: src = tmp [B][Hs][Ws = out], dst [B][Hd = out][Wd = out]
  const long total = (long)B * Hd * Wd;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int xo = id % Wd;
    const int yo = (id / Wd) % Hd;
    const int b = id / ((long)Wd * Hd);
    const int top = box[4 * b], left = box[4 * b + 1], h = box[4 * b + 2];
    if (!vertical && yo >= h) continue;  // rows below the crop are never read
    const int o = vertical ? yo : xo;
    const long ti = (long)b * out_size + o;
    const int lo = bounds[2 * ti], n = bounds[2 * ti + 1];
    const int* k = taps + ti * ksize;
    int ss0 = 1 << (kPrecisionBits - 1), ss1 = ss0, ss2 = ss0;
    const unsigned char* p = vertical ? src + (((long)b * Hs + lo) * Ws + xo) * 3
                                      : src + (((long)b * Hs + top + yo) * Ws + left + lo) * 3;
    const long step = vertical ? (long)Ws * 3 : 3;
    for (int t = 0; t < n; ++t) {
      const int kk = k[t];
      ss0 += p[0] * kk;
      ss1 += p[1] * kk;
      ss2 += p[2] * kk;
      p += step;
    }
    unsigned char* q = dst + (((long)b * Hd + yo) * Wd + xo) * 3;
    ss0 >>= kPrecisionBits; ss1 >>= kPrecisionBits; ss2 >>= kPrecisionBits;
    q[0] = (unsigned char)(ss0 < 0 ? 0 : (ss0 > 255 ? 255 : ss0));
    q[1] = (unsigned char)(ss1 < 0 ? 0 : (ss1 > 255 ? 255 : ss1));
    q[2] = (unsigned char)(ss2 < 0 ? 0 : (ss2 > 255 ? 255 : ss2));
  }
}

// ---------------------------------------------------------------------------------------------
// ColorJitter: four per-pixel ops in a per-sample order; contrast needs the image's mean luminance first
// ---------------------------------------------------------------------------------------------
struct Px { int r, g, b; };

__device__ __forceinline__ int rgb2l(const Px& p) { return (p.r * 19595 + p.g * 38470 + p.b * 7471 + 0x8000) >> 16; }  // Convert.c

// Blend.c ImagingBlend(in1 = degenerate, in2 = image, alpha): float arithmetic, separate multiply and add (no FMA: C float)
__device__ __forceinline__ int blend1(int deg, int v, float a, bool interp) {
  const float t = __fadd_rn((float)deg, __fmul_rn(a, __fsub_rn((float)v, (float)deg)));
  if (interp) return (int)t;  // 0 <= alpha <= 1: (UINT8) cast, truncation
  return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (int)t);
}
__device__ __forceinline__ Px blend3(int dr, int dg, int db, const Px& p, float a) {
  const bool interp = a >= 0.0f && a <= 1.0f;
  return Px{blend1(dr, p.r, a, interp), blend1(dg, p.g, a, interp), blend1(db, p.b, a, interp)};
}

// Convert.c rgb2hsv_row / hsv2rgb_row with the C source's float / double promotions (see oracle/augment_ref.py)
__device__ __forceinline__ Px hue_shift(const Px& p, int dh) {
  int maxc = max(p.r, max(p.g, p.b)), minc = min(p.r, min(p.g, p.b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = __fdiv_rn(cr, (float)maxc);
    const float rc = __fdiv_rn((float)(maxc - p.r), cr), gc = __fdiv_rn((float)(maxc - p.g), cr), bc = __fdiv_rn((float)(maxc - p.b), cr);
    float h;
    if (p.r == maxc) h = __fsub_rn(bc, gc);
    else if (p.g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    const int ih = (int)((double)h * 255.0), is = (int)((double)s * 255.0);
    uh = ih < 0 ? 0 : (ih > 255 ? 255 : ih);
    us = is < 0 ? 0 : (is > 255 ? 255 : is);
  }
  uh = (uh + dh) & 255;  // functional_pil.adjust_hue: np_h += np.uint8(hue_factor * 255), uint8 wrap-around
  if (us == 0) return Px{uv, uv, uv};
  const float fh = __fdiv_rn(__fmul_rn((float)uh, 6.0f), 255.0f);
  const int i = (int)floorf(fh);
  const float f = __fsub_rn(fh, (float)i);
  const float fs = __fdiv_rn((float)us, 255.0f), fv = (float)uv;
  auto rnd = [](float t) { const double r = floor((double)t + 0.5); return r < 0.0 ? 0 : (r > 255.0 ? 255 : (int)r); };
  const int pp = rnd(__fmul_rn(fv, __fsub_rn(1.0f, fs)));
  const int qq = rnd(__fmul_rn(fv, __fsub_rn(1.0f, __fmul_rn(fs, f))));
  const int tt = rnd(__fmul_rn(fv, __fsub_rn(1.0f, __fmul_rn(fs, __fsub_rn(1.0f, f)))));
  switch (i % 6) {
    case 0: return Px{uv, tt, pp};
    case 1: return Px{qq, uv, pp};
    case 2: return Px{pp, uv, tt};
    case 3: return Px{pp, qq, uv};
    case 4: return Px{tt, pp, uv};
    default: return Px{uv, pp, qq};
  }
}

// per-sample jitter record: order[4] (0 brightness, 1 contrast, 2 saturation, 3 hue; -1 = skip), factors, hue shift (uint8)
struct Jitter {
  int order[4];
  float brightness, contrast, saturation;
  int hue_shift;
};

__device__ __forceinline__ Px jitter_op(const Px& p, int op, const Jitter& j, int mean_l) {
  switch (op) {
    case 0: return blend3(0, 0, 0, p, j.brightness);
    case 1: return blend3(mean_l, mean_l, mean_l, p, j.contrast);
    case 2: { const int l = rgb2l(p); return blend3(l, l, l, p, j.saturation); }
    case 3: return hue_shift(p, j.hue_shift);
    default: return p;
  }
}

// pass 1: luminance sum of the image as it stands right before its contrast op (exact integer sum: order-independent)
__global__ __launch_bounds__(256) void jitter_lsum_kernel(const unsigned char* __restrict__ src, const Jitter* __restrict__ jit,
                                                          unsigned long long* __restrict__ lsum, int B, int HW) {
  __shared__ unsigned long long part[4];
  const int b = blockIdx.y;
  const Jitter j = jit[b];
  int npre = -1;
  for (int t = 0; t < 4; ++t)
    if (j.order[t] == 1) npre = t;
  if (npre < 0) return;  // no contrast op for this sample
  unsigned long long acc = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const unsigned char* p = src + ((long)b * HW + i) * 3;
    Px x{p[0], p[1], p[2]};
    for (int t = 0; t < npre; ++t) x = jitter_op(x, j.order[t], j, 0);
    acc += (unsigned)rgb2l(x);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(lsum + b, part[0] + part[1] + part[2] + part[3]);
}

// pass 2: the whole chain (ImageStat mean = sum / count in double, int(mean + 0.5))
__global__ __launch_bounds__(256) void jitter_apply_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                           const Jitter* __restrict__ jit,
                                                           const unsigned long long* __restrict__ lsum, int B, int HW) {
  const int b = blockIdx.y;
  const Jitter j = jit[b];
  const int mean_l = (int)((double)lsum[b] / (double)HW + 0.5);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const unsigned char* p = src + ((long)b * HW + i) * 3;
    Px x{p[0], p[1], p[2]};
#pragma unroll
    for (int t = 0; t < 4; ++t) x = jitter_op(x, j.order[t], j, mean_l);
    unsigned char* q = dst + ((long)b * HW + i) * 3;
    q[0] = (unsigned char)x.r; q[1] = (unsigned char)x.g; q[2] = (unsigned char)x.b;
  }
}

// ---------------------------------------------------------------------------------------------
// GaussianBlur: separable, reflect padding, f32 taps per sample, taps summed in index order with separate multiply / add
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// rows: u8 [B][H][W][3] -> f32 [B][H][W][3]
__global__ __launch_bounds__(256) void blur_rows_kernel(const unsigned char* __restrict__ src, float* __restrict__ tmp,
                                                        const float* __restrict__ taps, int ksize, int B, int H, int W) {
  const long total = (long)B * H * W;
  const int r = ksize >> 1;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int x = id % W;
    const long row = id / W;  // b * H + y
    const int b = row / H;
    const float* k = taps + (long)b * ksize;
    const unsigned char* line = src + row * W * 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int t = 0; t < ksize; ++t) {
      const unsigned char* p = line + reflect(x + t - r, W) * 3;
      const float kk = k[t];
      a0 = __fadd_rn(a0, __fmul_rn(kk, (float)p[0]));
      a1 = __fadd_rn(a1, __fmul_rn(kk, (float)p[1]));
      a2 = __fadd_rn(a2, __fmul_rn(kk, (float)p[2]));
    }
    float* q = tmp + id * 3;
    q[0] = a0; q[1] = a1; q[2] = a2;
  }
}

// columns: f32 [B][H][W][3] -> u8, rint (half to even, as torch.round) and clamp
__global__ __launch_bounds__(256) void blur_cols_kernel(const float* __restrict__ tmp, unsigned char* __restrict__ dst,
                                                        const float* __restrict__ taps, int ksize, int B, int H, int W) {
  const long total = (long)B * H * W;
  const int r = ksize >> 1;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int x = id % W;
    const int y = (id / W) % H;
    const int b = id / ((long)W * H);
    const float* k = taps + (long)b * ksize;
    const float* img = tmp + (long)b * H * W * 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int t = 0; t < ksize; ++t) {
      const float* p = img + ((long)reflect(y + t - r, H) * W + x) * 3;
      const float kk = k[t];
      a0 = __fadd_rn(a0, __fmul_rn(kk, p[0]));
      a1 = __fadd_rn(a1, __fmul_rn(kk, p[1]));
      a2 = __fadd_rn(a2, __fmul_rn(kk, p[2]));
    }
    unsigned char* q = dst + id * 3;
    a0 = fminf(fmaxf(rintf(a0), 0.f), 255.f); a1 = fminf(fmaxf(rintf(a1), 0.f), 255.f); a2 = fminf(fmaxf(rintf(a2), 0.f), 255.f);
    q[0] = (unsigned char)a0; q[1] = (unsigned char)a1; q[2] = (unsigned char)a2;
  }
}

// ---------------------------------------------------------------------------------------------
// flips + rotation (Pillow affine_fixed, nearest, fill 0) + ToTensor + Normalize -> f32 NCHW, or the rotated u8 frame
// ---------------------------------------------------------------------------------------------
// geom[b] = {mode, a0, a1, a2, a3, a4, a5, flips}: mode 0 = 16.16 fixed-point inverse map (xin = (a2 + a0 x + a1 y) >> 16, ...),
// 1 = identity, 2 = 180 degrees, 3 = 90 degrees (H == W), 4 = 270 degrees -- Image.rotate's transpose fast paths
struct Geom { int mode, a0, a1, a2, a3, a4, a5, flips; };

__device__ __forceinline__ bool geom_source(const Geom& g, int x, int y, int H, int W, int& sx, int& sy) {
  int xin, yin;
  switch (g.mode) {
    case 1: xin = x; yin = y; break;
    case 2: xin = W - 1 - x; yin = H - 1 - y; break;
    case 3: xin = W - 1 - y; yin = x; break;   // np.rot90(k=1): out[y][x] = in[x][W-1-y]
    case 4: xin = y; yin = H - 1 - x; break;   // np.rot90(k=3): out[y][x] = in[H-1-x][y]
    default: {
      xin = (g.a2 + g.a0 * x + g.a1 * y) >> 16;
      yin = (g.a5 + g.a3 * x + g.a4 * y) >> 16;
    }
  }
  if (xin < 0 || xin >= W || yin < 0 || yin >= H) return false;
  // the rotation reads the FLIPPED image (transforms.py:241-243: flips come first)
  sx = (g.flips & 1) ? W - 1 - xin : xin;
  sy = (g.flips & 2) ? H - 1 - yin : yin;
  return true;
}

template <bool TO_F32>
__global__ __launch_bounds__(256) void geom_kernel(const unsigned char* __restrict__ src, const Geom* __restrict__ geom,
                                                   void* __restrict__ dst, int B, int H, int W, f32x4 mean, f32x4 stdv) {
  const long total = (long)B * H * W;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int x = id % W;
    const int y = (id / W) % H;
    const int b = id / ((long)W * H);
    const Geom g = geom[b];
    int sx, sy, v0 = 0, v1 = 0, v2 = 0;  // fillcolor 0
    if (geom_source(g, x, y, H, W, sx, sy)) {
      const unsigned char* p = src + (((long)b * H + sy) * W + sx) * 3;
      v0 = p[0]; v1 = p[1]; v2 = p[2];
    }
    if constexpr (TO_F32) {
      float* out = reinterpret_cast<float*>(dst);
      const long plane = (long)H * W;
      const long o = (long)b * 3 * plane + (long)y * W + x;
      out[o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v0, 255.0f), mean[0]), stdv[0]);
      out[o + plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v1, 255.0f), mean[1]), stdv[1]);
      out[o + 2 * plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v2, 255.0f), mean[2]), stdv[2]);
    } else {
      unsigned char* q = reinterpret_cast<unsigned char*>(dst) + id * 3;
      q[0] = (unsigned char)v0; q[1] = (unsigned char)v1; q[2] = (unsigned char)v2;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The eval-time perturbations of PerRowPerturbations (classification/data/transforms.py:143-203) that are pixel arithmetic:
//   "blur*"  img.filter(ImageFilter.GaussianBlur(radius=sigma))  = Pillow BoxBlur.c ImagingGaussianBlur: three passes of an
//            "extended box" per axis -- integer radius r plus a fractional weight on the two pixels beyond it -- in UINT32 fixed
//            point: out = (ww * sum_{|d| <= r} in[x + d] + fw * (in[x - r - 1] + in[x + r + 1]) + 2^23) >> 24, indices clamped
//            to the line (the C code keeps a running sum along the line; every output is this window);
//   "occ*"   ImageDraw.rectangle([x0, y0, x1, y1], fill=0): both corners inclusive.
// ("bc*" is two ImageEnhance blends: pm_aug_color_jitter_u8 with order {0, 1, -1, -1}; "jpeg*" is a codec round trip and stays
// on the host like the decoding itself.)  r, ww, fw come from the host per sample (float32 arithmetic of ImagingHorizontalBoxBlur).
// ---------------------------------------------------------------------------------------------
struct BoxBlur {
  int radius;            // < 0: this sample is not blurred (the pass copies it)
  unsigned int ww, fw;
};

// one pass along x (axis = 0) or y (axis = 1) of every sample: u8 [B][H][W][3] -> the same
__global__ __launch_bounds__(256) void box_blur_pass_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                            const BoxBlur* __restrict__ prm, int axis, int B, int H, int W) {
  const long total = (long)B * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int b = i / ((long)H * W);
    const int rem = i % ((long)H * W);
    const int y = rem / W, x = rem % W;
    const unsigned char* img = src + (long)b * H * W * 3;
    const BoxBlur p = prm[b];
    unsigned char* o = dst + i * 3;
    if (p.radius < 0) {
      o[0] = img[(long)rem * 3];
      o[1] = img[(long)rem * 3 + 1];
      o[2] = img[(long)rem * 3 + 2];
      continue;
    }
    const int n = axis == 0 ? W : H, pos = axis == 0 ? x : y;
    const long step = axis == 0 ? 3 : (long)W * 3;
    const unsigned char* line = img + (axis == 0 ? (long)y * W * 3 : (long)x * 3);
    unsigned int acc[3] = {0u, 0u, 0u};
    for (int d = -p.radius; d <= p.radius; ++d) {
      int q = pos + d;
      q = q < 0 ? 0 : (q > n - 1 ? n - 1 : q);
      const unsigned char* px = line + q * step;
      acc[0] += px[0];
      acc[1] += px[1];
      acc[2] += px[2];
    }
    int ql = pos - p.radius - 1, qr = pos + p.radius + 1;
    ql = ql < 0 ? 0 : ql;
    qr = qr > n - 1 ? n - 1 : qr;
    const unsigned char* pl = line + ql * step;
    const unsigned char* pr = line + qr * step;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned int bulk = acc[c] * p.ww + ((unsigned int)pl[c] + (unsigned int)pr[c]) * p.fw;  // UINT32 wrap-around as in C
      o[c] = (unsigned char)((bulk + (1u << 23)) >> 24);
    }
  }
}

// rects: int [B][4] = x0, y0, x1, y1 (inclusive; x1 < x0: nothing drawn), in place
__global__ __launch_bounds__(256) void occlude_kernel(unsigned char* __restrict__ img, const int* __restrict__ rects, int B, int H, int W) {
  const int b = blockIdx.y;
  int x0 = rects[4 * b], y0 = rects[4 * b + 1], x1 = rects[4 * b + 2], y1 = rects[4 * b + 3];
  x0 = x0 < 0 ? 0 : x0;
  y0 = y0 < 0 ? 0 : y0;
  x1 = x1 > W - 1 ? W - 1 : x1;
  y1 = y1 > H - 1 ? H - 1 : y1;
  if (x1 < x0 || y1 < y0) return;
  const int rw = x1 - x0 + 1, rh = y1 - y0 + 1;
  const long total = (long)rw * rh * 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int yy = i / (rw * 3), r = i % (rw * 3);
    img[((long)b * H + y0 + yy) * W * 3 + (long)x0 * 3 + r] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// "jpeg*" rows: img.save(format="JPEG", quality=q, optimize=False, subsampling=0) and back (transforms.py:78-85).  The pixels
// that come back do not depend on the entropy coding (lossless) but on libjpeg's integer pipeline only, which is restated here step
// for step -- no bitstream is ever produced:  RGB -> YCbCr in 16-bit fixed point (jccolor.c), 4:4:4 (no subsampling), per 8 x 8
// block the "islow" forward DCT (jfdctint.c: Loeffler-Ligtenberg-Moschytz, CONST_BITS 13, PASS1_BITS 2, output scaled by 8),
// quantisation by the Annex-K tables scaled by the quality (jcparam.c: 5000 / q below 50, 200 - 2 q above; baseline clamp 1..255)
// with the divisor << 3 and round-half-away (jcdctmgr.c), dequantisation, the islow inverse DCT (jidctint.c) with its range limit,
// YCbCr -> RGB through the fixed-point tables of jdcolor.c.  Frames whose sides are not multiples of 8 are padded by replicating
// the last column / row (jcprepct.c expand edges) and cropped again.  One thread per 8 x 8 block, all three components.
// ---------------------------------------------------------------------------------------------
__device__ const unsigned char kStdLum[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                              14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                              18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                              49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
__device__ const unsigned char kStdChr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                              24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
constexpr int kF0298 = 2446, kF0390 = 3196, kF0541 = 4433, kF0765 = 6270, kF0899 = 7373, kF1175 = 9633, kF1501 = 12299, kF1847 = 15137,
              kF1961 = 16069, kF2053 = 16819, kF2562 = 20995, kF3072 = 25172;
__device__ __forceinline__ int jdescale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 8-point forward pass of jfdctint.c on d[0], d[s], ..., d[7 s] (first: rows, output << PASS1_BITS; second: columns)
__device__ __forceinline__ void jfdct8(int* d, int s, bool first) {
  const int t0 = d[0] + d[7 * s], t7 = d[0] - d[7 * s], t1 = d[s] + d[6 * s], t6 = d[s] - d[6 * s];
  const int t2 = d[2 * s] + d[5 * s], t5 = d[2 * s] - d[5 * s], t3 = d[3 * s] + d[4 * s], t4 = d[3 * s] - d[4 * s];
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  const int sh = first ? 13 - 2 : 13 + 2;
  d[0] = first ? (t10 + t11) << 2 : jdescale(t10 + t11, 2);
  d[4 * s] = first ? (t10 - t11) << 2 : jdescale(t10 - t11, 2);
  int z1 = (t12 + t13) * kF0541;
  d[2 * s] = jdescale(z1 + t13 * kF0765, sh);
  d[6 * s] = jdescale(z1 + t12 * (-kF1847), sh);
  z1 = t4 + t7;
  int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  const int z5 = (z3 + z4) * kF1175;
  const int a4 = t4 * kF0298, a5 = t5 * kF2053, a6 = t6 * kF3072, a7 = t7 * kF1501;
  z1 *= -kF0899;
  z2 *= -kF2562;
  z3 = z3 * -kF1961 + z5;
  z4 = z4 * -kF0390 + z5;
  d[7 * s] = jdescale(a4 + z1 + z3, sh);
  d[5 * s] = jdescale(a5 + z2 + z4, sh);
  d[3 * s] = jdescale(a6 + z2 + z3, sh);
  d[s] = jdescale(a7 + z1 + z4, sh);
}
// one 8-point inverse pass of jidctint.c (first: columns, descale CONST_BITS - PASS1_BITS; second: rows, + 3 more bits)
__device__ __forceinline__ void jidct8(int* c, int s, bool first) {
  int z2 = c[2 * s], z3 = c[6 * s];
  int z1 = (z2 + z3) * kF0541;
  const int e2 = z1 + z3 * (-kF1847), e3 = z1 + z2 * kF0765;
  z2 = c[0];
  z3 = c[4 * s];
  const int e0 = (z2 + z3) << 13, e1 = (z2 - z3) << 13;
  const int t10 = e0 + e3, t13 = e0 - e3, t11 = e1 + e2, t12 = e1 - e2;
  int o0 = c[7 * s], o1 = c[5 * s], o2 = c[3 * s], o3 = c[s];
  z1 = o0 + o3;
  z2 = o1 + o2;
  z3 = o0 + o2;
  int z4 = o1 + o3;
  const int z5 = (z3 + z4) * kF1175;
  o0 *= kF0298;
  o1 *= kF2053;
  o2 *= kF3072;
  o3 *= kF1501;
  z1 *= -kF0899;
  z2 *= -kF2562;
  z3 = z3 * -kF1961 + z5;
  z4 = z4 * -kF0390 + z5;
  o0 += z1 + z3;
  o1 += z2 + z4;
  o2 += z2 + z3;
  o3 += z1 + z4;
  const int sh = first ? 13 - 2 : 13 + 2 + 3;
  c[0] = jdescale(t10 + o3, sh);
  c[7 * s] = jdescale(t10 - o3, sh);
  c[s] = jdescale(t11 + o2, sh);
  c[6 * s] = jdescale(t11 - o2, sh);
  c[2 * s] = jdescale(t12 + o1, sh);
  c[5 * s] = jdescale(t12 - o1, sh);
  c[3 * s] = jdescale(t13 + o0, sh);
  c[4 * s] = jdescale(t13 - o0, sh);
}

// quality: int [B]; <= 0: the sample is copied.  In place is fine (a thread reads and writes its own block only).
__global__ __launch_bounds__(64) void jpeg_roundtrip_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                            const int* __restrict__ quality, int B, int H, int W) {
  const int bw = (W + 7) >> 3, bh = (H + 7) >> 3;
  const long total = (long)B * bh * bw;
  for (long id = (long)blockIdx.x * 64 + threadIdx.x; id < total; id += (long)gridDim.x * 64) {
    const int b = id / ((long)bh * bw), rem = id % ((long)bh * bw);
    const int by = rem / bw, bx = rem % bw;
    const unsigned char* img = src + (long)b * H * W * 3;
    unsigned char* out = dst + (long)b * H * W * 3;
    int q = quality[b];
    if (q <= 0) {
      for (int i = 0; i < 64; ++i) {
        const int y = by * 8 + (i >> 3), x = bx * 8 + (i & 7);
        if (y < H && x < W)
          for (int c = 0; c < 3; ++c) out[((long)y * W + x) * 3 + c] = img[((long)y * W + x) * 3 + c];
      }
      continue;
    }
    q = q > 100 ? 100 : q;
    const int scale = q < 50 ? 5000 / q : 200 - q * 2;
    unsigned char rec[3][64];
    for (int comp = 0; comp < 3; ++comp) {
      int v[64];
      for (int i = 0; i < 64; ++i) {
        int y = by * 8 + (i >> 3), x = bx * 8 + (i & 7);
        y = y < H ? y : H - 1;  // edge replication
        x = x < W ? x : W - 1;
        const unsigned char* p = img + ((long)y * W + x) * 3;
        const int r = p[0], g = p[1], bl = p[2];
        int s;
        if (comp == 0) s = (19595 * r + 38470 * g + 7471 * bl + 32768) >> 16;
        else if (comp == 1) s = (-11059 * r - 21709 * g + 32768 * bl + (128 << 16) + 32767) >> 16;
        else s = (32768 * r - 27439 * g - 5329 * bl + (128 << 16) + 32767) >> 16;
        v[i] = s - 128;
      }
      for (int r = 0; r < 8; ++r) jfdct8(v + 8 * r, 1, true);
      for (int c = 0; c < 8; ++c) jfdct8(v + c, 8, false);
      const unsigned char* base = comp == 0 ? kStdLum : kStdChr;
      for (int i = 0; i < 64; ++i) {
        int qt = ((int)base[i] * scale + 50) / 100;
        qt = qt < 1 ? 1 : (qt > 255 ? 255 : qt);
        const int qv = qt << 3, d = v[i];
        const int a = d < 0 ? -d : d;
        const int qc = (a + (qv >> 1)) / qv;
        v[i] = (d < 0 ? -qc : qc) * qt;   // quantise, dequantise
      }
      for (int c = 0; c < 8; ++c) jidct8(v + c, 8, true);
      for (int r = 0; r < 8; ++r) jidct8(v + 8 * r, 1, false);
      for (int i = 0; i < 64; ++i) {
        const int s = v[i] + 128;
        rec[comp][i] = (unsigned char)(s < 0 ? 0 : (s > 255 ? 255 : s));
      }
    }
    for (int i = 0; i < 64; ++i) {
      const int y = by * 8 + (i >> 3), x = bx * 8 + (i & 7);
      if (y >= H || x >= W) continue;
      const int yy = rec[0][i], cb = (int)rec[1][i] - 128, cr = (int)rec[2][i] - 128;
      int r = yy + ((91881 * cr + 32768) >> 16);
      int g = yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
      int bl = yy + ((116130 * cb + 32768) >> 16);
      unsigned char* o = out + ((long)y * W + x) * 3;
      o[0] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
      o[1] = (unsigned char)(g < 0 ? 0 : (g > 255 ? 255 : g));
      o[2] = (unsigned char)(bl < 0 ? 0 : (bl > 255 ? 255 : bl));
    }
  }
}

}  // namespace

static_assert(sizeof(Jitter) == sizeof(pm_aug_jitter), "pm_aug_jitter layout");
static_assert(sizeof(Geom) == sizeof(pm_aug_geom), "pm_aug_geom layout");
static_assert(sizeof(BoxBlur) == sizeof(pm_aug_boxblur), "pm_aug_boxblur layout");

extern "C" int pm_aug_resize_u8(const unsigned char* src, unsigned char* tmp, unsigned char* dst, const int* bounds_x,
                                const int* taps_x, int ksize_x, const int* bounds_y, const int* taps_y, int ksize_y, int B, int Hs,
                                int Ws, int Ho, int Wo, void* stream) {
  if (!src || !dst) return PM_EINVAL;
  if (B <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0) return PM_ESHAPE;
  const bool do_x = Wo != Ws, do_y = Ho != Hs;
  if ((do_x && (!bounds_x || !taps_x || ksize_x <= 0)) || (do_y && (!bounds_y || !taps_y || ksize_y <= 0))) return PM_EINVAL;
  if (do_x && do_y && !tmp) return PM_EINVAL;
  hipStream_t s = pm_stream(stream);
  if (!do_x && !do_y) {
    if (hipMemcpyAsync(dst, src, (size_t)B * Hs * Ws * 3, hipMemcpyDeviceToDevice, s) != hipSuccess) return PM_ELAUNCH;
    return PM_OK;
  }
  const unsigned char* cur = src;
  if (do_x) {  // horizontal pass first (ImagingResampleInner), into tmp when a vertical pass follows
    unsigned char* o = do_y ? tmp : dst;
    hipLaunchKernelGGL(resample_kernel, dim3(aug_grid((long)B * Hs * Wo)), dim3(256), 0, s, cur, o, bounds_x, taps_x, ksize_x, B, Hs,
                       Ws, Hs, Wo, 0);
    cur = o;
  }
  if (do_y)
    hipLaunchKernelGGL(resample_kernel, dim3(aug_grid((long)B * Ho * Wo)), dim3(256), 0, s, cur, dst, bounds_y, taps_y, ksize_y, B,
                       Hs, Wo, Ho, Wo, 1);
  return pm_check_launch();
}

extern "C" size_t pm_aug_resized_crop_workspace_bytes(int B, int Hs, int Ws, int out) {
  if (B <= 0 || Hs <= 0 || Ws <= 0 || out <= 0) return 0;
  const int big = Hs > Ws ? Hs : Ws;
  const int ksize = (int)((2.0 * big) / out + 1.0) * 2 + 3;   // >= ceil(2 * scale) * 2 + 1 for every crop inside the frame
  const size_t tables = 2 * ((size_t)B * out * 2 + (size_t)B * out * ksize) * sizeof(int);
  return tables + (size_t)B * Hs * out * 3 + 256;
}

extern "C" int pm_aug_resized_crop_u8(const unsigned char* src, const int* box, unsigned char* dst, int bicubic, int B, int Hs, int Ws,
                                      int out, void* workspace, size_t ws_bytes, void* stream) {
  if (!src || !box || !dst || !workspace) return PM_EINVAL;
  if (B <= 0 || Hs <= 0 || Ws <= 0 || out <= 0) return PM_ESHAPE;
  if (ws_bytes < pm_aug_resized_crop_workspace_bytes(B, Hs, Ws, out) || ((uintptr_t)workspace & 15)) return PM_EINVAL;
  const int big = Hs > Ws ? Hs : Ws;
  const int ksize = (int)((2.0 * big) / out + 1.0) * 2 + 3;
  int* bounds_x = reinterpret_cast<int*>(workspace);
  int* taps_x = bounds_x + (size_t)B * out * 2;
  int* bounds_y = taps_x + (size_t)B * out * ksize;
  int* taps_y = bounds_y + (size_t)B * out * 2;
  unsigned char* tmp = reinterpret_cast<unsigned char*>(taps_y + (size_t)B * out * ksize);
  hipStream_t s = pm_stream(stream);
  const int gc = (B * out + 255) / 256;
  hipLaunchKernelGGL(resample_coeffs_kernel, dim3(gc), dim3(256), 0, s, box, 0, out, bicubic ? 1 : 0, ksize, bounds_x, taps_x, B);
  hipLaunchKernelGGL(resample_coeffs_kernel, dim3(gc), dim3(256), 0, s, box, 1, out, bicubic ? 1 : 0, ksize, bounds_y, taps_y, B);
  hipLaunchKernelGGL(resample_crop_kernel, dim3(aug_grid((long)B * Hs * out)), dim3(256), 0, s, src, tmp, box, bounds_x, taps_x, ksize,
                     B, Hs, Ws, Hs, out, out, 0);
  hipLaunchKernelGGL(resample_crop_kernel, dim3(aug_grid((long)B * out * out)), dim3(256), 0, s, tmp, dst, box, bounds_y, taps_y, ksize,
                     B, Hs, out, out, out, out, 1);
  return pm_check_launch();
}

extern "C" int pm_aug_color_jitter_u8(const unsigned char* src, unsigned char* dst, const pm_aug_jitter* jitter,
                                      unsigned long long* lsum, int B, int H, int W, void* stream) {
  if (!src || !dst || !jitter || !lsum) return PM_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0) return PM_ESHAPE;
  hipStream_t s = pm_stream(stream);
  if (hipMemsetAsync(lsum, 0, (size_t)B * sizeof(unsigned long long), s) != hipSuccess) return PM_ELAUNCH;
  const int HW = H * W;
  int gx = (HW + 256 * 8 - 1) / (256 * 8);
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(jitter_lsum_kernel, dim3(gx, B), dim3(256), 0, s, src, reinterpret_cast<const Jitter*>(jitter), lsum, B, HW);
  hipLaunchKernelGGL(jitter_apply_kernel, dim3(gx, B), dim3(256), 0, s, src, dst, reinterpret_cast<const Jitter*>(jitter), lsum, B,
                     HW);
  return pm_check_launch();
}

extern "C" int pm_aug_gaussian_blur_u8(const unsigned char* src, float* tmp, unsigned char* dst, const float* taps, int ksize, int B,
                                       int H, int W, void* stream) {
  if (!src || !tmp || !dst || !taps) return PM_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0 || ksize <= 0 || !(ksize & 1)) return PM_ESHAPE;
  if ((ksize >> 1) >= H || (ksize >> 1) >= W) return PM_ESHAPE;  // reflect padding needs pad < size (as torch does)
  hipStream_t s = pm_stream(stream);
  const int g = aug_grid((long)B * H * W);
  hipLaunchKernelGGL(blur_rows_kernel, dim3(g), dim3(256), 0, s, src, tmp, taps, ksize, B, H, W);
  hipLaunchKernelGGL(blur_cols_kernel, dim3(g), dim3(256), 0, s, tmp, dst, taps, ksize, B, H, W);
  return pm_check_launch();
}

extern "C" int pm_aug_geometry_u8(const unsigned char* src, const pm_aug_geom* geom, void* dst, int to_f32, int B, int H, int W,
                                  float mean_r, float mean_g, float mean_b, float std_r, float std_g, float std_b, void* stream) {
  if (!src || !geom || !dst) return PM_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0) return PM_ESHAPE;
  if (to_f32 && (!(std_r > 0.f) || !(std_g > 0.f) || !(std_b > 0.f))) return PM_EINVAL;
  const int g = aug_grid((long)B * H * W);
  const f32x4 mean{mean_r, mean_g, mean_b, 0.f}, stdv{std_r, std_g, std_b, 1.f};
  if (to_f32)
    hipLaunchKernelGGL(geom_kernel<true>, dim3(g), dim3(256), 0, pm_stream(stream), src, reinterpret_cast<const Geom*>(geom), dst, B,
                       H, W, mean, stdv);
  else
    hipLaunchKernelGGL(geom_kernel<false>, dim3(g), dim3(256), 0, pm_stream(stream), src, reinterpret_cast<const Geom*>(geom), dst, B,
                       H, W, mean, stdv);
  return pm_check_launch();
}

extern "C" int pm_aug_pil_gaussian_blur_u8(const unsigned char* src, unsigned char* tmp, unsigned char* dst, const pm_aug_boxblur* prm,
                                           int passes, int B, int H, int W, void* stream) {
  if (!src || !tmp || !dst || !prm) return PM_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0 || passes < 1 || passes > 8) return PM_ESHAPE;
  if (src == tmp || tmp == dst) return PM_EINVAL;
  const BoxBlur* p = reinterpret_cast<const BoxBlur*>(prm);
  const int grid = aug_grid((long)B * H * W);
  hipStream_t s = pm_stream(stream);
  // 2 * passes launches (x passes, then y passes) ping-pong between tmp and dst so that the last one lands in dst; src is only
  // read by the first, so src == dst is fine
  const unsigned char* in = src;
  for (int k = 0; k < 2 * passes; ++k) {
    unsigned char* out = ((2 * passes - 1 - k) & 1) ? tmp : dst;
    hipLaunchKernelGGL(box_blur_pass_kernel, dim3(grid), dim3(256), 0, s, in, out, p, k < passes ? 0 : 1, B, H, W);
    in = out;
  }
  return pm_check_launch();
}

extern "C" int pm_aug_occlude_u8(unsigned char* img, const int* rects, int B, int H, int W, void* stream) {
  if (!img || !rects) return PM_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0) return PM_ESHAPE;
  hipLaunchKernelGGL(occlude_kernel, dim3(64, B), dim3(256), 0, pm_stream(stream), img, rects, B, H, W);
  return pm_check_launch();
}

extern "C" int pm_aug_jpeg_roundtrip_u8(const unsigned char* src, unsigned char* dst, const int* quality, int B, int H, int W,
                                        void* stream) {
  if (!src || !dst || !quality) return PM_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0) return PM_ESHAPE;
  const long blocks = (long)B * ((H + 7) / 8) * ((W + 7) / 8);
  long grid = (blocks + 63) / 64;
  grid = grid > 65535 ? 65535 : grid;
  hipLaunchKernelGGL(jpeg_roundtrip_kernel, dim3((int)grid), dim3(64), 0, pm_stream(stream), src, dst, quality, B, H, W);
  return pm_check_launch();
}
