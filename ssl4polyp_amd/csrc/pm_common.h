// pm_common.h -- shared device helpers for the gfx950 kernels (wave64, MFMA 32x32, bf16/f32 element traits).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/polypmae.h"

// Qualifier of the LDS image pointer in the functions that read MFMA fragments with ds_read_b64_tr_b16.
// The waitcnt pass of the compiler guards every LDS read whose memory operand carries no alias scope with
// `s_waitcnt vmcnt(0)` while LDS-DMA loads (global_load_lds) are in flight -- in a ring that keeps the next stages in
// flight BEHIND the reads of the current one (counted vmcnt + barrier, placed by hand) that wait serialises the
// prefetch: the stage issued in k-step t had to land by the top of k-step t+1.  `__restrict__` on the image pointer
// gives the transpose reads an alias scope (as the plain ds_read_b128 fragment reads already have), and the pass
// leaves them alone.  -DPM_AUTO_VMCNT restores the guarded form (A/B builds).
#ifdef PM_AUTO_VMCNT
#define PM_LDS_IMAGE
#else
#define PM_LDS_IMAGE __restrict__
#endif

#define PM_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// A 16-byte operand fragment: 8 bf16 or 8 f16 (one 32x32x16 MFMA) or 4 f32 (four 32x32x2 MFMAs).
union Frag16 {
  u32x4 u;
  bf16x8 h;
  f16x8 g;
  f32x4 f;
};

static inline hipStream_t pm_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// One-time (per kernel instantiation, per process = per GPU) opt-in to > 64 KiB of dynamic LDS.  Idempotent, so the
// unsynchronised flag is benign; keeps the per-launch host cost to the launch itself.
#define PM_ALLOW_LDS(kern, bytes)                                                                                  \
  do {                                                                                                             \
    static bool pm_lds_done_ = false;                                                                              \
    if (!pm_lds_done_) {                                                                                           \
      (void)hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);      \
      pm_lds_done_ = true;                                                                                         \
    }                                                                                                              \
  } while (0)

static inline int pm_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PM_OK : PM_ELAUNCH;
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<__bf16> {
  static constexpr int kDtype = PM_BF16;
  static constexpr int kPer16B = 8;
};
template <> struct ElemTraits<_Float16> {
  static constexpr int kDtype = PM_F16;
  static constexpr int kPer16B = 8;
};
template <> struct ElemTraits<float> {
  static constexpr int kDtype = PM_F32;
  static constexpr int kPer16B = 4;
};
template <typename T> struct IsHalf { static constexpr bool value = false; };
template <> struct IsHalf<_Float16> { static constexpr bool value = true; };

// The activation type of a dtype code, as a compile-time type inside `...`:  PM_DISPATCH_ACT(dtype, T, launch<T>(...));
// an unknown code returns PM_EINVAL from the enclosing function.
#define PM_DISPATCH_ACT(dtype, T, ...)                         \
  do {                                                         \
    if ((dtype) == PM_BF16) { using T = __bf16; __VA_ARGS__; }  \
    else if ((dtype) == PM_F16) { using T = _Float16; __VA_ARGS__; } \
    else if ((dtype) == PM_F32) { using T = float; __VA_ARGS__; }   \
    else return PM_EINVAL;                                     \
  } while (0)
// the same over the two 16-bit types only
#define PM_DISPATCH_16(dtype, T, ...)                          \
  do {                                                         \
    if ((dtype) == PM_BF16) { using T = __bf16; __VA_ARGS__; }  \
    else if ((dtype) == PM_F16) { using T = _Float16; __VA_ARGS__; } \
    else return PM_EINVAL;                                     \
  } while (0)

// acc[reg -> A-row][lane -> B-col] += A(16B frag) x B(16B frag).
// bf16: one v_mfma_f32_32x32x16_bf16 (lane (r,h) holds k = 8h..8h+7 of row/col r); f16: v_mfma_f32_32x32x16_f16, same map,
//       same rate (precision mode "fp16": 11 significant bits per operand instead of 8).
// f32 : four v_mfma_f32_32x32x2_f32; element e of lane half h stands for k = 4*(2j+h)+e in BOTH
//       operands, so any k permutation is consistent (exact f32 fma chain).
template <typename T>
__device__ __forceinline__ f32x16 mfma16B(const Frag16& a, const Frag16& b, f32x16 acc) {
  if constexpr (IsHalf<T>::value) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.g, b.g, acc, 0, 0, 0);
  } else if constexpr (sizeof(T) == 2) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[0], b.f[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[1], b.f[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[2], b.f[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[3], b.f[3], acc, 0, 0, 0);
    return acc;
  }
}
// 16 x 16 x 32 form of the 16-bit types (4 accumulator registers)
template <typename T>
__device__ __forceinline__ f32x4 mfma16x16(const Frag16& a, const Frag16& b, f32x4 acc) {
  if constexpr (IsHalf<T>::value) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.g, b.g, acc, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, acc, 0, 0, 0);
}
// element j of a 16-bit fragment <- f32 (the accumulator fed back as an MFMA operand)
template <typename T> __device__ __forceinline__ void frag_set(Frag16& f, int j, float v) {
  if constexpr (IsHalf<T>::value) f.g[j] = (_Float16)v;
  else f.h[j] = (__bf16)v;
}
template <typename T> __device__ __forceinline__ float frag_get(const Frag16& f, int j) {
  if constexpr (IsHalf<T>::value) return (float)f.g[j];
  else if constexpr (sizeof(T) == 2) return (float)f.h[j];
  else return f.f[j];
}
// two 1.0 of the 16-bit type in one dword
template <typename T> __device__ __forceinline__ unsigned ones2() { return IsHalf<T>::value ? 0x3C003C00u : 0x3F803F80u; }

// Row index (within a 32x32 accumulator tile) of register `reg` for lane half `h`.
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// Load / store 4 consecutive elements as f32x4.
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p) {
  if constexpr (sizeof(T) == 4) {
    return *reinterpret_cast<const f32x4*>(p);
  } else {
    typedef T __attribute__((ext_vector_type(4))) V4;
    const V4 v = *reinterpret_cast<const V4*>(p);
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
  }
}
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<f32x4*>(p) = v;
  } else {
    typedef T __attribute__((ext_vector_type(4))) V4;
    const V4 o = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
    *reinterpret_cast<V4*>(p) = o;
  }
}

// 8 consecutive bf16 as one 16-B access (a wave store instruction costs the CU's store path ~64 cycles whatever
// its width: 16 B per lane halves the epilogue's store time against 8 B per lane)
template <typename T> __device__ __forceinline__ void load8_16(const T* p, f32x4& lo, f32x4& hi) {
  typedef T __attribute__((ext_vector_type(8))) V8;
  const V8 v = *reinterpret_cast<const V8*>(p);
  lo = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  hi = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
}
template <typename T> __device__ __forceinline__ void store8_16(T* p, f32x4 lo, f32x4 hi) {
  typedef T __attribute__((ext_vector_type(8))) V8;
  const V8 o = {(T)lo[0], (T)lo[1], (T)lo[2], (T)lo[3], (T)hi[0], (T)hi[1], (T)hi[2], (T)hi[3]};
  *reinterpret_cast<V8*>(p) = o;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum of p[b * stride] for b = first, first + step, ... < n, in that fixed order (deterministic), with U loads in
// flight at a time (the second-stage reductions are latency-bound otherwise)
template <int U>
__device__ __forceinline__ float strided_sum(const float* __restrict__ p, long stride, int first, int step, int n) {
  float s = 0.f;
  int b = first;
  for (; b + (U - 1) * step < n; b += U * step) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[(long)(b + u * step) * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u];
  }
  for (; b < n; b += step) s += p[(long)b * stride];
  return s;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf-GELU (timm Mlp act_layer=nn.GELU, exact erf form) and its derivative.
// Phi(x) = 0.5 (1 + erf(x / sqrt 2)) with erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. f32
// round-off level): one v_exp + one v_rcp + 6 FMAs instead of libm's branchy erff; the exponential
// exp(-x^2/2) is shared with the Gaussian pdf that the derivative needs.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& e) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.2316418882f * ax);  // 0.3275911 / sqrt(2)
  e = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);       // exp(-x^2/2)
  float p = 1.061405429f;
  p = p * t - 1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t - 0.284496736f;
  p = p * t + 0.254829592f;
  const float half_erfc = 0.5f * p * t * e;  // 0.5 * erfc(|x|/sqrt2)
  cdf = x >= 0.f ? 1.0f - half_erfc : half_erfc;
}
// The same on two values at once: the polynomial / product chain as packed f32 math (v_pk_fma_f32, v_pk_mul_f32: two
// lanes' worth per issue slot), only rcp / exp2 / the sign select per component.  The GEMM epilogues that apply GELU or
// its derivative to a 256x256 tile are VALU-bound there (~20 ops per element, no MFMA to hide behind): ~13 slots per
// element instead of ~18.  Same operations in the same order per component as gelu_parts.
__device__ __forceinline__ void gelu_parts2(f32x2 x, f32x2& cdf, f32x2& e) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 den = __builtin_elementwise_fma(ax, f32x2{0.2316418882f, 0.2316418882f}, f32x2{1.0f, 1.0f});
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  const f32x2 a = (x * -0.72134752044448170368f) * x;
  e = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  f32x2 p = {1.061405429f, 1.061405429f};
  p = __builtin_elementwise_fma(p, t, f32x2{-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(p, t, f32x2{1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(p, t, f32x2{-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(p, t, f32x2{0.254829592f, 0.254829592f});
  const f32x2 half_erfc = ((p * 0.5f) * t) * e;
  const f32x2 upper = f32x2{1.0f, 1.0f} - half_erfc;
  cdf = f32x2{x[0] >= 0.f ? upper[0] : half_erfc[0], x[1] >= 0.f ? upper[1] : half_erfc[1]};
}
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  f32x2 cdf, e;
  gelu_parts2(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ f32x2 gelu_erf_grad2(f32x2 x) {
  f32x2 cdf, e;
  gelu_parts2(x, cdf, e);
  return __builtin_elementwise_fma(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float gelu_erf(float x) {
  float cdf, e;
  gelu_parts(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float cdf, e;
  gelu_parts(x, cdf, e);
  return cdf + x * 0.39894228040143267794f * e;
}

// f32x4 forms used by the GEMM epilogues
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
#ifdef PM_GELU_IDENTITY  // diagnostic build: what the epilogue costs without its exp / rcp (results are wrong)
  return v * 0.5f;
#elif defined(PM_GELU_SCALAR)  // A/B build (scratch/build_alt.sh): one value at a time
  return f32x4{gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
#else
  const f32x2 a = gelu_erf2(f32x2{v[0], v[1]}), b = gelu_erf2(f32x2{v[2], v[3]});
  return f32x4{a[0], a[1], b[0], b[1]};
#endif
}
__device__ __forceinline__ f32x4 gelu_erf_grad4(f32x4 v) {
#ifdef PM_GELU_IDENTITY
  return v * 0.5f;
#elif defined(PM_GELU_SCALAR)
  return f32x4{gelu_erf_grad(v[0]), gelu_erf_grad(v[1]), gelu_erf_grad(v[2]), gelu_erf_grad(v[3])};
#else
  const f32x2 a = gelu_erf_grad2(f32x2{v[0], v[1]}), b = gelu_erf_grad2(f32x2{v[2], v[3]});
  return f32x4{a[0], a[1], b[0], b[1]};
#endif
}
// 16-bit precision modes: Phi(x) without the quarter-rate rcp and exp2 -- an odd polynomial on the clamped argument,
//   Phi(x) ~= 0.5 + xc Q(t),  xc = clamp(x, -4.5, 4.5),  t = 2 xc^2 / 4.5^2 - 1,  Q of degree 10 by Horner (weighted least-squares
//   fit iterated towards minimax; evaluated in f32 with FMAs: |error| <= 2.8e-6 over all x, 1 - Phi(4.5) = 3.4e-6 being the floor of
//   the clamp).  The outputs of these epilogues are rounded to bf16 (relative 2e-3) or fp16 (5e-4) next, so the approximation sits two
//   to three orders below the rounding it feeds; fp32 mode keeps the 1.5e-7 form above.  8 issue slots per element as packed f32
//   math instead of 17 (experiment 14: the GELU epilogue is half of an fc1 launch, ~10 us of VALU per tile with no MFMA to hide behind).
//   The derivative keeps exp(-x^2/2) for the pdf term: one exp2, no rcp (14 slots instead of 19).
//   MEASURED (experiment 15, profiles/r4_exp15_*): fc1 + GELU 64.2 -> 59.7 us stand-alone at M = 6 304 (103.0 -> 99.3 at 12 608), dGELU
//   dgrad -1.7 %, the step +0.3 % -- and every bf16 rounding decision downstream is re-rolled, which moved the noisiest gated quantity
//   (worst weight-gradient rel-L2 of the bf16 classifier at B = 64: 3.0e-2 measured with the erf form, 3.9e-2 in the CPU emulation of
//   the rounding points) to 4.15e-2, past a gate that is "measured + 25 %".  Not worth a re-gate for +0.3 %: OFF by default,
//   -DPM_GELU_POLY (scratch/build_alt.sh) builds it.
#ifdef PM_GELU_POLY
__device__ __forceinline__ f32x2 phi_poly2(f32x2 x) {
  const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -4.5f, 4.5f), __builtin_amdgcn_fmed3f(x[1], -4.5f, 4.5f)};
  const f32x2 t = __builtin_elementwise_fma(xc * xc, f32x2{0.09876543283462524f, 0.09876543283462524f}, f32x2{-1.0f, -1.0f});
  f32x2 q = {8.193445974e-04f, 8.193445974e-04f};
  q = __builtin_elementwise_fma(q, t, f32x2{-2.259161090e-03f, -2.259161090e-03f});
  q = __builtin_elementwise_fma(q, t, f32x2{3.079207381e-03f, 3.079207381e-03f});
  q = __builtin_elementwise_fma(q, t, f32x2{-5.413250532e-03f, -5.413250532e-03f});
  q = __builtin_elementwise_fma(q, t, f32x2{1.115704700e-02f, 1.115704700e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2{-1.891482994e-02f, -1.891482994e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2{2.837207168e-02f, 2.837207168e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2{-4.013447464e-02f, -4.013447464e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2{5.469175428e-02f, 5.469175428e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2{-7.719228417e-02f, -7.719228417e-02f});
  q = __builtin_elementwise_fma(q, t, f32x2{1.569050848e-01f, 1.569050848e-01f});
  return __builtin_elementwise_fma(xc, q, f32x2{0.5f, 0.5f});
}
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) { return x * phi_poly2(x); }
__device__ __forceinline__ f32x2 gelu_poly_grad2(f32x2 x) {
  const f32x2 a = (x * -0.72134752044448170368f) * x;
  const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};  // exp(-x^2/2)
  return __builtin_elementwise_fma(x * 0.39894228040143267794f, e, phi_poly2(x));
}
#endif
// GELU / its derivative as the epilogues of activation type T apply them: the polynomial Phi for the 16-bit types, the erf form for f32
// when built with -DPM_GELU_POLY; the erf form everywhere otherwise (the default)
template <typename T> __device__ __forceinline__ f32x4 gelu_act4(f32x4 v) {
#ifdef PM_GELU_POLY
  if constexpr (sizeof(T) == 2) {
    const f32x2 a = gelu_poly2(f32x2{v[0], v[1]}), b = gelu_poly2(f32x2{v[2], v[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
  }
#endif
  return gelu_erf4(v);
}
template <typename T> __device__ __forceinline__ f32x4 gelu_act_grad4(f32x4 v) {
#ifdef PM_GELU_POLY
  if constexpr (sizeof(T) == 2) {
    const f32x2 a = gelu_poly_grad2(f32x2{v[0], v[1]}), b = gelu_poly_grad2(f32x2{v[2], v[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
  }
#endif
  return gelu_erf_grad4(v);
}
// the value the backward pass will see: the pre-activation after its round trip through the activation dtype
template <typename T> __device__ __forceinline__ f32x4 round_through(f32x4 v) {
  return f32x4{to_f32<T>(from_f32<T>(v[0])), to_f32<T>(from_f32<T>(v[1])), to_f32<T>(from_f32<T>(v[2])), to_f32<T>(from_f32<T>(v[3]))};
}
