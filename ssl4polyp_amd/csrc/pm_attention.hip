// pm_attention.hip -- fused multi-head self-attention core, forward and backward, for the short
// sequences of ViT-B/16 (N = 197, or 50 under MAE masking) and its MAE decoder (N = 197, dh = 32).
// Replaces timm 0.4.12 Attention.forward between the qkv and proj Linears
//   q,k,v = qkv.reshape(B,N,3,H,dh).permute(2,0,3,1,4); softmax(q k^T * dh^-0.5) v; transpose(1,2).reshape(B,N,C)
// (used by reference models_mae.py:39-41,53-55 through timm Block) and its autograd backward.
//
// MI355X design: a whole (batch, head) problem fits one CU -- K and V (or Q and dO) of the head are staged
// once into LDS (<= 2 x 224 x 64 bf16 = 56 KiB), scores / probabilities live only in MFMA accumulators, the
// [N,N] matrix never reaches HBM.  All contractions are v_mfma_f32_32x32x16_bf16 (f32 mode:
// v_mfma_f32_32x32x2_f32); products are oriented so that the reduction index of the NEXT product sits
// in the accumulator's register dimension, which lets the accumulator be fed back as the MFMA B operand
// with no LDS round trip ("accumulator as operand"):
//   fwd   : S^T[key][q] = K Q^T   (lane = q)  -> softmax over registers (+1 cross-half shuffle)
//           O^T[d][q]  += V^T[d][key] P^T[key][q]          (V^T fragments by ds_read_b64_tr_b16)
//   bwd   : bf16: ONE kernel per (batch, head) (attn_bwd_fused_kernel below): Q, K, V, dO staged once by LDS-DMA;
//           pass 1 (wave = query tile, lane = q):  S^T, dP^T[key][q] = V dO^T -> dS^T -> dQ^T[d][q] += K^T[d][key] dS^T[key][q]
//           pass 2 (wave = key tile, lane = key):  S[q][key] = Q K^T, dP[q][key] = dO V^T ->
//                   dV^T[d][key] += dO^T[d][q] P[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]
//           f32 mode keeps the two-kernel form (attn_bwd_q_kernel / attn_bwd_kv_kernel: the same two passes, each staging
//           half of the head and fetching the other half as per-lane global fragments).
// Every output tile therefore has (q or key) on the lane and 4 consecutive d per register quad: stores
// are 8-B (bf16) / 16-B (f32) vectors into the [B,N,(3,)H,dh] activations, the head transpose is free.
// S and dP are recomputed in pass 2 (7 products instead of 5) to avoid any cross-wave reduction or atomics:
// deterministic.  Roofline: HBM / MFMA; algorithmic FLOPs fwd 4*N^2*dh per head, bwd 10*N^2*dh.
#include "pm_common.h"
#include <stdlib.h>

namespace {

// tile loops of the backward kernels: two tiles per iteration at dh = 32 (half the registers per tile)
template <int DH> constexpr int kTileUnroll = DH == 32 ? 2 : 1;

template <int RB> __device__ __forceinline__ int swz(int row) {
  // XOR applied to the 16-B chunk index of LDS row `row` (row = RB bytes).  Chosen so that BOTH the
  // ds_read_b128 row reads of the 32x32x16 operand map and the ds_read_b64_tr_b16 transposed reads
  // (4 consecutive rows per 16-lane group) are bank-conflict free.
  if constexpr (RB == 64) return (row >> 2) & 3;
  else if constexpr (RB == 128) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
  else if constexpr (RB == 384) return row & 7;   // 24 slots: the XOR must stay inside a group of 8
  else return row & 15;
}

// Bytes between rows of an LDS image of DH-element rows.  dh = 32 / 64: the row itself.  dh = 80 (ViT-H: 1280 / 16 heads) is
// not a power of two: the row is padded to a pitch the chunk swizzle covers (16-bit: 128 elements = 256 B, 16 slots; f32: 96
// elements = 384 B, 24 slots); the products run over 5 k-steps of 16 (10 of 8 in f32) and 3 output tiles of 32 features, the
// last one half padding (zeros in the image, never stored).
template <typename T, int DH> constexpr int row_bytes() {
  return DH == 80 ? (sizeof(T) == 2 ? 256 : 384) : DH * (int)sizeof(T);
}

// Stage `rows_valid` rows of DH elements (global row stride `ld` elements) into an LDS image of `rows_total`
// rows; rows beyond rows_valid are zero filled.
template <typename T, int DH>
__device__ __forceinline__ void load_image(char* img, const T* __restrict__ src, long ld, int rows_valid,
                                           int rows_total, int tid, int nthreads) {
  constexpr int RB = row_bytes<T, DH>();
  constexpr int CPV = DH * sizeof(T) / 16;                      // 16-B chunks a row has in memory
  constexpr int CPF = ((DH + 31) / 32) * 32 * sizeof(T) / 16;   // chunks the products read (== CPV unless dh = 80)
  constexpr int EPC = 16 / sizeof(T);
  for (int id = tid; id < rows_total * CPF; id += nthreads) {
    const int row = id / CPF, c = id % CPF;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < rows_valid && c < CPV) v = *reinterpret_cast<const u32x4*>(src + (long)row * ld + c * EPC);
    *reinterpret_cast<u32x4*>(img + row * RB + 16 * (c ^ swz<RB>(row))) = v;
  }
}

// Row-read fragment: rows rb..rb+31 on lanes (l&31), k-step kk (chunk 2*kk + h).
template <typename T, int DH>
__device__ __forceinline__ Frag16 frag_rows(const char* img, int rb, int kk, int lane) {
  constexpr int RB = row_bytes<T, DH>();
  const int row = rb + (lane & 31);
  const int c = 2 * kk + (lane >> 5);
  Frag16 f;
  f.u = *reinterpret_cast<const u32x4*>(img + row * RB + 16 * (c ^ swz<RB>(row)));
  return f;
}

// Fragment straight from global memory (B operand: lane (r,h) holds elements of row r).
template <typename T>
__device__ __forceinline__ Frag16 frag_global(const T* __restrict__ rowptr, bool valid, int kk, int lane) {
  constexpr int EPC = 16 / sizeof(T);
  Frag16 f;
  f.u = (u32x4){0u, 0u, 0u, 0u};
  if (valid) f.u = *reinterpret_cast<const u32x4*>(rowptr + (2 * kk + (lane >> 5)) * EPC);
  return f;
}

// acc_out[d][lane] += sum over the 32 rows r of X:  Img[rb + r][d0 + d] * X[r][lane]
// where X is a 32x32 accumulator tile (rows in registers, column on the lane) used as the B operand.
template <typename T, int DH>
__device__ __forceinline__ f32x16 mma_imgT_acc(const char* PM_LDS_IMAGE img, int rb, int d0, const f32x16& x, f32x16 acc, int lane) {
  constexpr int RB = row_bytes<T, DH>();
  if constexpr (sizeof(T) == 2) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, hh = g >> 1;
    const int col = d0 + 16 * (g & 1) + 4 * p;
    const int c = col >> 3, sub = 8 * (p & 1);
    using lds_s4 = __attribute__((address_space(3))) short4v;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Frag16 a, b;
      const int r0 = rb + 16 * s + 4 * hh + q;
      const int r1 = r0 + 8;
      const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + r0 * RB + 16 * (c ^ swz<RB>(r0)) + sub));
      const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + r1 * RB + 16 * (c ^ swz<RB>(r1)) + sub));
      a.u[0] = ((unsigned)(unsigned short)lo[0]) | (((unsigned)(unsigned short)lo[1]) << 16);
      a.u[1] = ((unsigned)(unsigned short)lo[2]) | (((unsigned)(unsigned short)lo[3]) << 16);
      a.u[2] = ((unsigned)(unsigned short)hi[0]) | (((unsigned)(unsigned short)hi[1]) << 16);
      a.u[3] = ((unsigned)(unsigned short)hi[2]) | (((unsigned)(unsigned short)hi[3]) << 16);
      // element j of lane half hh  <->  row 16s + 8(j>>2) + 4hh + (j&3) of X  ==  registers 8s..8s+7 in order
#pragma unroll
      for (int j = 0; j < 8; ++j) frag_set<T>(b, j, x[8 * s + j]);
      acc = mfma16B<T>(a, b, acc);
    }
  } else {
    const int hh = lane >> 5;
    const int col = d0 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rb + acc_row(r, hh);
      const float a = *reinterpret_cast<const float*>(img + row * RB + 16 * ((col >> 2) ^ swz<RB>(row)) + 4 * (col & 3));
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x[r], acc, 0, 0, 0);
    }
  }
  return acc;
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// Store an accumulator tile whose lane is a token row and whose registers are 32 consecutive features (the first `ncols` of
// them exist: 16 for the last tile of an 80-wide head).
template <typename T>
__device__ __forceinline__ void store_tile_T(T* __restrict__ rowptr, bool valid, const f32x16& acc, float scale, int lane,
                                             int ncols = 32) {
  if (!valid) return;
  const int hh = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 v = {acc[4 * g] * scale, acc[4 * g + 1] * scale, acc[4 * g + 2] * scale, acc[4 * g + 3] * scale};
    if (8 * g + 4 * hh < ncols) store4<T>(rowptr + 8 * g + 4 * hh, v);
  }
}
// features of output tile dt that exist
template <int DH> __device__ __forceinline__ constexpr int tile_cols(int dt) { return DH - 32 * dt < 32 ? DH - 32 * dt : 32; }

constexpr float kLog2e = 1.4426950408889634f;

// Where Q / K / V rows of (batch b, head h) live inside the qkv buffer, in elements:
//   base(which, b, h) = which * which_stride + b * batch_stride + h * head_stride, consecutive tokens `ld` apart.
// token-major (timm's reshape of the qkv Linear's output, [B, N, 3, H, dh]): ld = 3 H dh, which = H dh, head = dh, batch = N ld.
// head-major ([3, B, H, N, dh]: a head's rows contiguous, 25 KB at N = 197, dh = 64): ld = dh, which = B H N dh, head = N dh,
// batch = H N dh -- what a qkv GEMM epilogue that scatters by head would write (experiment: PM_ATTN_HEADMAJOR=1 reads the
// buffer that way; scratch/bench_attn_layout.py permutes the input accordingly).
struct QkvLayout {
  long ld, which_stride, head_stride, batch_stride;
};
inline QkvLayout qkv_layout(int B, int N, int H, int DH, bool head_major) {
  if (head_major) return QkvLayout{(long)DH, (long)B * H * N * DH, (long)N * DH, (long)H * N * DH};
  return QkvLayout{3L * H * DH, (long)H * DH, (long)DH, (long)N * 3L * H * DH};
}
inline bool attn_head_major() {
  static const bool v = [] { const char* e = getenv("PM_ATTN_HEADMAJOR"); return e && e[0] == '1'; }();
  return v;
}

// Which (batch, head) problem a workgroup of the one-problem-per-workgroup kernels takes.  dh = 32 (the MAE decoder: 16 heads of 32):
// a head's rows are 64-B pieces of the [B, N, 3, H, dh] rows -- HALF a cache line, the other half belongs to the neighbouring head.
// The hardware deals workgroup ids to the 8 XCDs round-robin, so with the identity map heads 2p and 2p + 1 run on different XCDs and
// each L2 fetches the whole line for its half (PMC, MAE bs = 256: 147.8 MB fetched by the forward against 77.5 MB of qkv, 496 MB by
// the backward against 258 MB -- 1.9x; the dh = 64 kernels fetch 1.0x).  Map the two heads of a pair to workgroups b and b + 8: the
// same XCD, dispatched back to back, so the second finds the lines in that XCD's L2.  Needs an even H and B H % 16 == 0 (else identity).
template <int DH> __device__ __forceinline__ int problem_of_block(int bid, int nprob, int H) {
  if constexpr (DH == 32) {
#ifndef PM_ATTN_NO_PAIR_MAP
    if ((nprob & 15) == 0 && (H & 1) == 0) {
      const int xcd = bid & 7, j = bid >> 3;
      return (((j >> 1) << 3) + xcd) * 2 + (j & 1);
    }
#endif
  }
  return bid;
}

// one wave per 32-row tile (7 waves for N = 197): every wave does identical work, nothing idles on a tail tile
template <int NT> struct Waves { static constexpr int value = NT; };

// Tiles of the OTHER side of the product an LDS image pair holds at a time.  Normally all NT of them (one staging, one barrier).
// f32 rows of an 80-wide head at N = 257 (ViT-H/14 at 224^2) would need 2 x 288 x 384 B = 216 KiB: the images are then staged
// in chunks of CT tiles, the per-wave state (running max / sum and O^T; dQ^T; dK^T and dV^T) living in registers across chunks.
template <typename T, int DH, int NT> constexpr int chunk_tiles() {
  constexpr int per_tile = 2 * 32 * row_bytes<T, DH>() + 2 * 32 * (int)sizeof(float);
  if (NT * per_tile <= 150 * 1024) return NT;
  return (80 * 1024) / per_tile;   // two workgroups per CU
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <typename T, int DH, int NT, int CT>
__global__ __launch_bounds__(Waves<NT>::value * 64) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                                        float* __restrict__ lse, int N, int H,
                                                                        float scale) {
  constexpr int RB = row_bytes<T, DH>();
  constexpr int KS = DH * sizeof(T) / 32;   // 16-B fragment pairs along the head dim
  constexpr int DT = (DH + 31) / 32;        // 32-wide output tiles along the head dim
  constexpr int NW = Waves<NT>::value;
  static_assert(NW == NT, "one wave per query tile: its softmax state stays in registers across key chunks");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* imgK = smem;
  char* imgV = smem + CT * 32 * RB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int prob = problem_of_block<DH>(blockIdx.x, gridDim.x, H);
  const int b = prob / H, h = prob % H;
  const long ld = 3L * H * DH;
  const T* base = qkv + (long)b * N * ld + h * DH;
  const float c = scale * kLog2e;
  const int q = wave * 32 + (lane & 31);
  const bool qv = q < N;
  // online softmax over the key tiles: running max m (shared by the two lane halves of a query), running
  // sum l, O^T rescaled by exp2((m_old - m_new) c) -- only 16 score registers are live at a time.
  Frag16 fq[KS];
  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = zero16();
  float m = -INFINITY, l = 0.f;
  for (int c0 = 0; c0 < NT; c0 += CT) {
    if (c0) __syncthreads();  // every wave is done with the previous chunk's images
    load_image<T, DH>(imgK, base + H * DH + (long)c0 * 32 * ld, ld, N - c0 * 32, CT * 32, tid, NW * 64);
    load_image<T, DH>(imgV, base + 2 * H * DH + (long)c0 * 32 * ld, ld, N - c0 * 32, CT * 32, tid, NW * 64);
    __syncthreads();
    if (c0 == 0) {
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) fq[kk] = frag_global<T>(base + (long)q * ld, qv, kk, lane);
    }
    const int cend = c0 + CT < NT ? c0 + CT : NT;
#pragma unroll 1
    for (int kt = c0; kt < cend; ++kt) {
      const int rb = (kt - c0) * 32;
      f32x16 s = zero16();
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) s = mfma16B<T>(frag_rows<T, DH>(imgK, rb, kk, lane), fq[kk], s);
      float tm = -INFINITY;
      if (kt * 32 + 32 > N) {  // a tile with padded keys (zero rows of the K image; N = 100 on seven tiles has four of them): mask
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + acc_row(r, hh);
          s[r] = key < N ? s[r] : -INFINITY;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) tm = fmaxf(tm, s[r]);
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
      const float mn = fmaxf(m, tm);  // finite from the first tile on (key 0 is always valid)
      const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);
      const float mnc = mn * c;
      float ts = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] * c - mnc);
        ts += s[r];
      }
      l = l * alpha + ts;
      const bool moved = __builtin_amdgcn_ballot_w64(mn != m) != 0;  // after the first tiles the running max rarely moves
      m = mn;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        if (moved) {
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        o[dt] = mma_imgT_acc<T, DH>(imgV, rb, dt * 32, s, o[dt], lane);
      }
    }
  }
  l += __shfl_xor(l, 32, 64);
  if (qv && hh == 0) lse[((long)b * H + h) * N + q] = m * scale + __logf(l);
  const float inv = 1.0f / l;
  T* orow = out + ((long)b * N + q) * H * DH + h * DH;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) store_tile_T<T>(orow + dt * 32, qv, o[dt], inv, lane, tile_cols<DH>(dt));
}

// ------------------------------------------------------------------------------------------------
// backward, query side: dQ (and delta = rowsum(dO * O), written for the key-side kernel)
// ------------------------------------------------------------------------------------------------
template <typename T, int DH, int NT, int CT>
__global__ __launch_bounds__(Waves<NT>::value * 64) void attn_bwd_q_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                                          const T* __restrict__ dout,
                                                                          const float* __restrict__ lse,
                                                                          float* __restrict__ delta, T* __restrict__ dqkv,
                                                                          int N, int H, float scale) {
  constexpr int RB = row_bytes<T, DH>();
  constexpr int KS = DH * sizeof(T) / 32;
  constexpr int DT = (DH + 31) / 32;
  constexpr int NW = Waves<NT>::value;
  constexpr int EPC = 16 / sizeof(T);
  static_assert(NW == NT, "one wave per query tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* imgK = smem;
  char* imgV = smem + CT * 32 * RB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int prob = problem_of_block<DH>(blockIdx.x, gridDim.x, H);
  const int b = prob / H, h = prob % H;
  const long ld = 3L * H * DH, ldo = (long)H * DH;
  const T* base = qkv + (long)b * N * ld + h * DH;
  const float c = scale * kLog2e;
  const int q = wave * 32 + (lane & 31);
  const bool qv = q < N;
  const long sidx = ((long)b * H + h) * N + q;
  Frag16 fq[KS], fdo[KS];
  float l2 = 0.f, dls = 0.f;
  f32x16 dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) dq[dt] = zero16();
  for (int c0 = 0; c0 < NT; c0 += CT) {
    if (c0) __syncthreads();
    load_image<T, DH>(imgK, base + H * DH + (long)c0 * 32 * ld, ld, N - c0 * 32, CT * 32, tid, NW * 64);
    load_image<T, DH>(imgV, base + 2 * H * DH + (long)c0 * 32 * ld, ld, N - c0 * 32, CT * 32, tid, NW * 64);
    __syncthreads();
    if (c0 == 0) {
      const T* orow = out + ((long)b * N + q) * ldo + h * DH;
      const T* dorow = dout + ((long)b * N + q) * ldo + h * DH;
      float dl = 0.f;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        fq[kk] = frag_global<T>(base + (long)q * ld, qv, kk, lane);
        fdo[kk] = frag_global<T>(dorow, qv, kk, lane);
        const Frag16 fo = frag_global<T>(orow, qv, kk, lane);
#pragma unroll
        for (int e = 0; e < EPC; ++e) dl += frag_get<T>(fdo[kk], e) * frag_get<T>(fo, e);
      }
      dl += __shfl_xor(dl, 32, 64);
      if (qv && hh == 0) delta[sidx] = dl;
      l2 = qv ? lse[sidx] * kLog2e : 0.f;
      dls = dl * scale;
    }
    const int cend = c0 + CT < NT ? c0 + CT : NT;
    // dh = 32: two key tiles in flight per wave (the loop is a latency chain; registers allow it: 80 VGPRs)
#pragma unroll kTileUnroll<DH>
    for (int kt = c0; kt < cend; ++kt) {
      const int rb = (kt - c0) * 32;
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        s = mfma16B<T>(frag_rows<T, DH>(imgK, rb, kk, lane), fq[kk], s);
        dp = mfma16B<T>(frag_rows<T, DH>(imgV, rb, kk, lane), fdo[kk], dp);
      }
      // no masking: a padded key has a zero K row, so its dS is multiplied by zeros in the dQ product below -- PROVIDED it
      // is finite: its score is 0, so P = exp2(0 - l2) overflows once the row's log-sum-exp drops below -88 (every real
      // score strongly negative: seen after ~60 steps of the head+2 fine-tune at lr 1e-3), and inf x 0 = NaN in dQ.
      // P <= 1 for every real key (lse >= the row maximum), so the exponent is clamped at 0: exact for real keys, finite
      // for padded ones.  A padded query's lane is never stored.
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fminf(s[r] * c - l2, 0.f)) * (dp[r] * scale - dls);  // dS^T
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dq[dt] = mma_imgT_acc<T, DH>(imgK, rb, dt * 32, s, dq[dt], lane);
    }
  }
  T* dqrow = dqkv + ((long)b * N + q) * ld + h * DH;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) store_tile_T<T>(dqrow + dt * 32, qv, dq[dt], 1.0f, lane, tile_cols<DH>(dt));
}

// ------------------------------------------------------------------------------------------------
// backward, key side: dK and dV
// ------------------------------------------------------------------------------------------------
template <typename T, int DH, int NT, int CT>
__global__ __launch_bounds__(Waves<NT>::value * 64) void attn_bwd_kv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                                           const float* __restrict__ lse,
                                                                           const float* __restrict__ delta,
                                                                           T* __restrict__ dqkv, int N, int H, float scale) {
  constexpr int RB = row_bytes<T, DH>();
  constexpr int KS = DH * sizeof(T) / 32;
  constexpr int DT = (DH + 31) / 32;
  constexpr int NW = Waves<NT>::value;
  static_assert(NW == NT, "one wave per key tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* imgQ = smem;
  char* imgDO = smem + CT * 32 * RB;
  float* sl2 = reinterpret_cast<float*>(smem + 2 * CT * 32 * RB);  // lse * log2e per query of the chunk
  float* sdl = sl2 + CT * 32;                                      // delta * scale per query of the chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int prob = problem_of_block<DH>(blockIdx.x, gridDim.x, H);
  const int b = prob / H, h = prob % H;
  const long ld = 3L * H * DH, ldo = (long)H * DH;
  const T* base = qkv + (long)b * N * ld + h * DH;
  const float c = scale * kLog2e;
  const int key = wave * 32 + (lane & 31);
  const bool kv = key < N;
  Frag16 fk[KS], fv[KS];
  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    dk[dt] = zero16();
    dv[dt] = zero16();
  }
  for (int c0 = 0; c0 < NT; c0 += CT) {
    if (c0) __syncthreads();
    load_image<T, DH>(imgQ, base + (long)c0 * 32 * ld, ld, N - c0 * 32, CT * 32, tid, NW * 64);
    load_image<T, DH>(imgDO, dout + ((long)b * N + c0 * 32) * ldo + h * DH, ldo, N - c0 * 32, CT * 32, tid, NW * 64);
    for (int i = tid; i < CT * 32; i += NW * 64) {
      const int qi = c0 * 32 + i;
      const long sidx = ((long)b * H + h) * N + qi;
      sl2[i] = qi < N ? lse[sidx] * kLog2e : 0.f;
      sdl[i] = qi < N ? delta[sidx] * scale : 0.f;
    }
    __syncthreads();
    if (c0 == 0) {
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        fk[kk] = frag_global<T>(base + (long)key * ld + H * DH, kv, kk, lane);
        fv[kk] = frag_global<T>(base + (long)key * ld + 2 * H * DH, kv, kk, lane);
      }
    }
    const int cend = c0 + CT < NT ? c0 + CT : NT;
#pragma unroll kTileUnroll<DH>
    for (int qt = c0; qt < cend; ++qt) {
      const int rb = (qt - c0) * 32;
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        s = mfma16B<T>(frag_rows<T, DH>(imgQ, rb, kk, lane), fk[kk], s);
        dp = mfma16B<T>(frag_rows<T, DH>(imgDO, rb, kk, lane), fv[kk], dp);
      }
      // no masking: a padded query has zero Q and dO rows (and lse = delta = 0 in LDS), so its finite P / dS rows
      // meet zeros in both products below; a padded key's lane is never stored.  sdl holds delta * scale.
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 are queries qt*32 + 8g + 4hh + 0..3: one 16-B LDS read each
        const int qq = rb + 8 * g + 4 * hh;
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sl2 + qq);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(sdl + qq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float p = __builtin_amdgcn_exp2f(fminf(s[r] * c - l4[e], 0.f));
          s[r] = p;                                // P
          dp[r] = p * (dp[r] * scale - d4[e]);     // dS
        }
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dv[dt] = mma_imgT_acc<T, DH>(imgDO, rb, dt * 32, s, dv[dt], lane);
        dk[dt] = mma_imgT_acc<T, DH>(imgQ, rb, dt * 32, dp, dk[dt], lane);
      }
    }
  }
  T* drow = dqkv + ((long)b * N + key) * ld + h * DH;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    store_tile_T<T>(drow + H * DH + dt * 32, kv, dk[dt], 1.0f, lane, tile_cols<DH>(dt));
    store_tile_T<T>(drow + 2 * H * DH + dt * 32, kv, dv[dt], 1.0f, lane, tile_cols<DH>(dt));
  }
}

// ------------------------------------------------------------------------------------------------
// forward, persistent form (bf16, 193 <= N <= 224: the ViT-B/16 encoder at 224^2 and the MAE decoder)
// ------------------------------------------------------------------------------------------------
// The one-block-per-head kernel above spends 43 % of its wave cycles parked behind the K/V load of its head (PMC,
// DESIGN.md).  Here a workgroup walks several (batch, head) problems: K, V and Q of the NEXT head arrive by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave instruction, swizzle applied to the source address) into the other half of a
// double buffer while the current head computes, so only the first head of a workgroup sees its load latency.
// One wave per 32-query tile holds the whole score row block in registers (7 x 16 accumulators): no online rescaling,
// one max / one sum per row; S^T = K Q^T for all key tiles first (28 MFMAs back to back), then exp and P V per key tile.
template <int RB, typename T> __device__ __forceinline__ void dma_rows(char* img, const T* __restrict__ base, long ld, int rows_valid,
                                                                        int piece, int lane) {
  static_assert(sizeof(T) == 2, "16-bit rows");
  // piece = one wave instruction = 1 KiB of the image = 1024 / RB rows; lane -> (row, slot); source chunk = slot ^ swz(row)
  constexpr int CPR = RB / 16, RPP = 1024 / RB;
  const int row = piece * RPP + lane / CPR, cs = lane % CPR;
  const int c = cs ^ swz<RB>(row);
  const int gr = row < rows_valid ? row : rows_valid - 1;  // rows beyond N: a finite duplicate (masked / never stored)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (long)gr * ld + c * 8),
                                   (__attribute__((address_space(3))) void*)(img + 1024 * piece), 16, 0, 0);
}

template <typename T, int DH, int NT>
__global__ __launch_bounds__(NT * 64) void attn_fwd2_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                           float* __restrict__ lse, int N, int H, int BH, float scale,
                                                           QkvLayout lay) {
  constexpr int RB = DH * 2;
  constexpr int IMG = NT * 32 * RB;  // bytes per image
  constexpr int PCS = IMG / 1024;    // DMA pieces per image
  constexpr int KS = RB / 32;
  constexpr int DT = DH / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [K0 | V0 | K1 | V1 | Q]
  char* imgQ = smem + 4 * IMG;
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long ld = lay.ld;
  const float c = scale * kLog2e;
  auto issue = [&](int head, int buf) {
    const int b = head / H, h = head % H;
    const T* base = qkv + (long)b * lay.batch_stride + (long)h * lay.head_stride;
    char* imgK = smem + buf * 2 * IMG;
#pragma unroll
    for (int i = 0; i < (PCS + NT - 1) / NT; ++i) {
      const int p = wave + NT * i;
      if (p < PCS) {
        dma_rows<RB>(imgK, base + lay.which_stride, ld, N, p, lane);
        dma_rows<RB>(imgK + IMG, base + 2 * lay.which_stride, ld, N, p, lane);
        dma_rows<RB>(imgQ, base, ld, N, p, lane);
      }
    }
  };
  int head = blockIdx.x;
  if (head >= BH) return;
  issue(head, 0);
  for (int it = 0; head < BH; head += gridDim.x, ++it) {
    const int buf = it & 1;
    const char* imgK = smem + buf * 2 * IMG;
    const char* imgV = imgK + IMG;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the current head have landed
    __syncthreads();                                    // ... and everybody else's
    Frag16 fq[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) fq[kk] = frag_rows<T, DH>(imgQ, wave * 32, kk, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // every wave holds its Q fragments: the Q image and the other K/V buffer may be refilled
    const int next = head + gridDim.x;
    if (next < BH) issue(next, buf ^ 1);
    const int b = head / H, h = head % H;
    // ---- S^T = K Q^T for every key tile
    f32x16 s[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      s[kt] = zero16();
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) s[kt] = mfma16B<T>(frag_rows<T, DH>(imgK, kt * 32, kk, lane), fq[kk], s[kt]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // only the last key tile holds padded keys
      const int key = (NT - 1) * 32 + acc_row(r, hh);
      s[NT - 1][r] = key < N ? s[NT - 1][r] : -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) m = fmaxf(m, fmaxf(s[kt][r], s[kt][r + 1]));  // -> v_max3_f32
    m = fmaxf(m, __shfl_xor(m, 32, 64));  // the two lane halves of a query hold different keys
    const float mc = m * c;
    // ---- P = exp2(S c - m c), row sums, O^T += V^T P^T per key tile
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = zero16();
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] * c - mc);
        l += s[kt][r];
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[dt] = mma_imgT_acc<T, DH>(imgV, kt * 32, dt * 32, s[kt], o[dt], lane);
    }
    l += __shfl_xor(l, 32, 64);
    const int q = wave * 32 + (lane & 31);
    const bool qv = q < N;
    if (qv && hh == 0) lse[((long)b * H + h) * N + q] = m * scale + __logf(l);
    const float inv = 1.0f / l;
    T* orow = out + ((long)b * N + q) * H * DH + h * DH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) store_tile_T<T>(orow + dt * 32, qv, o[dt], inv, lane);
  }
}

template <typename T, int DH>
int launch_fwd2(const void* qkv, void* out, float* lse, int B, int N, int H, hipStream_t s) {
  constexpr int NT = 7;
  const int BH = B * H;
  constexpr int SLOTS = DH == 32 ? 512 : 256;   // workgroups the chip holds at once (72 KiB of LDS at dh = 32: two per CU)
  const int rounds = (BH + SLOTS - 1) / SLOTS;  // heads per workgroup ...
  const int grid = (BH + rounds - 1) / rounds;  // ... spread evenly: no workgroup walks one head more than another
  constexpr size_t lds = 5 * NT * 32 * DH * 2;
  const float scale = 1.0f / sqrtf((float)DH);
  auto kern = attn_fwd2_kernel<T, DH, NT>;
  PM_ALLOW_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT * 64), lds, s, (const T*)qkv, (T*)out, lse, N, H, BH, scale,
                     qkv_layout(B, N, H, DH, attn_head_major()));
  return pm_check_launch();
}

template <typename T, int DH, int NT>
int launch_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, hipStream_t s) {
  constexpr int NW = Waves<NT>::value;
  constexpr int CT = chunk_tiles<T, DH, NT>();
  const size_t lds = 2 * CT * 32 * row_bytes<T, DH>();
  const float scale = 1.0f / sqrtf((float)DH);
  auto kern = attn_fwd_kernel<T, DH, NT, CT>;
  PM_ALLOW_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(NW * 64), lds, s, (const T*)qkv, (T*)out, lse, N, H, scale);
  return pm_check_launch();
}

// ------------------------------------------------------------------------------------------------
// backward, fused (bf16): dQ, dK and dV of one (batch, head) from ONE set of LDS images
// ------------------------------------------------------------------------------------------------
// The two kernels above each stage half of the head (K, V | Q, dO) and fetch the other half as per-lane fragments from
// global memory, so every operand crosses HBM twice (231 MB per ViT-B launch against 155 MB algorithmic) and both spend
// half of their wave cycles parked behind those loads (PMC: 44-51 % s_waitcnt).  Here one workgroup stages
// Q, K, V and dO of its head once, by LDS-DMA (pad rows zeroed afterwards, which keeps the no-masking argument of the
// split kernels), and runs both passes out of LDS:
//   pass 1 (wave = query tile): delta = rowsum(dO O); S^T, dP^T -> dS^T -> dQ^T += K^T dS^T
//   pass 2 (wave = key tile):   S, dP -> P, dS -> dV^T += dO^T P, dK^T += Q^T dS
// S and dP are still recomputed in pass 2 (no cross-wave reduction, no atomics: deterministic); what disappears is the
// second trip of qkv / dO through HBM, the per-lane global fragment loads and the delta round trip.
template <typename T, int DH, int NT>
__global__ __launch_bounds__(NT * 64) void attn_bwd_fused_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                                const T* __restrict__ dout,
                                                                const float* __restrict__ lse, T* __restrict__ dqkv,
                                                                int N, int H, float scale, QkvLayout lay
#ifdef PM_ATTN_DEBUG
                                                                , int dbg  // diagnostic build: bit 0 skips pass 1, bit 1 pass 2
#endif
                                                                ) {
#ifndef PM_ATTN_DEBUG
  constexpr int dbg = 0;
#endif
  constexpr int RB = DH * 2;
  constexpr int IMG = NT * 32 * RB;
  constexpr int PCS = IMG / 1024;
  constexpr int KS = RB / 32;
  constexpr int DT = DH / 32;
  constexpr int CPR = RB / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [Q | K | V | dO | lse*log2e | delta*scale]
  char* imgQ = smem;
  char* imgK = smem + IMG;
  char* imgV = smem + 2 * IMG;
  char* imgDO = smem + 3 * IMG;
  float* sl2 = reinterpret_cast<float*>(smem + 4 * IMG);
  float* sdl = sl2 + NT * 32;
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int prob = problem_of_block<DH>(blockIdx.x, gridDim.x, H);
  const int b = prob / H, h = prob % H;
  const long ld = 3L * H * DH, ldo = (long)H * DH;   // (dqkv, out and dout stay token-major)
  const T* base = qkv + (long)b * lay.batch_stride + (long)h * lay.head_stride;
  const T* dobase = dout + (long)b * N * ldo + h * DH;
#pragma unroll
  for (int i = 0; i < (PCS + NT - 1) / NT; ++i) {
    const int p = wave + NT * i;
    if (p < PCS) {
      dma_rows<RB>(imgQ, base, lay.ld, N, p, lane);
      dma_rows<RB>(imgK, base + lay.which_stride, lay.ld, N, p, lane);
      dma_rows<RB>(imgV, base + 2 * lay.which_stride, lay.ld, N, p, lane);
      dma_rows<RB>(imgDO, dobase, ldo, N, p, lane);
    }
  }
  // while the images fly: this wave's O rows (only delta needs them) and the log-sum-exp of the head
  const int q = wave * 32 + (lane & 31);
  const bool qv = q < N;
  const long sidx = ((long)b * H + h) * N;
  Frag16 fo[KS];
  {
    const T* orow = out + ((long)b * N + (qv ? q : N - 1)) * ldo + h * DH;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) fo[kk].u = *reinterpret_cast<const u32x4*>(orow + (2 * kk + hh) * 8);
  }
  for (int i = tid; i < NT * 32; i += NT * 64) sl2[i] = i < N ? lse[sidx + i] * kLog2e : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // every wave's pieces have landed (a pad row may belong to another wave's piece)
  // zero the pad rows (the DMA filled them with a duplicate of the last valid row): a padded key / query then meets
  // zeros in every product it enters, exactly as in the split kernels
  {
    const int pad0 = N, npad = NT * 32 - N;
    for (int i = tid; i < npad * CPR; i += NT * 64) {
      const int off = (pad0 + i / CPR) * RB + 16 * (i % CPR);
      const u32x4 z = {0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(imgQ + off) = z;
      *reinterpret_cast<u32x4*>(imgK + off) = z;
      *reinterpret_cast<u32x4*>(imgV + off) = z;
      *reinterpret_cast<u32x4*>(imgDO + off) = z;
    }
  }
  __syncthreads();
  const float c = scale * kLog2e;
  // ---- pass 1: query side
  Frag16 fq[KS], fdo[KS];
  float dl = 0.f;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    fq[kk] = frag_rows<T, DH>(imgQ, wave * 32, kk, lane);
    fdo[kk] = frag_rows<T, DH>(imgDO, wave * 32, kk, lane);
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += frag_get<T>(fdo[kk], e) * frag_get<T>(fo[kk], e);
  }
  dl += __shfl_xor(dl, 32, 64);
  if (!qv) dl = 0.f;
  if (hh == 0) sdl[q] = dl * scale;
  {
    const float l2 = sl2[q];
    const float dls = dl * scale;
    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] = zero16();
#pragma unroll kTileUnroll<DH>
    for (int kt = 0; kt < ((dbg & 1) ? 0 : NT); ++kt) {
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        s = mfma16B<T>(frag_rows<T, DH>(imgK, kt * 32, kk, lane), fq[kk], s);
        dp = mfma16B<T>(frag_rows<T, DH>(imgV, kt * 32, kk, lane), fdo[kk], dp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fminf(s[r] * c - l2, 0.f)) * (dp[r] * scale - dls);  // dS^T
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dq[dt] = mma_imgT_acc<T, DH>(imgK, kt * 32, dt * 32, s, dq[dt], lane);
    }
    T* dqrow = dqkv + ((long)b * N + q) * ld + h * DH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) store_tile_T<T>(dqrow + dt * 32, qv, dq[dt], 1.0f, lane);
  }
  __syncthreads();  // every query tile's delta is in LDS
  // ---- pass 2: key side (this wave's key tile = its query tile index)
  {
    const int key = q;
    const bool kv = qv;
    Frag16 fk[KS], fv[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      fk[kk] = frag_rows<T, DH>(imgK, wave * 32, kk, lane);
      fv[kk] = frag_rows<T, DH>(imgV, wave * 32, kk, lane);
    }
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      dk[dt] = zero16();
      dv[dt] = zero16();
    }
#pragma unroll kTileUnroll<DH>
    for (int qt = 0; qt < ((dbg & 2) ? 0 : NT); ++qt) {
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        s = mfma16B<T>(frag_rows<T, DH>(imgQ, qt * 32, kk, lane), fk[kk], s);
        dp = mfma16B<T>(frag_rows<T, DH>(imgDO, qt * 32, kk, lane), fv[kk], dp);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int qq = qt * 32 + 8 * g + 4 * hh;
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sl2 + qq);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(sdl + qq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float pr = __builtin_amdgcn_exp2f(fminf(s[r] * c - l4[e], 0.f));
          s[r] = pr;                               // P
          dp[r] = pr * (dp[r] * scale - d4[e]);    // dS
        }
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dv[dt] = mma_imgT_acc<T, DH>(imgDO, qt * 32, dt * 32, s, dv[dt], lane);
        dk[dt] = mma_imgT_acc<T, DH>(imgQ, qt * 32, dt * 32, dp, dk[dt], lane);
      }
    }
    T* drow = dqkv + ((long)b * N + key) * ld + h * DH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      store_tile_T<T>(drow + H * DH + dt * 32, kv, dk[dt], 1.0f, lane);
      store_tile_T<T>(drow + 2 * H * DH + dt * 32, kv, dv[dt], 1.0f, lane);
    }
  }
}

template <typename T, int DH, int NT>
int launch_bwd_fused(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int N, int H,
                     hipStream_t s) {
  const size_t lds = 4 * NT * 32 * DH * 2 + 2 * NT * 32 * sizeof(float);
  const float scale = 1.0f / sqrtf((float)DH);
  auto kern = attn_bwd_fused_kernel<T, DH, NT>;
  PM_ALLOW_LDS(kern, lds);
#ifdef PM_ATTN_DEBUG
  static const int dbg = [] { const char* e = getenv("PM_ATTN_BWD_SKIP"); return e ? atoi(e) : 0; }();
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(NT * 64), lds, s, (const T*)qkv, (const T*)out, (const T*)dout, lse,
                     (T*)dqkv, N, H, scale, qkv_layout(B, N, H, DH, attn_head_major()), dbg);
#else
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(NT * 64), lds, s, (const T*)qkv, (const T*)out, (const T*)dout, lse,
                     (T*)dqkv, N, H, scale, qkv_layout(B, N, H, DH, attn_head_major()));
#endif
  return pm_check_launch();
}

template <typename T, int DH>
int dispatch_bwd_fused(int nt, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int N,
                       int H, hipStream_t s) {
  if (nt <= 1) return launch_bwd_fused<T, DH, 1>(qkv, out, dout, lse, dqkv, B, N, H, s);
  if (nt <= 2) return launch_bwd_fused<T, DH, 2>(qkv, out, dout, lse, dqkv, B, N, H, s);
  if (nt <= 7) return launch_bwd_fused<T, DH, 7>(qkv, out, dout, lse, dqkv, B, N, H, s);
  return launch_bwd_fused<T, DH, 9>(qkv, out, dout, lse, dqkv, B, N, H, s);   // N = 257: patch 14 at 224^2 (4 x 36 KiB images)
}

template <typename T, int DH, int NT>
int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B,
               int N, int H, hipStream_t s) {
  constexpr int NW = Waves<NT>::value;
  constexpr int CT = chunk_tiles<T, DH, NT>();
  const size_t lds = 2 * CT * 32 * row_bytes<T, DH>();
  const size_t lds_kv = lds + 2 * CT * 32 * sizeof(float);
  const float scale = 1.0f / sqrtf((float)DH);
  auto kq = attn_bwd_q_kernel<T, DH, NT, CT>;
  auto kkv = attn_bwd_kv_kernel<T, DH, NT, CT>;
  PM_ALLOW_LDS(kq, lds);
  PM_ALLOW_LDS(kkv, lds_kv);
  hipLaunchKernelGGL(kq, dim3(B * H), dim3(NW * 64), lds, s, (const T*)qkv, (const T*)out, (const T*)dout, lse, delta,
                     (T*)dqkv, N, H, scale);
  hipLaunchKernelGGL(kkv, dim3(B * H), dim3(NW * 64), lds_kv, s, (const T*)qkv, (const T*)dout, lse,
                     (const float*)delta, (T*)dqkv, N, H, scale);
  return pm_check_launch();
}

template <typename T, int DH>
int dispatch_fwd(int nt, const void* qkv, void* out, float* lse, int B, int N, int H, hipStream_t s) {
  if constexpr (DH == 80) {  // ViT-H: N = 65 under MAE masking, 257 unmasked
    if (nt <= 3) return launch_fwd<T, DH, 3>(qkv, out, lse, B, N, H, s);
    return launch_fwd<T, DH, 9>(qkv, out, lse, B, N, H, s);
  } else {
    if (nt <= 1) return launch_fwd<T, DH, 1>(qkv, out, lse, B, N, H, s);
    if (nt <= 2) return launch_fwd<T, DH, 2>(qkv, out, lse, B, N, H, s);
    if (nt <= 7) return launch_fwd<T, DH, 7>(qkv, out, lse, B, N, H, s);
    return launch_fwd<T, DH, 9>(qkv, out, lse, B, N, H, s);
  }
}
template <typename T, int DH>
int dispatch_bwd(int nt, const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                 int B, int N, int H, hipStream_t s) {
  if constexpr (DH == 80) {
    if (nt <= 3) return launch_bwd<T, DH, 3>(qkv, out, dout, lse, delta, dqkv, B, N, H, s);
    return launch_bwd<T, DH, 9>(qkv, out, dout, lse, delta, dqkv, B, N, H, s);
  } else {
    if (nt <= 1) return launch_bwd<T, DH, 1>(qkv, out, dout, lse, delta, dqkv, B, N, H, s);
    if (nt <= 2) return launch_bwd<T, DH, 2>(qkv, out, dout, lse, delta, dqkv, B, N, H, s);
    if (nt <= 7) return launch_bwd<T, DH, 7>(qkv, out, dout, lse, delta, dqkv, B, N, H, s);
    return launch_bwd<T, DH, 9>(qkv, out, dout, lse, delta, dqkv, B, N, H, s);
  }
}

// A/B switch for the tuning scripts (PM_ATTN_V1=1: the one-block-per-head kernels everywhere); read once.
bool attn_v1() {
  static const bool v = [] { const char* e = getenv("PM_ATTN_V1"); return e && e[0] == '1'; }();
  return v;
}
bool attn_fwd2_dh32() {  // the persistent forward for the 32-wide MAE decoder heads too (measured slower: 80 vs 71 us)
  static const bool v = [] { const char* e = getenv("PM_ATTN_FWD2_DH32"); return e && e[0] == '1'; }();
  return v;
}

inline int check_shape(int B, int N, int H, int dh, int dtype) {
  if (B <= 0 || N <= 0 || H <= 0) return PM_ESHAPE;
  if (N > 288) return PM_ESHAPE;
  if (dh != 32 && dh != 64 && dh != 80) return PM_ESHAPE;
  if (dtype != PM_BF16 && dtype != PM_F16 && dtype != PM_F32) return PM_EINVAL;
  return PM_OK;
}

}  // namespace

extern "C" int pm_attention_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int dh, int dtype,
                                void* stream) {
  if (!qkv || !out || !lse) return PM_EINVAL;
  const int st = check_shape(B, N, H, dh, dtype);
  if (st) return st;
  const int nt = (N + 31) / 32;
  hipStream_t s = pm_stream(stream);
  if (dtype != PM_F32) {
    // persistent double-buffered form for the ViT-B encoder heads (dh = 64, N = 197: 26 vs 28 us); the 32-wide MAE decoder
    // heads are faster one block per head, two blocks per CU (70 vs 77 us) -- both are bound by the strided qkv reads
    PM_DISPATCH_16(dtype, T, {
      if (nt == 7 && dh == 64 && !attn_v1()) return launch_fwd2<T, 64>(qkv, out, lse, B, N, H, s);
      if (nt == 7 && dh == 32 && attn_fwd2_dh32()) return launch_fwd2<T, 32>(qkv, out, lse, B, N, H, s);
      if (dh == 80) return dispatch_fwd<T, 80>(nt, qkv, out, lse, B, N, H, s);
      return dh == 64 ? dispatch_fwd<T, 64>(nt, qkv, out, lse, B, N, H, s) : dispatch_fwd<T, 32>(nt, qkv, out, lse, B, N, H, s);
    });
  }
  if (dh == 80) return dispatch_fwd<float, 80>(nt, qkv, out, lse, B, N, H, s);
  return dh == 64 ? dispatch_fwd<float, 64>(nt, qkv, out, lse, B, N, H, s)
                  : dispatch_fwd<float, 32>(nt, qkv, out, lse, B, N, H, s);
}

extern "C" int pm_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                                void* dqkv, int B, int N, int H, int dh, int dtype, void* stream) {
  if (!qkv || !out || !dout || !lse || !delta || !dqkv) return PM_EINVAL;
  const int st = check_shape(B, N, H, dh, dtype);
  if (st) return st;
  const int nt = (N + 31) / 32;
  hipStream_t s = pm_stream(stream);
  if (dtype != PM_F32) {
    PM_DISPATCH_16(dtype, T, {
      // 80-wide heads (ViT-H): the two-kernel form (the fused kernel's four images of 288 x 256 B would not fit the LDS)
      if (dh == 80) return dispatch_bwd<T, 80>(nt, qkv, out, dout, lse, delta, dqkv, B, N, H, s);
      if (!attn_v1())
        return dh == 64 ? dispatch_bwd_fused<T, 64>(nt, qkv, out, dout, lse, dqkv, B, N, H, s)
                        : dispatch_bwd_fused<T, 32>(nt, qkv, out, dout, lse, dqkv, B, N, H, s);
      return dh == 64 ? dispatch_bwd<T, 64>(nt, qkv, out, dout, lse, delta, dqkv, B, N, H, s)
                      : dispatch_bwd<T, 32>(nt, qkv, out, dout, lse, delta, dqkv, B, N, H, s);
    });
  }
  if (dh == 80) return dispatch_bwd<float, 80>(nt, qkv, out, dout, lse, delta, dqkv, B, N, H, s);
  return dh == 64 ? dispatch_bwd<float, 64>(nt, qkv, out, dout, lse, delta, dqkv, B, N, H, s)
                  : dispatch_bwd<float, 32>(nt, qkv, out, dout, lse, delta, dqkv, B, N, H, s);
}
