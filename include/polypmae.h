/* polypmae.h -- C-ABI of the MI355X (gfx950) hot path for SSL4POLYP's MAE pre-train /
 * ViT-B/16 fine-tune step.
 *
 * The reference (irconde/SSL4POLYP) has no FFI of its own: its hot path is the torch/timm op
 * sequence inside MaskedAutoencoderViT.forward (src/ssl4polyp/models/mae/models_mae.py:150-220),
 * ViT_from_MAE.forward / VisionTransformer_from_Any.forward (src/ssl4polyp/models/models.py:117-140,
 * 196-222) and autograd's backward of them (train_classification.py:4531-4533,
 * mae/engine_pretrain.py:52-65).  Each entry point below names the reference op(s) it replaces.
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory owned by the caller; the library never
 *     allocates, frees, retains pointers or synchronises; every launch goes to `stream`
 *     (a hipStream_t passed as void*).
 *   - return 0 on success, negative pm_status otherwise; no C++ exceptions cross the ABI.
 *   - dtype codes: PM_F32 = 0, PM_BF16 = 1, PM_F16 = 2.  "act" tensors (activations between kernels) use the
 *     precision mode's activation type: bf16 in PM_BF16 mode, IEEE half in PM_F16 mode, f32 in PM_F32 mode.
 *     PM_F16 is the reference's own AMP arithmetic (torch.cuda.amp.autocast runs every matmul in fp16 with f32
 *     accumulation: train_classification.py:4527-4546, engine_pretrain.py:52): same kernels, same MFMA rate
 *     (v_mfma_f32_32x32x16_f16), 11 significant bits per operand instead of 8.  The backward operands are fp16 too
 *     (an MFMA takes one type for both operands), so the caller scales the loss as the reference's GradScaler does;
 *     every backward entry point is linear in its incoming gradient and passes inf / nan through.
 *   - matrices are row-major with explicit leading dimensions in ELEMENTS.
 */
#ifndef POLYPMAE_H
#define POLYPMAE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum pm_status { PM_OK = 0, PM_EINVAL = -1, PM_ESHAPE = -2, PM_EARCH = -3, PM_ELAUNCH = -4, PM_EALIGN = -5 };
enum pm_dtype { PM_F32 = 0, PM_BF16 = 1, PM_F16 = 2 };

/* epilogue selector of pm_gemm (what happens to acc = op(A)*op(B) before it is stored) */
enum pm_epilogue {
  PM_EPI_STORE = 0,      /* C = acc (+bias)                                          (any Linear) */
  PM_EPI_GELU = 1,       /* aux = acc+bias (pre-activation; aux NULL: not stored), C = gelu_erf(acc+bias)  (timm Mlp.fc1 + act) */
  PM_EPI_RESIDUAL = 2,   /* C_f32 = resid_f32 + acc + bias                           (Block residual adds) */
  PM_EPI_DGELU = 3,      /* C = acc * gelu_erf'(aux)                                 (backward of Mlp.act) */
  PM_EPI_ACCUM = 4       /* C_f32 += acc                                             (grad accumulation) */
};

const char* pm_strerror(int status);
int pm_abi_version(void);

/* LayerNorm(eps) over the last dim -- replaces nn.LayerNorm in timm Block.norm1/norm2,
 * MaskedAutoencoderViT.norm / decoder_norm (models_mae.py:42,57,168,188).
 * x: f32 [M, D] with row stride ldx; y: out_dtype [M, D] contiguous; mean/rstd: f32 [M] (saved for backward).
 * D <= 1280 (ViT-H), D % 4 == 0. */
int pm_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, void* y, int out_dtype,
                     float* mean, float* rstd, int M, int D, float eps, void* stream);

/* Backward of the above fused with the residual-gradient add of the pre-LN block
 * (x_out = x + f(LN(x)) => dx = dres + LN'(dy)):
 *   dx_f32[M,D] (ld ldx) = (dres ? dres : 0) + LN_bwd(dy);  dx_act (optional) = cast(dx_f32)
 *   dgamma += sum_m dy*xhat ; dbeta += sum_m dy ; dcolsum (optional) += sum_m dx   (bias grad of the
 *   Linear whose output was added into this residual stream).  The three accumulators must be
 *   zero-initialised (or hold the running gradient) by the caller.  `workspace` (>= 512*3*D*4 bytes, optional):
 *   per-block partial column sums, reduced in a fixed order by a second launch; without it the sums use
 *   float atomics (order-dependent rounding). */
int pm_layernorm_bwd(const void* dy, int dy_dtype, const float* x, long ldx, const float* gamma, const float* mean,
                     const float* rstd, const float* dres, long lddres, float* dx, long lddx, void* dx_act,
                     int act_dtype, float* dgamma, float* dbeta, float* dcolsum, int M, int D, void* workspace,
                     size_t ws_bytes, void* stream);

/* General matrix product with fused epilogue -- replaces every nn.Linear / Conv2d-as-GEMM of the path
 * and their dgrad / wgrad.   acc[M,N] = sum_k A(m,k) * B(n,k)
 *   a_kmajor = 0: A stored [M][K] (lda);  1: A stored [K][M] (lda)   (wgrad: A = dY^T)
 *   b_kmajor = 0: B stored [N][K] (ldb);  1: B stored [K][N] (ldb)   (dgrad: B = W as stored; wgrad: B = X)
 * in_dtype: element type of A and B (PM_BF16 -> v_mfma_f32_32x32x16_bf16, PM_F32 -> v_mfma_f32_32x32x2_f32).
 * bias: f32 [N] or NULL.  C: c_dtype [M][N] (ldc).  aux: in_dtype [M][N] (ldc) for GELU / DGELU.
 * resid: f32 [M][N] (ldc) for PM_EPI_RESIDUAL (may alias C). */
int pm_gemm(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor, int in_dtype,
            const float* bias, void* C, long ldc, int c_dtype, int epilogue, void* aux, const float* resid,
            int M, int N, int K, void* stream);

/* Same as pm_gemm with a caller-owned scratch buffer: when the output has few 128x128 tiles and K is long (the wgrad
 * shapes: K = number of tokens) the k-loop is split over up to 16 blocks per tile, partial sums go to f32 slabs in
 * `workspace` and are reduced in a fixed order (deterministic).  ws_bytes >= 16*M*N*4 enables every split. */
int pm_gemm_ws(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor, int in_dtype,
               const float* bias, void* C, long ldc, int c_dtype, int epilogue, void* aux, const float* resid,
               int M, int N, int K, void* workspace, size_t ws_bytes, void* stream);

/* pm_gemm_ws with per-call options (no process-wide tuning state):
 *   max_blocks: workgroups a split-K weight-gradient GEMM (both operands k-major) spreads over; 0 = 256 = the whole
 *               chip (fastest alone).  A caller that runs weight gradients on a second stream beside the dgrad chain
 *               passes ~128 so that the chain keeps half of the CUs (the training engine does: +4.5 % step rate).
 *   variant:    0 = the dispatcher's heuristics; otherwise forces one kernel variant (tuning scripts and tests; the
 *               values are listed in pm_gemm.hip and are not stable across ABI versions). */
typedef struct pm_gemm_opts {
  int max_blocks;
  int variant;
} pm_gemm_opts;
int pm_gemm_ex(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor, int in_dtype,
               const float* bias, void* C, long ldc, int c_dtype, int epilogue, void* aux, const float* resid,
               int M, int N, int K, void* workspace, size_t ws_bytes, const pm_gemm_opts* opts, void* stream);

/* Every weight gradient of one transformer block in ONE launch (autograd of timm Block's four Linears: attn.qkv,
 * attn.proj, mlp.fc1, mlp.fc2 -- models_mae.py:39-41,53-55 through engine_pretrain.py:65 / tc.py:4533):
 *   dW_i[n_out][n_in] (+)= dY_i^T X_i,   dY_i act [K][n_out] (lddy), X_i act [K][n_in] (ldx), K = number of tokens.
 * The weight gradients have a huge reduction dimension and few output tiles; one GEMM at a time needs split-K (short
 * k-loops, f32 slabs, a reduce launch each).  Grouped, the ~100 tiles of a ViT-B block each run their whole K in one
 * workgroup: no slabs, no reduce, deterministic, and the launch occupies ~100 CUs beside the caller's dgrad chain.
 * `items` is a HOST array of n <= 8 descriptors (copied into the kernel arguments).  Returns PM_ESHAPE when the group
 * does not fit the ring kernel (f32 mode, K % 32 != 0, K < 2048, n_out < 256, n_in < 128): the caller then issues
 * pm_gemm_ws per gradient. */
typedef struct pm_wgrad_item {
  const void* dY;
  long lddy;
  const void* X;
  long ldx;
  float* dW;
  long lddw;
  int n_out, n_in;
  int accumulate; /* dW += (gradient accumulation) instead of dW = */
  float* dbias;   /* optional f32 [n_out]: += column sums of dY = the Linear's bias gradient, computed beside the GEMM on
                     the matrix cores (the dY fragments are already in registers) instead of by a separate pm_colsum pass */
} pm_wgrad_item;
/* max_blocks: workgroups (= CUs) the launch may occupy; 0 = one per work item.  Fewer workgroups walk several items each.
 * A group with fewer than 64 tiles of 256x256 and a long K (the 512-wide MAE decoder block: 48 tiles, K = 50 432 tokens)
 * is cut into k-slices so that tiles x slices fills the chip (48 x 4): f32 partials go to `workspace`
 * (pm_wgrad_group_workspace_bytes; 16-byte aligned) and ONE reduce launch on the same stream finishes every dW / dbias of
 * the group in a fixed order.  With workspace == NULL (or too small) such a group runs whole-K tiles of 256x128 instead.
 * max_blocks == PM_GROUP_WHOLE_K: never slice, whole-K tiles of 256x256 whatever the tile count -- for a caller that runs
 * this group beside another launch of the same kind (pm_vit_block_bwd's second group) and fills the chip that way. */
#define PM_GROUP_WHOLE_K (-1)
int pm_wgrad_group(const pm_wgrad_item* items, int n, int K, int in_dtype, int max_blocks, void* workspace, size_t ws_bytes,
                   void* stream);
size_t pm_wgrad_group_workspace_bytes(const pm_wgrad_item* items, int n, int K, int in_dtype);
/* Admission test and plan of pm_wgrad_group WITHOUT launching: returns the status pm_wgrad_group would return for these
 * shapes / alignments (PM_OK, PM_ESHAPE, PM_EALIGN, PM_EINVAL) and, when PM_OK, the scratch it wants (*ws_bytes), the number
 * of 256x256 output tiles of the group (*tiles256) and the k-slices per tile (*k_slices; 1 = whole-K tiles, no slabs).  The
 * pointers of the items are only checked for NULL / alignment.  pm_vit_block_bwd runs this before its first launch; a host
 * engine uses it instead of restating the rules. */
int pm_wgrad_group_plan(const pm_wgrad_item* items, int n, int K, int in_dtype, size_t* ws_bytes, int* tiles256,
                        int* k_slices);

/* pm_gemm followed by the column sums of the stored result: colsum[n] += sum_m C[m][n] (f32 [N]) -- the bias gradient of
 * the Linear whose output gradient C is (reference: autograd of nn.Linear).  Convenience composition (pm_gemm_ws +
 * pm_colsum_ws on the same stream; `workspace` is the column sum's).  The training engine does not use it: the bias
 * gradients of a block ride on pm_wgrad_group (dbias) and on pm_layernorm_bwd (dcolsum). */
int pm_gemm_colsum(const void* A, long lda, int a_kmajor, const void* B, long ldb, int b_kmajor, int in_dtype,
                   const float* bias, void* C, long ldc, int c_dtype, int epilogue, void* aux, const float* resid,
                   float* colsum, int M, int N, int K, void* workspace, size_t ws_bytes, void* stream);

/* Fused multi-head self-attention core -- replaces timm Attention.forward between qkv and proj:
 * softmax(q k^T * dh^-0.5) v, never materialising the [N,N] scores in HBM.
 * qkv: act [B, N, 3, H, dh] (the qkv Linear's output as stored); out: act [B, N, H*dh];
 * lse: f32 [B, H, N] (log-sum-exp of the scaled scores, saved for backward).
 * Supported: dh in {32, 64, 80} (ViT-B / MAE decoder / ViT-H), N <= 288 (257 = patch 14 at 224^2). */
int pm_attention_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int dh, int dtype, void* stream);
/* Backward: dqkv [B,N,3,H,dh] from dout [B,N,H*dh]; delta: f32 workspace [B,H,N]. */
int pm_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                     int B, int N, int H, int dh, int dtype, void* stream);

/* Column sums of a [M,N] act matrix into f32 [N] (+=) -- bias gradients of qkv / fc1 / decoder Linears. */
int pm_colsum(const void* x, long ldx, int dtype, float* out, int M, int N, void* stream);
/* Deterministic form: per-split partial rows go to `workspace` (>= 128*N*4 bytes) and are summed in a fixed order;
 * without a workspace the splits are combined with float atomics (order-dependent last bits). */
int pm_colsum_ws(const void* x, long ldx, int dtype, float* out, int M, int N, void* workspace, size_t ws_bytes,
                 void* stream);

/* Patch extraction (the im2col of timm PatchEmbed's Conv2d k=s=p), optionally only the kept patches
 * of MAE random masking (models_mae.py:141-142 applied BEFORE the projection, which is the same
 * linear map per patch): imgs f32 [B,C,Himg,Himg] NCHW -> cols act [B*keep, ldcols], the first C*p*p of a row in (c,py,px)
 * order, the rest zero (ldcols > C*p*p: the reduction dimension padded to the GEMM's k-step -- 3*14*14 = 588 -> 640 for
 * mae_vit_huge_patch14, models_mae.py:239-244).  Any p that divides img (16-B reads when p % 4 == 0).
 * ids_keep: int32 [B, keep] patch indices, or NULL for all patches in raster order (keep = L). */
int pm_patch_im2col(const float* imgs, const int* ids_keep, void* cols, long ldcols, int out_dtype, int B, int C, int img,
                    int p, int keep, void* stream);

/* Zero-padded operand copy and its inverse, for a Linear whose reduction dimension is not a multiple of the matrix-core
 * kernels' 16-byte chunks (patch 14: PatchEmbed.proj [D, 588], decoder_pred [588, Dd]; models_mae.py:36,57):
 *   pm_pad_cast : dst act [rows_pad, ldd] <- src f32 [rows, lds] in the top-left corner, zeros elsewhere (up to cols_pad)
 *   pm_unpad_add: dst f32 [rows, ldd] (+)= src f32 [rows, lds], the first `cols` columns (the valid part of a gradient
 *                 computed in the padded layout; accumulate = 0 stores). */
int pm_pad_cast(const float* src, long lds, void* dst, long ldd, int dst_dtype, int rows, int cols, int rows_pad,
                int cols_pad, void* stream);
int pm_unpad_add(const float* src, long lds, float* dst, long ldd, int rows, int cols, int accumulate, void* stream);

/* Token assembly: x[b,0,:] = cls + pos[0]; x[b,1+j,:] = emb[b*keep+j,:] + pos[1+id(b,j),:]
 * (models_mae.py:155-163; models.py:198-201 / 28-33).  emb f32 [B*keep, D]; x f32 [B, 1+keep, D]. */
int pm_assemble_tokens(const float* emb, const float* cls, const float* pos, const int* ids_keep, float* x,
                       int B, int keep, int D, void* stream);
/* Backward: demb[b*keep+j] = dx[b,1+j]; dcls += sum_b dx[b,0]; dpos (optional, learnable pos_embed)
 * += scatter of dx.  demb act-typed [B*keep, D]. */
int pm_assemble_tokens_bwd(const float* dx, const int* ids_keep, void* demb, int act_dtype, float* dcls, float* dpos,
                           int B, int keep, int D, void* stream);

/* MAE random masking from noise (models_mae.py:123-148): stable ascending argsort per sample.
 * noise f32 [B,L]; ids_shuffle/ids_restore int32 [B,L]; mask f32 [B,L] (1 = removed). L <= 1024. */
/* Masking noise for random_masking (models_mae.py:132 `torch.rand(N, L, device=x.device)`): noise[i] = U[0, 1) with 24 random bits,
 * Philox4x32-10 keyed by `seed`, counter = (i / 4, stream_id) -- a pure function of (seed, stream_id, i), whatever the launch
 * geometry, device or rank layout.  The caller advances stream_id per draw (per forward pass). */
int pm_mae_noise(float* noise, long n, unsigned long long seed, unsigned int stream_id, void* stream);
int pm_mae_masking(const float* noise, int* ids_shuffle, int* ids_restore, float* mask, int B, int L, int len_keep,
                   void* stream);

/* MAE decoder input (models_mae.py:177-183): mask-token fill + unshuffle + cls re-attach + pos add.
 * emb f32 [B, 1+keep, D]; out f32 [B, 1+L, D]. */
int pm_mae_unshuffle(const float* emb, const float* mask_token, const float* dpos, const int* ids_restore, float* out,
                     int B, int L, int keep, int D, void* stream);
/* Backward: demb act-typed [B,1+keep,D] (pure gather through ids_shuffle); dmask_token f32 [D] (+=).
 * workspace (pm_workspace_bytes(PM_WS_UNSHUFFLE_BWD, M, D) = 1024*D*4 bytes) makes the mask-token sum two-stage and deterministic;
 * NULL (or less than rows*D*4 for the launch's block rows) falls back to float atomics. */
int pm_mae_unshuffle_bwd(const float* dout, const int* ids_shuffle, void* demb, int act_dtype, float* dmask_token,
                         int B, int L, int keep, int D, void* workspace, size_t ws_bytes, void* stream);

/* MAE reconstruction loss (models_mae.py:95-107,198-214): fused patchify + (optional norm_pix) + per-patch MSE.
 * pred f32 rows of `ldp` elements; row (b*(L+1)+1+l) holds patch l of sample b when has_cls_row=1 (the
 * decoder_pred output incl. the cls row), row b*L+l otherwise.  patch_loss f32 [B*L] = mean((pred-target)^2, -1).
 * pm_mae_loss_finish reduces deterministically (single block): sums = {sum(loss*mask), sum(mask)},
 * loss = sums[0]/sums[1]  (models_mae.py:213). */
int pm_mae_loss_fwd(const float* imgs, const float* pred, long ldp, int has_cls_row, float* patch_loss, int B, int C,
                    int img, int p, int norm_pix, void* stream);
int pm_mae_loss_finish(const float* patch_loss, const float* mask, long n, float* sums, float* loss, void* stream);
/* Backward: dpred act-typed [B*(L+has_cls_row), lddp] (lddp >= p*p*C, a multiple of 4; columns beyond p*p*C zeroed),
 * same row order as pred (cls rows and kept patches zeroed); dloss f32 scalar on device. */
int pm_mae_loss_bwd(const float* imgs, const float* pred, long ldp, int has_cls_row, const float* mask,
                    const float* sums, const float* dloss, void* dpred, long lddp, int act_dtype, int B, int C, int img,
                    int p, int norm_pix, void* stream);

/* Rows of a [*, row_bytes] array by index, and back into an all-zero array (engine: the classifier's top block under out_token
 * "cls" -- models.py:134-136 -- whose incoming gradient lives in one row per sample):
 *   pm_gather_rows      : dst[r] = src[idx[r]] for r < R          (row_bytes % 4 == 0; source row pitch ld_bytes)
 *   pm_scatter_rows_zero: dst[m] = inv[m] >= 0 ? src[inv[m]] : 0  for m < M   (row_bytes % 16 == 0, 16-byte aligned buffers) */
int pm_gather_rows(const void* src, long ld_bytes, const int* idx, void* dst, int R, long row_bytes, void* stream);
int pm_scatter_rows_zero(const void* src, const int* inv, void* dst, int M, long row_bytes, void* stream);

/* f32 -> act cast (weight shadow copies for the bf16 MFMA path). */
int pm_cast(const float* src, void* dst, int dst_dtype, long n, void* stream);

/* Device-side tail of the input pipeline -- replaces torchvision `ToTensor()` + `Normalize(mean, std)`
 * and the two random flips of the reference's transform (classification/data/transforms.py:225-253) on frames that
 * arrive as decoded uint8: src u8 [B][H][W][3] (HWC, what PIL / numpy hold), dst f32 [B][3][H][W].
 *   dst = (float(src) / 255 - mean[c]) / std[c]   (same operations, same order, IEEE f32: bit-exact with torchvision)
 * flip_flags: u8 [B] or NULL; bit 0 = horizontal flip, bit 1 = vertical flip of that sample.  W % 4 == 0.
 * A uint8 batch is a quarter of the f32 batch on PCIe (9.6 MB instead of 38.5 MB at B = 64). */
int pm_preprocess_u8(const unsigned char* src, const unsigned char* flip_flags, float* dst, int B, int H, int W,
                     float mean_r, float mean_g, float mean_b, float std_r, float std_g, float std_b, void* stream);

/* ---- train-time augmentation on the device (classification/data/transforms.py:234-246) ----------------
 * The reference applies Resize, ColorJitter(0.4, 0.5, 0.25, 0.01), GaussianBlur((25, 25), sigma in [0.001, 2]), two random flips
 * and RandomRotation(180) per image in PIL / torchvision 0.10 on DataLoader workers.  These entry points do the same work on
 * uint8 HWC frames [B][H][W][3] resident in HBM, with the arithmetic of the library routines the reference ends up in (Pillow
 * Resample.c 8bpc, Blend.c, Convert.c rgb2l / rgb2hsv / hsv2rgb, Geometry.c affine_fixed; torchvision's tensor gaussian_blur):
 * results equal oracle/augment_ref.py bit for bit, which is pinned against Pillow itself.  Random parameters are the caller's. */

/* Resize (Image.resize(BILINEAR), antialiased): horizontal then vertical pass with Pillow's 22-bit fixed-point taps, which the
 * caller builds as Resample.c precompute_coeffs + normalize_coeffs_8bpc do: bounds_* i32 [out][2] = (first source index, count),
 * taps_* i32 [out][ksize].  tmp u8 [B][Hs][Wo][3] (needed when both sizes change).  A pass whose size does not change is skipped. */
int pm_aug_resize_u8(const unsigned char* src, unsigned char* tmp, unsigned char* dst, const int* bounds_x, const int* taps_x,
                     int ksize_x, const int* bounds_y, const int* taps_y, int ksize_y, int B, int Hs, int Ws, int Ho, int Wo,
                     void* stream);

/* RandomResizedCrop's arithmetic (mae/main_pretrain.py:157: RandomResizedCrop(224, scale=(0.2, 1.0), interpolation=bicubic) ->
 * functional.resized_crop = img.crop(box).resize((out, out), BICUBIC)): every sample has its own crop box = (top, left, h, w)
 * (i32 [B][4], inside the frame, drawn by the caller) and therefore its own resample taps, which are built ON THE DEVICE in the
 * double arithmetic of Resample.c (two small launches), followed by the horizontal and the vertical pass.  bicubic = 0 uses the
 * bilinear filter.  workspace: pm_aug_resized_crop_workspace_bytes(B, Hs, Ws, out) bytes, 16-byte aligned. */
size_t pm_aug_resized_crop_workspace_bytes(int B, int Hs, int Ws, int out);
int pm_aug_resized_crop_u8(const unsigned char* src, const int* box, unsigned char* dst, int bicubic, int B, int Hs, int Ws, int out,
                           void* workspace, size_t ws_bytes, void* stream);

/* ColorJitter: per sample, ops in `order` (0 brightness, 1 contrast, 2 saturation, 3 hue; -1 = none): ImageEnhance blends (float32,
 * truncating / clipping as Blend.c), contrast against int(mean luminance + 0.5) of the image as it stands before that op, hue as
 * an HSV round trip with H += hue_shift (uint8 wrap-around; hue_shift = uint8(hue_factor * 255)).  lsum: u64 [B] scratch.
 * Two launches (exact integer luminance sum, then the chain); src and dst may be the same buffer only if order has no contrast. */
typedef struct pm_aug_jitter {
  int order[4];
  float brightness, contrast, saturation;
  int hue_shift;
} pm_aug_jitter;
int pm_aug_color_jitter_u8(const unsigned char* src, unsigned char* dst, const pm_aug_jitter* jitter, unsigned long long* lsum,
                           int B, int H, int W, void* stream);

/* GaussianBlur: separable ksize-tap convolution with reflect padding in f32 (taps f32 [B][ksize], one row per sample: the
 * caller evaluates torchvision's _get_gaussian_kernel1d for its sigma), taps summed in index order with separate multiply and
 * add, rint + clamp back to u8.  tmp f32 [B][H][W][3].  ksize odd, ksize / 2 < min(H, W).
 * Exactness: bit for bit against oracle/augment_ref.py (two separable f32 passes in this summation order).  torchvision 0.10's
 * tensor path builds the 2-D kernel (outer product of the 1-D taps) and runs ONE depthwise conv2d: different f32 rounding, so
 * after rint a few pixels can differ from the reference transform by 1 grey level (the oracle agrees with a float64 2-D
 * convolution to <= 1 level, > 99 % of pixels exactly); torchvision is absent here, so this stage is tolerance +-1 LSB, unpinned. */
int pm_aug_gaussian_blur_u8(const unsigned char* src, float* tmp, unsigned char* dst, const float* taps, int ksize, int B, int H,
                            int W, void* stream);

/* Flips, then rotation (Image.rotate(angle, NEAREST, expand=False, fillcolor=0)), then either the rotated u8 frame (to_f32 = 0,
 * dst u8 [B][H][W][3]) or ToTensor + Normalize (to_f32 = 1, dst f32 [B][3][H][W], as pm_preprocess_u8).  Per sample:
 * mode 0 = Pillow's 16.16 fixed-point inverse map, xin = (a2 + a0 x + a1 y) >> 16, yin = (a5 + a3 x + a4 y) >> 16 (coefficients
 * built by the caller as Image.rotate + affine_fixed do); 1 = no rotation; 2 / 3 / 4 = 180 / 90 / 270 degrees (Image.rotate's
 * transpose fast paths; 90 / 270 need H == W).  flips: bit 0 horizontal, bit 1 vertical, applied BEFORE the rotation. */
typedef struct pm_aug_geom {
  int mode, a0, a1, a2, a3, a4, a5, flips;
} pm_aug_geom;
int pm_aug_geometry_u8(const unsigned char* src, const pm_aug_geom* geom, void* dst, int to_f32, int B, int H, int W, float mean_r,
                       float mean_g, float mean_b, float std_r, float std_g, float std_b, void* stream);

/* Eval-time perturbations of PerRowPerturbations (classification/data/transforms.py:143-203; Exp-5A/5B feed them to the evaluation
 * forward path) on the resized uint8 frames, u8 [B][H][W][3]:
 *   "blur*": ImageFilter.GaussianBlur(radius=sigma) = Pillow's BoxBlur.c, `passes` extended-box passes per axis in UINT32 fixed
 *            point.  Per sample: radius = (int) r, ww = (UINT32)((float) 2^24 / (r * 2 + 1)), fw = (2^24 - (2 radius + 1) ww) / 2
 *            with r = _gaussian_blur_radius(sigma, passes) in float32 (built by the caller: data.py pil_box_blur_params);
 *            radius < 0: the sample passes through unchanged.  tmp: u8 [B][H][W][3] scratch; src may equal dst, tmp must not.
 *   "occ*":  ImageDraw.rectangle([x0, y0, x1, y1], fill=0), corners inclusive, clipped to the frame; x1 < x0: nothing.  In place.
 *   "bc*":   pm_aug_color_jitter_u8 with order {0, 1, -1, -1}.
 *   "jpeg*": img.save(format="JPEG", quality=q, optimize=False, subsampling=0) and back (transforms.py:78-85) WITHOUT a bitstream:
 *            entropy coding is lossless, so the round trip is libjpeg's integer pipeline -- RGB -> YCbCr (jccolor.c), 4:4:4, islow
 *            forward DCT (jfdctint.c), Annex-K tables scaled by q (jcparam.c, baseline), quantise / dequantise (jcdctmgr.c), islow
 *            inverse DCT (jidctint.c), YCbCr -> RGB (jdcolor.c); sides that are not multiples of 8 are edge-replicated and cropped.
 *            quality: int [B], 1..100; <= 0: the sample is copied.  src may equal dst.
 * Bit for bit against oracle/augment_ref.py, which is pinned by running the reference's own PerRowPerturbations (Pillow 12.2 with
 * its bundled libjpeg-turbo). */
typedef struct pm_aug_boxblur {
  int radius;
  unsigned int ww, fw;
} pm_aug_boxblur;
int pm_aug_pil_gaussian_blur_u8(const unsigned char* src, unsigned char* tmp, unsigned char* dst, const pm_aug_boxblur* prm,
                                int passes, int B, int H, int W, void* stream);
int pm_aug_occlude_u8(unsigned char* img, const int* rects, int B, int H, int W, void* stream);
int pm_aug_jpeg_roundtrip_u8(const unsigned char* src, unsigned char* dst, const int* quality, int B, int H, int W, void* stream);

/* One transformer block forward for one range of samples in ONE call (timm Block: models_mae.py:39-41,53-55,166-167,
 * 186-187; models.py:122-123,204-205):  x_mid = x + proj(attn(LN1 x));  x_out = x_mid + fc2(gelu(fc1(LN2 x_mid))).
 * Host-side composition of pm_layernorm_fwd / pm_gemm_ex / pm_attention_fwd on `stream`, launch for launch what a caller
 * would issue itself -- for hosts where ~27 separate calls per block cost more than the block takes to run.  Every
 * pointer addresses row 0 of the range (rows = samples * N tokens); buffers are the saved-for-backward activations:
 * ln1 / qkv [rows, 3D] / attn / ln2 / h_pre / h_act [rows, Hd] act-typed, x / x_mid / x_out f32 [rows, D], mean / rstd f32
 * [rows], lse f32 [samples * heads * N].  h_pre may be NULL when no backward will run through the block (evaluation, a frozen block
 * under a frozen front): fc1's GELU epilogue then stores the activation only.  The descriptor is plain data: build it once, keep it,
 * pass it every step. */
typedef struct pm_block_fwd_desc {
  const float* x;
  float* x_mid;
  float* x_out;
  void* ln1;
  float* mean1;
  float* rstd1;
  void* qkv;
  float* lse;
  void* attn;
  void* ln2;
  float* mean2;
  float* rstd2;
  void* h_pre;
  void* h_act;
  const float *norm1_w, *norm1_b, *norm2_w, *norm2_b;  /* f32 [D] */
  const void *qkv_w, *proj_w, *fc1_w, *fc2_w;          /* act-typed [out, in] as nn.Linear stores them */
  const float *qkv_b, *proj_b, *fc1_b, *fc2_b;         /* f32 */
  int rows, samples, N, D, Hd, heads;
  int dtype;        /* PM_BF16 / PM_F32: activation and matrix type */
  int gemm_variant; /* pm_gemm_opts.variant for the four GEMMs (0 = the dispatcher's heuristics) */
  float eps;
} pm_block_fwd_desc;
int pm_vit_block_fwd(const pm_block_fwd_desc* d, void* stream);

/* The backward of that block in ONE call (autograd of the timm Block through engine_pretrain.py:65 / tc.py:4533), for a
 * fully trainable block whose four weight gradients fit pm_wgrad_group:
 *   on `stream`:       [wait ev_join]  dfc2 (dGELU) -> dfc1 -> LN2' -> dproj -> attention' -> dqkv -> LN1'
 *   on `side_stream`:  after ev_fork (recorded behind attention'): pm_wgrad_group(fc2, fc1 + bias, proj, qkv + bias) -> ev_done
 * With two_groups != 0 (and an MLP pair that alone has >= 64 whole-K tiles; otherwise the flag is ignored) the weight
 * gradients go out as two launches on two side streams, each as soon as its operands exist:
 *   on `side_stream`:   after ev_fork  (recorded behind dfc2):       pm_wgrad_group(fc2, fc1 + bias)  -> ev_done
 *   on `side_stream2`:  after ev_fork2 (recorded behind attention'): pm_wgrad_group(proj, qkv + bias) -> ev_done2
 * so that the 72 + 36 workgroups of a ViT-B block are spread over the whole dgrad chain (the second launch runs on into the
 * next block's dfc2, which then shares the chip with 36 workgroups instead of 108); the block's matrix gradients are final
 * once BOTH events have fired.
 * dx / dx_act: gradient of the block output (f32 + act copy); dmid / dmid_act: residual gradient between the branches;
 * din / din_act: gradient of the block input (outputs).  Vector gradients are += targets; g_below_bias receives the column
 * sums of din (bias gradient of the Linear that produced the block input), NULL when there is none; accumulate bit j
 * (0 qkv, 1 proj, 2 fc1, 3 fc2): dW += instead of dW =.  ev_join (optional), ev_fork, ev_done are hipEvent_t of the caller:
 * ev_done fires when this block's matrix gradients are final and its operands may be overwritten.  The library creates,
 * keeps and frees nothing.  The weight-gradient group is validated (pm_wgrad_group_plan: shapes, alignment; and a k-sliced
 * group's ws_group_bytes against the plan's slab size -- too small is PM_EINVAL, not a silent fall-back to whole-K 256x128 tiles)
 * BEFORE the first launch: a refusal (PM_ESHAPE / PM_EALIGN / PM_EINVAL) returns with nothing enqueued on either stream, and
 * the caller falls back to per-kernel calls.  Any later non-zero status is a launch failure (PM_ELAUNCH). */
typedef struct pm_block_bwd_desc {
  const float* x_in;
  const float* x_mid;
  const void *ln1, *qkv, *attn, *ln2, *h_pre, *h_act;
  const float *mean1, *rstd1, *mean2, *rstd2, *lse;
  const float *norm1_w, *norm2_w;
  const void *qkv_w, *proj_w, *fc1_w, *fc2_w;
  const float* dx;
  const void* dx_act;
  float* dmid;
  void* dmid_act;
  float* din;
  void* din_act;
  void *d_hidden, *d_qkv, *d_ln, *d_attn;
  float* delta;
  float *g_norm1_w, *g_norm1_b, *g_norm2_w, *g_norm2_b;
  float *g_qkv_w, *g_proj_w, *g_fc1_w, *g_fc2_w;
  float *g_qkv_b, *g_proj_b, *g_fc1_b, *g_below_bias;
  void* ws_ln;
  size_t ws_ln_bytes;
  void* ws_group;
  size_t ws_group_bytes;
  void *side_stream, *ev_join, *ev_fork, *ev_done;
  int samples, N, D, Hd, heads, dtype, gemm_variant, group_blocks, accumulate;
  int two_groups;                                /* see above; needs the three handles below */
  void *side_stream2, *ev_fork2, *ev_done2;
} pm_block_bwd_desc;
int pm_vit_block_bwd(const pm_block_bwd_desc* d, void* stream);

/* Top of the fine-tune forward (models.py:127,134-139 / 209,216-221): final LayerNorm, token selection, lin_head.
 *   pool = 0 (out_token "cls"):     feat[b] = LN(x[b, 0])                 (only row 0 is normalised)
 *   pool = 1 (out_token "spatial"): feat[b] = mean_{n >= 1} LN(x[b, n])   (x[:, 1:].mean(1))
 *   logits = feat W^T + bias when W != NULL (head=True); W == NULL returns the features only (head=False).
 * x f32 [B, N, D]; feat f32 [B, D]; mean / rstd f32 [B] (pool 0) or [B, N] (pool 1, row 0 unused); xhat_mean f32 [B, D]
 * (pool 1 only: the pooled normalised rows, saved for backward); logits f32 [B, n_class].  D <= 1024. */
int pm_vit_head_fwd(const float* x, int N, int pool, const float* gamma, const float* beta, const float* W,
                    const float* bias, float* feat, float* xhat_mean, float* mean, float* rstd, float* logits, int B, int D,
                    int n_class, float eps, void* stream);
/* Backward.  Gradient source: dlogits f32 [B, n_class] (with W), or dfeat f32 [B, D] (head=False; then dlogits / W / dW /
 * dbias are unused).  dx f32 [B,N,D] is fully written (rows that do not reach the feature get zeros) together with its
 * act-typed copy dx_act (optional); dx == NULL = frozen backbone, nothing below is needed.  dW / dbias / dgamma / dbeta
 * (+=, each optional) are summed over the batch in a fixed order. */
int pm_vit_head_bwd(const float* dlogits, const float* dfeat, const float* x, int N, int pool, const float* gamma,
                    const float* W, const float* feat, const float* xhat_mean, const float* mean, const float* rstd,
                    float* dx, void* dx_act, int act_dtype, float* dW, float* dbias, float* dgamma, float* dbeta, int B,
                    int D, int n_class, void* stream);
/* The same with pool = 0 and a head (the shipped configuration: utils/__init__.py:29,52 default out_token "cls"). */
int pm_cls_head_fwd(const float* x, int N, const float* gamma, const float* beta, const float* W, const float* bias,
                    float* xn, float* mean, float* rstd, float* logits, int B, int D, int n_class, float eps,
                    void* stream);
int pm_cls_head_bwd(const float* dlogits, const float* x, int N, const float* gamma, const float* W, const float* xn,
                    const float* mean, const float* rstd, float* dx, void* dx_act, int act_dtype, float* dW,
                    float* dbias, float* dgamma, float* dbeta, int B, int D, int n_class, void* stream);

/* Supervised loss of the fine-tune loop on the logits (train_classification.py:3347-3374 _compute_supervised_loss,
 * 6086-6104 loss construction), value and gradient in one launch:
 *   n_class == 2 ("binary_bce"): z = l[:,1] - l[:,0]; BCEWithLogitsLoss(pos_weight)(z, y), mean over the batch;
 *                                pos_weight: f32 DEVICE scalar or NULL (= 1)
 *   n_class  > 2: CrossEntropyLoss(weight = class_weights f32 [n_class] or NULL), weighted mean.
 * targets int64 [B]; loss f32 [1]; dlogits f32 [B, n_class] = d loss / d logits.  One block, fixed summation order. */
int pm_supervised_loss_fwd(const float* logits, const long long* targets, const float* pos_weight,
                           const float* class_weights, float* loss, float* dlogits, int B, int n_class, void* stream);
/* out[i] = x[i] * scale[0] (scale: f32 device scalar) -- chains an upstream d loss into the saved dlogits. */
int pm_scale(const float* x, const float* scale, float* out, long n, void* stream);

/* out[i] = dy[i] * gelu_erf'(pre[i]) over n elements of `dtype` (n % 4 == 0): the backward of timm Mlp's nn.GELU
 * (models_mae.py:39-41 through timm Block) as a stand-alone pass, for callers that compose blocks from the per-kernel ops
 * (the training step gets the same product from the dfc2 GEMM's PM_EPI_DGELU epilogue).  out may alias dy. */
int pm_dgelu(const void* dy, const void* pre, void* out, int dtype, long n, void* stream);

/* Fused multi-tensor AdamW over one flat f32 parameter range (torch.optim.AdamW semantics,
 * tc.py:5766-5768 / main_pretrain.py:218) that also refreshes the act-typed shadow copy used by the GEMMs.
 * grad_scale multiplies the gradient (1/world, 1/accum).  step >= 1. */
int pm_adamw(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, long n, float lr,
             float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* hipGraph-replayable form of the same update: the hyper-parameters live in DEVICE memory, one 16-float record per
 * param group: [0] lr [1] beta1 [2] beta2 [3] eps [4] weight_decay [5] grad_scale [6] step [7] bc1 [8] 1/sqrt(bc2)
 * [9] skip flag [10] 1 / loss scale (both written by pm_loss_scale_update; 0 = no loss scaling).
 * pm_adamw_tick advances step and the bias corrections of `n_groups` records; pm_adamw_dev applies one record to a
 * flat range (gradient x [5] x [10]); both return without touching anything while [9] is set.  A captured step therefore
 * replays correctly while the host changes lr between replays. */
int pm_adamw_tick(float* hyper, int n_groups, void* stream);
int pm_adamw_dev(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, long n,
                 const float* hyper, void* stream);

/* Dynamic loss scaling of precision mode fp16, on the device: GradScaler.step() + update() of the reference's AMP loop
 * (train_classification.py:4533-4546; engine_pretrain.py:65-72 via mae/util/misc.py:252-282) without the host read-back.
 * state: f32[8] = [0] scale [1] growth tracker [2] found_inf of the last step [3] skipped steps [4] steps [5] 1/scale used.
 * stats: the pm_grad_stats triple of THIS step's (scaled) gradients.  Writes the skip flag and 1/scale into the `n_groups`
 * AdamW records, then moves the scale: x backoff_factor after a non-finite step, x growth_factor after growth_interval
 * clean steps in a row.  Launch order on one stream: pm_grad_stats ..., pm_loss_scale_update, pm_adamw_tick, pm_adamw_dev ... */
int pm_loss_scale_update(float* state, const float* stats, float* hyper, int n_groups, float growth_factor,
                         float backoff_factor, int growth_interval, void* stream);

/* One-pass gradient statistics over a flat f32 range: out[0] += sum(g^2), out[1] += #NaN, out[2] += #Inf
 * (the device-side counterpart of tc.py:1437-1454 _compute_grad_norm and misc.py:387-400 detect_grad_anomalies). */
int pm_grad_stats(const float* g, long n, float* out, void* stream);

/* Scratch sizes.  Every workspace is caller-owned; these return the byte count that enables the deterministic
 * (two-stage, fixed-order) form of the op for the given shape, 0 when the op needs none.
 *   pm_gemm_workspace_bytes: the split-K slabs pm_gemm_ws / pm_gemm_ex would use for this GEMM under `opts`.
 *   pm_workspace_bytes(kind, M, N): PM_WS_LAYERNORM_BWD (M rows, N = D), PM_WS_COLSUM (M x N matrix),
 *   PM_WS_GEMM_COLSUM (pm_gemm_colsum of an M x N result), PM_WS_UNSHUFFLE_BWD (N = D). */
enum pm_ws_kind { PM_WS_LAYERNORM_BWD = 1, PM_WS_COLSUM = 2, PM_WS_GEMM_COLSUM = 3, PM_WS_UNSHUFFLE_BWD = 4 };
size_t pm_gemm_workspace_bytes(int a_kmajor, int b_kmajor, int in_dtype, int M, int N, int K, const pm_gemm_opts* opts);
size_t pm_workspace_bytes(int kind, int M, int N);

/* ---- gradient exchange for hosts without torch.distributed (SURVEY 8-b `pm_comm_*`) --------------------------------------
 * A handle over RCCL (bound at run time: PM_EARCH when no librccl can be loaded).  Replaces DistributedDataParallel's reducer of
 * the reference (mae/main_pretrain.py:212-214, train_classification.py:5746-5750) for a C / C++ host; the Python host does the
 * same through torch.distributed (parallel.GradSync).  One process per GPU: rank 0 calls pm_comm_unique_id and hands the 128
 * bytes to the other ranks out of band (file, socket, environment); every rank then calls pm_comm_create on ITS device.
 * pm_comm_allreduce_f32 sums, in place, n_buckets slices base[lo[i] .. hi[i]) (element offsets) of one f32 range over all ranks
 * in one RCCL group on `stream` -- the flat gradient range needs no packing, a bucket is a slice; the caller orders `stream`
 * against its compute stream with events and divides by the world size where it consumes the sums (pm_adamw_dev grad_scale).
 * The handle is the one object this library creates; pm_comm_destroy frees it. */
typedef struct pm_comm pm_comm;
int pm_comm_unique_id(void* id128);
int pm_comm_create(pm_comm** out, const void* id128, int rank, int world);
int pm_comm_world(const pm_comm* c, int* rank, int* world);
int pm_comm_allreduce_f32(pm_comm* c, float* base, const long* lo, const long* hi, int n_buckets, void* stream);
int pm_comm_destroy(pm_comm* c);

#ifdef __cplusplus
}
#endif
#endif
