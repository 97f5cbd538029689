// pm_block.hip -- block-level launchers: one C-ABI call enqueues every kernel of one transformer block (timm Block,
// models_mae.py:39-41,53-55 / models.py:122-123; x += proj(attn(LN1 x)); x += fc2(gelu(fc1(LN2 x)))) for one range of
// samples.  Host code only: it calls the per-kernel entry points of this library in the order the Python engine used to,
// so the device work is identical launch for launch.  Why: a ViT-B step is ~330 launches; issued one by one through
// ctypes the host spends 5 ms of Python per 10.8-ms step, which leaves little room on a busy host.  A block forward is 7
// launches -> 1 call with a descriptor the caller builds once and keeps (the workspaces it points into are persistent).
#include "pm_common.h"

extern "C" int pm_vit_block_fwd(const pm_block_fwd_desc* d, void* stream) {
  if (!d) return PM_EINVAL;
  if (!d->x || !d->x_mid || !d->x_out || !d->ln1 || !d->qkv || !d->attn || !d->ln2 || !d->h_act) return PM_EINVAL;   // (h_pre may be NULL: forward-only)
  if (d->rows <= 0 || d->samples <= 0 || d->N <= 0 || d->rows != d->samples * d->N) return PM_ESHAPE;
  if (d->D <= 0 || d->Hd <= 0 || d->heads <= 0 || (d->D % d->heads)) return PM_ESHAPE;
  if (d->dtype != PM_BF16 && d->dtype != PM_F16 && d->dtype != PM_F32) return PM_EINVAL;
  const int M = d->rows, D = d->D, Hd = d->Hd, dt = d->dtype;
  pm_gemm_opts opts;
  opts.max_blocks = 0;
  opts.variant = d->gemm_variant;
  int st;
  // x_mid = x + proj(attn(LN1 x))
  if ((st = pm_layernorm_fwd(d->x, D, d->norm1_w, d->norm1_b, d->ln1, dt, d->mean1, d->rstd1, M, D, d->eps, stream))) return st;
  if ((st = pm_gemm_ex(d->ln1, D, 0, d->qkv_w, D, 0, dt, d->qkv_b, d->qkv, 3 * D, dt, PM_EPI_STORE, nullptr, nullptr, M, 3 * D,
                       D, nullptr, 0, &opts, stream)))
    return st;
  if ((st = pm_attention_fwd(d->qkv, d->attn, d->lse, d->samples, d->N, d->heads, D / d->heads, dt, stream))) return st;
  if ((st = pm_gemm_ex(d->attn, D, 0, d->proj_w, D, 0, dt, d->proj_b, d->x_mid, D, PM_F32, PM_EPI_RESIDUAL, nullptr, d->x, M, D,
                       D, nullptr, 0, &opts, stream)))
    return st;
  // x_out = x_mid + fc2(gelu(fc1(LN2 x_mid)))
  if ((st = pm_layernorm_fwd(d->x_mid, D, d->norm2_w, d->norm2_b, d->ln2, dt, d->mean2, d->rstd2, M, D, d->eps, stream))) return st;
  if ((st = pm_gemm_ex(d->ln2, D, 0, d->fc1_w, D, 0, dt, d->fc1_b, d->h_act, Hd, dt, PM_EPI_GELU, d->h_pre, nullptr, M, Hd, D,
                       nullptr, 0, &opts, stream)))
    return st;
  return pm_gemm_ex(d->h_act, Hd, 0, d->fc2_w, Hd, 0, dt, d->fc2_b, d->x_out, D, PM_F32, PM_EPI_RESIDUAL, nullptr, d->x_mid, M, D,
                    Hd, nullptr, 0, &opts, stream);
}

// Backward of the same block: the dgrad / attention / LayerNorm chain on `stream`, the grouped weight gradients (with the
// qkv / fc1 bias gradients inside) on d->side_stream behind an event, exactly as BlockStack.backward orders them:
//   [wait ev_join]  dfc2(dGELU) -> dfc1 -> LN2' -> dproj -> attention' -> {fork: side: wgrad_group -> ev_done} -> dqkv -> LN1'
// The events are the caller's (created once, reused every step); nothing here allocates or keeps state.
extern "C" int pm_vit_block_bwd(const pm_block_bwd_desc* d, void* stream) {
  if (!d) return PM_EINVAL;
  if (!d->x_in || !d->x_mid || !d->ln1 || !d->qkv || !d->attn || !d->ln2 || !d->h_pre || !d->h_act || !d->dx || !d->dx_act ||
      !d->dmid || !d->dmid_act || !d->din || !d->din_act || !d->d_hidden || !d->d_qkv || !d->d_ln || !d->d_attn || !d->delta)
    return PM_EINVAL;
  if (!d->g_qkv_w || !d->g_proj_w || !d->g_fc1_w || !d->g_fc2_w || !d->side_stream || !d->ev_fork || !d->ev_done) return PM_EINVAL;
  if (d->samples <= 0 || d->N <= 0 || d->D <= 0 || d->Hd <= 0 || d->heads <= 0 || (d->D % d->heads)) return PM_ESHAPE;
  if (d->dtype != PM_BF16 && d->dtype != PM_F16 && d->dtype != PM_F32) return PM_EINVAL;
  const int M = d->samples * d->N, D = d->D, Hd = d->Hd, dt = d->dtype;
  hipStream_t main = pm_stream(stream), side = pm_stream(d->side_stream);
  pm_gemm_opts opts;
  opts.max_blocks = 0;
  opts.variant = d->gemm_variant;
  int st;
  // the grouped weight gradients are the only launch of this block that can refuse its shapes: plan them BEFORE anything is
  // enqueued, so that a refusal leaves both streams untouched (the caller then issues the block kernel by kernel)
  pm_wgrad_item it[4];
  size_t ws_need = 0;  // slab scratch of the one-launch k-sliced group (0: whole-K tiles)
  {
    const void* dys[4] = {d->dx_act, d->d_hidden, d->dmid_act, d->d_qkv};
    const void* xs[4] = {d->h_act, d->ln2, d->attn, d->ln1};
    float* dws[4] = {d->g_fc2_w, d->g_fc1_w, d->g_proj_w, d->g_qkv_w};
    float* dbs[4] = {nullptr, d->g_fc1_b, nullptr, d->g_qkv_b};
    const int n_out[4] = {D, Hd, D, 3 * D}, n_in[4] = {Hd, D, D, D};
    const int acc_bit[4] = {3, 2, 1, 0};  // accumulate bits: 0 qkv, 1 proj, 2 fc1, 3 fc2
    for (int j = 0; j < 4; ++j) {
      it[j].dY = dys[j]; it[j].lddy = n_out[j]; it[j].X = xs[j]; it[j].ldx = n_in[j]; it[j].dW = dws[j]; it[j].lddw = n_in[j];
      it[j].n_out = n_out[j]; it[j].n_in = n_in[j]; it[j].accumulate = (d->accumulate >> acc_bit[j]) & 1; it[j].dbias = dbs[j];
    }
    if ((st = pm_wgrad_group_plan(it, 4, M, dt, &ws_need, nullptr, nullptr))) return st;
  }
  // two launches (fc2, fc1 | proj, qkv) only when the MLP pair alone is a whole-K group; the small pair then runs whole-K too
  bool two = false;
  if (d->two_groups) {
    if (!d->side_stream2 || !d->ev_fork2 || !d->ev_done2) return PM_EINVAL;
    int slices = 0;
    if ((st = pm_wgrad_group_plan(it, 2, M, dt, nullptr, nullptr, &slices))) return st;
    if ((st = pm_wgrad_group_plan(it + 2, 2, M, dt, nullptr, nullptr, nullptr))) return st;
    two = slices == 1;
  }
  // a k-sliced group without room for its slabs would silently run as whole-K 256x128 tiles (pm_wgrad_group's fallback): a
  // different launch geometry than the caller planned for -- refuse instead, before anything is enqueued
  if (!two && ws_need > 0 && (!d->ws_group || d->ws_group_bytes < ws_need)) return PM_EINVAL;
  hipStream_t side2 = two ? pm_stream(d->side_stream2) : nullptr;
  if (d->ev_join && hipStreamWaitEvent(main, (hipEvent_t)d->ev_join, 0) != hipSuccess) return PM_ELAUNCH;
  // ---- MLP branch
  if ((st = pm_gemm_ex(d->dx_act, D, 0, d->fc2_w, Hd, 1, dt, nullptr, d->d_hidden, Hd, dt, PM_EPI_DGELU, (void*)d->h_pre, nullptr, M,
                       Hd, D, nullptr, 0, &opts, stream)))
    return st;
  if (two) {  // fc2 / fc1 weight gradients: their operands exist from here on
    if (hipEventRecord((hipEvent_t)d->ev_fork, main) != hipSuccess) return PM_ELAUNCH;
    if (hipStreamWaitEvent(side, (hipEvent_t)d->ev_fork, 0) != hipSuccess) return PM_ELAUNCH;
    if ((st = pm_wgrad_group(it, 2, M, dt, d->group_blocks, nullptr, 0, d->side_stream))) return st;
    if (hipEventRecord((hipEvent_t)d->ev_done, side) != hipSuccess) return PM_ELAUNCH;
  }
  if ((st = pm_gemm_ex(d->d_hidden, Hd, 0, d->fc1_w, D, 1, dt, nullptr, d->d_ln, D, dt, PM_EPI_STORE, nullptr, nullptr, M, D, Hd,
                       nullptr, 0, &opts, stream)))
    return st;
  if ((st = pm_layernorm_bwd(d->d_ln, dt, d->x_mid, D, d->norm2_w, d->mean2, d->rstd2, d->dx, D, d->dmid, D, d->dmid_act, dt,
                             d->g_norm2_w, d->g_norm2_b, d->g_proj_b, M, D, d->ws_ln, d->ws_ln_bytes, stream)))
    return st;
  // ---- attention branch
  if ((st = pm_gemm_ex(d->dmid_act, D, 0, d->proj_w, D, 1, dt, nullptr, d->d_attn, D, dt, PM_EPI_STORE, nullptr, nullptr, M, D, D,
                       nullptr, 0, &opts, stream)))
    return st;
  if ((st = pm_attention_bwd(d->qkv, d->attn, d->d_attn, d->lse, d->delta, d->d_qkv, d->samples, d->N, d->heads, D / d->heads, dt,
                             stream)))
    return st;
  // ---- weight gradients of the block: one grouped launch on the side stream, beside the rest of this chain and the next block's
  if (two) {
    if (hipEventRecord((hipEvent_t)d->ev_fork2, main) != hipSuccess) return PM_ELAUNCH;
    if (hipStreamWaitEvent(side2, (hipEvent_t)d->ev_fork2, 0) != hipSuccess) return PM_ELAUNCH;
    if ((st = pm_wgrad_group(it + 2, 2, M, dt, PM_GROUP_WHOLE_K, nullptr, 0, d->side_stream2))) return st;
    if (hipEventRecord((hipEvent_t)d->ev_done2, side2) != hipSuccess) return PM_ELAUNCH;
  } else {
    if (hipEventRecord((hipEvent_t)d->ev_fork, main) != hipSuccess) return PM_ELAUNCH;
    if (hipStreamWaitEvent(side, (hipEvent_t)d->ev_fork, 0) != hipSuccess) return PM_ELAUNCH;
    if ((st = pm_wgrad_group(it, 4, M, dt, d->group_blocks, d->ws_group, d->ws_group_bytes, d->side_stream))) return st;
    if (hipEventRecord((hipEvent_t)d->ev_done, side) != hipSuccess) return PM_ELAUNCH;
  }
  // ---- back on the main chain
  if ((st = pm_gemm_ex(d->d_qkv, 3 * D, 0, d->qkv_w, D, 1, dt, nullptr, d->d_ln, D, dt, PM_EPI_STORE, nullptr, nullptr, M, D, 3 * D,
                       nullptr, 0, &opts, stream)))
    return st;
  return pm_layernorm_bwd(d->d_ln, dt, d->x_in, D, d->norm1_w, d->mean1, d->rstd1, d->dmid, D, d->din, D, d->din_act, dt,
                          d->g_norm1_w, d->g_norm1_b, d->g_below_bias, M, D, d->ws_ln, d->ws_ln_bytes, stream);
}
