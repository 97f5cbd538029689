"""Differentiable bf16 emulation of the HIP path's numerics, forward AND backward, with one switch per rounding point
(TEST INFRASTRUCTURE, see oracle/__init__.py -- CPU, plain torch, no kernel involved).

Same algorithm as oracle/vit_mae_ref.py (i.e. the reference's: models.py:196-222, models_mae.py:150-220 + timm 0.4.12
Block / Attention / Mlp), with every tensor that the MI355X kernels store in bf16, or hand to a bf16 MFMA, rounded at the
same point -- in the forward pass (as oracle/vit_bf16_sim.py does) and in the backward pass (activation gradients, the
probabilities P and the score gradients dS in front of their MFMAs, the saved GELU pre-activation).  Accumulation, the
residual stream, LayerNorm / softmax statistics, the head and the losses stay f32, as in the kernels.

Use (scratch/rounding_cost.py, tests/test_gpu_parity_large.py): switch ONE rounding point off at a time and look at what
it costs against the fp32 oracle -- the table in DESIGN.md section 2 -- and derive the bf16 gates of the parity tests from the
all-on emulation instead of from a multiple of a yardstick.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import vit_mae_ref as O


@dataclass(frozen=True)
class Rounding:
    """Which bf16 rounding points are ON (all True = what the kernels do)."""
    operands: bool = True     # forward: LayerNorm outputs, weights, qkv, attention output, GELU output, im2col patches
    p: bool = True            # attention: probabilities P -> bf16 in front of P V (forward) and P^T dO (backward)
    ds: bool = True           # attention backward: dS -> bf16 in front of dS K and dS^T Q
    h_pre: bool = True        # fc1 pre-activation saved in bf16: GELU and its derivative see the rounded value
    act_grads: bool = True    # backward: activation gradients stored in bf16 (dx copy, d_hidden, d_ln, d_attn, d_qkv)
    # round 4: the 16-bit TYPE of each side.  "fp16" forward = what precision="fp16" of the HIP path stores and multiplies
    # (11 significant bits instead of 8; every forward tensor of this network is far inside fp16's range).  An MFMA takes one
    # type for both operands, so fp16 X / W in the forward means fp16 dY in dgrad (dY W) and wgrad (dY^T X): "fp16" backward,
    # whose range (6e-8 ... 65504) needs the reference's loss scaling (train_classification.py:4527-4546: GradScaler):
    # every activation gradient is S x the true one, rounded to fp16 (overflow -> inf, as the hardware does), and the f32 weight
    # gradients are unscaled afterwards -- emulated here as round(g * S) / S at every backward rounding point.
    fwd_dtype: str = "bf16"
    bwd_dtype: str = "bf16"
    loss_scale: float = 1.0


ALL_ON = Rounding()
VARIANTS = {
    "all roundings (the kernels)": ALL_ON,
    "P kept f32": Rounding(p=False),
    "dS kept f32": Rounding(ds=False),
    "GELU pre-activation kept f32": Rounding(h_pre=False),
    "P, dS, pre-activation kept f32": Rounding(p=False, ds=False, h_pre=False),
    "activation gradients kept f32": Rounding(act_grads=False),
    "forward operands only (no backward rounding)": Rounding(p=True, ds=False, h_pre=True, act_grads=False),
}
# round 4 (scratch/rounding_cost.py --family fp16): the candidates for a 16-bit mode that meets SURVEY 8-d as written
FP16_VARIANTS = {
    "bf16 forward, bf16 backward (precision='bf16')": ALL_ON,
    "fp16 forward, bf16 backward (needs bf16 copies of X and W for the backward MFMAs)": Rounding(fwd_dtype="fp16"),
    "fp16 forward, fp16 backward, no loss scaling": Rounding(fwd_dtype="fp16", bwd_dtype="fp16"),
    "fp16 forward, fp16 backward, loss scale 2^8": Rounding(fwd_dtype="fp16", bwd_dtype="fp16", loss_scale=256.0),
    "fp16 forward, fp16 backward, loss scale 2^12": Rounding(fwd_dtype="fp16", bwd_dtype="fp16", loss_scale=4096.0),
    "fp16 forward, fp16 backward, loss scale 2^16 (GradScaler's initial scale)": Rounding(fwd_dtype="fp16", bwd_dtype="fp16", loss_scale=65536.0),
    "fp16 forward only (no backward rounding)": Rounding(fwd_dtype="fp16", ds=False, act_grads=False),
}
FP16 = Rounding(fwd_dtype="fp16", bwd_dtype="fp16", loss_scale=4096.0)  # what precision="fp16" does at its default static scale


def _r(x: torch.Tensor, dtype: str = "bf16", scale: float = 1.0) -> torch.Tensor:
    """x -> the 16-bit type and back (round to nearest even; fp16 saturates to inf and underflows gradually, as v_cvt_f16_f32
    does with the default denormal mode); scale: the loss scale the stored value carries."""
    if dtype == "bf16":
        return x.bfloat16().float()
    if scale == 1.0:
        return x.half().float()
    return (x * scale).half().float() / scale


class _RoundFwd(torch.autograd.Function):  # value -> 16 bit, gradient passes
    @staticmethod
    def forward(ctx, x, dtype):
        return _r(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _RoundBwd(torch.autograd.Function):  # value passes, gradient -> 16 bit (carrying the loss scale)
    @staticmethod
    def forward(ctx, x, dtype, scale):
        ctx.dtype, ctx.scale = dtype, scale
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g, ctx.dtype, ctx.scale), None, None


def _q(x, on, rnd=None):
    return _RoundFwd.apply(x, rnd.fwd_dtype if rnd is not None else "bf16") if on else x


def _qg(x, on, rnd=None):
    return _RoundBwd.apply(x, rnd.bwd_dtype if rnd is not None else "bf16", rnd.loss_scale if rnd is not None else 1.0) if on else x


class _Attention(torch.autograd.Function):
    """softmax(q k^T dh^-0.5) v on [B, H, N, dh] with the kernels' rounding points (pm_attention.hip: attn_fwd2_kernel,
    attn_bwd_fused_kernel): forward P = exp(s - rowmax) un-normalised, rounded in front of P V, normalised afterwards by the
    f32 row sum; backward P = exp(s - lse) recomputed, rounded in front of P^T dO; dS = P (dP - delta) scale rounded in front
    of dS K / dS^T Q; delta = rowsum(dO O) from the stored (bf16) tensors."""

    @staticmethod
    def forward(ctx, q, k, v, rnd: Rounding):
        scale = q.shape[-1] ** -0.5
        s = q @ k.transpose(-2, -1)
        m = s.amax(-1, keepdim=True)
        p = torch.exp((s - m) * scale)
        l = p.sum(-1, keepdim=True)
        o = ((_r(p, rnd.fwd_dtype) if rnd.p else p) @ v) / l
        if rnd.operands:
            o = _r(o, rnd.fwd_dtype)
        ctx.save_for_backward(q, k, v, o, m * scale + torch.log(l))
        ctx.rnd = rnd
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        rnd = ctx.rnd
        scale = q.shape[-1] ** -0.5
        delta = (do * o).sum(-1, keepdim=True)
        p = torch.exp((q @ k.transpose(-2, -1)) * scale - lse)
        dp = do @ v.transpose(-2, -1)
        ds = p * (dp - delta) * scale
        dv = (_r(p, rnd.bwd_dtype) if rnd.p else p).transpose(-2, -1) @ do   # (P^T dO: P takes dO's type)
        if rnd.ds:
            ds = _r(ds, rnd.bwd_dtype, rnd.loss_scale)
        return ds @ k, ds.transpose(-2, -1) @ q, dv, None


class _Gelu(torch.autograd.Function):
    """fc1 epilogue (pm_gemm.hip EPI_GELU / EPI_DGELU): the pre-activation is SAVED in bf16 and GELU is taken of the saved
    value, so forward and derivative see the same number."""

    @staticmethod
    def forward(ctx, h, rnd: Rounding):
        hp = _r(h, rnd.fwd_dtype) if (rnd.h_pre and rnd.operands) else h
        ctx.save_for_backward(hp)
        ctx.rnd = rnd
        a = F.gelu(hp)
        return _r(a, rnd.fwd_dtype) if rnd.operands else a

    @staticmethod
    def backward(ctx, da):
        (hp,) = ctx.saved_tensors
        cdf = 0.5 * (1.0 + torch.erf(hp * (1.0 / math.sqrt(2.0))))
        pdf = torch.exp(-0.5 * hp * hp) * (1.0 / math.sqrt(2.0 * math.pi))
        d = da * (cdf + hp * pdf)
        return (_r(d, ctx.rnd.bwd_dtype, ctx.rnd.loss_scale) if ctx.rnd.act_grads else d), None


def _linear(x, sd, pre, rnd):
    return F.linear(x, _q(sd[pre + "weight"], rnd.operands, rnd), sd[pre + "bias"])


def block(x, sd, pre, heads, rnd: Rounding = ALL_ON):
    B, N, C = x.shape
    dh = C // heads
    ln1 = _qg(_q(O.layer_norm(x, sd, pre + "norm1."), rnd.operands, rnd), rnd.act_grads, rnd)           # d_ln (LN1')
    qkv = _qg(_q(_linear(ln1, sd, pre + "attn.qkv.", rnd), rnd.operands, rnd), rnd.act_grads, rnd)       # d_qkv
    qkv = qkv.reshape(B, N, 3, heads, dh).permute(2, 0, 3, 1, 4)
    a = _Attention.apply(qkv[0], qkv[1], qkv[2], rnd)
    a = _qg(a.transpose(1, 2).reshape(B, N, C), rnd.act_grads, rnd)                                  # d_attn
    x = x + _qg(_linear(a, sd, pre + "attn.proj.", rnd), rnd.act_grads, rnd)                         # bf16 copy of the residual gradient
    ln2 = _qg(_q(O.layer_norm(x, sd, pre + "norm2."), rnd.operands, rnd), rnd.act_grads, rnd)            # d_ln (LN2')
    h = _Gelu.apply(_linear(ln2, sd, pre + "mlp.fc1.", rnd), rnd)                               # d_hidden inside
    return x + _qg(_linear(h, sd, pre + "mlp.fc2.", rnd), rnd.act_grads, rnd)


def _tokens(sd, imgs, cfg, rnd, ids_keep=None):
    x = F.conv2d(_q(imgs, rnd.operands, rnd), _q(sd["patch_embed.proj.weight"], rnd.operands, rnd), sd["patch_embed.proj.bias"],
                 stride=cfg.patch_size)
    x = _qg(x, rnd.act_grads, rnd)  # the embedding's output gradient is handed to its weight-gradient GEMM in bf16
    x = x.flatten(2).transpose(1, 2) + sd["pos_embed"][:, 1:, :]
    if ids_keep is not None:
        x = torch.gather(x, 1, ids_keep.unsqueeze(-1).repeat(1, 1, x.shape[-1]))
    cls = sd["cls_token"] + sd["pos_embed"][:, :1, :]
    return torch.cat((cls.expand(x.shape[0], -1, -1), x), dim=1)


def vit_classify(sd, imgs, cfg=O.VIT_BASE, rnd: Rounding = ALL_ON):
    x = _tokens(sd, imgs, cfg, rnd)
    for i in range(cfg.depth):
        x = block(x, sd, f"blocks.{i}.", cfg.num_heads, rnd)
    x = O.layer_norm(x, sd, "norm.")[:, 0]
    return F.linear(x, sd["lin_head.weight"], sd["lin_head.bias"])  # final LayerNorm + head run in f32 (pm_head.hip)


def mae_forward(sd, imgs, noise, cfg=O.VIT_BASE, mask_ratio=0.75, norm_pix_loss=False, rnd: Rounding = ALL_ON):
    ids_keep, mask, ids_restore = O.masking_from_noise(noise, mask_ratio)
    x = _tokens(sd, imgs, cfg, rnd, ids_keep)
    for i in range(cfg.depth):
        x = block(x, sd, f"blocks.{i}.", cfg.num_heads, rnd)
    latent = _qg(_q(O.layer_norm(x, sd, "norm."), rnd.operands, rnd), rnd.act_grads, rnd)
    x = _qg(_linear(latent, sd, "decoder_embed.", rnd), rnd.act_grads, rnd)
    mask_tokens = sd["mask_token"].repeat(x.shape[0], ids_restore.shape[1] + 1 - x.shape[1], 1)
    x_ = torch.cat([x[:, 1:, :], mask_tokens], dim=1)
    x_ = torch.gather(x_, 1, ids_restore.unsqueeze(-1).repeat(1, 1, x.shape[2]))
    x = torch.cat([x[:, :1, :], x_], dim=1) + sd["decoder_pos_embed"]
    for i in range(cfg.decoder_depth):
        x = block(x, sd, f"decoder_blocks.{i}.", cfg.decoder_num_heads, rnd)
    yn = _qg(_q(O.layer_norm(x, sd, "decoder_norm."), rnd.operands, rnd), rnd.act_grads, rnd)
    pred = _qg(_linear(yn, sd, "decoder_pred.", rnd), rnd.act_grads, rnd)[:, 1:, :]
    return O.mae_loss(imgs, pred, mask, cfg, norm_pix_loss), pred, mask
