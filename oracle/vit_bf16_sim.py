"""bf16-operand emulation of the HIP path's numerics on the CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

Same algorithm as oracle/vit_mae_ref.py (i.e. the reference's), but every tensor that the MI355X kernels hand to a
bf16 MFMA is rounded to bf16 at the same point, accumulation stays f32, the residual stream / LayerNorm /
softmax statistics stay f32, and the attention follows the kernel's online softmax over 32-key tiles
(un-normalised probabilities rounded to bf16 against the RUNNING maximum).  Two uses:
  * it shows, with no kernel involved, how far bf16 operand rounding alone moves the logits / loss away from the
    fp32 reference (5e-3 .. 1e-2 on ViT-B/16 logits) -- the bound used by the bf16-vs-fp32 tests;
  * the HIP bf16 path must agree with it to <= 1e-3 (only accumulation order and 1-ulp transcendental
    differences remain), which is the north-star tolerance applied to like-for-like arithmetic.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import vit_mae_ref as O


def q(x: torch.Tensor) -> torch.Tensor:
    return x.bfloat16().float()


def _linear(x, sd, pre):  # bf16 operands, f32 accumulate, f32 bias
    return F.linear(q(x), q(sd[pre + "weight"]), sd[pre + "bias"])


def attention(x_ln, sd, pre, heads):
    B, N, C = x_ln.shape
    dh = C // heads
    qkv = q(_linear(x_ln, sd, pre + "qkv."))  # stored bf16
    qkv = qkv.reshape(B, N, 3, heads, dh).permute(2, 0, 3, 1, 4)
    qq, kk, vv = qkv[0], qkv[1], qkv[2]
    c = dh ** -0.5 * math.log2(math.e)
    s = qq @ kk.transpose(-2, -1)  # f32 accumulate of bf16 operands
    m = torch.full((B, heads, N, 1), float("-inf"))
    l = torch.zeros(B, heads, N, 1)
    o = torch.zeros(B, heads, N, dh)
    for t0 in range(0, N, 32):  # the kernel's 32-key tiles
        st = s[..., t0:t0 + 32]
        mn = torch.maximum(m, st.amax(-1, keepdim=True))
        alpha = torch.exp2((m - mn) * c)
        p = torch.exp2((st - mn) * c)
        l = l * alpha + p.sum(-1, keepdim=True)
        o = o * alpha + q(p) @ vv[..., t0:t0 + 32, :]
        m = mn
    out = q((o / l).transpose(1, 2).reshape(B, N, C))  # stored bf16
    return _linear(out, sd, pre + "proj.")


def block(x, sd, pre, heads):
    x = x + attention(q(O.layer_norm(x, sd, pre + "norm1.")), sd, pre + "attn.", heads)
    h_pre = q(_linear(q(O.layer_norm(x, sd, pre + "norm2.")), sd, pre + "mlp.fc1."))  # stored bf16
    g = q(F.gelu(h_pre))
    return x + _linear(g, sd, pre + "mlp.fc2.")


def _tokens(sd, imgs, cfg, ids_keep=None):
    x = F.conv2d(q(imgs), q(sd["patch_embed.proj.weight"]), sd["patch_embed.proj.bias"], stride=cfg.patch_size)
    x = x.flatten(2).transpose(1, 2) + sd["pos_embed"][:, 1:, :]
    if ids_keep is not None:
        x = torch.gather(x, 1, ids_keep.unsqueeze(-1).repeat(1, 1, x.shape[-1]))
    cls = sd["cls_token"] + sd["pos_embed"][:, :1, :]
    return torch.cat((cls.expand(x.shape[0], -1, -1), x), dim=1)


def vit_classify(sd, imgs, cfg=O.VIT_BASE):
    x = _tokens(sd, imgs, cfg)
    for i in range(cfg.depth):
        x = block(x, sd, f"blocks.{i}.", cfg.num_heads)
    x = O.layer_norm(x, sd, "norm.")[:, 0]
    return F.linear(x, sd["lin_head.weight"], sd["lin_head.bias"])  # the head runs in f32


def mae_forward(sd, imgs, noise, cfg=O.VIT_BASE, mask_ratio=0.75, norm_pix_loss=False):
    ids_keep, mask, ids_restore = O.masking_from_noise(noise, mask_ratio)
    x = _tokens(sd, imgs, cfg, ids_keep)
    for i in range(cfg.depth):
        x = block(x, sd, f"blocks.{i}.", cfg.num_heads)
    latent = q(O.layer_norm(x, sd, "norm."))
    x = _linear(latent, sd, "decoder_embed.")
    mask_tokens = sd["mask_token"].repeat(x.shape[0], ids_restore.shape[1] + 1 - x.shape[1], 1)
    x_ = torch.cat([x[:, 1:, :], mask_tokens], dim=1)
    x_ = torch.gather(x_, 1, ids_restore.unsqueeze(-1).repeat(1, 1, x.shape[2]))
    x = torch.cat([x[:, :1, :], x_], dim=1) + sd["decoder_pos_embed"]
    for i in range(cfg.decoder_depth):
        x = block(x, sd, f"decoder_blocks.{i}.", cfg.decoder_num_heads)
    pred = _linear(q(O.layer_norm(x, sd, "decoder_norm.")), sd, "decoder_pred.")[:, 1:, :]
    return O.mae_loss(imgs, pred, mask, cfg, norm_pix_loss), pred, mask
