"""fp32 CPU restatement of the MAE pre-train / ViT-B/16 fine-tune forward.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional style: every entry
point takes a state dict with the reference's key names and
plain tensors, so the same code checks the tiny fixture config and ViT-B/16.

Reference files restated (paths relative to the reference checkout):
  src/ssl4polyp/models/mae/models_mae.py:22-228   MaskedAutoencoderViT
  src/ssl4polyp/models/models.py:26-33,117-140    VisionTransformer_from_Any
  src/ssl4polyp/models/models.py:143-222          ViT_from_MAE
  src/ssl4polyp/models/mae/util/pos_embed.py:20-67
  src/ssl4polyp/classification/train_classification.py:3347-3374,6086-6104 (loss)
  timm==0.4.12 vision_transformer.{PatchEmbed,Attention,Mlp,Block} (absent dep)
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LN_EPS = 1e-6  # models_mae.py:227 / models.py:164 -- partial(nn.LayerNorm, eps=1e-6)


@dataclass(frozen=True)
class ViTConfig:
    """Geometry of one model.  Defaults = mae_vit_base_patch16 (models_mae.py:223-228)."""

    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    decoder_embed_dim: int = 512
    decoder_depth: int = 8
    decoder_num_heads: int = 16
    mlp_ratio: float = 4.0

    @property
    def grid(self) -> int:
        return self.img_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid


VIT_BASE = ViTConfig()
VIT_LARGE = ViTConfig(embed_dim=1024, depth=24, num_heads=16)                  # models_mae.py:231-236
VIT_HUGE = ViTConfig(patch_size=14, embed_dim=1280, depth=32, num_heads=16)    # models_mae.py:239-244
# Small geometry used by the committed fixtures (weights travel in the fixture).
VIT_TINY = ViTConfig(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2,
                     decoder_embed_dim=32, decoder_depth=1, decoder_num_heads=1)


# --------------------------------------------------------------------------
# pos_embed.py:20-67
# --------------------------------------------------------------------------
def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """pos_embed.py:48-67 -- omega_k = 10000^(-k/(D/2)); [sin | cos]; float64."""
    assert embed_dim % 2 == 0
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim: int, grid_size: int, cls_token: bool = False) -> np.ndarray:
    """pos_embed.py:20-45.  meshgrid(w, h) puts the *w* coordinate in grid[0], so the
    first D/2 channels encode w and the second D/2 encode h."""
    grid_h = np.arange(grid_size, dtype=np.float32)
    grid_w = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape(2, 1, grid_size, grid_size)
    emb = np.concatenate([sincos_1d(embed_dim // 2, grid[0]), sincos_1d(embed_dim // 2, grid[1])], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


# --------------------------------------------------------------------------
# timm 0.4.12 blocks
# --------------------------------------------------------------------------
def patch_embed(imgs: Tensor, w: Tensor, b: Tensor, patch: int) -> Tensor:
    """timm PatchEmbed.forward: Conv2d(k=s=patch) -> flatten(2) -> transpose(1,2)."""
    x = F.conv2d(imgs, w, b, stride=patch)
    return x.flatten(2).transpose(1, 2)


def attention(x: Tensor, sd: Dict[str, Tensor], pre: str, heads: int) -> Tensor:
    """timm Attention.forward (qkv_bias=True, no dropout)."""
    B, N, C = x.shape
    dh = C // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (dh ** -0.5)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def mlp(x: Tensor, sd: Dict[str, Tensor], pre: str) -> Tensor:
    """timm Mlp.forward: fc1 -> GELU(erf) -> fc2."""
    x = F.linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"])
    x = F.gelu(x)
    return F.linear(x, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def layer_norm(x: Tensor, sd: Dict[str, Tensor], pre: str) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[pre + "weight"], sd[pre + "bias"], LN_EPS)


def block(x: Tensor, sd: Dict[str, Tensor], pre: str, heads: int) -> Tensor:
    """timm Block.forward, pre-LN residual, drop_path = identity."""
    x = x + attention(layer_norm(x, sd, pre + "norm1."), sd, pre + "attn.", heads)
    x = x + mlp(layer_norm(x, sd, pre + "norm2."), sd, pre + "mlp.")
    return x


# --------------------------------------------------------------------------
# models_mae.py
# --------------------------------------------------------------------------
def patchify(imgs: Tensor, p: int) -> Tensor:
    """models_mae.py:95-107 -- [N,3,H,W] -> [N, L, p*p*3], pixel order (p, q, c)."""
    n, c, hh, ww = imgs.shape
    assert hh == ww and hh % p == 0
    h = w = hh // p
    x = imgs.reshape(n, c, h, p, w, p)
    x = torch.einsum("nchpwq->nhwpqc", x)
    return x.reshape(n, h * w, p * p * c)


def masking_from_noise(noise: Tensor, mask_ratio: float) -> Tuple[Tensor, Tensor, Tensor]:
    """models_mae.py:129-146 with the noise passed in: returns (ids_keep, mask, ids_restore)."""
    n, L = noise.shape
    len_keep = int(L * (1 - mask_ratio))
    ids_shuffle = torch.argsort(noise, dim=1)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    ids_keep = ids_shuffle[:, :len_keep]
    mask = torch.ones([n, L], dtype=torch.float32)
    mask[:, :len_keep] = 0
    mask = torch.gather(mask, dim=1, index=ids_restore)
    return ids_keep, mask, ids_restore


def mae_forward_encoder(sd, imgs, ids_keep, cfg: ViTConfig) -> Tensor:
    """models_mae.py:150-170 (masking indices supplied)."""
    x = patch_embed(imgs, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], cfg.patch_size)
    x = x + sd["pos_embed"][:, 1:, :]
    D = x.shape[-1]
    x = torch.gather(x, dim=1, index=ids_keep.unsqueeze(-1).repeat(1, 1, D))
    cls = sd["cls_token"] + sd["pos_embed"][:, :1, :]
    x = torch.cat((cls.expand(x.shape[0], -1, -1), x), dim=1)
    for i in range(cfg.depth):
        x = block(x, sd, f"blocks.{i}.", cfg.num_heads)
    return layer_norm(x, sd, "norm.")


def mae_forward_decoder(sd, latent, ids_restore, cfg: ViTConfig) -> Tensor:
    """models_mae.py:172-196."""
    x = F.linear(latent, sd["decoder_embed.weight"], sd["decoder_embed.bias"])
    n_mask = ids_restore.shape[1] + 1 - x.shape[1]
    mask_tokens = sd["mask_token"].repeat(x.shape[0], n_mask, 1)
    x_ = torch.cat([x[:, 1:, :], mask_tokens], dim=1)
    x_ = torch.gather(x_, dim=1, index=ids_restore.unsqueeze(-1).repeat(1, 1, x.shape[2]))
    x = torch.cat([x[:, :1, :], x_], dim=1)
    x = x + sd["decoder_pos_embed"]
    for i in range(cfg.decoder_depth):
        x = block(x, sd, f"decoder_blocks.{i}.", cfg.decoder_num_heads)
    x = layer_norm(x, sd, "decoder_norm.")
    x = F.linear(x, sd["decoder_pred.weight"], sd["decoder_pred.bias"])
    return x[:, 1:, :]


def mae_loss(imgs, pred, mask, cfg: ViTConfig, norm_pix_loss: bool = False) -> Tensor:
    """models_mae.py:198-214 (unbiased variance when norm_pix_loss)."""
    target = patchify(imgs, cfg.patch_size)
    if norm_pix_loss:
        mean = target.mean(dim=-1, keepdim=True)
        var = target.var(dim=-1, keepdim=True)
        target = (target - mean) / (var + 1.0e-6) ** 0.5
    loss = ((pred - target) ** 2).mean(dim=-1)
    return (loss * mask).sum() / mask.sum()


def mae_forward(sd, imgs, noise, cfg: ViTConfig = VIT_BASE, mask_ratio: float = 0.75,
                norm_pix_loss: bool = False):
    """models_mae.py:216-220 -> (loss, pred, mask); `noise` replaces torch.rand (:132)."""
    ids_keep, mask, ids_restore = masking_from_noise(noise, mask_ratio)
    latent = mae_forward_encoder(sd, imgs, ids_keep, cfg)
    pred = mae_forward_decoder(sd, latent, ids_restore, cfg)
    return mae_loss(imgs, pred, mask, cfg, norm_pix_loss), pred, mask


# --------------------------------------------------------------------------
# models.py classifiers
# --------------------------------------------------------------------------
def vit_features(sd, imgs, cfg: ViTConfig, learned_pos: bool) -> Tensor:
    """ViT_from_MAE.forward_encoder (models.py:196-209) when learned_pos=False;
    VisionTransformer_from_Any.forward_features + _pos_embed (models.py:28-33,117-127)
    when True.  The two differ only in whether cls_token gets pos_embed[0] added
    before (MAE) or after (timm) the concat -- numerically the same expression."""
    x = patch_embed(imgs, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], cfg.patch_size)
    if learned_pos:
        x = torch.cat((sd["cls_token"].expand(x.shape[0], -1, -1), x), dim=1)
        x = x + sd["pos_embed"]
    else:
        x = x + sd["pos_embed"][:, 1:, :]
        cls = sd["cls_token"] + sd["pos_embed"][:, :1, :]
        x = torch.cat((cls.expand(x.shape[0], -1, -1), x), dim=1)
    for i in range(cfg.depth):
        x = block(x, sd, f"blocks.{i}.", cfg.num_heads)
    return layer_norm(x, sd, "norm.")


def vit_classify(sd, imgs, cfg: ViTConfig = VIT_BASE, learned_pos: bool = False,
                 out_token: str = "cls", head: bool = True) -> Tensor:
    """models.py:129-140 / 211-222 with dense=None."""
    x = vit_features(sd, imgs, cfg, learned_pos)
    if out_token == "cls":
        x = x[:, 0]
    elif out_token == "spatial":
        x = x[:, 1:].mean(1)
    if head:
        x = F.linear(x, sd["lin_head.weight"], sd["lin_head.bias"])
    return x


def supervised_loss(logits: Tensor, targets: Tensor, pos_weight: float = 1.0,
                    class_weights: Optional[Tensor] = None) -> Tensor:
    """tc.py:3347-3374 + 6086-6104: two classes -> BCEWithLogits(pos_weight) on
    logits[:,1]-logits[:,0]; otherwise weighted cross entropy."""
    if logits.ndim == 2 and logits.size(1) == 2:
        z = logits[:, 1] - logits[:, 0]
        return F.binary_cross_entropy_with_logits(
            z, targets.to(z.dtype), pos_weight=torch.tensor(pos_weight, dtype=z.dtype))
    return F.cross_entropy(logits, targets, weight=class_weights)


# --------------------------------------------------------------------------
# state-dict construction (oracle-owned, platform-stable generator)
# --------------------------------------------------------------------------
def _block_shapes(pre: str, D: int, hidden: int):
    return [
        (pre + "norm1.weight", (D,)), (pre + "norm1.bias", (D,)),
        (pre + "attn.qkv.weight", (3 * D, D)), (pre + "attn.qkv.bias", (3 * D,)),
        (pre + "attn.proj.weight", (D, D)), (pre + "attn.proj.bias", (D,)),
        (pre + "norm2.weight", (D,)), (pre + "norm2.bias", (D,)),
        (pre + "mlp.fc1.weight", (hidden, D)), (pre + "mlp.fc1.bias", (hidden,)),
        (pre + "mlp.fc2.weight", (D, hidden)), (pre + "mlp.fc2.bias", (D,)),
    ]


def param_shapes(cfg: ViTConfig, decoder: bool, n_class: Optional[int]):
    """Ordered (name, shape) list with the reference's state-dict keys."""
    D, p, L = cfg.embed_dim, cfg.patch_size, cfg.num_patches
    out = [("cls_token", (1, 1, D)), ("pos_embed", (1, L + 1, D)),
           ("patch_embed.proj.weight", (D, cfg.in_chans, p, p)), ("patch_embed.proj.bias", (D,))]
    for i in range(cfg.depth):
        out += _block_shapes(f"blocks.{i}.", D, int(D * cfg.mlp_ratio))
    out += [("norm.weight", (D,)), ("norm.bias", (D,))]
    if decoder:
        Dd = cfg.decoder_embed_dim
        out += [("decoder_embed.weight", (Dd, D)), ("decoder_embed.bias", (Dd,)),
                ("mask_token", (1, 1, Dd)), ("decoder_pos_embed", (1, L + 1, Dd))]
        for i in range(cfg.decoder_depth):
            out += _block_shapes(f"decoder_blocks.{i}.", Dd, int(Dd * cfg.mlp_ratio))
        out += [("decoder_norm.weight", (Dd,)), ("decoder_norm.bias", (Dd,)),
                ("decoder_pred.weight", (p * p * cfg.in_chans, Dd)),
                ("decoder_pred.bias", (p * p * cfg.in_chans,))]
    if n_class is not None:
        out += [("lin_head.weight", (n_class, D)), ("lin_head.bias", (n_class,))]
    return out


def generated_state_dict(cfg: ViTConfig, seed: int, decoder: bool = True,
                         n_class: Optional[int] = 2) -> Dict[str, Tensor]:
    """Deterministic, platform-stable weights (numpy PCG64): xavier-uniform-scaled
    matrices, *non-trivial* biases / LN affine (so that every term is exercised),
    sincos positional tables.  Used by fixtures whose weights are too big to commit."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, Tensor] = {}
    for name, shape in param_shapes(cfg, decoder, n_class):
        if name == "pos_embed":
            arr = sincos_2d(cfg.embed_dim, cfg.grid, cls_token=True)[None]
        elif name == "decoder_pos_embed":
            arr = sincos_2d(cfg.decoder_embed_dim, cfg.grid, cls_token=True)[None]
        elif name.endswith(("norm1.weight", "norm2.weight")) or name in ("norm.weight", "decoder_norm.weight"):
            arr = 1.0 + 0.1 * rng.standard_normal(shape)
        elif len(shape) == 1:
            arr = 0.02 * rng.standard_normal(shape)
        elif name in ("cls_token", "mask_token"):
            arr = 0.02 * rng.standard_normal(shape)
        else:
            fan_out = shape[0]
            fan_in = int(np.prod(shape[1:]))
            a = math.sqrt(6.0 / (fan_in + fan_out))
            arr = rng.uniform(-a, a, size=shape)
        sd[name] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).float()
    return sd


def generated_batch(cfg: ViTConfig, batch: int, seed: int):
    """Synthetic post-Normalize batch: imgs~N(0,1), labels~Bern(.5), noise~U[0,1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    imgs = rng.standard_normal((batch, cfg.in_chans, cfg.img_size, cfg.img_size)).astype(np.float32)
    labels = (rng.random(batch) < 0.5).astype(np.int64)
    noise = rng.random((batch, cfg.num_patches)).astype(np.float32)
    return torch.from_numpy(imgs), torch.from_numpy(labels), torch.from_numpy(noise)
