"""Drop-in model classes for the SSL4POLYP hot path, executed by the MI355X engine.

Mirrors (same class names, constructor arguments, attribute surface and state-dict keys):
  src/ssl4polyp/models/mae/models_mae.py:22-250   MaskedAutoencoderViT, mae_vit_{base,large,huge}_*
  src/ssl4polyp/models/models.py:26-140           VisionTransformer_from_Any
  src/ssl4polyp/models/models.py:143-222          ViT_from_MAE
The sub-modules (nn.Conv2d / nn.Linear / nn.LayerNorm) are PARAMETER CONTAINERS with the reference's
names and initialisation order -- their own forward is never called; the arithmetic runs as one autograd
node over the HIP kernels (engine.py).  There is no eager fallback: on a CPU tensor forward raises.
"""
from __future__ import annotations

import math
from functools import partial
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .engine import (BLOCK_PARAM_NAMES, EPI_STORE, BlockStack, Kernels, StackGeom, StackWorkspace, _ptr, _stream)
from .flat import FlatParams

_DEFAULT_PRECISION = "bf16"


# --------------------------------------------------------------------------------------------------
# positional tables (pos_embed.py:20-67): numpy float64, first D/2 channels from grid w, second from h
# --------------------------------------------------------------------------------------------------
def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    assert embed_dim % 2 == 0
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    grid_h = np.arange(grid_size, dtype=np.float32)
    grid_w = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, grid_size, grid_size])
    assert embed_dim % 2 == 0
    emb = np.concatenate([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0]),
                          get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


# --------------------------------------------------------------------------------------------------
# parameter containers with timm 0.4.12's module / parameter names and construction order
# --------------------------------------------------------------------------------------------------
class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


# --------------------------------------------------------------------------------------------------
# runtime: binds a module's parameters to the engine
# --------------------------------------------------------------------------------------------------
class _Runtime:
    """Per-model engine state: kernels, flat storage, block stacks, workspace pool, grad-sync hook."""

    def __init__(self, module: nn.Module, precision: str, enc: StackGeom, dec: Optional[StackGeom]):
        self.module = module
        self.precision = precision
        self.enc_geom, self.dec_geom = enc, dec
        self.k: Optional[Kernels] = None
        self.flat: Optional[FlatParams] = None
        self.pool: Dict[tuple, List[StackWorkspace]] = {}
        self.grad_sync = None  # parallel.GradSync, set by DataParallel wrapper
        self._wcache: Dict[str, tuple] = {}
        self.eps = 1e-6
        # optimizer updates still running on the side stream (FusedAdamW(overlap_forward=True)): (region, lo, hi, event)
        # in launch order; the forward waits for a block's range right before it reads the block's weights
        self.pending_updates: List[tuple] = []
        # parameters of the module that take no part in its forward and therefore never get a gradient, whatever their
        # requires_grad flag says (ViT_from_MAE keeps the deleted decoder's positional table, models.py:171-175)
        self.unused: tuple = ()

    def wait_updates(self, mat_through: Optional[int] = None, also=()) -> None:
        """Make the current stream (and the streams in `also`) wait for pending optimizer updates: all of them
        (default), or every `vec` update and the `mat` updates starting below offset `mat_through`.  The updates run in
        order on one stream, so waiting for the last needed one covers everything launched before it."""
        if not self.pending_updates:
            return
        last = -1
        for j, (r, lo, hi, ev) in enumerate(self.pending_updates):
            if mat_through is None or r == "vec" or lo < mat_through:
                last = j
        if last >= 0:
            ev = self.pending_updates[last][3]
            torch.cuda.current_stream().wait_event(ev)
            for st in also:
                st.wait_event(ev)
            del self.pending_updates[:last + 1]

    def block_end(self, prefix: str, i: int) -> int:
        """End offset (in `mat`) of block i's matrices: what `wait_updates` needs before the block runs."""
        f = self.flat
        j = f.index[f"{prefix}{i}.mlp.fc2.weight"]
        return f.offset[j] + f.numel[j]

    def ensure(self, device: torch.device) -> None:
        if device.type != "cuda":
            raise _lib.PolypMaeError("ssl4polyp_amd models run on the MI355X HIP path only (got a CPU tensor); "
                                     "there is no eager fallback")
        if self.k is None:
            self.k = Kernels(self.precision, self.eps)
        if self.flat is None or set(self.flat.names) != {n for n, _ in self.module.named_parameters()}:
            self.flat = FlatParams(self.module, self.k.act_dtype)
        if not self.flat.bound(device):
            self.flat.materialize(device)
            self.pool.clear()
            self._wcache = {}
        if self.flat.shadow_stale():
            self.wait_updates()
            self.k.cast(self.flat.P["mat"], self.flat.S)
            self.flat.mark_shadow_fresh()
        if self.pending_updates:  # vectors, tokens and everything in front of the first block (patch / decoder embed)
            f = self.flat
            self.wait_updates(mat_through=f.offset[f.index["blocks.0.attn.qkv.weight"]])

    # weights / grads of one block stack as dicts of persistent views
    def stack_weights(self, prefix: str, depth: int):
        hit = self._wcache.get(prefix)
        if hit is not None:
            return hit
        f = self.flat
        W, G = [], []
        for i in range(depth):
            w, g = {}, {}
            for n in BLOCK_PARAM_NAMES:
                full = f"{prefix}{i}.{n}"
                w[n] = f.shadow_view(full) if n.endswith("weight") and "norm" not in n else f.param_view(full)
                g[n] = f.grad_view(full)
            W.append(w)
            G.append(g)
        self._wcache[prefix] = (W, G)
        return W, G

    def get_ws(self, geom: StackGeom, B: int, N: int, training: bool) -> StackWorkspace:
        key = (geom.dim, geom.depth, B, N, training)
        lst = self.pool.setdefault(key, [])
        if lst:
            return lst.pop()
        return StackWorkspace(geom, B, N, self.k.act_dtype, self.flat.device, training)

    def put_ws(self, geom: StackGeom, ws: StackWorkspace) -> None:
        self.pool.setdefault((geom.dim, geom.depth, ws.B, ws.N, ws.training), []).append(ws)


def _update_gate(rt: _Runtime, prefix: str):
    """before_block hook of BlockStack.forward: wait for the optimizer update of block i's weights (if one is pending)."""
    if not rt.pending_updates:
        return None
    return lambda i, also=(): rt.wait_updates(mat_through=rt.block_end(prefix, i), also=also)


def _plan_grads(rt: _Runtime, names: List[str], needs: List[bool]):
    """Decide fresh vs accumulate-in-place for this backward (see flat.py) and prepare the flat ranges."""
    f = rt.flat
    tr = [n for n, need in zip(names, needs) if need and n not in rt.unused]
    flat_state = [f.grad_is_flat(n) for n in tr]
    if tr and all(flat_state):
        return True
    if any(flat_state):
        raise _lib.PolypMaeError("some parameter .grad tensors alias the engine's flat gradient buffer and others do "
                                 "not; call optimizer.zero_grad() for all parameters consistently")
    f.G["vec"].zero_()
    return False


class _EncoderFrontMixin:
    """Patch-embed + token assembly shared by both paths."""

    @staticmethod
    def front_fwd(rt: _Runtime, imgs, ids_keep, keep, pos_name="pos_embed"):
        k, f, mod = rt.k, rt.flat, rt.module
        B, C, img = imgs.shape[0], imgs.shape[1], imgs.shape[2]
        p = mod.patch_embed.patch_size[0]
        D = rt.enc_geom.dim
        PE = C * p * p
        PEp = k.padded_k(PE)  # (patch 14: 588 -> 640, zero columns in both operands)
        dev = imgs.device
        cols = torch.empty(B * keep, PEp, dtype=k.act_dtype, device=dev)
        _lib.check(k.lib.pm_patch_im2col(_ptr(imgs), _ptr(ids_keep), _ptr(cols), PEp, k.act, B, C, img, p, keep, _stream()),
                   "pm_patch_im2col")
        emb = torch.empty(B * keep, D, dtype=torch.float32, device=dev)
        if PEp == PE:
            w = f.shadow_view("patch_embed.proj.weight").view(D, PE)
        else:
            w = k.pad_cast(f.param_view("patch_embed.proj.weight").view(D, PE), D, PEp)
        k.linear_fwd(cols, w, f.param_view("patch_embed.proj.bias"), emb, B * keep, D, PEp)
        x0 = torch.empty(B * (keep + 1), D, dtype=torch.float32, device=dev)
        _lib.check(k.lib.pm_assemble_tokens(_ptr(emb), _ptr(f.param_view("cls_token")), _ptr(f.param_view(pos_name)),
                                            _ptr(ids_keep), _ptr(x0), B, keep, D, _stream()), "pm_assemble_tokens")
        return cols, x0

    @staticmethod
    def front_bwd(rt: _Runtime, dx0, cols, ids_keep, B, keep, accumulate, need, learn_pos):
        k, f = rt.k, rt.flat
        D = rt.enc_geom.dim
        PEp = cols.shape[1]
        PE = f.param_view("patch_embed.proj.weight").numel() // D
        demb = torch.empty(B * keep, D, dtype=k.act_dtype, device=cols.device)
        dcls = f.grad_view("cls_token") if need("cls_token") else None
        dpos = f.grad_view("pos_embed") if learn_pos and need("pos_embed") else None
        if dpos is not None and ids_keep is None:
            # every sample holds every position: d pos_embed = sum_b dx0[b] -- one fixed-order column sum over the batch of
            # the [B, (L+1) D] view (cls row included) instead of the scatter's float atomics
            _lib.check(k.lib.pm_assemble_tokens_bwd(_ptr(dx0), None, _ptr(demb), k.act, _ptr(dcls), None, B, keep, D,
                                                    _stream()), "pm_assemble_tokens_bwd")
            k.colsum(dx0.view(B, (keep + 1) * D), dpos.view(-1), B, (keep + 1) * D)
        else:
            _lib.check(k.lib.pm_assemble_tokens_bwd(_ptr(dx0), _ptr(ids_keep), _ptr(demb), k.act, _ptr(dcls), _ptr(dpos), B,
                                                    keep, D, _stream()), "pm_assemble_tokens_bwd")
        if need("patch_embed.proj.weight"):
            # (its own split-K scratch: the last blocks' weight gradients may still be running on the side stream)
            if PEp == PE:
                k.linear_wgrad(demb, cols, f.grad_view("patch_embed.proj.weight").view(D, PE), B * keep, D, PE, accumulate,
                               ws_name="_ws_front")
            else:  # the gradient in the padded layout, its valid columns into the parameter's gradient
                dw = torch.empty(D, PEp, dtype=torch.float32, device=cols.device)
                k.linear_wgrad(demb, cols, dw, B * keep, D, PEp, False, ws_name="_ws_front")
                k.unpad_add(dw, f.grad_view("patch_embed.proj.weight").view(D, PE), accumulate)
        if need("patch_embed.proj.bias"):
            k.colsum(demb, f.grad_view("patch_embed.proj.bias"), B * keep, D)


class _VitClsFn(torch.autograd.Function):
    """imgs -> logits (head) or features (head=False) for ViT_from_MAE / VisionTransformer_from_Any
    (models.py:129-140, 211-222), one node.  pool: 0 = out_token "cls", 1 = "spatial"."""

    @staticmethod
    def forward(ctx, rt: _Runtime, imgs: torch.Tensor, pool: int, head: bool, names, *params):
        k, f, mod, g = rt.k, rt.flat, rt.module, rt.enc_geom
        B = imgs.shape[0]
        L = mod.patch_embed.num_patches
        N = L + 1
        D = g.dim
        needs = ctx.needs_input_grad[5:] if getattr(rt, "grad_enabled", True) else (False,) * len(params)   # (set by ops._vit_forward)
        # (grad mode is already off inside Function.forward.)  Linear probe (finetune.py mode "none": only lin_head
        # trains): nothing below the head needs saved activations -> the forward-only workspace (2 blocks instead of 12).
        below = any(nd for n, nd in zip(names, needs) if not n.startswith("lin_head"))
        training = any(needs)
        imgs = imgs.contiguous().float()
        cols, x0 = _EncoderFrontMixin.front_fwd(rt, imgs, None, L)
        ws = rt.get_ws(g, B, N, below)
        W, _ = rt.stack_weights("blocks.", g.depth)
        # staged fine-tuning (finetune.py:49-91: head + the last one or two blocks): with the front frozen too, no backward reaches
        # the blocks below the lowest trainable one -- their fc1 pre-activations need not be stored
        need_map = dict(zip(names, needs))
        front_needs = any(need_map.get(n, False) for n in ("cls_token", "pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias"))
        keep_from = 0 if front_needs else min((i for i in range(g.depth)
                                                if any(need_map.get(f"blocks.{i}.{n}", False) for n in BLOCK_PARAM_NAMES)), default=g.depth)
        # out_token "cls" (models.py:134-136): the head reads row 0 of every sample -- behind its attention the top block runs on those
        # rows only, forward and backward (engine.BlockStack._top_block_forward_cls / _top_block_sparse).  Training needs B % 8 == 0
        # (16-byte rows of the k-major weight-gradient operands); otherwise the dense block.
        cls_top = pool == 0 and k.SPARSE_TOP and N > 1 and (B % 8 == 0 or not any(needs))
        x = BlockStack(k, g).forward(ws, x0, W, before_block=_update_gate(rt, "blocks."), keep_from=keep_from, cls_top=cls_top)
        Nh = 1 if cls_top else N   # rows per sample of what the head kernels see
        rt.wait_updates()  # norm / lin_head and anything else still pending
        dev = imgs.device
        f32 = torch.float32
        n_class = mod.lin_head.weight.shape[0] if head else 0
        feat = torch.empty(B, D, dtype=f32, device=dev)
        xhm = torch.empty(B, D, dtype=f32, device=dev) if pool else None
        mean = torch.empty(B * (N if pool else 1), dtype=f32, device=dev)
        rstd = torch.empty_like(mean)
        logits = torch.empty(B, n_class, dtype=f32, device=dev) if head else None
        _lib.check(k.lib.pm_vit_head_fwd(_ptr(x), Nh, pool, _ptr(f.param_view("norm.weight")), _ptr(f.param_view("norm.bias")),
                                         _ptr(f.param_view("lin_head.weight")) if head else None,
                                         _ptr(f.param_view("lin_head.bias")) if head else None, _ptr(feat), _ptr(xhm),
                                         _ptr(mean), _ptr(rstd), _ptr(logits), B, D, n_class, rt.eps, _stream()),
                   "pm_vit_head_fwd")
        if training:
            ctx.rt, ctx.names, ctx.ws = rt, names, ws
            if not below:  # the head's backward reads only the last block's output rows: keep those, release the rest
                x = x.clone() if pool else x.view(B, Nh, D)[:, 0].clone()
                rt.put_ws(g, ws)
                ctx.ws = None
                cols = x0 = None
            ctx.saved = (cols, x0, x, feat, xhm, mean, rstd)
            ctx.dims = (B, L, N, D, n_class, pool, head, below)
            ctx.cls_top = cls_top
        else:
            rt.put_ws(g, ws)
        return logits if head else feat

    @staticmethod
    def backward(ctx, dout):
        rt, names, ws = ctx.rt, ctx.names, ctx.ws
        k, f, mod, g = rt.k, rt.flat, rt.module, rt.enc_geom
        cols, x0, x, feat, xhm, mean, rstd = ctx.saved
        B, L, N, D, n_class, pool, head, below = ctx.dims
        needs = list(ctx.needs_input_grad[5:])
        need_map = dict(zip(names, needs))
        need = lambda n: need_map.get(n, False)
        accumulate = _plan_grads(rt, names, needs)
        if not accumulate and need("lin_head.weight"):
            f.grad_view("lin_head.weight").zero_()
        trainable = [any(need(f"blocks.{i}.{n}") for n in BLOCK_PARAM_NAMES) for i in range(g.depth)]
        # timm's table is learnable (models.py:28-33); the MAE-derived classifier's is a frozen sincos buffer-like Parameter
        # (models_mae.py:37) -- until finetune.py:52-55 (mode "full": requires_grad_(True) on EVERY parameter, applied by
        # tc.py:5740 to whatever the factory returned) makes it trainable too.  Either way: follow the flag.
        learn_pos = need("pos_embed")
        front = need("cls_token") or need("patch_embed.proj.weight") or need("patch_embed.proj.bias") or \
            (learn_pos and need("pos_embed"))
        below_head = front or any(trainable)
        dout = dout.contiguous().float()
        cls_top = bool(getattr(ctx, "cls_top", False))
        if below_head and cls_top:   # the head's gradient on the B cls rows only ([B, 1, D]): what the cls-row top block takes
            tc = BlockStack.top_compact(ws, g, k.act_dtype, dout.device)
            dx, dx_act = tc["dx"], tc["dx_act"]
        else:
            dx = ws.dx[0] if below_head else None
            dx_act = ws.dx_act[0] if below_head else None
        x_rows = 1 if cls_top else (N if (below or pool) else 1)   # linear probe, cls token: only the saved cls rows [B, 1, D]
        _lib.check(k.lib.pm_vit_head_bwd(
            _ptr(dout) if head else None, None if head else _ptr(dout), _ptr(x), x_rows, pool,
            _ptr(f.param_view("norm.weight")), _ptr(f.param_view("lin_head.weight")) if head else None, _ptr(feat), _ptr(xhm),
            _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dx_act), k.act,
            _ptr(f.grad_view("lin_head.weight")) if need("lin_head.weight") else None,
            _ptr(f.grad_view("lin_head.bias")) if need("lin_head.bias") else None,
            _ptr(f.grad_view("norm.weight")) if need("norm.weight") else None,
            _ptr(f.grad_view("norm.bias")) if need("norm.bias") else None, B, D, n_class, _stream()), "pm_vit_head_bwd")
        sync = rt.grad_sync
        if below_head:
            W, G = rt.stack_weights("blocks.", g.depth)
            cb = (lambda i: sync.block_done("blocks.", i)) if sync is not None else None
            # the incoming dx lives in ws.dx[0]; an odd-depth stack would start writing ws.dx[1] first: fine either way
            # the embedding's backward does not read the blocks' weight gradients: it runs BEFORE the main stream joins the
            # side stream, beside the tail block's last weight-gradient launches instead of behind them
            dx0, _ = BlockStack(k, g).backward(ws, x0, W, G, dx, dx_act, False, trainable, front,
                                               lambda n, i: accumulate, cb, defer_join=k.DEFER_JOIN,
                                               sparse_top=True if cls_top else None)
            if front and dx0 is not None:
                _EncoderFrontMixin.front_bwd(rt, dx0, cols, None, B, L, accumulate, need, learn_pos)
            BlockStack.join_deferred(ws)
        if sync is not None:
            sync.backward_done()
        if ws is not None:
            rt.put_ws(g, ws)
        ctx.saved = None
        # decoder_pos_embed survives in ViT_from_MAE (models.py:171-175) but takes no part in the forward: like autograd in the
        # reference it gets NO gradient (None, not zeros -- AdamW must skip it, not decay it) even when mode "full" flags it
        grads = [None if (accumulate or not nd or n in rt.unused) else f.grad_view(n) for n, nd in zip(names, needs)]
        return (None, None, None, None, None, *grads)


class _SupervisedLossFn(torch.autograd.Function):
    """logits [B, n_class], targets int64 [B] -> scalar loss (tc.py:3347-3374, 6086-6104), value and d loss / d logits
    from one launch (pm_supervised_loss_fwd); backward scales the saved gradient by the upstream scalar (pm_scale)."""

    @staticmethod
    def forward(ctx, logits, targets, pos_weight, class_weights):
        if not logits.is_cuda:
            raise _lib.PolypMaeError("supervised_loss runs on the MI355X HIP path only (got a CPU tensor)")
        lib = _lib.load()
        logits = logits.contiguous().float()
        targets = targets.contiguous().to(torch.int64)
        B, C = logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(logits)
        _lib.check(lib.pm_supervised_loss_fwd(_ptr(logits), _ptr(targets), _ptr(pos_weight), _ptr(class_weights), _ptr(loss),
                                              _ptr(dlogits), B, C, _stream()), "pm_supervised_loss_fwd")
        ctx.dlogits = dlogits
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        lib = _lib.load()
        d = ctx.dlogits
        out = torch.empty_like(d)
        dloss = dloss.reshape(1).contiguous().float()
        _lib.check(lib.pm_scale(_ptr(d), _ptr(dloss), _ptr(out), d.numel(), _stream()), "pm_scale")
        return out, None, None, None


def supervised_loss(logits: torch.Tensor, targets: torch.Tensor, pos_weight=None, class_weights=None) -> torch.Tensor:
    """The fine-tune loss of train_classification.py:3347-3374 / 6086-6104 as one HIP op: two classes ->
    BCEWithLogitsLoss(pos_weight) on logits[:,1]-logits[:,0]; more -> CrossEntropyLoss(weight=class_weights).
    pos_weight: python float or a DEVICE scalar tensor; class_weights: f32 [n_class] on the device."""
    dev = logits.device
    if pos_weight is not None and not torch.is_tensor(pos_weight):
        pos_weight = torch.tensor(float(pos_weight), dtype=torch.float32, device=dev)
    if pos_weight is not None:
        pos_weight = pos_weight.to(device=dev, dtype=torch.float32).reshape(1)
    if class_weights is not None:
        class_weights = class_weights.to(device=dev, dtype=torch.float32).contiguous()
        if class_weights.numel() != logits.shape[1]:
            raise ValueError("class_weights must have one entry per class")
    if logits.ndim != 2 or logits.shape[1] < 2:
        raise ValueError("supervised_loss expects logits [B, n_class >= 2]")
    if not logits.is_cuda:
        raise _lib.PolypMaeError("supervised_loss runs on the MI355X HIP path only (got a CPU tensor)")
    from . import ops  # noqa: F401  (registers torch.ops.polypmae.*)
    return torch.ops.polypmae.supervised_loss(logits, targets, pos_weight, class_weights)


class _MaeFn(torch.autograd.Function):
    """imgs (+noise) -> (loss, pred, mask) for MaskedAutoencoderViT.forward (models_mae.py:216-220), one node."""

    @staticmethod
    def forward(ctx, rt: _Runtime, imgs, noise, mask_ratio, names, *params):
        k, f, mod, ge, gd = rt.k, rt.flat, rt.module, rt.enc_geom, rt.dec_geom
        B, C, img = imgs.shape[0], imgs.shape[1], imgs.shape[2]
        p = mod.patch_embed.patch_size[0]
        L = mod.patch_embed.num_patches
        keep = int(L * (1 - mask_ratio))  # models_mae.py:130
        if keep < 1:
            raise ValueError("mask_ratio leaves no visible patch")
        De, Dd, PE = ge.dim, gd.dim, p * p * C
        dev = imgs.device
        f32 = torch.float32
        training = any(ctx.needs_input_grad) and getattr(rt, "grad_enabled", True)  # (grad mode is already off inside Function.forward)
        imgs = imgs.contiguous().float()
        ctx.set_materialize_grads(False)
        # -- masking (models_mae.py:123-148)
        ids_shuffle = torch.empty(B, L, dtype=torch.int32, device=dev)
        ids_restore = torch.empty(B, L, dtype=torch.int32, device=dev)
        mask = torch.empty(B, L, dtype=f32, device=dev)
        noise = noise.contiguous().float()
        _lib.check(k.lib.pm_mae_masking(_ptr(noise), _ptr(ids_shuffle), _ptr(ids_restore), _ptr(mask), B, L, keep,
                                        _stream()), "pm_mae_masking")
        ids_keep = ids_shuffle[:, :keep].contiguous()
        # -- encoder on the kept patches only (the projection is per patch, so gather-then-embed == embed-then-gather)
        cols, x0 = _EncoderFrontMixin.front_fwd(rt, imgs, ids_keep, keep)
        ws_e = rt.get_ws(ge, B, keep + 1, training)
        We, _ = rt.stack_weights("blocks.", ge.depth)
        xe = BlockStack(k, ge).forward(ws_e, x0, We, before_block=_update_gate(rt, "blocks."))
        rt.wait_updates()  # the decoder side is short: take the rest at once
        Me, Md = B * (keep + 1), B * (L + 1)
        latent = torch.empty(Me, De, dtype=k.act_dtype, device=dev)
        mean_e, rstd_e = torch.empty(Me, dtype=f32, device=dev), torch.empty(Me, dtype=f32, device=dev)
        k.layernorm_fwd(xe, f.param_view("norm.weight"), f.param_view("norm.bias"), latent, mean_e, rstd_e, Me, De)
        # -- decoder (models_mae.py:172-196)
        demb = torch.empty(Me, Dd, dtype=f32, device=dev)
        k.linear_fwd(latent, f.shadow_view("decoder_embed.weight"), f.param_view("decoder_embed.bias"), demb, Me, Dd, De)
        xd0 = torch.empty(Md, Dd, dtype=f32, device=dev)
        _lib.check(k.lib.pm_mae_unshuffle(_ptr(demb), _ptr(f.param_view("mask_token")),
                                          _ptr(f.param_view("decoder_pos_embed")), _ptr(ids_restore), _ptr(xd0), B, L,
                                          keep, Dd, _stream()), "pm_mae_unshuffle")
        ws_d = rt.get_ws(gd, B, L + 1, training)
        Wd, _ = rt.stack_weights("decoder_blocks.", gd.depth)
        xd = BlockStack(k, gd).forward(ws_d, xd0, Wd)
        yn = torch.empty(Md, Dd, dtype=k.act_dtype, device=dev)
        mean_d, rstd_d = torch.empty(Md, dtype=f32, device=dev), torch.empty(Md, dtype=f32, device=dev)
        k.layernorm_fwd(xd, f.param_view("decoder_norm.weight"), f.param_view("decoder_norm.bias"), yn, mean_d, rstd_d,
                        Md, Dd)
        pred_full = torch.empty(B, L + 1, PE, dtype=f32, device=dev)
        k.linear_fwd(yn, f.shadow_view("decoder_pred.weight"), f.param_view("decoder_pred.bias"), pred_full, Md, PE, Dd)
        # -- loss (models_mae.py:198-214)
        patch_loss = torch.empty(B * L, dtype=f32, device=dev)
        sums = torch.empty(2, dtype=f32, device=dev)
        loss = torch.empty(1, dtype=f32, device=dev)
        npx = 1 if mod.norm_pix_loss else 0
        _lib.check(k.lib.pm_mae_loss_fwd(_ptr(imgs), _ptr(pred_full), PE, 1, _ptr(patch_loss), B, C, img, p, npx,
                                         _stream()), "pm_mae_loss_fwd")
        _lib.check(k.lib.pm_mae_loss_finish(_ptr(patch_loss), _ptr(mask), B * L, _ptr(sums), _ptr(loss), _stream()),
                   "pm_mae_loss_finish")
        pred = pred_full[:, 1:, :]
        if training:
            ctx.rt, ctx.names = rt, names
            ctx.ws = (ws_e, ws_d)
            ctx.saved = (imgs, ids_keep, ids_shuffle, mask, cols, x0, xe, latent, mean_e, rstd_e, xd0, xd, yn, mean_d,
                         rstd_d, pred_full, sums)
            ctx.dims = (B, C, img, p, L, keep, npx)
            ctx.mark_non_differentiable(mask)
        else:
            rt.put_ws(ge, ws_e)
            rt.put_ws(gd, ws_d)
        return loss.reshape(()), pred, mask

    @staticmethod
    def backward(ctx, dloss, dpred, dmask):
        if dpred is not None:
            raise NotImplementedError("gradients flowing into `pred` from outside the MAE loss are not supported")
        rt, names = ctx.rt, ctx.names
        k, f, mod, ge, gd = rt.k, rt.flat, rt.module, rt.enc_geom, rt.dec_geom
        ws_e, ws_d = ctx.ws
        (imgs, ids_keep, ids_shuffle, mask, cols, x0, xe, latent, mean_e, rstd_e, xd0, xd, yn, mean_d, rstd_d, pred_full,
         sums) = ctx.saved
        B, C, img, p, L, keep, npx = ctx.dims
        De, Dd, PE = ge.dim, gd.dim, p * p * C
        Me, Md = B * (keep + 1), B * (L + 1)
        dev = imgs.device
        needs = list(ctx.needs_input_grad[5:])
        need_map = dict(zip(names, needs))
        need = lambda n: need_map.get(n, False)
        accumulate = _plan_grads(rt, names, needs)
        acc_fn = lambda n, i: accumulate
        sync = rt.grad_sync
        if dloss is None:
            dloss = torch.zeros((), dtype=torch.float32, device=dev)
        dloss = dloss.reshape(1).contiguous().float()
        # -- loss + decoder_pred
        PEp = k.padded_k(PE)  # (patch 14: the 588 outputs of decoder_pred are the reduction dimension of its backward GEMMs)
        dpred_act = torch.empty(Md, PEp, dtype=k.act_dtype, device=dev)
        _lib.check(k.lib.pm_mae_loss_bwd(_ptr(imgs), _ptr(pred_full), PE, 1, _ptr(mask), _ptr(sums), _ptr(dloss),
                                         _ptr(dpred_act), PEp, k.act, B, C, img, p, npx, _stream()), "pm_mae_loss_bwd")
        if PEp == PE:
            if need("decoder_pred.weight"):
                k.linear_wgrad(dpred_act, yn, f.grad_view("decoder_pred.weight"), Md, PE, Dd, accumulate)
            if need("decoder_pred.bias"):
                k.colsum(dpred_act, f.grad_view("decoder_pred.bias"), Md, PE)
            k.linear_dgrad(dpred_act, f.shadow_view("decoder_pred.weight"), ws_d.d_ln, Md, PE, Dd)
        else:
            if need("decoder_pred.weight"):
                dw = torch.empty(PEp, Dd, dtype=torch.float32, device=dev)
                k.linear_wgrad(dpred_act, yn, dw, Md, PEp, Dd, False)
                k.unpad_add(dw, f.grad_view("decoder_pred.weight").view(PE, Dd), accumulate)
            if need("decoder_pred.bias"):
                db = torch.zeros(1, PEp, dtype=torch.float32, device=dev)
                k.colsum(dpred_act, db, Md, PEp)
                k.unpad_add(db, f.grad_view("decoder_pred.bias").view(1, PE), True)  # (vector gradients always accumulate)
            wpad = k.pad_cast(f.param_view("decoder_pred.weight").view(PE, Dd), PEp, Dd)
            k.linear_dgrad(dpred_act, wpad, ws_d.d_ln, Md, PEp, Dd)
        Wd, Gd = rt.stack_weights("decoder_blocks.", gd.depth)
        tr_d = [any(need(f"decoder_blocks.{i}.{n}") for n in BLOCK_PARAM_NAMES) for i in range(gd.depth)]
        last = gd.depth - 1
        k.layernorm_bwd(ws_d.d_ln, xd, f.param_view("decoder_norm.weight"), mean_d, rstd_d, None, ws_d.dx[last & 1 ^ 1],
                        ws_d.dx_act[last & 1 ^ 1],
                        f.grad_view("decoder_norm.weight") if need("decoder_norm.weight") else None,
                        f.grad_view("decoder_norm.bias") if need("decoder_norm.bias") else None,
                        Gd[last]["mlp.fc2.bias"] if tr_d[last] else None, Md, Dd)
        cb_d = (lambda i: sync.block_done("decoder_blocks.", i)) if sync is not None else None
        dxd0, _ = BlockStack(k, gd).backward(ws_d, xd0, Wd, Gd, ws_d.dx[last & 1 ^ 1], ws_d.dx_act[last & 1 ^ 1], True, tr_d,
                                             True, acc_fn, cb_d, ends_pass=False)
        # -- un-shuffle + decoder_embed
        demb_act = torch.empty(Me, Dd, dtype=k.act_dtype, device=dev)
        # partial rows of the mask-token sum (fixed-order second stage)
        cs_ws = k._scratch("_ws_cs_main", k._need(("unshuf", Dd), lambda: k.lib.pm_workspace_bytes(_lib.WS_UNSHUFFLE_BWD, 1, Dd)), dev)
        _lib.check(k.lib.pm_mae_unshuffle_bwd(_ptr(dxd0), _ptr(ids_shuffle), _ptr(demb_act), k.act,
                                              _ptr(f.grad_view("mask_token")) if need("mask_token") else None, B, L, keep,
                                              Dd, _ptr(cs_ws), cs_ws.numel(), _stream()), "pm_mae_unshuffle_bwd")
        if need("decoder_embed.weight"):
            k.linear_wgrad(demb_act, latent, f.grad_view("decoder_embed.weight"), Me, Dd, De, accumulate)
        if need("decoder_embed.bias"):
            k.colsum(demb_act, f.grad_view("decoder_embed.bias"), Me, Dd)
        k.linear_dgrad(demb_act, f.shadow_view("decoder_embed.weight"), ws_e.d_ln, Me, Dd, De)
        # -- encoder
        We, Ge = rt.stack_weights("blocks.", ge.depth)
        tr_e = [any(need(f"blocks.{i}.{n}") for n in BLOCK_PARAM_NAMES) for i in range(ge.depth)]
        last = ge.depth - 1
        k.layernorm_bwd(ws_e.d_ln, xe, f.param_view("norm.weight"), mean_e, rstd_e, None, ws_e.dx[last & 1 ^ 1],
                        ws_e.dx_act[last & 1 ^ 1], f.grad_view("norm.weight") if need("norm.weight") else None,
                        f.grad_view("norm.bias") if need("norm.bias") else None,
                        Ge[last]["mlp.fc2.bias"] if tr_e[last] else None, Me, De)
        front = need("cls_token") or need("patch_embed.proj.weight") or need("patch_embed.proj.bias")
        cb_e = (lambda i: sync.block_done("blocks.", i)) if sync is not None else None
        dx0, _ = BlockStack(k, ge).backward(ws_e, x0, We, Ge, ws_e.dx[last & 1 ^ 1], ws_e.dx_act[last & 1 ^ 1], True, tr_e,
                                            front, acc_fn, cb_e, defer_join=k.DEFER_JOIN)
        if front and dx0 is not None:
            _EncoderFrontMixin.front_bwd(rt, dx0, cols, ids_keep, B, keep, accumulate, need, False)
        BlockStack.join_deferred(ws_e)
        if sync is not None:
            sync.backward_done()
        rt.put_ws(ge, ws_e)
        rt.put_ws(gd, ws_d)
        ctx.saved = None
        grads = [None if (accumulate or not nd) else f.grad_view(n) for n, nd in zip(names, needs)]
        return (None, None, None, None, None, *grads)


# --------------------------------------------------------------------------------------------------
# public modules
# --------------------------------------------------------------------------------------------------
class MaskedAutoencoderViT(nn.Module):
    """Masked Autoencoder with VisionTransformer backbone (models_mae.py:22-220), HIP-executed."""

    def state_dict(self, *args, **kwargs):
        self._rt.wait_updates()  # an overlapped optimizer update may still be writing the parameters
        return super().state_dict(*args, **kwargs)

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4.0, norm_layer=nn.LayerNorm,
                 norm_pix_loss=False, precision: Optional[str] = None):
        super().__init__()
        eps = self._check_norm(norm_layer)
        # construction order follows models_mae.py:31-59 so that a given torch seed yields identical weights
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer)
                                     for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([
            Block(decoder_embed_dim, decoder_num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer)
            for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.decoder_pred = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
        self.norm_pix_loss = norm_pix_loss
        self.initialize_weights()
        enc = StackGeom(embed_dim, num_heads, depth, int(embed_dim * mlp_ratio))
        dec = StackGeom(decoder_embed_dim, decoder_num_heads, decoder_depth, int(decoder_embed_dim * mlp_ratio))
        object.__setattr__(self, "_rt", _Runtime(self, precision or _DEFAULT_PRECISION, enc, dec))
        self._rt.eps = eps

    @staticmethod
    def _check_norm(norm_layer):
        probe = norm_layer(8)
        if not isinstance(probe, nn.LayerNorm) or not probe.elementwise_affine:
            raise ValueError("only affine nn.LayerNorm is supported by the HIP path")
        return float(probe.eps)

    def initialize_weights(self):
        """models_mae.py:65-82 (same RNG consumption order)."""
        gs = int(self.patch_embed.num_patches ** 0.5)
        pos_embed = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], gs, cls_token=True)
        self.pos_embed.data.copy_(torch.from_numpy(pos_embed).float().unsqueeze(0))
        dpos = get_2d_sincos_pos_embed(self.decoder_pos_embed.shape[-1], gs, cls_token=True)
        self.decoder_pos_embed.data.copy_(torch.from_numpy(dpos).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        torch.nn.init.normal_(self.cls_token, std=0.02)
        torch.nn.init.normal_(self.mask_token, std=0.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        """models_mae.py:84-93."""
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # Pure data-movement helpers of the reference's public surface (models_mae.py:95-121: imgs [N, 3, H, W] <-> patches
    # [N, L, p*p*3] with the channel fastest inside a patch).  Off the hot path -- the loss kernel fuses this layout
    # (pm_mae_loss_fwd) -- and written as one view + permute each way; the reference-made ramp fixture pins the pixel order.
    def patchify(self, imgs):
        p = self.patch_embed.patch_size[0]
        n, c, hh, ww = imgs.shape
        if hh != ww or hh % p or c != 3:
            raise ValueError(f"patchify expects square 3-channel images whose side is a multiple of {p} (got {tuple(imgs.shape)})")
        g = hh // p
        return imgs.view(n, c, g, p, g, p).permute(0, 2, 4, 3, 5, 1).reshape(n, g * g, p * p * c)

    def unpatchify(self, x):
        p = self.patch_embed.patch_size[0]
        n, L, _ = x.shape
        g = math.isqrt(L)
        if g * g != L:
            raise ValueError(f"unpatchify expects a square number of patches (got {L})")
        return x.view(n, g, g, p, p, 3).permute(0, 5, 1, 3, 2, 4).reshape(n, 3, g * p, g * p)

    def _draw_noise(self, B: int, device) -> torch.Tensor:
        """The masking noise of models_mae.py:132 (`torch.rand(N, L, device=x.device)`) from the library's counter-based generator
        (pm_mae_noise: Philox4x32-10), keyed like the draw it replaces: seed and running offset of torch's DEVICE generator
        (what `torch.manual_seed` / main_pretrain.py:147 `seed + rank` set), and the offset is advanced by the numbers drawn, as
        torch.rand would.  So the noise is a pure function of (seed, numbers drawn before on this device, element): re-seeding
        replays the same masks, consecutive forwards differ, ranks differ by their seeds -- and no ATen kernel runs."""
        if torch.cuda.is_current_stream_capturing():
            raise _lib.PolypMaeError("MAE forward under stream capture: the masking draw would be baked into the graph (the same "
                                     "masks at every replay) -- pass `noise=` from a buffer the caller refills between replays")
        L = self.patch_embed.num_patches
        n = B * L
        gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        seed = int(gen.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        off = int(gen.get_offset())
        gen.set_offset(off + 4 * ((n + 3) // 4))   # (multiples of 4, as the generator requires)
        # the 64-bit offset picks the stream: low word = pm_mae_noise's stream id, high word folded into the key
        key = (seed ^ ((off >> 32) * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
        noise = torch.empty(B, L, dtype=torch.float32, device=device)
        _lib.check(self._rt.k.lib.pm_mae_noise(_ptr(noise), n, key, off & 0xFFFFFFFF, _stream()), "pm_mae_noise")
        return noise

    def forward(self, imgs, mask_ratio=0.75, noise: Optional[torch.Tensor] = None):
        """-> (loss, pred [N, L, p*p*3], mask [N, L]).  `noise` replaces the generator's draw (tests; models_mae.py:132)."""
        rt = self._rt
        rt.ensure(imgs.device)
        assert imgs.shape[2] == self.patch_embed.img_size[0] and imgs.shape[3] == self.patch_embed.img_size[1]
        if noise is None:
            noise = self._draw_noise(imgs.shape[0], imgs.device)
        from . import ops  # registers torch.ops.polypmae.* on first use
        return torch.ops.polypmae.mae_forward(imgs, noise, float(mask_ratio), ops.register_runtime(rt), list(rt.flat.params))


def mae_vit_base_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=16, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512,
                                decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_large_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=16, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512,
                                decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_huge_patch14_dec512d8b(**kwargs):
    """models_mae.py:239-244: ViT-H/14 -- D = 1280, 32 blocks, 16 heads of 80, 256 + 1 tokens at 224^2, a 588-element patch.
    What differs from the base path: 80-wide heads run the attention kernels on LDS rows padded to 128 (3 feature tiles, the
    last half empty), N = 257 takes nine 32-row tiles, LayerNorm keeps 5 float4 per lane, and the two Linears whose reduction
    dimension is 588 (PatchEmbed.proj forward / weight gradient, decoder_pred backward) run zero-padded to 640."""
    return MaskedAutoencoderViT(patch_size=14, embed_dim=1280, depth=32, num_heads=16, decoder_embed_dim=512,
                                decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


mae_vit_base_patch16 = mae_vit_base_patch16_dec512d8b
mae_vit_large_patch16 = mae_vit_large_patch16_dec512d8b
mae_vit_huge_patch14 = mae_vit_huge_patch14_dec512d8b


class _ClassifierBase(nn.Module):
    def state_dict(self, *args, **kwargs):
        self._rt.wait_updates()  # an overlapped optimizer update may still be writing the parameters
        return super().state_dict(*args, **kwargs)

    def _classify(self, imgs):
        if self.dense:
            raise NotImplementedError("dense (DPT decoder) mode is dead code on the reference's path (every caller "
                                      "passes dense=None, tc.py:5646-5654) and is not built")
        if self.out_token not in ("cls", "spatial"):
            raise ValueError(f"out_token must be 'cls' or 'spatial' (got {self.out_token!r})")  # models.py:134-137,216-219
        head = bool(self.head if isinstance(self.head, bool) else self.head_bool)
        rt = self._rt
        rt.ensure(imgs.device)
        assert imgs.shape[2] == self.patch_embed.img_size[0] and imgs.shape[3] == self.patch_embed.img_size[1]
        from . import ops  # registers torch.ops.polypmae.* on first use
        return torch.ops.polypmae.vit_forward(imgs, ops.register_runtime(rt), 1 if self.out_token == "spatial" else 0, head,
                                              list(rt.flat.params))


class ViT_from_MAE(_ClassifierBase):
    """models.py:143-222: the MAE encoder (decoder deleted) + lin_head.  `.head` is a bool (models.py:177);
    `decoder_pos_embed` survives in the state dict exactly as in the reference."""

    def __init__(self, weight_path, head, num_classes, frozen, dense, embed_dim, depth, num_heads, out_token,
                 precision: Optional[str] = None):
        super().__init__()
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        # build the full MAE first (models.py:155-165) so that seeded initialisation matches, then drop the decoder
        full = MaskedAutoencoderViT(patch_size=16, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                                    decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                    norm_layer=norm_layer)
        if weight_path is not None:
            # the MAE checkpoint also pickles an argparse.Namespace under "args" (misc.py:311-318): a trusted local file,
            # read the way torch 1.9's torch.load did (full unpickle)
            weights = torch.load(weight_path, map_location="cpu", weights_only=False)["model"]
            own = full.state_dict()
            n = 0
            for name, param in weights.items():
                if name in own:
                    own[name].copy_(param)
                    n += 1
            print(f"Successfully loaded params for {n} items")
        self.patch_embed = full.patch_embed
        self.cls_token = full.cls_token
        self.pos_embed = full.pos_embed
        self.blocks = full.blocks
        self.norm = full.norm
        self.decoder_pos_embed = full.decoder_pos_embed
        self.norm_pix_loss = False
        self.head = head
        if head:
            self.lin_head = nn.Linear(embed_dim, num_classes)
        self.frozen = frozen
        self.dense = dense
        self.out_token = out_token
        self._learned_pos = False
        enc = StackGeom(embed_dim, num_heads, depth, int(embed_dim * 4))
        object.__setattr__(self, "_rt", _Runtime(self, precision or _DEFAULT_PRECISION, enc, None))
        self._rt.unused = ("decoder_pos_embed",)

    def load_my_state_dict(self, state_dict):
        own_state = self.state_dict()
        i = 0
        for name, param in state_dict.items():
            if name not in own_state:
                continue
            own_state[name].copy_(param)
            i += 1
        print(f"Successfully loaded params for {i} items")

    def forward(self, imgs):
        return self._classify(imgs)


class VisionTransformer_from_Any(_ClassifierBase):
    """models.py:26-140: timm VisionTransformer (learnable pos_embed, trunc-normal init) + lin_head."""

    def __init__(self, head, num_classes, frozen, dense, embed_dim, depth, num_heads, out_token, ImageNet_weights=False,
                 precision: Optional[str] = None):
        super().__init__()
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(224, 16, 3, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=0.0)
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, 4.0, qkv_bias=True, norm_layer=norm_layer)
                                      for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        _timm_head = nn.Linear(embed_dim, 1000)  # consumed RNG like timm's classifier head, replaced below
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        nn.init.zeros_(_timm_head.weight)  # timm 0.4.12 _init_vit_weights: name.startswith('head') -> zeros
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.zeros_(m.bias)
                nn.init.ones_(m.weight)
        if ImageNet_weights:
            # models.py:51-55 downloads B_16-i21k-...-res_224.npz from storage.googleapis.com; there is no network on this
            # path, so the same file is taken from a local path: ImageNet_weights="<file.npz>" or $SSL4POLYP_AUGREG_NPZ
            import os
            loc = ImageNet_weights if isinstance(ImageNet_weights, (str, os.PathLike)) else os.environ.get("SSL4POLYP_AUGREG_NPZ")
            if not loc or not os.path.exists(str(loc)):
                raise FileNotFoundError("ImageNet_weights=True needs the augreg ViT-B/16 .npz of models.py:51-55 on local disk: "
                                        "pass its path as ImageNet_weights or set SSL4POLYP_AUGREG_NPZ (no download here)")
            self.load_pretrained(loc)
        self.head = nn.Identity()
        self.head_bool = head
        if head:
            self.lin_head = nn.Linear(embed_dim, num_classes)
        self.frozen = frozen
        self.dense = dense
        self.out_token = out_token
        self._learned_pos = True
        enc = StackGeom(embed_dim, num_heads, depth, int(embed_dim * 4))
        object.__setattr__(self, "_rt", _Runtime(self, precision or _DEFAULT_PRECISION, enc, None))

    def load_pretrained(self, checkpoint_path):
        """models.py:68-115 for torch checkpoints (dict with state_dict/model/... or a bare state dict)."""
        if str(checkpoint_path).lower().endswith(".npz"):
            return load_augreg_npz(self, str(checkpoint_path))
        checkpoint = torch.load(str(checkpoint_path), map_location="cpu", weights_only=False)
        state_dict = None
        if isinstance(checkpoint, dict):
            for key in ("state_dict", "model", "model_state", "weights", "params"):
                if isinstance(checkpoint.get(key), dict):
                    state_dict = checkpoint[key]
                    break
            if state_dict is None and all(isinstance(k, str) for k in checkpoint.keys()):
                state_dict = checkpoint
        if not isinstance(state_dict, dict):
            print(f"Warning: unsupported checkpoint format for {checkpoint_path!s}: {type(checkpoint)}")
            return
        missing, unexpected = self.load_state_dict(state_dict, strict=False)
        if missing:
            print(f"Missing keys when loading pretrained weights: {missing}")
        if unexpected:
            print(f"Unexpected keys when loading pretrained weights: {unexpected}")

    def forward(self, x):
        return self._classify(x)


def load_augreg_npz(model, path: str, prefix: str = "") -> None:
    """Official JAX / augreg ViT checkpoint (.npz) -> this module's parameters: the mapping of timm 0.4.12
    ``vision_transformer._load_weights`` (absent dependency, restated), which the reference calls for SUP-imnet
    (models.py:68-77).  Dense kernels are stored [in, out], attention query/key/value kernels [D, H, dh], the
    attention output kernel [H, dh, D], the patch embedding HWIO."""
    w = np.load(path)
    if not prefix and "opt/target/embedding/kernel" in w:
        prefix = "opt/target/"

    def t(a):  # numpy kernel -> torch weight layout
        if a.ndim == 4:
            a = a.transpose(3, 2, 0, 1)
        elif a.ndim == 3:
            a = a.transpose(2, 0, 1)
        elif a.ndim == 2:
            a = a.transpose(1, 0)
        return torch.from_numpy(np.ascontiguousarray(a))

    def raw(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    with torch.no_grad():
        ew = t(w[f"{prefix}embedding/kernel"])
        if ew.shape != model.patch_embed.proj.weight.shape:
            raise ValueError(f"patch embedding {tuple(ew.shape)} does not fit {tuple(model.patch_embed.proj.weight.shape)}")
        model.patch_embed.proj.weight.copy_(ew)
        model.patch_embed.proj.bias.copy_(raw(w[f"{prefix}embedding/bias"]))
        model.cls_token.copy_(raw(w[f"{prefix}cls"]).reshape(model.cls_token.shape))
        pos = raw(w[f"{prefix}Transformer/posembed_input/pos_embedding"])
        if pos.shape != model.pos_embed.shape:
            raise ValueError(f"pos_embedding {tuple(pos.shape)} != {tuple(model.pos_embed.shape)} (bicubic resize of the "
                             "table is not on this path: the reference runs at 224^2 / patch 16)")
        model.pos_embed.copy_(pos)
        model.norm.weight.copy_(raw(w[f"{prefix}Transformer/encoder_norm/scale"]))
        model.norm.bias.copy_(raw(w[f"{prefix}Transformer/encoder_norm/bias"]))
        for i, blk in enumerate(model.blocks.children()):
            bp = f"{prefix}Transformer/encoderblock_{i}/"
            mp = bp + "MultiHeadDotProductAttention_1/"
            blk.norm1.weight.copy_(raw(w[bp + "LayerNorm_0/scale"]))
            blk.norm1.bias.copy_(raw(w[bp + "LayerNorm_0/bias"]))
            blk.attn.qkv.weight.copy_(torch.cat([raw(w[f"{mp}{n}/kernel"]).flatten(1).T for n in ("query", "key", "value")]))
            blk.attn.qkv.bias.copy_(torch.cat([raw(w[f"{mp}{n}/bias"]).reshape(-1) for n in ("query", "key", "value")]))
            blk.attn.proj.weight.copy_(t(w[mp + "out/kernel"]).flatten(1))
            blk.attn.proj.bias.copy_(raw(w[mp + "out/bias"]))
            for r in range(2):
                fc = getattr(blk.mlp, f"fc{r + 1}")
                fc.weight.copy_(t(w[f"{bp}MlpBlock_3/Dense_{r}/kernel"]))
                fc.bias.copy_(raw(w[f"{bp}MlpBlock_3/Dense_{r}/bias"]))
            blk.norm2.weight.copy_(raw(w[bp + "LayerNorm_2/scale"]))
            blk.norm2.bias.copy_(raw(w[bp + "LayerNorm_2/bias"]))


# factories with the reference's names / signatures (src/ssl4polyp/utils/__init__.py:29-67)
def get_MAE_backbone(weight_path, head, num_classes, frozen, dense, out_token="cls", precision=None):
    return ViT_from_MAE(weight_path, head, num_classes, frozen, dense, embed_dim=768, depth=12, num_heads=12,
                        out_token=out_token, precision=precision)


def get_ImageNet_or_random_ViT(head, num_classes, frozen, dense, ImageNet_weights, out_token="cls", precision=None):
    return VisionTransformer_from_Any(head, num_classes, frozen, dense, 768, 12, 12, out_token, ImageNet_weights,
                                      precision=precision)
