"""`torch.ops.polypmae.*` -- the HIP kernels registered as custom torch ops over the C-ABI (include/polypmae.h).

North-star wording: "hand-written HIP for CDNA4 exposed as custom torch ops through a thin C-ABI extension".  Two layers:

  * whole-model ops, the ones the drop-in modules call once per step:
        polypmae::vit_forward(imgs, runtime, pool, head, params) -> logits | features       (models.py:129-140, 211-222)
        polypmae::mae_forward(imgs, noise, mask_ratio, runtime, params) -> (loss, pred, mask) (models_mae.py:216-220)
        polypmae::supervised_loss(logits, targets, pos_weight?, class_weights?) -> loss       (tc.py:3347-3374, 6086-6104)
    Their autograd kernels are the one-node autograd.Functions of models.py (the engine owns every intermediate buffer,
    so a step is one node, not ~200);
  * per-kernel functional ops with autograd, for callers that compose their own blocks (the list of SURVEY 8-b):
        polypmae::layernorm, ::linear_bias, ::linear_bias_gelu, ::linear_bias_residual, ::attention.

Inside a step the engine calls the C-ABI directly (ctypes, ~3 us per launch): going through the dispatcher for each of the
~330 launches of a ViT-B step costs ~12 us per op on this host (measured, DESIGN.md "Boundary"), i.e. ~4 ms per step against
a 12 ms step -- the op registry is the public surface, not the inner loop.
There is no CPU kernel registered for any of these ops: on a CPU tensor the dispatcher itself raises.
"""
from __future__ import annotations

import weakref
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from .engine import EPI_GELU, EPI_RESIDUAL, EPI_STORE, Kernels, _ptr, _stream

_DEF = torch.library.Library("polypmae", "DEF")
_RUNTIMES: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()
_KERNELS = {}


def register_runtime(rt) -> int:
    """Handle under which a module's engine state (_Runtime) is passed through the dispatcher (ops take tensors and scalars)."""
    h = id(rt)
    _RUNTIMES[h] = rt
    return h


def _rt(handle: int):
    rt = _RUNTIMES.get(handle)
    if rt is None:
        raise _lib.PolypMaeError("polypmae: stale runtime handle (the module that owned it is gone)")
    return rt


def _k(t_or_dtype, eps: float = 1e-6) -> Kernels:
    dt = t_or_dtype.dtype if torch.is_tensor(t_or_dtype) else t_or_dtype
    key = ("bf16" if dt == torch.bfloat16 else "fp16" if dt == torch.float16 else "fp32", float(eps))
    k = _KERNELS.get(key)
    if k is None:
        k = _KERNELS[key] = Kernels(key[0], key[1])
    return k


# ---------------------------------------------------------------------------------------------------------------------
# whole-model ops
# ---------------------------------------------------------------------------------------------------------------------
_DEF.define("vit_forward(Tensor imgs, int runtime, int pool, bool head, Tensor[] params) -> Tensor")
_DEF.define("mae_forward(Tensor imgs, Tensor noise, float mask_ratio, int runtime, Tensor[] params) -> (Tensor, Tensor, Tensor)")
_DEF.define("supervised_loss(Tensor logits, Tensor targets, Tensor? pos_weight, Tensor? class_weights) -> Tensor")


def _vit_forward(imgs, runtime, pool, head, params):
    from .models import _VitClsFn
    rt = _rt(runtime)
    # grad mode is switched off inside Function.forward, and ctx.needs_input_grad follows the parameters' requires_grad flags whatever
    # the caller's mode: an evaluation pass under torch.no_grad() over a model with trainable blocks must still take the forward-only path
    rt.grad_enabled = torch.is_grad_enabled()
    return _VitClsFn.apply(rt, imgs, pool, head, rt.flat.names, *params)


def _mae_forward(imgs, noise, mask_ratio, runtime, params):
    from .models import _MaeFn
    rt = _rt(runtime)
    rt.grad_enabled = torch.is_grad_enabled()
    return _MaeFn.apply(rt, imgs, noise, float(mask_ratio), rt.flat.names, *params)


def _supervised_loss(logits, targets, pos_weight, class_weights):
    from .models import _SupervisedLossFn
    return _SupervisedLossFn.apply(logits, targets, pos_weight, class_weights)


# AutogradCUDA: the kernel that runs for CUDA (HIP) tensors whether or not a gradient is required; it builds the
# one-node graph itself.  No CPU / CompositeImplicit registration exists: a CPU tensor fails in the dispatcher.
_IMPL = torch.library.Library("polypmae", "IMPL")
def _impl(name, fn):
    # AutogradCUDA: tensors that may require a gradient; CUDA: the same kernel when autograd dispatch is excluded
    # (inference_mode, or a call from inside another autograd.Function).  Nothing is registered for CPU.
    _IMPL.impl(name, fn, "AutogradCUDA")
    _IMPL.impl(name, fn, "CUDA")


_impl("vit_forward", _vit_forward)
_impl("mae_forward", _mae_forward)
_impl("supervised_loss", _supervised_loss)


# ---------------------------------------------------------------------------------------------------------------------
# per-kernel functional ops (autograd.Function over the C-ABI, registered the same way)
# ---------------------------------------------------------------------------------------------------------------------
class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, bf16_out, out_dtype=None):
        x = x.contiguous().float()
        D = x.shape[-1]
        M = x.numel() // D
        k = _k(out_dtype if out_dtype is not None else (torch.bfloat16 if bf16_out else torch.float32), eps)
        y = torch.empty(x.shape, dtype=k.act_dtype, device=x.device)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        k.layernorm_fwd(x, gamma.float().contiguous(), beta.float().contiguous(), y, mean, rstd, M, D)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.k = k
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        k = ctx.k
        D = x.shape[-1]
        M = x.numel() // D
        dy = dy.contiguous().to(k.act_dtype)
        dx = torch.empty_like(x)
        dg, db = torch.zeros(D, device=x.device), torch.zeros(D, device=x.device)
        k.layernorm_bwd(dy, x, gamma.float().contiguous(), mean, rstd, None, dx, None, dg, db, None, M, D)
        return dx, dg, db, None, None, None


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b with the fused epilogues of pm_gemm: plain, erf-GELU (pre-activation kept for backward), f32 residual."""

    @staticmethod
    def forward(ctx, x, W, bias, resid, mode):
        k = _k(x)
        x, W = x.contiguous(), W.contiguous().to(x.dtype)
        K = x.shape[-1]
        M, N = x.numel() // K, W.shape[0]
        b = bias.float().contiguous() if bias is not None else None
        aux = None
        if mode == EPI_RESIDUAL:
            out = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)
            k.linear_fwd(x, W, b, out, M, N, K, EPI_RESIDUAL, resid=resid.contiguous().float())
        elif mode == EPI_GELU:
            out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
            aux = torch.empty_like(out)
            k.linear_fwd(x, W, b, out, M, N, K, EPI_GELU, aux=aux)
        else:
            out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
            k.linear_fwd(x, W, b, out, M, N, K)
        ctx.save_for_backward(x, W, aux)
        ctx.k, ctx.mode, ctx.has_bias = k, mode, bias is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        from ._lib import EPI_DGELU
        x, W, aux = ctx.saved_tensors
        k = ctx.k
        K = x.shape[-1]
        M, N = x.numel() // K, W.shape[0]
        dres = dy if ctx.mode == EPI_RESIDUAL else None
        dy = dy.contiguous().to(x.dtype)
        if ctx.mode == EPI_GELU:  # dy * gelu'(saved pre-activation): one elementwise HIP pass (pm_dgelu)
            dact = torch.empty_like(dy)
            _lib.check(k.lib.pm_dgelu(_ptr(dy), _ptr(aux), _ptr(dact), _lib.dtype_code(dy.dtype), dy.numel(), _stream()), "pm_dgelu")
            dy = dact
        dx = torch.empty_like(x)
        k.linear_dgrad(dy, W, dx, M, N, K)
        dW = torch.empty(N, K, dtype=torch.float32, device=x.device)
        k.linear_wgrad(dy, x, dW, M, N, K, False)
        db = None
        if ctx.has_bias:
            db = torch.zeros(N, dtype=torch.float32, device=x.device)
            k.colsum(dy, db, M, N)
        return dx, dW, db, dres, None


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, heads):
        k = _k(qkv)
        B, N, C3 = qkv.shape
        D = C3 // 3
        dh = D // heads
        qkv = qkv.contiguous()
        out = torch.empty(B, N, D, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B * heads * N, dtype=torch.float32, device=qkv.device)
        k.attention_fwd(qkv, out, lse, B, N, heads, dh)
        ctx.save_for_backward(qkv, out, lse)
        ctx.k, ctx.heads = k, heads
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        k, H = ctx.k, ctx.heads
        B, N, C3 = qkv.shape
        dh = C3 // 3 // H
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        k.attention_bwd(qkv, out, dout.contiguous().to(qkv.dtype), lse, delta, dqkv, B, N, H, dh)
        return dqkv, None


_DEF.define("layernorm(Tensor x, Tensor gamma, Tensor beta, float eps=1e-6, bool bf16_out=True, ScalarType? out_dtype=None) -> Tensor")
_DEF.define("linear_bias(Tensor x, Tensor weight, Tensor? bias) -> Tensor")
_DEF.define("linear_bias_gelu(Tensor x, Tensor weight, Tensor? bias) -> Tensor")
_DEF.define("linear_bias_residual(Tensor x, Tensor weight, Tensor? bias, Tensor residual) -> Tensor")
_DEF.define("attention(Tensor qkv, int heads) -> Tensor")
_impl("layernorm", lambda x, g, b, eps=1e-6, bf16_out=True, out_dtype=None: _LayerNormFn.apply(x, g, b, float(eps), bool(bf16_out), out_dtype))
_impl("linear_bias", lambda x, w, b: _LinearFn.apply(x, w, b, None, EPI_STORE))
_impl("linear_bias_gelu", lambda x, w, b: _LinearFn.apply(x, w, b, None, EPI_GELU))
_impl("linear_bias_residual", lambda x, w, b, r: _LinearFn.apply(x, w, b, r, EPI_RESIDUAL))
_impl("attention", lambda qkv, heads: _AttentionFn.apply(qkv, int(heads)))

OP_NAMES = ("vit_forward", "mae_forward", "supervised_loss", "layernorm", "linear_bias", "linear_bias_gelu",
            "linear_bias_residual", "attention")
