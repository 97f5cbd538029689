"""Flat parameter / gradient / shadow storage for the engine.

Parameters stay ordinary ``nn.Parameter`` objects under the reference's state-dict key names
 but their storage is two contiguous f32 ranges on the device:

    vec : every parameter with ndim <= 1, the cls / mask tokens and the positional tables
    mat : every matrix (Linear / patch-embed weights), in forward order

so that (i) bf16 shadow copies for the MFMA path are one cast over ``mat``, (ii) the gradient of
every matrix is written by its wgrad GEMM straight into a flat f32 gradient range that RCCL
all-reduces bucket-by-bucket without packing, (iii) the "+=" gradient targets (bias / LayerNorm / token
gradients, written with atomics) are zeroed by a single memset over ``vec``, and (iv) the fused AdamW
runs over a handful of contiguous segments instead of ~150 tensors.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn


def _is_vec(name: str, p: torch.Tensor) -> bool:
    return p.ndim <= 1 or name.endswith(("cls_token", "mask_token", "pos_embed"))


class FlatParams:
    ALIGN = 64  # elements; keeps every segment 256-B aligned (16-B vector access, bucket boundaries)

    def __init__(self, module: nn.Module, act_dtype: torch.dtype):
        self.module = module
        self.act_dtype = act_dtype
        self.names: List[str] = []
        self.params: List[nn.Parameter] = []
        self.region: List[str] = []
        self.offset: List[int] = []
        self.numel: List[int] = []
        sizes = {"vec": 0, "mat": 0}
        for name, p in module.named_parameters():
            r = "vec" if _is_vec(name, p) else "mat"
            self.names.append(name)
            self.params.append(p)
            self.region.append(r)
            self.offset.append(sizes[r])
            self.numel.append(p.numel())
            sizes[r] += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.sizes = sizes
        self.index = {n: i for i, n in enumerate(self.names)}
        self.device: Optional[torch.device] = None
        self.P: Dict[str, torch.Tensor] = {}
        self.G: Dict[str, torch.Tensor] = {}
        self.S: Optional[torch.Tensor] = None  # act-typed shadow of P["mat"] (bf16 mode)
        self._shadow_versions: Optional[Tuple[int, ...]] = None

    # ------------------------------------------------------------------------------------------
    def materialize(self, device: torch.device) -> None:
        """(Re)bind every parameter's storage to the flat ranges on `device` (after .to()/load)."""
        P = {r: torch.zeros(max(n, self.ALIGN), dtype=torch.float32, device=device) for r, n in self.sizes.items()}
        with torch.no_grad():
            for i, p in enumerate(self.params):
                view = P[self.region[i]][self.offset[i]: self.offset[i] + self.numel[i]].view(p.shape)
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = view
        self.P = P
        self.G = {r: torch.zeros_like(t) for r, t in P.items()}
        self.S = torch.empty_like(P["mat"], dtype=self.act_dtype) if self.act_dtype != torch.float32 else None
        self.device = device
        self._shadow_versions = None

    def bound(self, device: torch.device) -> bool:
        if self.device != device or not self.P:
            return False
        for i, p in enumerate(self.params):
            base = self.P[self.region[i]]
            if p.data_ptr() != base.data_ptr() + 4 * self.offset[i] or p.dtype != torch.float32:
                return False
        return True

    # ------------------------------------------------------------------------------------------
    def param_view(self, name: str) -> torch.Tensor:
        i = self.index[name]
        return self.P[self.region[i]][self.offset[i]: self.offset[i] + self.numel[i]].view(self.params[i].shape)

    def grad_view(self, name: str) -> torch.Tensor:
        """A FRESH view tensor each call (autograd may steal it as .grad without a copy)."""
        i = self.index[name]
        return self.G[self.region[i]][self.offset[i]: self.offset[i] + self.numel[i]].view(self.params[i].shape)

    def shadow_view(self, name: str) -> torch.Tensor:
        """The tensor the GEMMs read for a matrix: act-typed shadow (bf16 mode) or the f32 parameter."""
        i = self.index[name]
        if self.S is None:
            return self.param_view(name)
        return self.S[self.offset[i]: self.offset[i] + self.numel[i]].view(self.params[i].shape)

    def grad_is_flat(self, name: str) -> bool:
        """True when param.grad already aliases the flat gradient range (=> accumulate in place)."""
        i = self.index[name]
        g = self.params[i].grad
        return g is not None and g.data_ptr() == self.G[self.region[i]].data_ptr() + 4 * self.offset[i]

    # ------------------------------------------------------------------------------------------
    def shadow_stale(self) -> bool:
        if self.S is None:
            return False
        v = tuple(p._version for p in self.params if p.ndim > 1)
        return v != self._shadow_versions

    def mark_shadow_fresh(self) -> None:
        self._shadow_versions = tuple(p._version for p in self.params if p.ndim > 1)
