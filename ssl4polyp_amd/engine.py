"""Host-side execution engine: sequences the HIP kernels of one ViT / MAE training step.

This is the MI355X-native counterpart of the torch/timm op graph that the reference builds in
``MaskedAutoencoderViT.forward`` (src/ssl4polyp/models/mae/models_mae.py:150-220) and
``ViT_from_MAE.forward`` / ``VisionTransformer_from_Any.forward`` (src/ssl4polyp/models/models.py:117-140,
196-222) plus autograd's backward of it.  Instead of ~14 library launches per block with the [N,N]
attention scores, GELU and residual tensors round-tripping HBM, a block is 7 launches forward
(LN, qkv GEMM, fused attention, proj GEMM+residual, LN, fc1 GEMM+GELU, fc2 GEMM+residual) and the whole
forward/backward is ONE autograd node, so the engine owns every intermediate buffer, their dtypes
(bf16 activations, f32 residual stream and statistics) and the order in which gradients become ready
(which drives the RCCL bucket all-reduce, see parallel.py).

PyTorch is used for device memory (caching allocator), streams and autograd plumbing only; all
arithmetic is in libpolypmae.so.  No CPU / eager fallback exists.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import contextlib
import ctypes
import os

import torch

from . import _lib
from ._lib import EPI_ACCUM, EPI_DGELU, EPI_GELU, EPI_RESIDUAL, EPI_STORE, PM_BF16, PM_F16, PM_F32

LN_EPS = 1e-6  # models_mae.py:227, models.py:164: partial(nn.LayerNorm, eps=1e-6)

BLOCK_PARAM_NAMES = (
    "norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
    "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """hipStream_t of torch's current stream.  Called once per launch (~450 times per step, forward + backward): the raw getters
    cost ~0.3 us, `torch.cuda.current_stream().cuda_stream` ~8 us (device-index resolution + a Stream object per call) --
    1.8 ms of a ViT-B step's 6 ms host enqueue time."""
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


_STREAMS: Dict[Tuple[int, str], "torch.cuda.Stream"] = {}
# The second weight-gradient stream (busy in the backward only) IS the second forward chain's stream (busy in the forward only):
# three engine streams + the main stream = four busy streams at most, one per hardware queue of the runtime's default, and one queue
# left for RCCL in a data-parallel rank (scratch/archive_r3/r3_exp23.sh; PM_MERGE_AUX_SIDE2=0 gives the weight gradients a stream of their own)
MERGE_AUX_SIDE2 = os.environ.get("PM_MERGE_AUX_SIDE2", "1") == "1"


def _shared_stream(device, role: str):
    """The process-wide side stream of this role ("aux0", "side", "side2", ...) on `device`, created on first use."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), role)
    st = _STREAMS.get(key)
    if st is None:
        st = _STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


def reserve_streams(device) -> None:
    """Create the engine's side streams on `device` and put one tiny launch on each, NOW.  The HIP runtime binds a stream to one
    of its few hardware queues (GPU_MAX_HW_QUEUES, 4) when the stream is first used -- a new queue while there are fewer than
    that, the least-loaded one afterwards: called before anything else uses streams -- in particular before torch.distributed /
    RCCL initialise, which use half a dozen of their own -- the main stream and the two side streams each get a queue to
    themselves and the fourth is left for RCCL.  Called late, two of them can land on one queue and run one after the other
    (measured with RCCL initialised first: the second forward chain behind the main stream, cls step 15.0 instead of 10.6 ms)."""
    dev = torch.device(device)
    torch.zeros(1, device=dev)  # (the main stream first)
    for role in ("aux0", "side") + (() if MERGE_AUX_SIDE2 else ("side2",)):
        with torch.cuda.stream(_shared_stream(dev, role)):
            torch.zeros(1, device=dev)
    torch.cuda.synchronize(dev)


# precision mode -> (activation / matrix-shadow dtype, its C-ABI code).  bf16: v_mfma_f32_32x32x16_bf16, no loss scaling.
# fp16: the reference's own AMP arithmetic (torch.cuda.amp.autocast = fp16 matmuls with f32 accumulation + GradScaler:
# train_classification.py:4527-4546, engine_pretrain.py:52-72) on v_mfma_f32_32x32x16_f16 -- same rate, 11 significant bits per
# operand instead of 8 (logits 4x closer to the fp32 path, gradients 7x: profiles/r4_rounding_fp16_*.json); the backward runs on
# fp16 operands too, so the loss must be scaled (optim.LossScaler, or torch's GradScaler).  fp32: exact-f32 MFMA, 1/16 of the rate.
PRECISIONS = {"bf16": (torch.bfloat16, PM_BF16), "fp16": (torch.float16, PM_F16), "fp32": (torch.float32, PM_F32)}


class Kernels:
    """Thin typed wrappers: torch tensors in, C-ABI calls out (include/polypmae.h)."""

    def __init__(self, precision: str, eps: float = LN_EPS):
        self.eps = float(eps)
        if precision not in PRECISIONS:
            raise ValueError("precision must be 'bf16', 'fp16' or 'fp32'")
        self.lib = _lib.load()
        self.precision = precision
        self.act_dtype, self.act = PRECISIONS[precision]

    # -- helpers ------------------------------------------------------------------------------
    def layernorm_fwd(self, x, gamma, beta, y, mean, rstd, M, D):
        _lib.check(self.lib.pm_layernorm_fwd(_ptr(x), D, _ptr(gamma), _ptr(beta), _ptr(y), _lib.dtype_code(y.dtype),
                                             _ptr(mean), _ptr(rstd), M, D, self.eps, _stream()), "pm_layernorm_fwd")

    def layernorm_bwd(self, dy, x, gamma, mean, rstd, dres, dx, dx_act, dgamma, dbeta, dcolsum, M, D):
        ws = self._scratch("_ws_ln", self._need(("ln", M, D), lambda: self.lib.pm_workspace_bytes(_lib.WS_LAYERNORM_BWD, M, D)),
                           x.device)
        _lib.check(self.lib.pm_layernorm_bwd(_ptr(dy), _lib.dtype_code(dy.dtype), _ptr(x), D, _ptr(gamma), _ptr(mean),
                                             _ptr(rstd), _ptr(dres), D, _ptr(dx), D, _ptr(dx_act), self.act,
                                             _ptr(dgamma), _ptr(dbeta), _ptr(dcolsum), M, D, _ptr(ws), ws.numel(), _stream()),
                   "pm_layernorm_bwd")

    SPLIT_FORWARD = int(os.environ.get("PM_SPLIT_FWD", "2"))  # forward: this many sub-batches as concurrent chains (1 = off)
    # workgroups a split-K weight gradient spreads over (pm_gemm_opts.max_blocks): the weight gradients run beside the
    # dgrad chain on a side stream, so each takes ~half of the CUs and leaves the rest to LayerNorm / attention backward /
    # dgrad GEMMs (measured: 128 best of 96..256)
    WGRAD_BLOCKS = int(os.environ.get("PM_WGRAD_BLOCKS", "128"))
    # fc1.bias gradient inside the dGELU dgrad epilogue (pm_gemm_colsum) instead of a column-sum kernel on the side
    # stream.  Off: the HBM-bound column sum overlaps the MFMA-bound GEMMs for free, while the fused reduction
    # lengthens the dgrad chain (measured -2.5 % step rate when fused).
    FUSE_COLSUM = os.environ.get("PM_FUSE_COLSUM", "0") == "1"
    # pm_vit_block_bwd.two_groups: a whole-K block's weight gradients as (fc2, fc1) behind dfc2 and (proj, qkv) behind the
    # attention backward, on two side streams: each launch starts as soon as its operands exist, and the 72 + 36 workgroups
    # of a ViT-B block spread over the whole dgrad chain instead of 108 behind the attention backward (measured, 5 pairs of
    # runs: MAE +0.5 %, cls +0.1 %; gradients bit-identical to the single launch)
    TWO_GROUPS = os.environ.get("PM_TWO_GROUPS", "1") == "1"
    # bench.py's kernel statistics: the fork / done events of pm_vit_block_bwd carry timestamps, so that the weight-gradient launches
    # of the one-call-per-block path can be timed in place (fork -> done on an idle side stream = the launch)
    TIME_BLOCK_EVENTS = False
    gemm_variant = int(os.environ.get("PM_GEMM_VARIANT", "0"))  # pm_gemm_opts.variant: 0 = the dispatcher's heuristics (tuning scripts set it per Kernels object)

    # -- scratch buffers: sized by the library's own queries (pm_gemm_workspace_bytes / pm_workspace_bytes), cached per
    #    shape; a buffer only ever grows, and growing it (first step of a new shape) drains the device first because the
    #    side stream may still be using the old one
    def _need(self, key, query) -> int:
        cache = self.__dict__.setdefault("_need_cache", {})
        n = cache.get(key)
        if n is None:
            n = cache[key] = int(query())
        return n

    def _scratch(self, name: str, nbytes: int, device) -> torch.Tensor:
        ws = getattr(self, name, None)
        if ws is None or ws.device != device or ws.numel() < nbytes:
            if ws is not None and ws.device == device:
                torch.cuda.synchronize(device)
            ws = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
            setattr(self, name, ws)
        return ws

    # The side streams are shared by every Kernels object of the process (one set per device): the HIP runtime multiplexes
    # streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and two streams on one queue run their
    # kernels one after the other.  A stream set per model put the third model's weight-gradient stream onto the main
    # stream's queue (bench.py builds several models in one process: the MAE step lost 11 %).
    def aux_stream(self, device, j: int = 0):
        return _shared_stream(device, f"aux{j}")

    def side_stream(self, device):
        st = getattr(self, "_side", None)
        if st is None or st.device != device:
            st = self._side = _shared_stream(device, "side")
        return st

    def side_stream2(self, device):
        st = getattr(self, "_side2", None)
        if st is None or st.device != device:
            st = self._side2 = _shared_stream(device, "aux0" if MERGE_AUX_SIDE2 else "side2")
        return st

    def _opts(self, wgrad: bool):
        o = self.__dict__.get("_opts_cache")
        if o is None or o[0] != (self.WGRAD_BLOCKS, self.gemm_variant):
            key = (self.WGRAD_BLOCKS, self.gemm_variant)
            o = self._opts_cache = (key, _lib.GemmOpts(self.WGRAD_BLOCKS, self.gemm_variant), _lib.GemmOpts(0, self.gemm_variant))
        return o[1] if wgrad else o[2]

    def gemm(self, A, lda, a_kmajor, B, ldb, b_kmajor, bias, C, ldc, epilogue, M, N, K, aux=None, resid=None, colsum=None,
             ws_name="_ws"):
        in_dtype = _lib.dtype_code(A.dtype)
        if _lib.dtype_code(B.dtype) != in_dtype:
            raise _lib.PolypMaeError("pm_gemm: operand dtypes differ")
        if colsum is not None:  # colsum[n] += sum_m C[m][n], fused into the epilogue where the kernel supports it
            ws = self._colsum_workspace(A.device, M, N, True)
            _lib.check(self.lib.pm_gemm_colsum(_ptr(A), lda, int(a_kmajor), _ptr(B), ldb, int(b_kmajor), in_dtype, _ptr(bias),
                                               _ptr(C), ldc, _lib.dtype_code(C.dtype), epilogue, _ptr(aux), _ptr(resid),
                                               _ptr(colsum), M, N, K, _ptr(ws), ws.numel(), _stream()), "pm_gemm_colsum")
            return
        wgrad = bool(a_kmajor and b_kmajor)
        opts = self._opts(wgrad)
        ws = None
        if wgrad:
            need = self._need(("gemm", in_dtype, M, N, K, opts.max_blocks, opts.variant),
                              lambda: self.lib.pm_gemm_workspace_bytes(1, 1, in_dtype, M, N, K, ctypes.byref(opts)))
            ws = self._scratch(ws_name, need, A.device)
        _lib.check(self.lib.pm_gemm_ex(_ptr(A), lda, int(a_kmajor), _ptr(B), ldb, int(b_kmajor), in_dtype, _ptr(bias),
                                       _ptr(C), ldc, _lib.dtype_code(C.dtype), epilogue, _ptr(aux), _ptr(resid), M, N, K,
                                       _ptr(ws), ws.numel() if ws is not None else 0, ctypes.byref(opts), _stream()), "pm_gemm")

    def linear_fwd(self, x, W, bias, out, M, N, K, epilogue=EPI_STORE, aux=None, resid=None):
        """out[M,N] = x[M,K] @ W[N,K]^T + bias  (nn.Linear forward)."""
        self.gemm(x, K, 0, W, K, 0, bias, out, N, epilogue, M, N, K, aux=aux, resid=resid)

    def linear_dgrad(self, dy, W, dx, M, N_out, K_in, epilogue=EPI_STORE, aux=None, colsum=None):
        """dx[M,K_in] = dy[M,N_out] @ W[N_out,K_in]  (W read as stored: k-major B operand); colsum += column sums of dx."""
        self.gemm(dy, N_out, 0, W, K_in, 1, None, dx, K_in, epilogue, M, K_in, N_out, aux=aux, colsum=colsum)

    DEFER_JOIN = os.environ.get("PM_DEFER_JOIN", "1") != "0"    # A/B switch: the embedding's backward before the side-stream join
    BLOCK_CALLS = os.environ.get("PM_BLOCK_CALLS", "1") != "0"  # block forward as ONE C call (pm_vit_block_fwd) instead of seven
    GROUP_WGRAD = os.environ.get("PM_GROUP_WGRAD", "1") != "0"  # A/B switch: one grouped weight-gradient launch per block
    GROUP_BLOCKS = int(os.environ.get("PM_GROUP_BLOCKS", "0"))  # CUs the grouped launch may take (0 = one workgroup per tile)
    # the same for a k-sliced group (MAE decoder block: 48 tiles x 4 slices = 192 work items; 0 = one workgroup per item)
    GROUP_BLOCKS_SLICED = int(os.environ.get("PM_GROUP_BLOCKS_SLICED", "0"))
    GROUP_MIN_TILES = int(os.environ.get("PM_GROUP_MIN_TILES", "64"))
    GROUP_BIAS = os.environ.get("PM_GROUP_BIAS", "1") != "0"    # qkv / fc1 bias gradients inside that launch (no pm_colsum pass)
    GROUP_SPLIT = os.environ.get("PM_GROUP_SPLIT", "1") != "0"  # A/B switch: groups of < 64 tiles as k-sliced grouped launches
    # The last blocks of a backward pass (lowest trainable ones) have no dgrad chain below them to run beside: a grouped
    # launch there holds 108 CUs for ~0.4 ms while the rest idle, so those blocks issue their four gradients one by one
    # (whole-chip split-K launches, each as soon as its dY exists).
    UNGROUP_TAIL = int(os.environ.get("PM_UNGROUP_TAIL", "1"))

    def wgrad_group(self, items, K, whole_k: bool = False) -> bool:
        """items: [(dy [K, n_out], x [K, n_in], dW f32 [n_out, n_in], accumulate[, dbias f32 [n_out] (+=)])].  One launch for all
        (pm_wgrad_group; a group of few tiles and long K is cut into k-slices: slabs in a scratch buffer sized by
        pm_wgrad_group_workspace_bytes, one reduce launch); False when the shapes do not fit the grouped kernel (the caller
        then uses linear_wgrad).  whole_k: PM_GROUP_WHOLE_K -- never slice (the second launch of a two-launch block)."""
        n = len(items)
        arr = (_lib.WgradItem * n)()
        for j, (dy, x, dW, acc, *rest) in enumerate(items):
            n_out, n_in = dW.shape
            arr[j] = _lib.WgradItem(_ptr(dy), n_out, _ptr(x), n_in, _ptr(dW), n_in, n_out, n_in, int(bool(acc)),
                                    _ptr(rest[0]) if rest and rest[0] is not None else None)
        dt = _lib.dtype_code(items[0][0].dtype)
        key = ("group", K, dt, tuple((tuple(it[2].shape), len(it) > 4 and it[4] is not None) for it in items))
        need = 0 if whole_k else self._need(key, lambda: self.lib.pm_wgrad_group_workspace_bytes(arr, n, K, dt))
        ws = self._scratch("_ws_group", need, items[0][0].device) if need else None
        blocks = _lib.PM_GROUP_WHOLE_K if whole_k else (self.GROUP_BLOCKS_SLICED if need else self.GROUP_BLOCKS)
        st = self.lib.pm_wgrad_group(arr, n, K, dt, blocks, _ptr(ws), ws.numel() if ws is not None else 0, _stream())
        if st == _lib.PM_ESHAPE:
            return False
        _lib.check(st, "pm_wgrad_group")
        return True

    def can_group_wgrad(self, K: int, dims) -> bool:
        """Would pm_wgrad_group take these (n_out, n_in) gradients over K tokens, and does the engine want it to?  The
        admission rules are the LIBRARY's (pm_wgrad_group_plan: bf16, whole 32-token k-steps, K >= 2048, tiles of at least
        256 x 128, 16-byte alignment) -- asked, not restated here; the policy on top is the engine's."""
        if not self.GROUP_WGRAD:
            return False
        key = ("can_group", K, self.act, tuple(dims), self.GROUP_MIN_TILES, self.GROUP_SPLIT)
        cache = self.__dict__.setdefault("_need_cache", {})
        hit = cache.get(key)
        if hit is not None:
            return hit
        n = len(dims)
        arr = (_lib.WgradItem * n)()
        for j, (o, i) in enumerate(dims):
            arr[j] = _lib.WgradItem(64, o, 64, i, 64, i, o, i, 0, None)  # (shapes only: the pointers are checked for alignment)
        tiles, slices = ctypes.c_int(0), ctypes.c_int(0)
        st = self.lib.pm_wgrad_group_plan(arr, n, K, self.act, None, ctypes.byref(tiles), ctypes.byref(slices))
        # >= 64 tiles of 256x256 (ViT-B block: 108): one full-K tile per workgroup.  Fewer (the 512-wide MAE decoder block:
        # 48) group as well when the library's plan cuts every tile into k-slices (48 tiles x 4 slices of >= 128 k-steps);
        # otherwise the gradients keep the per-GEMM split-K path (PM_GROUP_MIN_TILES=0 forces grouping: whole-K 256x128 tiles).
        ok = st == 0 and (tiles.value >= self.GROUP_MIN_TILES or (self.GROUP_SPLIT and slices.value > 1))
        cache[key] = ok
        return ok

    def two_launch_group(self, K: int, dims) -> bool:
        """May a grouped block (dims = fc2, fc1, proj, qkv) go out as two launches -- (fc2, fc1) and (proj, qkv) -- on two side
        streams (pm_block_bwd_desc.two_groups)?  Yes when the first pair alone is a whole-K group by the library's plan (no
        slabs: the two launches must not share a workspace); the second pair then runs whole-K too (PM_GROUP_WHOLE_K)."""
        if not self.TWO_GROUPS:
            return False
        key = ("two_launch", K, self.act, tuple(dims))
        cache = self.__dict__.setdefault("_need_cache", {})
        hit = cache.get(key)
        if hit is None:
            ok = True
            for first, part in ((True, dims[:2]), (False, dims[2:])):
                arr = (_lib.WgradItem * 2)()
                for j, (o, i) in enumerate(part):
                    arr[j] = _lib.WgradItem(64, o, 64, i, 64, i, o, i, 0, None)
                slices = ctypes.c_int(0)
                st = self.lib.pm_wgrad_group_plan(arr, 2, K, self.act, None, None, ctypes.byref(slices))
                ok = ok and st == 0 and (slices.value == 1 or not first)
            hit = cache[key] = ok
        return hit

    def linear_wgrad(self, dy, x, dW, M, N_out, K_in, accumulate, ws_name="_ws"):
        """dW[N_out,K_in] (+)= dy[M,N_out]^T @ x[M,K_in]  (both operands k-major, f32 output).  ws_name: the split-K slab
        scratch -- "_ws" belongs to the weight-gradient side stream; a launch on another stream that may run beside it names
        its own."""
        self.gemm(dy, N_out, 1, x, K_in, 1, None, dW, K_in, EPI_ACCUM if accumulate else EPI_STORE, N_out, K_in, M, ws_name=ws_name)

    def _colsum_workspace(self, device, M: int, N: int, fused: bool = False):
        # partial rows: one scratch per stream (main / wgrad side stream) so concurrent column sums never share it
        on_side = getattr(self, "_side", None) is not None and torch.cuda.current_stream() == self._side
        kind = _lib.WS_GEMM_COLSUM if fused else _lib.WS_COLSUM
        need = self._need(("cs", kind, M, N), lambda: self.lib.pm_workspace_bytes(kind, M, N))
        return self._scratch("_ws_cs_side" if on_side else "_ws_cs_main", need, device)

    def colsum(self, x, out, M, N):
        ws = self._colsum_workspace(x.device, M, N)
        _lib.check(self.lib.pm_colsum_ws(_ptr(x), N, _lib.dtype_code(x.dtype), _ptr(out), M, N, _ptr(ws), ws.numel(),
                                         _stream()), "pm_colsum")

    def attention_fwd(self, qkv, out, lse, B, N, H, dh):
        _lib.check(self.lib.pm_attention_fwd(_ptr(qkv), _ptr(out), _ptr(lse), B, N, H, dh, self.act, _stream()),
                   "pm_attention_fwd")

    def attention_bwd(self, qkv, out, dout, lse, delta, dqkv, B, N, H, dh):
        _lib.check(self.lib.pm_attention_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(delta), _ptr(dqkv), B, N,
                                             H, dh, self.act, _stream()), "pm_attention_bwd")

    def gather_rows(self, src: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dst[r] = src[idx[r]] for a 2-D (or 1-D) contiguous src; idx int32 on the device; out: a contiguous [R, width] destination."""
        rows = src if src.ndim == 2 else src.view(-1, 1)
        dst = out if out is not None else torch.empty(idx.numel(), rows.shape[1], dtype=src.dtype, device=src.device)
        rb = rows.shape[1] * src.element_size()
        _lib.check(self.lib.pm_gather_rows(_ptr(rows), rb, _ptr(idx), _ptr(dst), idx.numel(), rb, _stream()), "pm_gather_rows")
        return dst if src.ndim == 2 else dst.view(-1)

    def scatter_rows_zero(self, src: torch.Tensor, inv: torch.Tensor, dst: torch.Tensor) -> None:
        """dst[m] = src[inv[m]] where inv[m] >= 0, zeros elsewhere (every row of dst is written)."""
        _lib.check(self.lib.pm_scatter_rows_zero(_ptr(src), _ptr(inv), _ptr(dst), dst.shape[0], dst.shape[1] * dst.element_size(),
                                                 _stream()), "pm_scatter_rows_zero")

    SPARSE_TOP = os.environ.get("PM_SPARSE_TOP", "1") != "0"   # A/B switch: the classifier's top block on its cls rows (backward)

    def cast(self, src, dst):
        _lib.check(self.lib.pm_cast(_ptr(src), _ptr(dst), _lib.dtype_code(dst.dtype), src.numel(), _stream()), "pm_cast")

    # A Linear whose reduction dimension is not a multiple of the GEMM k-step (patch 14: 3 * 14 * 14 = 588) runs in a layout
    # padded with zeros to the next multiple of 64: operands by pad_cast, the valid part of a gradient back by unpad_add.
    @staticmethod
    def padded_k(K: int) -> int:
        return K if K % 64 == 0 else (K + 63) // 64 * 64

    def pad_cast(self, src: torch.Tensor, rows_pad: int, cols_pad: int) -> torch.Tensor:
        """f32 [rows, cols] -> act [rows_pad, cols_pad], zeros outside the source."""
        rows, cols = src.shape
        dst = torch.empty(rows_pad, cols_pad, dtype=self.act_dtype, device=src.device)
        _lib.check(self.lib.pm_pad_cast(_ptr(src), cols, _ptr(dst), cols_pad, self.act, rows, cols, rows_pad, cols_pad, _stream()),
                   "pm_pad_cast")
        return dst

    def unpad_add(self, src: torch.Tensor, dst: torch.Tensor, accumulate: bool) -> None:
        """dst f32 [rows, cols] (+)= src f32 [>= rows, >= cols][:rows, :cols]."""
        rows, cols = dst.shape
        _lib.check(self.lib.pm_unpad_add(_ptr(src), src.shape[1], _ptr(dst), cols, rows, cols, 1 if accumulate else 0, _stream()),
                   "pm_unpad_add")


@dataclass
class StackGeom:
    """Geometry of one stack of pre-LN transformer blocks (timm Block)."""
    dim: int
    heads: int
    depth: int
    hidden: int

    @property
    def dh(self) -> int:
        return self.dim // self.heads


class BlockWorkspace:
    """Saved-for-backward activations of one block at (B, N).  Everything the backward needs, nothing else."""

    def __init__(self, g: StackGeom, B: int, N: int, act_dtype, dev):
        M, D, Hd = B * N, g.dim, g.hidden
        f32 = torch.float32
        e = lambda *s, dt=act_dtype: torch.empty(*s, dtype=dt, device=dev)
        self.mean1, self.rstd1 = e(M, dt=f32), e(M, dt=f32)
        self.ln1 = e(M, D)
        self.qkv = e(M, 3 * D)
        self.lse = e(B * g.heads * N, dt=f32)
        self.attn = e(M, D)
        self.x_mid = e(M, D, dt=f32)
        self.mean2, self.rstd2 = e(M, dt=f32), e(M, dt=f32)
        self.ln2 = e(M, D)
        self.h_pre = e(M, Hd)
        self.h_act = e(M, Hd)
        self.x_out = e(M, D, dt=f32)


class StackWorkspace:
    """Per-(B,N) buffers of a whole block stack: per-block saves + shared backward temporaries."""

    def __init__(self, g: StackGeom, B: int, N: int, act_dtype, dev, training: bool):
        self.B, self.N, self.M = B, N, B * N
        M, D, Hd = self.M, g.dim, g.hidden
        f32 = torch.float32
        nblk = g.depth if training else min(g.depth, 2)
        self.blocks = [BlockWorkspace(g, B, N, act_dtype, dev) for _ in range(nblk)]
        self.training = training
        if training:
            e = lambda *s, dt=act_dtype: torch.empty(*s, dtype=dt, device=dev)
            # The weight-gradient GEMMs of block i run on a side stream and may still be in flight while the main
            # stream is one block further down (see BlockStack.backward): what they read is double-buffered by block
            # parity (d_hidden, d_qkv) or rotates through five buffers (act copies of the residual gradient: block i
            # uses slots p, p+1, p+2, block i-1 slots p+2, p+3, p+4).
            self.d_hidden = [e(M, Hd), e(M, Hd)]
            self.d_qkv = [e(M, 3 * D), e(M, 3 * D)]
            self.d_ln = e(M, D)
            self.d_attn = e(M, D)
            self.delta = e(B * g.heads * N, dt=f32)
            self.dx = [e(M, D, dt=f32), e(M, D, dt=f32)]
            self.dx_act = [e(M, D) for _ in range(5)]

    def block(self, i: int) -> BlockWorkspace:
        return self.blocks[i if self.training else i % len(self.blocks)]


class BlockStack:
    """Runs `depth` transformer blocks forward / backward over an f32 residual stream."""

    def __init__(self, k: Kernels, g: StackGeom):
        self.k, self.g = k, g

    def forward(self, ws: StackWorkspace, x_in: torch.Tensor, W: Sequence[Dict[str, torch.Tensor]],
                before_block: Optional[Callable[[int], None]] = None, keep_from: int = 0, cls_top: bool = False) -> torch.Tensor:
        """x_in f32 [M, D]; W[i] maps BLOCK_PARAM_NAMES -> tensors (matrices act-typed, vectors f32).
        before_block(i) runs before block i's first kernel (gate on a pending optimizer update of its weights).
        keep_from: no backward will run through blocks below this index (frozen blocks under a frozen front: finetune.py:49-91), so
        fc1's GELU epilogue does not store their pre-activations (38.7 MB per launch at ViT-B, bs = 64); a forward-only workspace
        (evaluation, linear probe) keeps none.
        cls_top: the caller reads row 0 of every sample of the result only (out_token "cls", models.py:134-136).  Behind its attention,
        the TOP block is then per-token work whose other rows nobody reads: proj + residual, LayerNorm2, fc1 + GELU, fc2 + residual run
        on the B cls rows (gathered behind the attention), and the result comes back as [B, D] (= [B, 1, D] for the head kernels)."""
        k, g = self.k, self.g
        if not ws.training:
            keep_from = g.depth
        cls_top = cls_top and ws.N > 1 and g.depth >= 1
        B, N, M, D, Hd = ws.B, ws.N, ws.M, g.dim, g.hidden
        # The forward has no second stream of its own work to share the CUs with, so the batch is cut into two halves
        # that run as independent chains on two streams: one half's GEMM tails / attention / LayerNorm fill the CUs
        # the other half's kernels leave idle (every op is per token or per (sample, head), and the halves write
        # disjoint row ranges of the same workspaces, so the backward sees one batch).
        main = torch.cuda.current_stream() if x_in.is_cuda else None
        nparts = min(k.SPLIT_FORWARD, B)
        split = nparts > 1 and main is not None and not torch.cuda.is_current_stream_capturing()
        if split:
            auxs = [k.aux_stream(x_in.device, j) for j in range(nparts - 1)]
            ev = torch.cuda.Event()
            ev.record(main)
            for aux in auxs:
                aux.wait_event(ev)
            cuts = [B * j // nparts for j in range(nparts + 1)]
            parts = [((main if j == 0 else auxs[j - 1]), cuts[j], cuts[j + 1]) for j in range(nparts)]
        else:
            parts = [(None, 0, B)]
        # PM_BLOCK_CALLS=0: every kernel of a block as its own C call from Python (the same launches; A/B and a fallback)
        fast = k.BLOCK_CALLS and x_in.is_cuda
        if fast:
            lib = k.lib
            cache = ws.__dict__.setdefault("_fwd_descs", {})
            raw = [(st.cuda_stream if st is not None and st is not main else None) for st, _, _ in parts]
        x = x_in
        for i in range(g.depth):
            if before_block is not None:
                if split:
                    before_block(i, also=auxs)
                else:
                    before_block(i)
            bw, p = ws.block(i), W[i]
            if cls_top and i == g.depth - 1:
                x = self._top_block_forward_cls(ws, bw, p, x, parts, main, i >= keep_from)
                continue
            if fast:
                # one C call per (block, sample range): pm_vit_block_fwd issues the same seven launches from a descriptor that
                # is built once and kept (every buffer it names is persistent: workspace, flat parameters, bf16 shadow)
                for j, (st, b0, b1) in enumerate(parts):
                    key = (b0, b1, x.data_ptr(), p["attn.qkv.weight"].data_ptr(), p["norm1.weight"].data_ptr(), k.gemm_variant,
                           i >= keep_from)
                    ent = cache.get((i, j))
                    if ent is None or ent[0] != key:
                        ent = cache[(i, j)] = (key, self._fwd_desc(ws, bw, p, x, b0, b1, keep=i >= keep_from))
                    _lib.check(lib.pm_vit_block_fwd(ctypes.byref(ent[1]), raw[j] if raw[j] is not None else _stream()),
                               "pm_vit_block_fwd")
                x = bw.x_out
                continue
            for st, b0, b1 in parts:
                r0, r1, Bh = b0 * N, b1 * N, b1 - b0
                Mh = r1 - r0
                h0, h1 = b0 * g.heads * N, b1 * g.heads * N
                with (torch.cuda.stream(st) if st is not None and st is not main else contextlib.nullcontext()):
                    xs = x[r0:r1]
                    k.layernorm_fwd(xs, p["norm1.weight"], p["norm1.bias"], bw.ln1[r0:r1], bw.mean1[r0:r1], bw.rstd1[r0:r1],
                                    Mh, D)
                    k.linear_fwd(bw.ln1[r0:r1], p["attn.qkv.weight"], p["attn.qkv.bias"], bw.qkv[r0:r1], Mh, 3 * D, D)
                    k.attention_fwd(bw.qkv[r0:r1], bw.attn[r0:r1], bw.lse[h0:h1], Bh, N, g.heads, g.dh)
                    k.linear_fwd(bw.attn[r0:r1], p["attn.proj.weight"], p["attn.proj.bias"], bw.x_mid[r0:r1], Mh, D, D,
                                 EPI_RESIDUAL, resid=xs)
                    k.layernorm_fwd(bw.x_mid[r0:r1], p["norm2.weight"], p["norm2.bias"], bw.ln2[r0:r1], bw.mean2[r0:r1],
                                    bw.rstd2[r0:r1], Mh, D)
                    k.linear_fwd(bw.ln2[r0:r1], p["mlp.fc1.weight"], p["mlp.fc1.bias"], bw.h_act[r0:r1], Mh, Hd, D, EPI_GELU,
                                 aux=bw.h_pre[r0:r1] if i >= keep_from else None)
                    k.linear_fwd(bw.h_act[r0:r1], p["mlp.fc2.weight"], p["mlp.fc2.bias"], bw.x_out[r0:r1], Mh, D, Hd,
                                 EPI_RESIDUAL, resid=bw.x_mid[r0:r1])
            x = bw.x_out
        if split:
            for aux in auxs:
                ev = torch.cuda.Event()
                ev.record(aux)
                main.wait_event(ev)
        return x

    @staticmethod
    def top_compact(ws: StackWorkspace, g: StackGeom, act_dtype, dev) -> dict:
        """Compact [B, ...] buffers of the top block's cls rows (forward saves + the head's gradient), kept on the workspace."""
        tc = ws.__dict__.get("_top_c")
        if tc is None:
            B, N, D, Hd, f32 = ws.B, ws.N, g.dim, g.hidden, torch.float32
            e = lambda *s, dt=act_dtype: torch.empty(*s, dtype=dt, device=dev)
            rows = torch.arange(B, dtype=torch.int32, device=dev) * N          # (built on the device: no host copy, no sync)
            inv = torch.full((B * N,), -1, dtype=torch.int32, device=dev)
            inv[rows.long()] = torch.arange(B, dtype=torch.int32, device=dev)
            tc = ws.__dict__["_top_c"] = dict(
                rows=rows, inv=inv, x=e(B, D, dt=f32), attn=e(B, D), x_mid=e(B, D, dt=f32), ln2=e(B, D), mean2=e(B, dt=f32),
                rstd2=e(B, dt=f32), h_pre=e(B, Hd), h_act=e(B, Hd), x_out=e(B, D, dt=f32), dx=e(B, D, dt=f32), dx_act=e(B, D))
        return tc

    def _top_block_forward_cls(self, ws, bw, p, x, parts, main, keep: bool) -> torch.Tensor:
        k, g = self.k, self.g
        N, D, Hd = ws.N, g.dim, g.hidden
        tc = self.top_compact(ws, g, k.act_dtype, x.device)
        for st, b0, b1 in parts:
            r0, r1, Bh = b0 * N, b1 * N, b1 - b0
            Mh = r1 - r0
            h0, h1 = b0 * g.heads * N, b1 * g.heads * N
            with (torch.cuda.stream(st) if st is not None and st is not main else contextlib.nullcontext()):
                xs = x[r0:r1]
                k.layernorm_fwd(xs, p["norm1.weight"], p["norm1.bias"], bw.ln1[r0:r1], bw.mean1[r0:r1], bw.rstd1[r0:r1], Mh, D)
                k.linear_fwd(bw.ln1[r0:r1], p["attn.qkv.weight"], p["attn.qkv.bias"], bw.qkv[r0:r1], Mh, 3 * D, D)
                k.attention_fwd(bw.qkv[r0:r1], bw.attn[r0:r1], bw.lse[h0:h1], Bh, N, g.heads, g.dh)
                idx = tc["rows"][b0:b1]
                k.gather_rows(x.view(-1, D), idx, out=tc["x"][b0:b1])
                k.gather_rows(bw.attn.view(-1, D), idx, out=tc["attn"][b0:b1])
                k.linear_fwd(tc["attn"][b0:b1], p["attn.proj.weight"], p["attn.proj.bias"], tc["x_mid"][b0:b1], Bh, D, D, EPI_RESIDUAL,
                             resid=tc["x"][b0:b1])
                k.layernorm_fwd(tc["x_mid"][b0:b1], p["norm2.weight"], p["norm2.bias"], tc["ln2"][b0:b1], tc["mean2"][b0:b1],
                                tc["rstd2"][b0:b1], Bh, D)
                k.linear_fwd(tc["ln2"][b0:b1], p["mlp.fc1.weight"], p["mlp.fc1.bias"], tc["h_act"][b0:b1], Bh, Hd, D, EPI_GELU,
                             aux=tc["h_pre"][b0:b1] if keep else None)
                k.linear_fwd(tc["h_act"][b0:b1], p["mlp.fc2.weight"], p["mlp.fc2.bias"], tc["x_out"][b0:b1], Bh, D, Hd, EPI_RESIDUAL,
                             resid=tc["x_mid"][b0:b1])
        return tc["x_out"]

    def _fwd_desc(self, ws: StackWorkspace, bw: BlockWorkspace, p, x: torch.Tensor, b0: int, b1: int, keep: bool = True):
        """pm_block_fwd_desc of block `bw` for samples [b0, b1): every pointer at the first row of the range.  keep = False:
        h_pre = NULL (the pre-activation of fc1 is not stored: no backward through this block)."""
        k, g = self.k, self.g
        N, D, Hd = ws.N, g.dim, g.hidden
        r0, h0 = b0 * N, b0 * g.heads * N
        a = lambda t, row, width: t.data_ptr() + row * width * t.element_size()
        v = lambda t: t.data_ptr() if t is not None else None
        return _lib.BlockFwdDesc(
            a(x, r0, D), a(bw.x_mid, r0, D), a(bw.x_out, r0, D), a(bw.ln1, r0, D), a(bw.mean1, r0, 1), a(bw.rstd1, r0, 1),
            a(bw.qkv, r0, 3 * D), a(bw.lse, h0, 1), a(bw.attn, r0, D), a(bw.ln2, r0, D), a(bw.mean2, r0, 1), a(bw.rstd2, r0, 1),
            a(bw.h_pre, r0, Hd) if keep else None, a(bw.h_act, r0, Hd),
            v(p["norm1.weight"]), v(p["norm1.bias"]), v(p["norm2.weight"]), v(p["norm2.bias"]),
            v(p["attn.qkv.weight"]), v(p["attn.proj.weight"]), v(p["mlp.fc1.weight"]), v(p["mlp.fc2.weight"]),
            v(p["attn.qkv.bias"]), v(p["attn.proj.bias"]), v(p["mlp.fc1.bias"]), v(p["mlp.fc2.bias"]),
            (b1 - b0) * N, b1 - b0, N, D, Hd, g.heads, k.act, k.gemm_variant, k.eps)

    def _bwd_desc(self, ws, bw, p, gr, xin, dx, dx_act, dmid, dmid_act, din, din_act, d_hidden, d_qkv, below_bias, acc, side, main):
        """(pm_block_bwd_desc, done event, objects to keep alive) of one block: the events are created once and reused."""
        k, g = self.k, self.g
        M, D, Hd = ws.M, g.dim, g.hidden
        v = lambda t: t.data_ptr() if t is not None else None
        ws_ln = k._scratch("_ws_ln", k._need(("ln", M, D), lambda: k.lib.pm_workspace_bytes(_lib.WS_LAYERNORM_BWD, M, D)), dx.device)
        items = (_lib.WgradItem * 4)()
        for j, (n_out, n_in, bias) in enumerate(((D, Hd, False), (Hd, D, True), (D, D, False), (3 * D, D, True))):
            items[j] = _lib.WgradItem(64, n_out, 64, n_in, 64, n_in, n_out, n_in, 0, 64 if bias else None)  # (sizes only)
        need = k._need(("groupws", M, k.act, D, Hd), lambda: k.lib.pm_wgrad_group_workspace_bytes(items, 4, M, k.act))
        ws_group = k._scratch("_ws_group", need, dx.device) if need else None
        mk = lambda: torch.cuda.Event(enable_timing=bool(k.TIME_BLOCK_EVENTS))
        ev_fork, ev_done = mk(), mk()
        ev_fork.record(main)   # (materialises the hipEvent_t handles; re-recorded by every call)
        ev_done.record(main)
        two = not need and k.two_launch_group(M, ((D, Hd), (Hd, D), (D, D), (3 * D, D)))
        ev_fork2 = ev_done2 = side2 = None
        if two:
            side2 = k.side_stream2(dx.device)
            ev_fork2, ev_done2 = mk(), mk()
            ev_fork2.record(main)
            ev_done2.record(main)
        desc = _lib.BlockBwdDesc(
            v(xin), v(bw.x_mid), v(bw.ln1), v(bw.qkv), v(bw.attn), v(bw.ln2), v(bw.h_pre), v(bw.h_act), v(bw.mean1), v(bw.rstd1),
            v(bw.mean2), v(bw.rstd2), v(bw.lse), v(p["norm1.weight"]), v(p["norm2.weight"]), v(p["attn.qkv.weight"]),
            v(p["attn.proj.weight"]), v(p["mlp.fc1.weight"]), v(p["mlp.fc2.weight"]), v(dx), v(dx_act), v(dmid), v(dmid_act), v(din),
            v(din_act), v(d_hidden), v(d_qkv), v(ws.d_ln), v(ws.d_attn), v(ws.delta),
            v(gr["norm1.weight"]), v(gr["norm1.bias"]), v(gr["norm2.weight"]), v(gr["norm2.bias"]), v(gr["attn.qkv.weight"]),
            v(gr["attn.proj.weight"]), v(gr["mlp.fc1.weight"]), v(gr["mlp.fc2.weight"]), v(gr["attn.qkv.bias"]),
            v(gr["attn.proj.bias"]), v(gr["mlp.fc1.bias"]), v(below_bias),
            v(ws_ln), ws_ln.numel(), v(ws_group), ws_group.numel() if ws_group is not None else 0,
            side.cuda_stream, None, ev_fork.cuda_event, ev_done.cuda_event,
            ws.B, ws.N, D, Hd, g.heads, k.act, k.gemm_variant, k.GROUP_BLOCKS_SLICED if need else k.GROUP_BLOCKS, acc,
            int(two), side2.cuda_stream if two else None, ev_fork2.cuda_event if two else None, ev_done2.cuda_event if two else None)
        return desc, (ev_done, ev_done2) if two else (ev_done,), (ev_fork, ev_fork2, ws_ln, ws_group)

    def backward(self, ws: StackWorkspace, x_in: torch.Tensor, W, G, dx: torch.Tensor, dx_act: torch.Tensor,
                 last_bias_grad_done: bool, trainable: Sequence[bool], need_input_grad: bool,
                 accumulate: Callable[[str, int], bool], on_block_done: Optional[Callable[[int], None]] = None,
                 prev_bias_grad: Optional[torch.Tensor] = None,
                 ends_pass: bool = True, defer_join: bool = False,
                 sparse_top: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        """dx / dx_act: gradient w.r.t. the stack output (f32 + act copy).  G[i][name] = f32 gradient
        tensors (vectors are += targets and must be zeroed or hold the running sum; matrices follow
        accumulate(name, i)).  `last_bias_grad_done`: the producer of dx already added colsum(dx) into the
        last block's fc2.bias gradient.  `prev_bias_grad`: bias gradient of the Linear that produced the
        stack input (receives colsum of the input gradient), if any.  `ends_pass`: nothing but a short tail follows this
        stack in the backward pass (the encoder; not the MAE decoder, whose weight gradients run beside the encoder's chain).
        `defer_join`: do not make the main stream wait for the last blocks' weight gradients here -- the caller still has
        main-stream work that does not read them (the embedding's backward) and calls `join_deferred(ws)` after it; the side
        stream's last launches then run beside that work instead of in front of it.
        `sparse_top` (any non-None value): the forward ran with cls_top and dx / dx_act are the head's gradient on the B cls rows
        ([B, D]); the top block's backward then runs on those rows up to its attention (see _top_block_sparse).
        Returns (dx_in f32, dx_in act) or (None, None) when nothing below needs it."""
        k, g = self.k, self.g
        B, N, M, D, Hd = ws.B, ws.N, ws.M, g.dim, g.hidden
        lowest = min([i for i, t in enumerate(trainable) if t], default=g.depth)
        main = torch.cuda.current_stream()
        pending: Dict[int, Tuple[torch.cuda.Event, ...]] = {}  # block -> side-stream event(s) after its last weight-gradient kernels

        def join(down_to: int):
            for j in sorted((j for j in pending if j >= down_to), reverse=True):
                for ev in pending.pop(j):
                    main.wait_event(ev)

        # PM_BLOCK_CALLS: fully trainable, grouped blocks go through pm_vit_block_bwd (one C call per block)
        fast = (k.BLOCK_CALLS and k.GROUP_BIAS and not k.FUSE_COLSUM and dx.is_cuda and
                not torch.cuda.is_current_stream_capturing())
        bcache = ws.__dict__.setdefault("_bwd_descs", {})
        ptr_of = lambda t: t.data_ptr() if t is not None else 0
        for i in reversed(range(g.depth)):
            if i < lowest and not need_input_grad:
                join(0)
                return None, None
            bw, p, gr, tr = ws.block(i), W[i], G[i], trainable[i]
            xin = x_in if i == 0 else ws.block(i - 1).x_out
            need_dx_in = need_input_grad or i > lowest
            o = i & 1
            dmid, din = ws.dx[o], ws.dx[o ^ 1]  # f32 residual gradients: main stream only, ping-pong (in-place add is fine)
            d_hidden, d_qkv = ws.d_hidden[o], ws.d_qkv[o]
            # act copies rotate through five buffers, two steps per block: wgrad(fc2) / wgrad(proj) of this block read
            # the incoming one / dmid_act on the side stream, possibly until the main stream has finished block i-1,
            # which writes the next two slots.
            pin = next((j for j in range(5) if dx_act is ws.dx_act[j]), 4)
            dmid_act, din_act = ws.dx_act[(pin + 1) % 5], ws.dx_act[(pin + 2) % 5]
            join(i + 2)  # block i overwrites what the side stream read for block i+2
            if i == g.depth - 1 and sparse_top is not None:
                if tr and ws.B % 8:
                    raise _lib.PolypMaeError("the cls-row top block needs a batch that is a multiple of 8 to train (16-byte k-major rows)")
                below_bias = (G[i - 1]["mlp.fc2.bias"] if trainable[i - 1] else None) if i > 0 else prev_bias_grad
                ev_last = self._top_block_sparse(ws, bw, p, gr, xin, dx, dx_act, sparse_top, last_bias_grad_done, dmid, din, din_act,
                                                 d_qkv, below_bias, lambda n: accumulate(n, i), main, on_block_done, i, tr)
                if ev_last is not None:
                    pending[i] = ev_last
                dx, dx_act = din, din_act
                continue
            if i == g.depth - 1 and not last_bias_grad_done and tr:
                k.colsum(dx, gr["mlp.fc2.bias"], M, D)
            # Weight gradients run on a side stream, concurrently with the dgrad chain on the main stream: the two
            # kernels' blocks share the CUs out of phase, so one's epilogue / partial last round hides under the
            # other's k-loop (both read the same dY; every buffer the side stream reads stays untouched until the
            # join below).
            side = k.side_stream(main.device) if tr else None

            def fork():
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)

            # One grouped launch for the block's four weight gradients (full-K tiles, no split-K slabs), issued once the
            # whole dgrad chain of the block is enqueued; it runs beside block i-1's chain.
            grouped = tr and (not ends_pass or i - lowest >= k.UNGROUP_TAIL) and k.can_group_wgrad(M, ((D, Hd), (Hd, D), (D, D), (3 * D, D)))
            if grouped and fast:
                # the whole block in ONE C call (pm_vit_block_bwd): the same launches, the same fork / done events, from a
                # descriptor built once per block and kept (every buffer it names is persistent)
                below_bias = (G[i - 1]["mlp.fc2.bias"] if trainable[i - 1] else None) if i > 0 else prev_bias_grad
                acc = (int(accumulate("attn.qkv.weight", i)) | int(accumulate("attn.proj.weight", i)) << 1 |
                       int(accumulate("mlp.fc1.weight", i)) << 2 | int(accumulate("mlp.fc2.weight", i)) << 3)
                key = (dx.data_ptr(), dx_act.data_ptr(), xin.data_ptr(), gr["attn.qkv.weight"].data_ptr(),
                       p["attn.qkv.weight"].data_ptr(), gr["norm1.weight"].data_ptr(),
                       below_bias.data_ptr() if below_bias is not None else 0, acc, k.gemm_variant, k.GROUP_BLOCKS, k.GROUP_BLOCKS_SLICED, k.TWO_GROUPS, k.TIME_BLOCK_EVENTS,
                       ptr_of(k.__dict__.get("_ws_ln")), ptr_of(k.__dict__.get("_ws_group")))  # (scratch is replaced when it grows)
                ent = bcache.get(i)
                if ent is None or ent[0] != key:
                    ent = bcache[i] = (key,) + self._bwd_desc(ws, bw, p, gr, xin, dx, dx_act, dmid, dmid_act, din, din_act,
                                                              d_hidden, d_qkv, below_bias, acc, side, main)
                _, desc, evs_done, keep = ent
                _lib.check(k.lib.pm_vit_block_bwd(ctypes.byref(desc), _stream()), "pm_vit_block_bwd")
                pending[i] = evs_done
                if on_block_done is not None:
                    if len(evs_done) == 2:  # the block's gradients are final behind BOTH launches: the later one's stream
                        last = k.side_stream2(main.device)
                        last.wait_event(evs_done[0])
                    else:
                        last = side
                    with torch.cuda.stream(last):
                        on_block_done(i)
                dx, dx_act = din, din_act
                continue
            # ---- MLP branch ----
            if tr and not grouped:
                fork()
                with torch.cuda.stream(side):
                    k.linear_wgrad(dx_act, bw.h_act, gr["mlp.fc2.weight"], M, D, Hd, accumulate("mlp.fc2.weight", i))
            # fc1.bias gradient = column sums of d_hidden: optionally fused into the dGELU epilogue that produces it
            fuse_cs = tr and k.FUSE_COLSUM
            # the launcher's two-launch schedule, kernel by kernel: (fc2, fc1) behind dfc2 on the side stream, (proj, qkv) behind
            # the attention backward on the second one
            two = (grouped and k.GROUP_BIAS and not fuse_cs and
                   k.two_launch_group(M, ((D, Hd), (Hd, D), (D, D), (3 * D, D))))
            group_mlp = [(dx_act, bw.h_act, gr["mlp.fc2.weight"], accumulate("mlp.fc2.weight", i)),
                         (d_hidden, bw.ln2, gr["mlp.fc1.weight"], accumulate("mlp.fc1.weight", i),
                          gr["mlp.fc1.bias"] if (grouped and k.GROUP_BIAS and not fuse_cs) else None)] if grouped else None
            ev_mlp = None
            k.linear_dgrad(dx_act, p["mlp.fc2.weight"], d_hidden, M, D, Hd, EPI_DGELU, aux=bw.h_pre,
                           colsum=gr["mlp.fc1.bias"] if fuse_cs else None)
            if tr:
                fork()
                with torch.cuda.stream(side):
                    if not grouped:
                        k.linear_wgrad(d_hidden, bw.ln2, gr["mlp.fc1.weight"], M, Hd, D, accumulate("mlp.fc1.weight", i))
                    if not fuse_cs and not (grouped and k.GROUP_BIAS):
                        k.colsum(d_hidden, gr["mlp.fc1.bias"], M, Hd)
                    if two:
                        if not k.wgrad_group(group_mlp, M):
                            raise _lib.PolypMaeError("pm_wgrad_group refused a group that two_launch_group admitted")
                        ev_mlp = torch.cuda.Event()
                        ev_mlp.record(side)
            k.linear_dgrad(d_hidden, p["mlp.fc1.weight"], ws.d_ln, M, Hd, D)
            k.layernorm_bwd(ws.d_ln, bw.x_mid, p["norm2.weight"], bw.mean2, bw.rstd2, dx, dmid, dmid_act,
                            gr["norm2.weight"] if tr else None, gr["norm2.bias"] if tr else None,
                            gr["attn.proj.bias"] if tr else None, M, D)
            # ---- attention branch ----
            if tr and not grouped:
                fork()
                with torch.cuda.stream(side):
                    k.linear_wgrad(dmid_act, bw.attn, gr["attn.proj.weight"], M, D, D, accumulate("attn.proj.weight", i))
            k.linear_dgrad(dmid_act, p["attn.proj.weight"], ws.d_attn, M, D, D)
            k.attention_bwd(bw.qkv, bw.attn, ws.d_attn, bw.lse, ws.delta, d_qkv, B, N, g.heads, g.dh)
            if tr:
                last = k.side_stream2(main.device) if two else side
                ev = torch.cuda.Event()
                ev.record(main)
                last.wait_event(ev)
                with torch.cuda.stream(last):
                    gb = grouped and k.GROUP_BIAS  # bias gradients (column sums of dY) inside the grouped launch
                    if not gb:
                        k.colsum(d_qkv, gr["attn.qkv.bias"], M, 3 * D)
                    if grouped:
                        group_attn = [(dmid_act, bw.attn, gr["attn.proj.weight"], accumulate("attn.proj.weight", i)),
                                      (d_qkv, bw.ln1, gr["attn.qkv.weight"], accumulate("attn.qkv.weight", i),
                                       gr["attn.qkv.bias"] if gb else None)]
                        ok = k.wgrad_group(group_attn, M, whole_k=True) if two else k.wgrad_group(group_mlp + group_attn, M)
                        if not ok:
                            raise _lib.PolypMaeError("pm_wgrad_group refused a group that can_group_wgrad admitted")
                    else:
                        k.linear_wgrad(d_qkv, bw.ln1, gr["attn.qkv.weight"], M, 3 * D, D, accumulate("attn.qkv.weight", i))
                    # no join here: the main stream runs on into block i-1 and waits for these events only before
                    # block i-2.  The block's matrix gradients are final behind the last launch of each stream.
                    ev_last = torch.cuda.Event()
                    ev_last.record(last)
                    pending[i] = (ev_mlp, ev_last) if two else (ev_last,)
                    if on_block_done is not None:
                        if two:
                            last.wait_event(ev_mlp)
                        on_block_done(i)
            run_ln1 = need_dx_in or tr
            if run_ln1:
                k.linear_dgrad(d_qkv, p["attn.qkv.weight"], ws.d_ln, M, 3 * D, D)
            if not run_ln1:
                join(0)
                return None, None
            below_bias = None
            if i > 0 and trainable[i - 1]:
                below_bias = G[i - 1]["mlp.fc2.bias"]
            elif i == 0:
                below_bias = prev_bias_grad
            k.layernorm_bwd(ws.d_ln, xin, p["norm1.weight"], bw.mean1, bw.rstd1, dmid, din, din_act,
                            gr["norm1.weight"] if tr else None, gr["norm1.bias"] if tr else None, below_bias, M, D)
            dx, dx_act = din, din_act
            if on_block_done is not None and not tr:
                on_block_done(i)
        if defer_join:
            ws._deferred_join = [ev for j in sorted(pending, reverse=True) for ev in pending.pop(j)]
        else:
            join(0)
        return dx, dx_act

    def _top_block_sparse(self, ws, bw, p, gr, xin, dx, dx_act, sparse_top, last_bias_grad_done, dmid, din, din_act, d_qkv, below_bias,
                          acc, main, on_block_done, i, tr):
        """Backward of the top block of a classifier that reads the cls row only (forward: _top_block_forward_cls).  dx / dx_act are the
        head's gradient on the B cls rows ([B, D], pm_vit_head_bwd with N = 1); the block's saves behind the attention are compact too.
        The MLP branch and the proj Linear -- dfc2 + dGELU, dfc1, LayerNorm2', proj dgrad, the fc2 / fc1 / proj weight gradients -- run at
        M = B; d mid and d attn are scattered into zero-filled dense buffers (a zero row of dY adds nothing anywhere, and the residual
        path adds zero to zero), and from the attention backward on -- which spreads the gradient over all keys -- the block is dense:
        qkv weight gradient on the side stream, qkv dgrad, LayerNorm1'.  The reference's autograd multiplies those zeros; skipping them
        is the same kind of identity as embedding only the kept patches in MAE."""
        k, g = self.k, self.g
        tc = self.top_compact(ws, g, k.act_dtype, dx.device)
        rows, inv = tc["rows"], tc["inv"]
        B, N, M, D, Hd = ws.B, ws.N, ws.M, g.dim, g.hidden
        R = B
        dev, f32 = dx.device, torch.float32
        e = lambda *s, dt=k.act_dtype: torch.empty(*s, dtype=dt, device=dev)
        dxc, dxac = dx.view(R, D), dx_act.view(R, D)
        G_ = (lambda n: gr[n]) if tr else (lambda n: None)
        if tr and not last_bias_grad_done:
            k.colsum(dxc, gr["mlp.fc2.bias"], R, D)
        # ---- MLP branch on the live rows
        d_hidden_c = e(R, Hd)
        k.linear_dgrad(dxac, p["mlp.fc2.weight"], d_hidden_c, R, D, Hd, EPI_DGELU, aux=tc["h_pre"])
        if tr:
            k.linear_wgrad(dxac, tc["h_act"], gr["mlp.fc2.weight"], R, D, Hd, acc("mlp.fc2.weight"), ws_name="_ws_front")
            k.linear_wgrad(d_hidden_c, tc["ln2"], gr["mlp.fc1.weight"], R, Hd, D, acc("mlp.fc1.weight"), ws_name="_ws_front")
            k.colsum(d_hidden_c, gr["mlp.fc1.bias"], R, Hd)
        d_ln_c = e(R, D)
        k.linear_dgrad(d_hidden_c, p["mlp.fc1.weight"], d_ln_c, R, Hd, D)
        dmid_c, dmid_act_c = e(R, D, dt=f32), e(R, D)
        k.layernorm_bwd(d_ln_c, tc["x_mid"], p["norm2.weight"], tc["mean2"], tc["rstd2"], dxc, dmid_c, dmid_act_c, G_("norm2.weight"),
                        G_("norm2.bias"), G_("attn.proj.bias"), R, D)
        # ---- proj on the live rows, then dense again for the attention backward
        if tr:
            k.linear_wgrad(dmid_act_c, tc["attn"], gr["attn.proj.weight"], R, D, D, acc("attn.proj.weight"), ws_name="_ws_front")
        d_attn_c = e(R, D)
        k.linear_dgrad(dmid_act_c, p["attn.proj.weight"], d_attn_c, R, D, D)
        k.scatter_rows_zero(dmid_c, inv, dmid.view(M, D))
        k.scatter_rows_zero(d_attn_c, inv, ws.d_attn.view(M, D))
        k.attention_bwd(bw.qkv, bw.attn, ws.d_attn, bw.lse, ws.delta, d_qkv, B, N, g.heads, g.dh)
        ev_last = None
        if tr:
            side = k.side_stream(main.device)
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                k.colsum(d_qkv, gr["attn.qkv.bias"], M, 3 * D)
                k.linear_wgrad(d_qkv, bw.ln1, gr["attn.qkv.weight"], M, 3 * D, D, acc("attn.qkv.weight"))
                ev_last = torch.cuda.Event()
                ev_last.record(side)
                if on_block_done is not None:
                    on_block_done(i)
        k.linear_dgrad(d_qkv, p["attn.qkv.weight"], ws.d_ln, M, 3 * D, D)
        k.layernorm_bwd(ws.d_ln, xin, p["norm1.weight"], bw.mean1, bw.rstd1, dmid, din, din_act, G_("norm1.weight"), G_("norm1.bias"),
                        below_bias, M, D)
        if not tr and on_block_done is not None:
            on_block_done(i)
        return (ev_last,) if ev_last is not None else None

    @staticmethod
    def join_deferred(ws: StackWorkspace) -> None:
        """Make the current stream wait for the weight-gradient launches a backward(..., defer_join=True) left running."""
        main = torch.cuda.current_stream()
        for ev in ws.__dict__.pop("_deferred_join", []):
            main.wait_event(ev)
