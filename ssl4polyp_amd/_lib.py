"""ctypes binding of the C-ABI library (include/polypmae.h).

The product path has NO CPU or eager fallback: if the HIP library is missing or a call
fails, this module raises.  Tensors cross the boundary as raw device pointers + sizes.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_void_p

PM_F32, PM_BF16, PM_F16 = 0, 1, 2
EPI_STORE, EPI_GELU, EPI_RESIDUAL, EPI_DGELU, EPI_ACCUM = 0, 1, 2, 3, 4

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POLYPMAE_LIB") or os.path.join(_HERE, "lib", "libpolypmae.so")  # override: kernel A/B builds

P, I, L, F = c_void_p, c_int, c_long, c_float

# name -> argtypes, exactly the prototypes of include/polypmae.h
SIGNATURES = {
    "pm_layernorm_fwd": [P, L, P, P, P, I, P, P, I, I, F, P],
    "pm_layernorm_bwd": [P, I, P, L, P, P, P, P, L, P, L, P, I, P, P, P, I, I, P, ctypes.c_size_t, P],
    "pm_vit_block_fwd": [P, P],
    "pm_vit_block_bwd": [P, P],
    "pm_gemm": [P, L, I, P, L, I, I, P, P, L, I, I, P, P, I, I, I, P],
    "pm_gemm_ws": [P, L, I, P, L, I, I, P, P, L, I, I, P, P, I, I, I, P, ctypes.c_size_t, P],
    "pm_gemm_ex": [P, L, I, P, L, I, I, P, P, L, I, I, P, P, I, I, I, P, ctypes.c_size_t, P, P],
    "pm_wgrad_group": [P, I, I, I, I, P, ctypes.c_size_t, P],
    "pm_wgrad_group_plan": [P, I, I, I, P, P, P],
    "pm_gemm_colsum": [P, L, I, P, L, I, I, P, P, L, I, I, P, P, P, I, I, I, P, ctypes.c_size_t, P],
    "pm_attention_fwd": [P, P, P, I, I, I, I, I, P],
    "pm_attention_bwd": [P, P, P, P, P, P, I, I, I, I, I, P],
    "pm_colsum": [P, L, I, P, I, I, P],
    "pm_colsum_ws": [P, L, I, P, I, I, P, ctypes.c_size_t, P],
    "pm_patch_im2col": [P, P, P, L, I, I, I, I, I, I, P],
    "pm_pad_cast": [P, L, P, L, I, I, I, I, I, P],
    "pm_unpad_add": [P, L, P, L, I, I, I, P],
    "pm_assemble_tokens": [P, P, P, P, P, I, I, I, P],
    "pm_assemble_tokens_bwd": [P, P, P, I, P, P, I, I, I, P],
    "pm_mae_noise": [P, L, ctypes.c_ulonglong, ctypes.c_uint, P],
    "pm_mae_masking": [P, P, P, P, I, I, I, P],
    "pm_mae_unshuffle": [P, P, P, P, P, I, I, I, I, P],
    "pm_mae_unshuffle_bwd": [P, P, P, I, P, I, I, I, I, P, ctypes.c_size_t, P],
    "pm_mae_loss_fwd": [P, P, L, I, P, I, I, I, I, I, P],
    "pm_mae_loss_finish": [P, P, L, P, P, P],
    "pm_mae_loss_bwd": [P, P, L, I, P, P, P, P, L, I, I, I, I, I, I, P],
    "pm_cast": [P, P, I, L, P],
    "pm_gather_rows": [P, L, P, P, I, L, P],
    "pm_scatter_rows_zero": [P, P, P, I, L, P],
    "pm_preprocess_u8": [P, P, P, I, I, I, F, F, F, F, F, F, P],
    "pm_aug_resize_u8": [P, P, P, P, P, I, P, P, I, I, I, I, I, I, P],
    "pm_aug_resized_crop_u8": [P, P, P, I, I, I, I, I, P, ctypes.c_size_t, P],
    "pm_aug_color_jitter_u8": [P, P, P, P, I, I, I, P],
    "pm_aug_gaussian_blur_u8": [P, P, P, P, I, I, I, I, P],
    "pm_aug_geometry_u8": [P, P, P, I, I, I, I, F, F, F, F, F, F, P],
    "pm_aug_pil_gaussian_blur_u8": [P, P, P, P, I, I, I, I, P],
    "pm_aug_occlude_u8": [P, P, I, I, I, P],
    "pm_aug_jpeg_roundtrip_u8": [P, P, P, I, I, I, P],
    "pm_comm_unique_id": [P],
    "pm_comm_create": [P, P, I, I],
    "pm_comm_world": [P, P, P],
    "pm_comm_allreduce_f32": [P, P, P, P, I, P],
    "pm_comm_destroy": [P],
    "pm_vit_head_fwd": [P, I, I, P, P, P, P, P, P, P, P, P, I, I, I, F, P],
    "pm_vit_head_bwd": [P, P, P, I, I, P, P, P, P, P, P, P, P, I, P, P, P, P, I, I, I, P],
    "pm_supervised_loss_fwd": [P, P, P, P, P, P, I, I, P],
    "pm_scale": [P, P, P, L, P],
    "pm_cls_head_fwd": [P, I, P, P, P, P, P, P, P, P, I, I, I, F, P],
    "pm_cls_head_bwd": [P, P, I, P, P, P, P, P, P, P, I, P, P, P, P, I, I, I, P],
    "pm_adamw": [P, P, P, P, P, I, L, F, F, F, F, F, I, F, P],
    "pm_adamw_tick": [P, I, P],
    "pm_adamw_dev": [P, P, P, P, P, I, L, P, P],
    "pm_grad_stats": [P, L, P, P],
    "pm_loss_scale_update": [P, P, P, I, F, F, I, P],
    "pm_dgelu": [P, P, P, I, L, P],
}

ABI_VERSION = 14  # pm_abi_version() of the library these signatures describe
PM_GROUP_WHOLE_K = -1  # pm_wgrad_group(max_blocks=...): never slice, whole-K 256x256 tiles

WS_LAYERNORM_BWD, WS_COLSUM, WS_GEMM_COLSUM, WS_UNSHUFFLE_BWD = 1, 2, 3, 4


class GemmOpts(ctypes.Structure):
    """pm_gemm_opts of include/polypmae.h."""
    _fields_ = [("max_blocks", c_int), ("variant", c_int)]


class WgradItem(ctypes.Structure):
    """pm_wgrad_item of include/polypmae.h."""
    _fields_ = [("dY", c_void_p), ("lddy", c_long), ("X", c_void_p), ("ldx", c_long), ("dW", c_void_p), ("lddw", c_long),
                ("n_out", c_int), ("n_in", c_int), ("accumulate", c_int), ("dbias", c_void_p)]


class BlockBwdDesc(ctypes.Structure):
    """pm_block_bwd_desc of include/polypmae.h."""
    _fields_ = ([(n, c_void_p) for n in ("x_in", "x_mid", "ln1", "qkv", "attn", "ln2", "h_pre", "h_act", "mean1", "rstd1", "mean2",
                                        "rstd2", "lse", "norm1_w", "norm2_w", "qkv_w", "proj_w", "fc1_w", "fc2_w", "dx", "dx_act",
                                        "dmid", "dmid_act", "din", "din_act", "d_hidden", "d_qkv", "d_ln", "d_attn", "delta",
                                        "g_norm1_w", "g_norm1_b", "g_norm2_w", "g_norm2_b", "g_qkv_w", "g_proj_w", "g_fc1_w",
                                        "g_fc2_w", "g_qkv_b", "g_proj_b", "g_fc1_b", "g_below_bias")] +
                [("ws_ln", c_void_p), ("ws_ln_bytes", ctypes.c_size_t), ("ws_group", c_void_p), ("ws_group_bytes", ctypes.c_size_t)] +
                [(n, c_void_p) for n in ("side_stream", "ev_join", "ev_fork", "ev_done")] +
                [(n, c_int) for n in ("samples", "N", "D", "Hd", "heads", "dtype", "gemm_variant", "group_blocks", "accumulate",
                                      "two_groups")] +
                [(n, c_void_p) for n in ("side_stream2", "ev_fork2", "ev_done2")])


class BlockFwdDesc(ctypes.Structure):
    """pm_block_fwd_desc of include/polypmae.h."""
    _fields_ = ([(n, c_void_p) for n in ("x", "x_mid", "x_out", "ln1", "mean1", "rstd1", "qkv", "lse", "attn", "ln2", "mean2",
                                        "rstd2", "h_pre", "h_act", "norm1_w", "norm1_b", "norm2_w", "norm2_b", "qkv_w", "proj_w",
                                        "fc1_w", "fc2_w", "qkv_b", "proj_b", "fc1_b", "fc2_b")] +
                [(n, c_int) for n in ("rows", "samples", "N", "D", "Hd", "heads", "dtype", "gemm_variant")] +
                [("eps", ctypes.c_float)])


PM_ESHAPE = -2
_lib = None


class PolypMaeError(RuntimeError):
    pass


def load():
    """Load libpolypmae.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PolypMaeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU/eager fallback for the MI355X hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.pm_strerror.restype = c_char_p
    lib.pm_strerror.argtypes = [c_int]
    lib.pm_abi_version.restype = c_int
    lib.pm_abi_version.argtypes = []
    if lib.pm_abi_version() != ABI_VERSION:  # a stale build would be called with the wrong argument lists
        raise PolypMaeError(f"{LIB_PATH} has ABI version {lib.pm_abi_version()}, this package binds version {ABI_VERSION}: "
                            "rebuild with `python -c 'import __graft_entry__ as g; g.build(force=True)'`")
    lib.pm_gemm_workspace_bytes.restype = ctypes.c_size_t
    lib.pm_gemm_workspace_bytes.argtypes = [I, I, I, I, I, I, P]
    lib.pm_workspace_bytes.restype = ctypes.c_size_t
    lib.pm_workspace_bytes.argtypes = [I, I, I]
    lib.pm_aug_resized_crop_workspace_bytes.restype = ctypes.c_size_t
    lib.pm_aug_resized_crop_workspace_bytes.argtypes = [I, I, I, I]
    lib.pm_wgrad_group_workspace_bytes.restype = ctypes.c_size_t
    lib.pm_wgrad_group_workspace_bytes.argtypes = [P, I, I, I]
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = c_int
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        raise PolypMaeError(f"{what} failed: {load().pm_strerror(status).decode()} (status {status})")


def dtype_code(torch_dtype) -> int:
    import torch

    if torch_dtype == torch.float32:
        return PM_F32
    if torch_dtype == torch.bfloat16:
        return PM_BF16
    if torch_dtype == torch.float16:
        return PM_F16
    raise PolypMaeError(f"unsupported dtype {torch_dtype}")
