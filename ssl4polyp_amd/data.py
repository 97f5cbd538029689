"""Device-side tail of the input pipeline.

The reference decodes, resizes and augments with PIL on DataLoader workers and hands float32 NCHW batches to the GPU
(classification/data/transforms.py:225-253, mae/main_pretrain.py:156-191).  Here the workers stop at decoded uint8
HWC frames; `DevicePrefetcher` stages them through pinned host buffers, copies them on a dedicated stream while the
previous step computes (a uint8 batch is a quarter of the float32 bytes on PCIe) and runs the last three transform
stages on the device in one HBM-bound kernel (`pm_preprocess_u8`): optional horizontal / vertical flip, ToTensor,
Normalize -- bit-exact with torchvision's float32 arithmetic.  `DeviceAugmenter` (round 3) moves the rest of the train transform
there as well -- Resize, ColorJitter, GaussianBlur((25, 25)), flips, RandomRotation(180) -- with Pillow's own integer / float
arithmetic (`pm_aug_*`, csrc/pm_augment.hip), so the workers are left with JPEG decoding only.
"""
from __future__ import annotations

import concurrent.futures
import os
from typing import Iterable, Iterator, Optional, Sequence, Tuple

import torch

from . import _lib

IMAGENET_MEAN: Sequence[float] = (0.485, 0.456, 0.406)   # transforms.py:16
IMAGENET_STD: Sequence[float] = (0.229, 0.224, 0.225)    # transforms.py:17


def preprocess_u8(frames: torch.Tensor, flips: Optional[torch.Tensor] = None, mean: Sequence[float] = IMAGENET_MEAN,
                  std: Sequence[float] = IMAGENET_STD, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """frames: uint8 [B, H, W, 3] on the GPU; flips: uint8 [B] (bit 0 horizontal, bit 1 vertical) or None.
    Returns float32 [B, 3, H, W] = Normalize(mean, std)(ToTensor(frame)) of the (flipped) frames."""
    if frames.dtype != torch.uint8 or frames.ndim != 4 or frames.shape[-1] != 3 or not frames.is_contiguous():
        raise ValueError("frames must be a contiguous uint8 [B, H, W, 3] tensor")
    if not frames.is_cuda:
        raise _lib.PolypMaeError("preprocess_u8 runs on the GPU only (no CPU fallback)")
    B, H, W, _ = frames.shape
    if out is None:
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=frames.device)
    if flips is not None and (flips.dtype != torch.uint8 or flips.numel() != B or flips.device != frames.device):
        raise ValueError("flips must be uint8 [B] on the frames' device")
    lib = _lib.load()
    st = lib.pm_preprocess_u8(frames.data_ptr(), flips.data_ptr() if flips is not None else None, out.data_ptr(), B, H, W,
                              float(mean[0]), float(mean[1]), float(mean[2]), float(std[0]), float(std[1]), float(std[2]),
                              torch.cuda.current_stream(frames.device).cuda_stream)
    _lib.check(st, "pm_preprocess_u8")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# train-time augmentation on the device (classification/data/transforms.py:234-246)
# ---------------------------------------------------------------------------------------------------------------------
def _resample_coeffs(in_size: int, out_size: int):
    """Pillow Resample.c precompute_coeffs (bilinear filter, support 1, antialiased) + normalize_coeffs_8bpc: for every output
    index the first source index, the tap count and the 22-bit fixed-point taps.  Double arithmetic, as in the C source."""
    import math

    import numpy as np
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)], dtype=np.float64)
        ww = 0.0
        for v in w:   # (the C loop's summation order)
            ww += v
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    taps = np.where(kk < 0, (-0.5 + kk * (1 << 22)).astype(np.int64), (0.5 + kk * (1 << 22)).astype(np.int64)).astype(np.int32)
    return bounds, taps, ksize


def _gaussian_taps(ksize: int, sigma):
    """torchvision functional_tensor._get_gaussian_kernel1d in float32, one row per sigma."""
    import numpy as np
    half = (ksize - 1) * 0.5
    x = np.linspace(-half, half, ksize, dtype=np.float32)
    rows = []
    for sg in sigma:
        pdf = np.exp(np.float32(-0.5) * (x / np.float32(sg)) ** 2).astype(np.float32)
        rows.append((pdf / pdf.sum(dtype=np.float32)).astype(np.float32))
    return np.stack(rows)


def _rotation_geom(angle_deg: float, w: int, h: int, flips: int):
    """Image.rotate(angle, NEAREST, expand=False, center=None) -> the pm_aug_geom record: Pillow's inverse affine map about the
    image centre (coefficients rounded to 15 decimals, then FIX(v) = floor(v * 65536 + 0.5), half-pixel offsets folded into
    a2 / a5 as Geometry.c affine_fixed does), or one of its transpose fast paths."""
    import math
    ang = angle_deg % 360.0
    if ang == 0:
        return (1, 0, 0, 0, 0, 0, 0, flips)
    if ang == 180:
        return (2, 0, 0, 0, 0, 0, 0, flips)
    if ang in (90, 270) and w == h:
        return (3 if ang == 90 else 4, 0, 0, 0, 0, 0, 0, flips)
    rad = -math.radians(ang)
    m = [round(math.cos(rad), 15), round(math.sin(rad), 15), 0.0, round(-math.sin(rad), 15), round(math.cos(rad), 15), 0.0]
    cx, cy = w / 2.0, h / 2.0
    m[2] = m[0] * -cx + m[1] * -cy + m[2] + cx
    m[5] = m[3] * -cx + m[4] * -cy + m[5] + cy
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return (0, fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]),
            fix(m[5] + m[3] * 0.5 + m[4] * 0.5), flips)


def draw_train_params(B: int, generator: Optional[torch.Generator] = None, brightness=0.4, contrast=0.5, saturation=0.25, hue=0.01,
                      sigma=(0.001, 2.0), flip_p=0.5, degrees=180.0) -> dict:
    """The random draws of the reference's train transform for B samples (transforms.py:238-245; torchvision 0.10 get_params:
    ColorJitter -> a permutation of the four ops + one uniform factor each, GaussianBlur -> sigma ~ U(0.001, 2),
    RandomHorizontal/VerticalFlip -> rand < 0.5, RandomRotation -> angle ~ U(-180, 180)).  The reference draws per image inside
    its DataLoader workers; here one host generator serves the batch (the streams differ, the distributions do not)."""
    g = generator
    u = lambda lo, hi: torch.empty(B, dtype=torch.float64).uniform_(lo, hi, generator=g)
    order = torch.stack([torch.randperm(4, generator=g) for _ in range(B)])
    return {"order": order.numpy(), "brightness": u(max(0.0, 1 - brightness), 1 + brightness).numpy(),
            "contrast": u(max(0.0, 1 - contrast), 1 + contrast).numpy(), "saturation": u(max(0.0, 1 - saturation), 1 + saturation).numpy(),
            "hue": u(-hue, hue).numpy(), "sigma": u(sigma[0], sigma[1]).numpy(),
            "hflip": (torch.rand(B, generator=g) < flip_p).numpy(), "vflip": (torch.rand(B, generator=g) < flip_p).numpy(),
            "angle": u(-degrees, degrees).numpy()}


def draw_rrc_boxes(B: int, height: int, width: int, generator: Optional[torch.Generator] = None, scale=(0.2, 1.0),
                   ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision 0.10 RandomResizedCrop.get_params for B frames of one size -> int32 [B, 4] = (top, left, h, w)
    (mae/main_pretrain.py:157: scale (0.2, 1.0), default ratio): up to ten tries of area ~ U(scale) x aspect ~ logU(ratio), then
    the central-crop fallback."""
    import math

    import numpy as np
    g = generator
    area = height * width
    lr0, lr1 = math.log(ratio[0]), math.log(ratio[1])
    out = np.zeros((B, 4), dtype=np.int32)
    for b in range(B):
        for _ in range(10):
            target = area * torch.empty(1).uniform_(scale[0], scale[1], generator=g).item()
            aspect = math.exp(torch.empty(1).uniform_(lr0, lr1, generator=g).item())
            w = int(round(math.sqrt(target * aspect)))
            h = int(round(math.sqrt(target / aspect)))
            if 0 < w <= width and 0 < h <= height:
                i = torch.randint(0, height - h + 1, (1,), generator=g).item()
                j = torch.randint(0, width - w + 1, (1,), generator=g).item()
                out[b] = (i, j, h, w)
                break
        else:
            in_ratio = width / height
            if in_ratio < min(ratio):
                w = width
                h = int(round(w / min(ratio)))
            elif in_ratio > max(ratio):
                h = height
                w = int(round(h * max(ratio)))
            else:
                w, h = width, height
            out[b] = ((height - h) // 2, (width - w) // 2, h, w)
    return out


class DeviceAugmenter:
    """The reference's whole train transform after decoding, on the device: [Resize ->] ColorJitter -> GaussianBlur(25) -> flips
    -> RandomRotation(180) -> ToTensor -> Normalize, six launches over a uint8 batch (pm_aug_*), f32 NCHW out.  The host does
    what is scalar and per sample: the random draws, Pillow's resample taps (cached per size pair), the blur taps, the rotation's
    fixed-point matrix -- a few hundred bytes per batch, uploaded from pinned memory on the caller's stream."""

    KSIZE = 25  # transforms.py:239

    def __init__(self, device, size: int = 224, mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD):
        self.device, self.size, self.mean, self.std = torch.device(device), int(size), tuple(mean), tuple(std)
        self._coeffs = {}
        self._bufs = {}

    def _buf(self, name, shape, dtype):
        t = self._bufs.get(name)
        if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
            t = self._bufs[name] = torch.empty(shape, dtype=dtype, device=self.device)
        return t

    def _upload(self, name, arr):
        """numpy -> device through a pinned staging tensor kept per name (non-blocking; rewritten only after a host sync of the
        previous copy's event)."""
        import numpy as np
        host = torch.from_numpy(np.ascontiguousarray(arr))
        slot = self._bufs.get("pin_" + name)
        if slot is None or slot[0].shape != host.shape or slot[0].dtype != host.dtype:
            slot = self._bufs["pin_" + name] = [torch.empty(host.shape, dtype=host.dtype).pin_memory(), None]
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0].copy_(host)
        dev = self._buf("dev_" + name, host.shape, host.dtype)
        dev.copy_(slot[0], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        slot[1] = ev
        return dev

    def resize(self, frames: torch.Tensor) -> torch.Tensor:
        """uint8 [B, Hs, Ws, 3] -> uint8 [B, size, size, 3] (T.Resize((size, size)) on PIL images)."""
        B, Hs, Ws, _ = frames.shape
        S = self.size
        if (Hs, Ws) == (S, S):
            return frames
        key = (Hs, Ws, S)
        c = self._coeffs.get(key)
        if c is None:
            bx, tx, kx = _resample_coeffs(Ws, S)
            by, ty, ky = _resample_coeffs(Hs, S)
            c = self._coeffs[key] = tuple(torch.from_numpy(a).to(self.device) for a in (bx, tx, by, ty)) + (kx, ky)
        bx, tx, by, ty, kx, ky = c
        tmp = self._buf("rs_tmp", (B, Hs, S, 3), torch.uint8)
        out = self._buf("rs_out", (B, S, S, 3), torch.uint8)
        lib = _lib.load()
        _lib.check(lib.pm_aug_resize_u8(frames.data_ptr(), tmp.data_ptr(), out.data_ptr(), bx.data_ptr(), tx.data_ptr(), kx,
                                        by.data_ptr(), ty.data_ptr(), ky, B, Hs, Ws, S, S,
                                        torch.cuda.current_stream(self.device).cuda_stream), "pm_aug_resize_u8")
        return out

    def random_resized_crop(self, frames: torch.Tensor, boxes=None, generator: Optional[torch.Generator] = None,
                            bicubic: bool = True) -> torch.Tensor:
        """uint8 [B, Hs, Ws, 3] -> uint8 [B, size, size, 3]: RandomResizedCrop(size, scale=(0.2, 1.0), interpolation=bicubic) of
        the MAE pre-train transform (main_pretrain.py:157), one crop box per sample (drawn here unless given)."""
        import numpy as np
        if frames.dtype != torch.uint8 or frames.ndim != 4 or frames.shape[-1] != 3 or not frames.is_contiguous() or not frames.is_cuda:
            raise ValueError("frames must be a contiguous uint8 [B, H, W, 3] tensor on the GPU")
        B, Hs, Ws, _ = frames.shape
        S = self.size
        if boxes is None:
            boxes = draw_rrc_boxes(B, Hs, Ws, generator)
        boxes = np.ascontiguousarray(boxes, dtype=np.int32)
        if boxes.shape != (B, 4) or (boxes[:, 2:] <= 0).any() or (boxes[:, :2] < 0).any() or \
                (boxes[:, 0] + boxes[:, 2] > Hs).any() or (boxes[:, 1] + boxes[:, 3] > Ws).any():
            raise ValueError("crop boxes must be (top, left, h, w) inside the frame, one per sample")
        lib = _lib.load()
        need = int(lib.pm_aug_resized_crop_workspace_bytes(B, Hs, Ws, S))
        ws = self._buf("rrc_ws", (need,), torch.uint8)
        box_d = self._upload("rrc_box", boxes)
        out = self._buf("rrc_out", (B, S, S, 3), torch.uint8)
        _lib.check(lib.pm_aug_resized_crop_u8(frames.data_ptr(), box_d.data_ptr(), out.data_ptr(), 1 if bicubic else 0, B, Hs, Ws, S,
                                              ws.data_ptr(), ws.numel(), torch.cuda.current_stream(self.device).cuda_stream),
                   "pm_aug_resized_crop_u8")
        return out

    def mae_transform(self, frames: torch.Tensor, boxes=None, hflip=None, generator: Optional[torch.Generator] = None,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The MAE pre-train transform after decoding (main_pretrain.py:156-160): RandomResizedCrop(bicubic) -> RandomHorizontalFlip
        -> ToTensor -> Normalize, f32 [B, 3, size, size]."""
        B = frames.shape[0]
        x = self.random_resized_crop(frames, boxes, generator)
        if hflip is None:
            hflip = (torch.rand(B, generator=generator) < 0.5)
        flips = self._upload("mae_flips", torch.as_tensor(hflip).to(torch.uint8).numpy())
        return preprocess_u8(x, flips, self.mean, self.std, out=out)

    def __call__(self, frames: torch.Tensor, params: Optional[dict] = None, generator: Optional[torch.Generator] = None,
                 to_f32: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """frames uint8 [B, H, W, 3] on the device.  Returns f32 [B, 3, size, size] (normalised) or, with to_f32=False, the
        augmented uint8 frames [B, size, size, 3] (written into `out` when given)."""
        import numpy as np
        if frames.dtype != torch.uint8 or frames.ndim != 4 or frames.shape[-1] != 3 or not frames.is_contiguous():
            raise ValueError("frames must be a contiguous uint8 [B, H, W, 3] tensor")
        if not frames.is_cuda:
            raise _lib.PolypMaeError("DeviceAugmenter runs on the GPU only (no CPU fallback)")
        lib = _lib.load()
        st = torch.cuda.current_stream(self.device).cuda_stream
        x = self.resize(frames)
        B, H, W, _ = x.shape
        p = params if params is not None else draw_train_params(B, generator)
        # -- ColorJitter
        jit = np.zeros((B, 8), dtype=np.int32)
        jit[:, :4] = np.asarray(p["order"], dtype=np.int32)
        jit[:, 4:7] = np.stack([np.asarray(p[k], dtype=np.float32) for k in ("brightness", "contrast", "saturation")], 1).view(np.int32)
        jit[:, 7] = [int(np.int64(float(h) * 255)) & 0xFF for h in p["hue"]]   # np.uint8(hue_factor * 255): C cast, wraps
        jit_d = self._upload("jit", jit)
        lsum = self._buf("lsum", (B,), torch.int64)
        a = self._buf("aug_a", (B, H, W, 3), torch.uint8)
        _lib.check(lib.pm_aug_color_jitter_u8(x.data_ptr(), a.data_ptr(), jit_d.data_ptr(), lsum.data_ptr(), B, H, W, st),
                   "pm_aug_color_jitter_u8")
        # -- GaussianBlur((25, 25))
        taps_d = self._upload("taps", _gaussian_taps(self.KSIZE, p["sigma"]))
        tmp = self._buf("blur_tmp", (B, H, W, 3), torch.float32)
        b = self._buf("aug_b", (B, H, W, 3), torch.uint8)
        _lib.check(lib.pm_aug_gaussian_blur_u8(a.data_ptr(), tmp.data_ptr(), b.data_ptr(), taps_d.data_ptr(), self.KSIZE, B, H, W,
                                               st), "pm_aug_gaussian_blur_u8")
        # -- flips + rotation (+ ToTensor + Normalize)
        geom = np.array([_rotation_geom(float(p["angle"][i]), W, H, int(bool(p["hflip"][i])) | (int(bool(p["vflip"][i])) << 1))
                         for i in range(B)], dtype=np.int32)
        geom_d = self._upload("geom", geom)
        want = (B, 3, H, W) if to_f32 else (B, H, W, 3)
        if out is None:
            out = torch.empty(want, dtype=torch.float32 if to_f32 else torch.uint8, device=self.device)
        elif tuple(out.shape) != want or out.dtype != (torch.float32 if to_f32 else torch.uint8) or not out.is_contiguous():
            raise ValueError("`out` has the wrong shape / dtype")
        m, s = self.mean, self.std
        _lib.check(lib.pm_aug_geometry_u8(b.data_ptr(), geom_d.data_ptr(), out.data_ptr(), 1 if to_f32 else 0, B, H, W, float(m[0]),
                                          float(m[1]), float(m[2]), float(s[0]), float(s[1]), float(s[2]), st),
                   "pm_aug_geometry_u8")
        return out


# ------------------------------------------------------------------------------------------------------------------------------------
# Eval-time perturbations (classification/data/transforms.py:21-203): what PerRowPerturbations does to the resized PIL image of a
# row of an Exp-5A / 5B pack, planned on the host from the row's metadata and applied to the uint8 batch on the device
# ------------------------------------------------------------------------------------------------------------------------------------
DEFAULT_HMAC_KEY = b"ssl4polyp"   # transforms.py:18
_UNSET = (None, "", -1, "-1")
_UNSET_F = (None, "", -1, "-1", "-1.0")


def _variant_number(token: str):
    """A number as the variant names spell it (transforms.py:30-40): 'p' is the decimal point, 'minus' / 'neg' the sign."""
    t = token.strip().lower()
    if not t:
        return None
    t = t.replace("minus", "-").replace("neg", "-").replace("p", ".")
    try:
        return float(t)
    except ValueError:
        return None


def _last_number(variant: str):
    """The last '_'-separated token that reads as a number (transforms.py:43-49)."""
    for part in reversed(variant.split("_")):
        v = _variant_number(part)
        if v is not None:
            return v
    return None


def _row_seed(row, key: bytes) -> int:
    """transforms.py:123-140: HMAC-SHA256 over five metadata fields, first 8 bytes big-endian."""
    import hashlib
    import hmac
    msg = "|".join(str(row.get(f, "")) for f in ("frame_path", "frame_id", "case_id", "variant", "perturbation_id"))
    return int.from_bytes(hmac.new(key, msg.encode("utf-8"), hashlib.sha256).digest()[:8], "big", signed=False)


def occlusion_rect(area_fraction: float, seed: int, width: int, height: int):
    """transforms.py:99-120: the black rectangle of an "occ" row as (x0, y0, x1, y1), corners inclusive as ImageDraw.rectangle draws
    them, or None.  The draws come from Python's own random.Random(seed), in the reference's order."""
    import math
    import random
    a = max(0.0, min(float(area_fraction), 1.0))
    if a <= 0:
        return None
    rng = random.Random(seed)
    occ_area = max(1.0, a * (width * height))
    aspect = rng.uniform(0.5, 2.0)
    ow = max(1, min(width, int(round(math.sqrt(occ_area * aspect)))))
    oh = max(1, min(height, int(round(math.sqrt(occ_area / aspect)))))
    max_x, max_y = max(0, width - ow), max(0, height - oh)
    x0 = rng.randint(0, max_x) if max_x > 0 else 0
    y0 = rng.randint(0, max_y) if max_y > 0 else 0
    return x0, y0, min(width, x0 + ow), min(height, y0 + oh)


def perturbation_plan(row, key: bytes = DEFAULT_HMAC_KEY):
    """What PerRowPerturbations.__call__ (transforms.py:149-203) would do for this row, as a tuple:
    ("none",) | ("blur", sigma) | ("jpeg", quality) | ("bc", brightness or None, contrast or None) | ("occ", area_fraction, seed).
    Field values win over what the variant name spells; a variant whose number does not parse leaves the frame alone, as there."""
    if not row:
        return ("none",)
    flag = row.get("render_in_pipeline", True)
    if flag is None or not (flag if isinstance(flag, bool) else str(flag).strip().lower() in {"1", "true", "yes", "y"}):
        return ("none",)
    variant = str(row.get("variant") or row.get("perturbation_id") or "").strip()
    if not variant or variant.lower() == "clean":
        return ("none",)
    v = variant.lower()
    if v.startswith("blur"):
        f = row.get("blur_sigma")
        sigma = float(f) if f not in _UNSET_F else _last_number(v)
        return ("blur", sigma) if sigma is not None and sigma > 0 else ("none",)
    if v.startswith("jpeg"):
        f = row.get("jpeg_q")
        q = float(f) if f not in _UNSET else _last_number(v)
        if q is not None and f in _UNSET:
            q = float(int(round(q)))          # (_parse_quality rounds once, the caller once more)
        return ("jpeg", max(1, min(int(round(q)), 100))) if q is not None else ("none",)
    if v.startswith("bc"):
        fb, fc = row.get("brightness"), row.get("contrast")
        b = float(fb) if fb not in _UNSET_F else None
        c = float(fc) if fc not in _UNSET_F else None
        pb = pc = None
        for part in v.split("_"):
            if part.startswith("b"):
                pb = _variant_number(part[1:])
            elif part.startswith("c"):
                pc = _variant_number(part[1:])
        return ("bc", b if b is not None else pb, c if c is not None else pc)
    if v.startswith("occ"):
        f = row.get("bbox_area_frac")
        if f not in _UNSET_F:
            area = float(f)
        else:
            area = _variant_number(v.split("a", 1)[1] if "a" in v else v.split("_")[-1])
        if area is None or area <= 0:
            return ("none",)
        rs = row.get("rng_seed")
        return ("occ", area, int(rs) if rs not in _UNSET else _row_seed(row, key))
    return ("none",)


def pil_box_blur_params(sigma: float, passes: int = 3):
    """(radius, ww, fw) of Pillow's ImagingGaussianBlur for ImageFilter.GaussianBlur(radius=sigma) (BoxBlur.c: _gaussian_blur_radius, then
    ImagingHorizontalBoxBlur's fixed-point weights), in the float32 / UINT32 arithmetic of the C code.  radius = -1: nothing to blur."""
    import math
    import numpy as np
    f32 = np.float32
    s = f32(sigma)
    s2 = f32(s * s / f32(passes))
    big_l = f32(math.sqrt(12.0 * float(s2) + 1.0))
    l = f32(math.floor((float(big_l) - 1.0) / 2.0))
    a = f32((f32(2) * l + f32(1)) * (l * (l + f32(1)) - f32(3) * s2))
    a = f32(a / f32(f32(6) * (s2 - (l + f32(1)) * (l + f32(1)))))
    r = f32(l + a)
    if not r > 0:
        return -1, 0, 0
    radius = int(r)
    ww = int(np.uint32(f32(16777216.0) / f32(r * f32(2) + f32(1))))
    fw = ((1 << 24) - (radius * 2 + 1) * ww) // 2
    return radius, ww, fw & 0xFFFFFFFF


class DevicePerturber:
    """PerRowPerturbations (transforms.py:143-203) for a whole uint8 batch on the device: blur (Pillow's box-blur GaussianBlur),
    brightness / contrast (ImageEnhance blends), occlusion and the JPEG round trip (libjpeg's integer pipeline without a bitstream:
    the entropy coding is lossless), each bit for bit what Pillow returns for the row's image.  `jpeg_fn(frame_u8_hwc_numpy, quality)
    -> numpy`, when given, replaces the device JPEG stage by a host codec (A/B and other codecs).
    Frames are the RESIZED images (the perturbation sits between Resize and ToTensor: transforms.py:249-256)."""

    PASSES = 3  # ImageFilter.GaussianBlur -> ImagingGaussianBlur(..., passes=3)

    def __init__(self, device, key: bytes = DEFAULT_HMAC_KEY, jpeg_fn=None):
        self.device, self.key, self.jpeg_fn = torch.device(device), key, jpeg_fn
        self._aug = DeviceAugmenter(device)   # (its staging / scratch helpers)

    def eval_transform(self, frames: torch.Tensor, rows=None, size: int = 224, mean: Sequence[float] = IMAGENET_MEAN,
                       std: Sequence[float] = IMAGENET_STD) -> torch.Tensor:
        """ClassificationTransforms(stage="val" / "test", enable_perturbations=rows is not None) for a decoded uint8 batch
        [B, Hs, Ws, 3] of one frame size (transforms.py:234-256): Resize((size, size)) -> [the rows' perturbations] -> ToTensor ->
        Normalize, f32 [B, 3, size, size] out; three to ten launches, nothing leaves the device."""
        if self._aug.size != size:
            self._aug = DeviceAugmenter(self.device, size=size)
        x = self._aug.resize(frames)
        if rows is not None:
            x = self(x, rows)
        return preprocess_u8(x, None, mean, std)

    def batches(self, loader: Iterable, size: int = 224) -> Iterator[Tuple]:
        """For the evaluation loop: `loader` yields (decoded uint8 frames [B, Hs, Ws, 3] on the host, labels, rows) -- what
        PackDataset + pack_collate hand over before the transform (classification/data/packs.py:70-80) -- and this yields
        (f32 [B, 3, size, size] on the device, labels on the device, rows), i.e. what train.evaluate_cls iterates over, with the
        rows' perturbations rendered on the way (transforms.py:249-256)."""
        for frames, labels, rows in loader:
            x = self.eval_transform(torch.as_tensor(frames).to(self.device, non_blocking=True).contiguous(), rows, size=size)
            yield x, torch.as_tensor(labels).to(self.device, non_blocking=True), rows

    def __call__(self, frames: torch.Tensor, rows) -> torch.Tensor:
        """frames uint8 [B, H, W, 3] on the device, rows: one metadata mapping (or None) per frame.  Returns a new uint8 tensor."""
        import numpy as np
        if frames.dtype != torch.uint8 or frames.ndim != 4 or frames.shape[-1] != 3 or not frames.is_contiguous():
            raise ValueError("frames must be a contiguous uint8 [B, H, W, 3] tensor")
        if not frames.is_cuda:
            raise _lib.PolypMaeError("DevicePerturber runs on the GPU only (no CPU fallback)")
        B, H, W, _ = frames.shape
        if len(rows) != B:
            raise ValueError("one row per frame")
        plans = [perturbation_plan(r, self.key) for r in rows]
        lib = _lib.load()
        st = torch.cuda.current_stream(self.device).cuda_stream
        up, buf = self._aug._upload, self._aug._buf
        out = frames.clone()
        kinds = {p[0] for p in plans}
        if "jpeg" in kinds:
            if self.jpeg_fn is not None:
                for i, p in enumerate(plans):
                    if p[0] == "jpeg":
                        out[i].copy_(torch.from_numpy(np.array(self.jpeg_fn(frames[i].cpu().numpy(), p[1]), dtype=np.uint8)))
            else:
                qual = np.array([p[1] if p[0] == "jpeg" else 0 for p in plans], dtype=np.int32)   # 0: the sample is copied
                _lib.check(lib.pm_aug_jpeg_roundtrip_u8(out.data_ptr(), out.data_ptr(), up("pt_jpeg", qual).data_ptr(), B, H, W, st),
                           "pm_aug_jpeg_roundtrip_u8")
        if "bc" in kinds:
            jit = np.zeros((B, 8), dtype=np.int32)
            jit[:, :4] = -1
            fac = np.ones((B, 3), dtype=np.float32)
            for i, p in enumerate(plans):
                if p[0] != "bc":
                    continue
                ops = []
                if p[1] is not None and p[1] > 0:   # transforms.py:92-96: brightness first, then contrast, each only if > 0
                    ops.append(0)
                    fac[i, 0] = p[1]
                if p[2] is not None and p[2] > 0:
                    ops.append(1)
                    fac[i, 1] = p[2]
                jit[i, :len(ops)] = ops
            jit[:, 4:7] = fac.view(np.int32)
            src = buf("pt_src", (B, H, W, 3), torch.uint8)
            src.copy_(out)
            _lib.check(lib.pm_aug_color_jitter_u8(src.data_ptr(), out.data_ptr(), up("pt_jit", jit).data_ptr(),
                                                  buf("pt_lsum", (B,), torch.int64).data_ptr(), B, H, W, st), "pm_aug_color_jitter_u8")
        if "blur" in kinds:
            prm = np.zeros((B, 3), dtype=np.uint32)
            prm[:, 0] = np.uint32(0xFFFFFFFF)   # radius -1: pass through
            for i, p in enumerate(plans):
                if p[0] == "blur":
                    r, ww, fw = pil_box_blur_params(p[1], self.PASSES)
                    prm[i] = (np.uint32(r & 0xFFFFFFFF), ww, fw)
            tmp = buf("pt_tmp", (B, H, W, 3), torch.uint8)
            _lib.check(lib.pm_aug_pil_gaussian_blur_u8(out.data_ptr(), tmp.data_ptr(), out.data_ptr(), up("pt_blur", prm.view(np.int32)).data_ptr(),
                                                       self.PASSES, B, H, W, st), "pm_aug_pil_gaussian_blur_u8")
        if "occ" in kinds:
            rects = np.zeros((B, 4), dtype=np.int32)
            rects[:, 2] = -1   # x1 < x0: nothing
            for i, p in enumerate(plans):
                if p[0] == "occ":
                    r = occlusion_rect(p[1], p[2], W, H)
                    if r is not None:
                        rects[i] = r
            _lib.check(lib.pm_aug_occlude_u8(out.data_ptr(), up("pt_rects", rects).data_ptr(), B, H, W, st), "pm_aug_occlude_u8")
        return out


class DevicePrefetcher:
    """Wraps a loader that yields (frames uint8 [B,H,W,3] on the host, *rest): copies batch i+1 to the device on a
    side stream (pinned staging, two slots) while batch i is consumed, and yields (imgs float32 [B,3,H,W] on the
    device, *rest on the device).  The yielded image tensor is a per-slot buffer that is refilled two batches later
    (consume it within the step, as a training loop does).  `flip_p > 0` draws per-sample horizontal / vertical flips (RandomHorizontalFlip /
    RandomVerticalFlip of the reference's train transform) from `generator`."""

    def __init__(self, loader: Iterable, device, mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD,
                 flip_p: float = 0.0, generator: Optional[torch.Generator] = None, augment: Optional["DeviceAugmenter"] = None,
                 stream: str = "auto"):
        """augment: a DeviceAugmenter -> the loader may yield decoded frames of any (batch-uniform) size and the WHOLE train
        transform of the reference (Resize, ColorJitter, GaussianBlur(25), flips, RandomRotation(180), ToTensor, Normalize) runs
        on the copy stream; `flip_p` is then ignored (the augmenter draws its own flips from `generator`).
        LIMITATION: every frame of a batch must have the SAME decoded size (a batch is one [B, H, W, 3] array and the Resize taps
        are cached per (H, W)); the reference resizes image by image, and Hyperkvasir / SUN mix several native resolutions.  A
        loader over mixed-size data has to bucket frames by source size per batch (a batch sampler keyed on the manifest's size
        column) -- or take the per-sample path that `pm_aug_resized_crop_u8` already has for the MAE transform.  The device path
        is wired into bench.py and the tests; the cls training entry points still take whatever loader the caller passes.
        stream: where the copies and the transform run -- "own": a stream of the prefetcher (a fourth busy stream beside the
        engine's three: one hardware queue each, fastest on a single GPU); "side": the engine's weight-gradient stream (idle
        during the forward pass, when the next batch is staged) -- for data-parallel ranks, where RCCL's stream is the fourth busy
        one and a fifth would share a queue with the main stream (13.9 instead of 11.6 ms per step measured);
        "auto": "side" when a torch.distributed process group is initialised, "own" otherwise."""
        if stream not in ("auto", "own", "side"):
            raise ValueError("stream must be 'auto', 'own' or 'side'")
        self.stream_mode = stream
        self.loader, self.device = loader, torch.device(device)
        self.mean, self.std, self.flip_p, self.generator = mean, std, float(flip_p), generator
        self.augment = augment
        self._pinned = [None, None]
        self._flip_pin = [None, None]   # per slot: pinned flip flags
        self._dev = [None, None]        # per slot: (uint8 frames, float32 images) on the device
        self._consumed = [None, None]   # per slot: event recorded on the consumer's stream after it used the batch
        self._slot_copied = [None, None]  # per slot: event after the slot's host-to-device copies were enqueued
        self._stream: Optional[torch.cuda.Stream] = None

    def __len__(self):
        return len(self.loader)

    def _stage(self, slot: int, batch) -> Tuple:
        frames, rest = batch[0], tuple(batch[1:])
        if frames.dtype != torch.uint8:
            raise ValueError("DevicePrefetcher expects uint8 HWC frames from the loader")
        if frames.is_pinned():  # e.g. DataLoader(pin_memory=True): no staging copy
            pin = frames
        else:
            pin = self._pinned[slot]
            if pin is None or pin.shape != frames.shape:
                pin = torch.empty(frames.shape, dtype=torch.uint8).pin_memory()
                self._pinned[slot] = pin
            pin.copy_(frames)
        flips = None
        if self.flip_p > 0 and self.augment is None:
            r = torch.rand(2, frames.shape[0], generator=self.generator)
            f8 = ((r[0] < self.flip_p).to(torch.uint8) | ((r[1] < self.flip_p).to(torch.uint8) << 1))
            # pinned too: a pageable host-to-device copy is synchronous and would stall the enqueue of the step
            flips = self._flip_pin[slot]
            if flips is None or flips.numel() != f8.numel():
                flips = torch.empty(f8.numel(), dtype=torch.uint8).pin_memory()
                self._flip_pin[slot] = flips
            flips.copy_(f8)
        # device buffers are owned per slot and reused (no allocator traffic on the copy stream): the copy stream first
        # waits until the consumer's work on the batch that last used this slot has been enqueued AND executed
        bufs = self._dev[slot]
        if bufs is None or bufs[0].shape != frames.shape:
            B, H, W, _ = frames.shape
            S = self.augment.size if self.augment is not None else None
            bufs = (torch.empty(frames.shape, dtype=torch.uint8, device=self.device),
                    torch.empty((B, 3, S, S) if S else (B, 3, H, W), dtype=torch.float32, device=self.device))
            self._dev[slot] = bufs
        with torch.cuda.stream(self._stream):
            if self._consumed[slot] is not None:
                self._stream.wait_event(self._consumed[slot])
            bufs[0].copy_(pin, non_blocking=True)
            if self.augment is not None:
                imgs = self.augment(bufs[0], generator=self.generator, out=bufs[1])
            else:
                fl = flips.to(self.device, non_blocking=True) if flips is not None else None
                imgs = preprocess_u8(bufs[0], fl, self.mean, self.std, out=bufs[1])
            rest_dev = tuple(t.to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in rest)
            ev = torch.cuda.Event()
            ev.record(self._stream)
            self._slot_copied[slot] = ev
        return imgs, rest_dev, ev

    def __iter__(self) -> Iterator:
        if self.device.type != "cuda":
            raise _lib.PolypMaeError("DevicePrefetcher needs a GPU (the transform tail is a HIP kernel)")
        if self._stream is None:
            mode = {"0": "own", "1": "side"}.get(os.environ.get("PM_PREFETCH_ON_SIDE", ""), self.stream_mode)  # (A/B switch)
            if mode == "auto":
                import torch.distributed as dist
                mode = "side" if dist.is_available() and dist.is_initialized() else "own"
            if mode == "side":
                from .engine import _shared_stream
                self._stream = _shared_stream(self.device, "side")
            else:
                self._stream = torch.cuda.Stream(device=self.device)
        it = iter(self.loader)
        # Staging runs on a worker thread: the pinned-memory copy and the host-to-device enqueue block their caller for
        # about as long as a CPU memcpy of the batch (1 ms per 64 frames measured), and the main thread must spend that
        # time enqueueing the training step instead.
        pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)

        def job(slot, batch):
            torch.cuda.set_device(self.device)
            # this slot's pinned buffers (frames and / or flip flags) were last read by the copies issued two batches ago,
            # which wait on the consumer's event and may not have executed yet: drain them before the host rewrites them
            if self._slot_copied[slot] is not None:
                self._slot_copied[slot].synchronize()
            return self._stage(slot, batch)

        try:
            slot = 0
            try:
                fut = pool.submit(job, slot, next(it))
            except StopIteration:
                return
            while fut is not None:
                imgs, rest, ev = fut.result()
                cur, slot = slot, slot ^ 1
                try:
                    fut = pool.submit(job, slot, next(it))
                except StopIteration:
                    fut = None
                main = torch.cuda.current_stream(self.device)
                main.wait_event(ev)
                for t in rest:
                    if torch.is_tensor(t):
                        t.record_stream(main)
                yield (imgs,) + rest
                # the consumer has enqueued its work on this batch: the slot's buffers may be refilled once that work ran
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(self.device))
                self._consumed[cur] = done
        finally:
            pool.shutdown(wait=True)
