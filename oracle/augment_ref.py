"""CPU oracle for the device-side AUGMENTATION stages of the input pipeline (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates, on uint8 HWC frames, what the reference's train transform does to a PIL image before ToTensor
(classification/data/transforms.py:234-246):
    T.Resize((224, 224)); T.ColorJitter(brightness=0.4, contrast=0.5, saturation=0.25, hue=0.01);
    T.GaussianBlur((25, 25), sigma=(0.001, 2.0)); T.RandomHorizontalFlip(); T.RandomVerticalFlip(); T.RandomRotation(180)
torchvision 0.10 (absent here) runs these on PIL images through thin wrappers around Pillow (present here: the pin):
    Resize          -> img.resize((w, h), Image.BILINEAR)                     Pillow src/libImaging/Resample.c (8bpc path)
    adjust_brightness / contrast / saturation -> ImageEnhance.{Brightness, Contrast, Color}(img).enhance(f) = Image.blend
                                                                              Pillow src/libImaging/Blend.c, Convert.c (rgb2l)
    adjust_hue      -> img.convert("HSV"), h += uint8(f * 255) (wraps), convert("RGB")   Convert.c rgb2hsv / hsv2rgb
    rotate          -> img.rotate(angle, Image.NEAREST, expand=False, center=None, fillcolor=0)   Geometry.c affine_fixed
    gaussian_blur   -> tensor path of torchvision (functional_tensor.gaussian_blur): float32 conv2d with the outer product of
                       two 1-D gaussians, reflect padding, torch.round back to uint8   (NOT Pillow: restated, unpinned)
PINNED against Pillow 12.2 run in the build container (tests/golden/make_augment_fixtures.py -> tests/golden/augment.npz):
resize, brightness, contrast, saturation, hue, rotation -- bit for bit.  UNPINNED: the gaussian blur (torchvision absent; the
summation order of its conv2d is not reproducible anyway: this file fixes a separable f32 order and says so) and the RANDOM
DRAWS of the transforms (torch RNG consumption order of torchvision 0.10, restated in `draw_train_params`).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2   # Resample.c


# ---------------------------------------------------------------------------------------------------------------------
# Resize: Pillow ImagingResample, bilinear (triangle) filter with antialiasing, 8 bits per channel
# ---------------------------------------------------------------------------------------------------------------------
def _bicubic(x: float) -> float:
    """Resample.c bicubic_filter (a = -0.5), double arithmetic."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _resample_coeffs(in_size: int, out_size: int, filt: str = "bilinear"):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc over the whole input: bilinear (triangle, support 1) or bicubic
    (support 2)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = (1.0 if filt == "bilinear" else 2.0) * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ww = 0.0
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            w = (1.0 - abs(a) if abs(a) < 1.0 else 0.0) if filt == "bilinear" else _bicubic(a)
            kk[xx, x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                kk[xx, x] /= ww
        bounds[xx] = (xmin, xmax)
    # normalize_coeffs_8bpc: fixed point, round half away from zero
    ik = np.where(kk < 0, (-0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64))
    return bounds, ik.astype(np.int64), ksize


def _clip8(ss: np.ndarray) -> np.ndarray:
    return np.clip(ss >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resized_crop(frames: np.ndarray, boxes, out_h: int, out_w: int, filt: str = "bicubic") -> np.ndarray:
    """torchvision functional.resized_crop on PIL images: img.crop((left, top, left + w, top + h)).resize((out_w, out_h), filt),
    per sample.  boxes: [B][4] = (top, left, h, w).  (main_pretrain.py:157: RandomResizedCrop(224, scale=(0.2, 1), bicubic).)"""
    return np.concatenate([resize(frames[b:b + 1, t:t + h, l:l + w], out_h, out_w, filt) for b, (t, l, h, w) in enumerate(boxes)])


def resize_bilinear(frames: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    return resize(frames, out_h, out_w, "bilinear")


def resize(frames: np.ndarray, out_h: int, out_w: int, filt: str = "bilinear") -> np.ndarray:
    """uint8 [B, H, W, C] -> uint8 [B, out_h, out_w, C]: horizontal pass, then vertical pass, each rounded to uint8
    (ImagingResampleInner: a pass is skipped when the size does not change)."""
    x = frames
    B, H, W, C = x.shape
    if out_w != W:
        bounds, ik, _ = _resample_coeffs(W, out_w, filt)
        out = np.empty((B, H, out_w, C), dtype=np.uint8)
        xi = x.astype(np.int64)
        for xx in range(out_w):
            xmin, n = bounds[xx]
            ss = (1 << (PRECISION_BITS - 1)) + np.tensordot(xi[:, :, xmin:xmin + n, :], ik[xx, :n], axes=([2], [0]))
            out[:, :, xx, :] = _clip8(ss)
        x = out
    if out_h != H:
        bounds, ik, _ = _resample_coeffs(H, out_h, filt)
        out = np.empty((B, out_h, x.shape[2], C), dtype=np.uint8)
        xi = x.astype(np.int64)
        for yy in range(out_h):
            ymin, n = bounds[yy]
            ss = (1 << (PRECISION_BITS - 1)) + np.tensordot(xi[:, ymin:ymin + n, :, :], ik[yy, :n], axes=([1], [0]))
            out[:, yy, :, :] = _clip8(ss)
        x = out
    return x


# ---------------------------------------------------------------------------------------------------------------------
# ColorJitter pieces
# ---------------------------------------------------------------------------------------------------------------------
def rgb_to_l(x: np.ndarray) -> np.ndarray:
    """Convert.c rgb2l: L = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16."""
    xi = x.astype(np.int64)
    return ((xi[..., 0] * 19595 + xi[..., 1] * 38470 + xi[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(degenerate: np.ndarray, img: np.ndarray, factor: float) -> np.ndarray:
    """Blend.c ImagingBlend(degenerate, img, alpha): float32 arithmetic; interpolation truncates, extrapolation clips first."""
    a = np.float32(factor)
    d, i = degenerate.astype(np.float32), img.astype(np.float32)
    t = d + a * (i - d)          # (float) in1 + alpha * ((float) in2 - (float) in1)
    if 0.0 <= factor <= 1.0:
        return t.astype(np.uint8)  # (UINT8) cast: truncation
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int64))).astype(np.uint8)


def adjust_brightness(img: np.ndarray, f: float) -> np.ndarray:
    return blend(np.zeros_like(img), img, f)


def adjust_contrast(img: np.ndarray, f: float) -> np.ndarray:
    """ImageEnhance.Contrast: the degenerate image is a flat grey at int(mean(L) + 0.5), per image."""
    out = np.empty_like(img)
    for b in range(img.shape[0]):
        l = rgb_to_l(img[b])
        mean = int(l.astype(np.float64).sum() / l.size + 0.5)   # ImageStat.Stat(...).mean[0]: sum / count in Python floats
        out[b] = blend(np.full_like(img[b], mean), img[b], f)
    return out


def adjust_saturation(img: np.ndarray, f: float) -> np.ndarray:
    l = rgb_to_l(img)
    return blend(np.repeat(l[..., None], 3, axis=-1), img, f)


def rgb_to_hsv(x: np.ndarray) -> np.ndarray:
    """Convert.c rgb2hsv_row (uint8 in, uint8 out).  The C source mixes `float` variables with double literals, so the
    promotions matter: s, rc, gc, bc are float32 quotients; `2.0 + rc - bc`, `h / 6.0 + 1.0`, fmod and the `* 255.0` run in
    double and are rounded to float32 where they are assigned to `float h`."""
    r, g, b = (x[..., i].astype(np.int64) for i in range(3))
    maxc = np.maximum(np.maximum(r, g), b)
    minc = np.minimum(np.minimum(r, g), b)
    cr = (maxc - minc).astype(np.float32)
    safe = np.where(cr == 0, np.float32(1), cr)
    sat = (cr / np.where(maxc == 0, 1, maxc).astype(np.float32)).astype(np.float32)
    rc = ((maxc - r).astype(np.float32) / safe).astype(np.float32)
    gc = ((maxc - g).astype(np.float32) / safe).astype(np.float32)
    bc = ((maxc - b).astype(np.float32) / safe).astype(np.float32)
    h_r = (bc - gc).astype(np.float32)                                                    # float - float
    h_g = (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32)        # double, stored to float
    h_b = (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32)
    h = np.where(r == maxc, h_r, np.where(g == maxc, h_g, h_b)).astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)               # incorrect hue happens if h/6 is negative
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    us = np.clip((sat.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    grey = maxc == minc
    out = np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], axis=-1)
    return out.astype(np.uint8)


def hsv_to_rgb(x: np.ndarray) -> np.ndarray:
    """Convert.c hsv2rgb (float32 intermediates, round-half-up via (int)(v + 0.5) of CLIP8)."""
    h, s, v = x[..., 0], x[..., 1], x[..., 2]
    fh = (h.astype(np.float32) * np.float32(6.0) / np.float32(255.0)).astype(np.float32)
    i = np.floor(fh).astype(np.int64)
    f = (fh - i.astype(np.float32)).astype(np.float32)
    fs = (s.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    fv = v.astype(np.float32)

    def rnd(t):  # CLIP8(round(t)): Convert.c uses (UINT8) round()
        return np.clip(np.floor(t.astype(np.float64) + 0.5), 0, 255).astype(np.int64)
    p = rnd(fv * (np.float32(1.0) - fs))
    q = rnd(fv * (np.float32(1.0) - fs * f))
    t = rnd(fv * (np.float32(1.0) - fs * (np.float32(1.0) - f)))
    vv = v.astype(np.int64)
    i6 = i % 6
    r = np.choose(i6, [vv, q, p, p, t, vv])
    g = np.choose(i6, [t, vv, vv, q, p, p])
    b = np.choose(i6, [p, p, t, vv, vv, q])
    grey = s == 0
    out = np.stack([np.where(grey, vv, r), np.where(grey, vv, g), np.where(grey, vv, b)], axis=-1)
    return out.astype(np.uint8)


def adjust_hue(img: np.ndarray, f: float) -> np.ndarray:
    """functional_pil.adjust_hue: H channel += uint8(f * 255) with uint8 wrap-around."""
    hsv = rgb_to_hsv(img)
    with np.errstate(over="ignore"):
        hsv[..., 0] = (hsv[..., 0].astype(np.int64) + int(np.uint8(np.int64(f * 255) & 0xFF))).astype(np.uint8)  # np.uint8(f*255): C cast, wraps
    return hsv_to_rgb(hsv)


# ---------------------------------------------------------------------------------------------------------------------
# Rotation: Pillow Image.rotate -> transform(AFFINE, NEAREST) -> Geometry.c affine_fixed (16.16 fixed point)
# ---------------------------------------------------------------------------------------------------------------------
def rotate_matrix(angle_deg: float, w: int, h: int):
    """Image.rotate: the inverse map (output pixel -> input position) about the image centre, rounded to 15 decimals."""
    angle = angle_deg % 360.0
    rad = -math.radians(angle)
    m = [round(math.cos(rad), 15), round(math.sin(rad), 15), 0.0, round(-math.sin(rad), 15), round(math.cos(rad), 15), 0.0]
    cx, cy = w / 2.0, h / 2.0

    def tr(x, y):
        a, b, c, d, e, f = m
        return a * x + b * y + c, d * x + e * y + f
    m[2], m[5] = tr(-cx, -cy)
    m[2] += cx
    m[5] += cy
    return m


def rotate_nearest(img: np.ndarray, angle_deg: float, fill: int = 0) -> np.ndarray:
    """uint8 [B, H, W, C]; every sample by the same angle (callers loop for per-sample angles)."""
    B, H, W, C = img.shape
    ang = angle_deg % 360.0
    if ang == 0:
        return img.copy()
    if ang == 180:
        return img[:, ::-1, ::-1].copy()      # Image.rotate fast paths: transpose(ROTATE_180)
    if ang in (90, 270) and H == W:
        return np.rot90(img, k=1 if ang == 90 else 3, axes=(1, 2)).copy()
    a = rotate_matrix(angle_deg, W, H)

    def fix(v):
        return int(math.floor(v * 65536.0 + 0.5))
    a0, a1, a3, a4 = fix(a[0]), fix(a[1]), fix(a[3]), fix(a[4])
    a2 = fix(a[2] + a[0] * 0.5 + a[1] * 0.5)
    a5 = fix(a[5] + a[3] * 0.5 + a[4] * 0.5)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.int64)
    xx = a2 + a0 * xs + a1 * ys
    yy = a5 + a3 * xs + a4 * ys
    xin, yin = xx >> 16, yy >> 16                     # floor division of the 16.16 value
    ok = (xin >= 0) & (xin < W) & (yin >= 0) & (yin < H)
    out = np.full_like(img, fill)
    out[:, ok] = img[:, yin[ok], xin[ok]]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Gaussian blur (torchvision tensor path, restated: UNPINNED)
# ---------------------------------------------------------------------------------------------------------------------
def gaussian_kernel1d(ksize: int, sigma: float) -> np.ndarray:
    """functional_tensor._get_gaussian_kernel1d, float32."""
    half = (ksize - 1) * 0.5
    x = np.linspace(-half, half, ksize, dtype=np.float32)
    pdf = np.exp(np.float32(-0.5) * (x / np.float32(sigma)) ** 2).astype(np.float32)
    return (pdf / pdf.sum(dtype=np.float32)).astype(np.float32)


def gaussian_blur(img: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """uint8 [B, H, W, C] -> uint8: reflect padding, SEPARABLE float32 convolution (rows first, then columns, taps summed left to
    right / top to bottom), round half to even (torch.round), cast.  torchvision convolves with the 2-D outer-product kernel in
    one conv2d whose summation order is an implementation detail: the two agree to <= 1 grey level (tested)."""
    k = gaussian_kernel1d(ksize, sigma)
    r = ksize // 2
    x = img.astype(np.float32)
    xp = np.pad(x, ((0, 0), (0, 0), (r, r), (0, 0)), mode="reflect")
    acc = np.zeros_like(x)
    for t in range(ksize):
        acc = (acc + k[t] * xp[:, :, t:t + x.shape[2], :]).astype(np.float32)
    yp = np.pad(acc, ((0, 0), (r, r), (0, 0), (0, 0)), mode="reflect")
    out = np.zeros_like(x)
    for t in range(ksize):
        out = (out + k[t] * yp[:, t:t + x.shape[1], :, :]).astype(np.float32)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------------------------------------------------------
# the whole train-time chain on a batch, with explicit parameters
# ---------------------------------------------------------------------------------------------------------------------
def train_augment(frames: np.ndarray, params: dict) -> np.ndarray:
    """frames uint8 [B, H, W, 3] already at the target size; params (all per sample, length B):
    order [B, 4] (permutation of 0 brightness, 1 contrast, 2 saturation, 3 hue), brightness / contrast / saturation / hue
    factors, sigma, hflip, vflip (0/1), angle (degrees).  transforms.py:238-245 order: jitter, blur, flips, rotation."""
    out = np.empty_like(frames)
    fns = [adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue]
    names = ["brightness", "contrast", "saturation", "hue"]
    for b in range(frames.shape[0]):
        x = frames[b:b + 1]
        for op in params["order"][b]:
            x = fns[int(op)](x, float(params[names[int(op)]][b]))
        x = gaussian_blur(x, 25, float(params["sigma"][b]))
        if params["hflip"][b]:
            x = x[:, :, ::-1]
        if params["vflip"][b]:
            x = x[:, ::-1]
        x = rotate_nearest(np.ascontiguousarray(x), float(params["angle"][b]))
        out[b] = x[0]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Eval-time perturbations (classification/data/transforms.py:78-120, applied by PerRowPerturbations :143-203 between Resize and
# ToTensor): the pixel arithmetic of the three that are not a codec.  PINNED by running the reference's own PerRowPerturbations
# in the build container (tests/golden/make_perturb_fixtures.py -> tests/golden/perturb.npz; Pillow 12.2 underneath).
#   blur  img.filter(ImageFilter.GaussianBlur(radius=sigma))      Pillow src/libImaging/BoxBlur.c: ImagingGaussianBlur = three
#         "extended box" passes per axis (Gwosdek et al.): integer radius + a fractional weight on the two pixels beyond it
#   bc    ImageEnhance.Brightness(img).enhance(b), then ImageEnhance.Contrast(img).enhance(c)   (adjust_brightness / _contrast above)
#   occ   ImageDraw.Draw(img).rectangle([x0, y0, x1, y1], fill=(0, 0, 0)): both corners inclusive
# ---------------------------------------------------------------------------------------------------------------------
def pil_gaussian_box_radius(sigma: float, passes: int = 3) -> np.float32:
    """BoxBlur.c _gaussian_blur_radius: float variables, double intermediates where the C constants are double."""
    f32 = np.float32
    s = f32(sigma)
    sigma2 = f32(s * s / f32(passes))
    big_l = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(big_l) - 1.0) / 2.0))
    a = f32((f32(2) * l + f32(1)) * (l * (l + f32(1)) - f32(3) * sigma2))
    a = f32(a / f32(f32(6) * (sigma2 - (l + f32(1)) * (l + f32(1)))))
    return f32(l + a)


def _box_blur_lines(x: np.ndarray, fr: np.float32) -> np.ndarray:
    """One ImagingHorizontalBoxBlur pass along axis -2 of uint8 [..., n, C].  The C code carries a running sum along the line; each
    output equals  (ww * sum_{|d| <= radius} in[clamp(x + d)] + fw * (in[clamp(x - radius - 1)] + in[clamp(x + radius + 1)]) + 2^23) >> 24
    in UINT32 arithmetic, with  ww = (UINT32)((float) 2^24 / (fr * 2 + 1)),  fw = (2^24 - (2 radius + 1) * ww) / 2."""
    f32 = np.float32
    radius = int(fr)
    ww = np.uint64(np.uint32(f32(16777216.0) / f32(fr * f32(2) + f32(1))))
    fw = np.uint64(((1 << 24) - (radius * 2 + 1) * int(ww)) // 2 & 0xFFFFFFFF)
    n = x.shape[-2]
    idx = np.arange(n)
    xi = x.astype(np.uint64)
    acc = np.zeros_like(xi)
    for d in range(-radius, radius + 1):
        acc += np.take(xi, np.clip(idx + d, 0, n - 1), axis=-2)
    far = np.take(xi, np.clip(idx - radius - 1, 0, n - 1), axis=-2) + np.take(xi, np.clip(idx + radius + 1, 0, n - 1), axis=-2)
    mask = np.uint64(0xFFFFFFFF)
    bulk = (acc * ww + far * fw) & mask
    return (((bulk + np.uint64(1 << 23)) & mask) >> np.uint64(24)).astype(np.uint8)


def pil_gaussian_blur(img: np.ndarray, sigma: float, passes: int = 3) -> np.ndarray:
    """ImageFilter.GaussianBlur(radius=sigma) on uint8 [H, W, 3]: `passes` box passes along x, then along y (ImagingBoxBlur)."""
    r = pil_gaussian_box_radius(sigma, passes)
    out = img
    if r > 0:
        for _ in range(passes):
            out = _box_blur_lines(out, r)
        out = np.swapaxes(out, 0, 1)
        for _ in range(passes):
            out = _box_blur_lines(out, r)
        out = np.swapaxes(out, 0, 1)
    return np.ascontiguousarray(out)


def occlude(img: np.ndarray, rect) -> np.ndarray:
    """ImageDraw.rectangle([x0, y0, x1, y1], fill=0): corners inclusive, clipped to the image."""
    out = img.copy()
    if rect is not None:
        x0, y0, x1, y1 = rect
        out[max(0, y0):min(img.shape[0] - 1, y1) + 1, max(0, x0):min(img.shape[1] - 1, x1) + 1] = 0
    return out


def brightness_contrast(img: np.ndarray, brightness, contrast) -> np.ndarray:
    """transforms.py:88-96: each enhancement only when its factor is given and positive, brightness first."""
    out = img
    if brightness is not None and brightness > 0:
        out = adjust_brightness(out[None], brightness)[0]
    if contrast is not None and contrast > 0:
        out = adjust_contrast(out[None], contrast)[0]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# "jpeg" rows (transforms.py:78-85): img.save(buffer, format="JPEG", quality=q, optimize=False, subsampling=0), Image.open, convert.
# Entropy coding is lossless, so what comes back is libjpeg's integer pipeline and nothing else -- restated here without a bitstream:
#   jccolor.c   RGB -> YCbCr, 16-bit fixed point;  4:4:4 (subsampling=0)
#   jfdctint.c  "islow" forward DCT (Loeffler-Ligtenberg-Moschytz; CONST_BITS 13, PASS1_BITS 2; output scaled by 8)
#   jcparam.c   Annex-K tables scaled by the quality: 5000 / q below 50, 200 - 2 q from 50 on; (t * s + 50) / 100, baseline clamp 1..255
#   jcdctmgr.c  quantise with divisor (q << 3), round half away from zero;  jdcoefct / jddctmgr: coefficient * q
#   jidctint.c  islow inverse DCT, + 128, clamp;  jdcolor.c YCbCr -> RGB through its fixed-point tables
# Sides that are not multiples of 8 are padded by edge replication (jcprepct.c) and cropped.  PINNED by the jpeg rows of
# tests/golden/perturb.npz (made by the reference's own _apply_jpeg through PerRowPerturbations; Pillow 12.2 + its libjpeg-turbo).
# ---------------------------------------------------------------------------------------------------------------------
_JPEG_LUM = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                      18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101,
                      72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int64).reshape(8, 8)
_JPEG_CHR = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99]
                     + [99] * 32, dtype=np.int64).reshape(8, 8)
_JC = dict(a=2446, b=3196, c=4433, d=6270, e=7373, f=9633, g=12299, h=15137, i=16069, j=16819, k=20995, l=25172)


def _jdescale(x, n):
    return (x + (1 << (n - 1))) >> n


def _jfdct8(d, first):
    d0, d1, d2, d3, d4, d5, d6, d7 = (d[..., i] for i in range(8))
    t0, t7, t1, t6, t2, t5, t3, t4 = d0 + d7, d0 - d7, d1 + d6, d1 - d6, d2 + d5, d2 - d5, d3 + d4, d3 - d4
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    sh = 11 if first else 15
    o = [None] * 8
    o[0] = (t10 + t11) << 2 if first else _jdescale(t10 + t11, 2)
    o[4] = (t10 - t11) << 2 if first else _jdescale(t10 - t11, 2)
    z1 = (t12 + t13) * _JC["c"]
    o[2] = _jdescale(z1 + t13 * _JC["d"], sh)
    o[6] = _jdescale(z1 - t12 * _JC["h"], sh)
    z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
    z5 = (z3 + z4) * _JC["f"]
    a4, a5, a6, a7 = t4 * _JC["a"], t5 * _JC["j"], t6 * _JC["l"], t7 * _JC["g"]
    z1, z2, z3, z4 = -z1 * _JC["e"], -z2 * _JC["k"], -z3 * _JC["i"] + z5, -z4 * _JC["b"] + z5
    o[7], o[5], o[3], o[1] = (_jdescale(a4 + z1 + z3, sh), _jdescale(a5 + z2 + z4, sh), _jdescale(a6 + z2 + z3, sh),
                              _jdescale(a7 + z1 + z4, sh))
    return np.stack(o, -1)


def _jidct8(c, first):
    i0, i1, i2, i3, i4, i5, i6, i7 = (c[..., i] for i in range(8))
    z1 = (i2 + i6) * _JC["c"]
    e2, e3 = z1 - i6 * _JC["h"], z1 + i2 * _JC["d"]
    e0, e1 = (i0 + i4) << 13, (i0 - i4) << 13
    t10, t13, t11, t12 = e0 + e3, e0 - e3, e1 + e2, e1 - e2
    o0, o1, o2, o3 = i7, i5, i3, i1
    z1, z2, z3, z4 = o0 + o3, o1 + o2, o0 + o2, o1 + o3
    z5 = (z3 + z4) * _JC["f"]
    o0, o1, o2, o3 = o0 * _JC["a"], o1 * _JC["j"], o2 * _JC["l"], o3 * _JC["g"]
    z1, z2, z3, z4 = -z1 * _JC["e"], -z2 * _JC["k"], -z3 * _JC["i"] + z5, -z4 * _JC["b"] + z5
    o0, o1, o2, o3 = o0 + z1 + z3, o1 + z2 + z4, o2 + z2 + z3, o3 + z1 + z4
    sh = 11 if first else 18
    return np.stack([_jdescale(t10 + o3, sh), _jdescale(t11 + o2, sh), _jdescale(t12 + o1, sh), _jdescale(t13 + o0, sh),
                     _jdescale(t13 - o0, sh), _jdescale(t12 - o1, sh), _jdescale(t11 - o2, sh), _jdescale(t10 - o3, sh)], -1)


def jpeg_roundtrip(img: np.ndarray, quality: int) -> np.ndarray:
    """uint8 [H, W, 3] through a baseline 4:4:4 JPEG of the given quality and back, as Pillow (libjpeg-turbo) returns it."""
    q = max(1, min(100, int(quality)))
    scale = 5000 // q if q < 50 else 200 - 2 * q
    H, W, _ = img.shape
    Hp, Wp = (H + 7) // 8 * 8, (W + 7) // 8 * 8
    x = np.pad(img, ((0, Hp - H), (0, Wp - W), (0, 0)), mode="edge").astype(np.int64)
    r, g, b = x[..., 0], x[..., 1], x[..., 2]
    comps = ((19595 * r + 38470 * g + 7471 * b + 32768) >> 16,
             (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16,
             (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16)
    rec = []
    for comp, base in zip(comps, (_JPEG_LUM, _JPEG_CHR, _JPEG_CHR)):
        qt = np.clip((base * scale + 50) // 100, 1, 255)
        blk = comp.reshape(Hp // 8, 8, Wp // 8, 8).transpose(0, 2, 1, 3) - 128
        d = _jfdct8(blk, True)                                              # rows
        d = np.swapaxes(_jfdct8(np.swapaxes(d, -1, -2), False), -1, -2)     # columns
        qv = qt << 3
        qc = (np.abs(d) + (qv >> 1)) // qv
        coef = np.where(d < 0, -qc, qc) * qt
        ws = np.swapaxes(_jidct8(np.swapaxes(coef, -1, -2), True), -1, -2)  # columns
        rec.append(np.clip(_jidct8(ws, False) + 128, 0, 255).transpose(0, 2, 1, 3).reshape(Hp, Wp))
    y, cb, cr = rec[0], rec[1] - 128, rec[2] - 128
    out = np.stack([y + ((91881 * cr + 32768) >> 16), y + ((-22554 * cb + 32768 - 46802 * cr) >> 16), y + ((116130 * cb + 32768) >> 16)], -1)
    return np.clip(out, 0, 255).astype(np.uint8)[:H, :W]
