#!/usr/bin/env python3
"""Golden vectors for the augmentation oracle (oracle/augment_ref.py) by RUNNING PILLOW -- the library the reference's
torchvision 0.10 transforms end up in for PIL images (classification/data/transforms.py:234-246):

    python tests/golden/make_augment_fixtures.py          (build container: Pillow is installed, torchvision is not)

torchvision's PIL paths are thin wrappers (functional_pil.py): resize -> Image.resize(BILINEAR); adjust_brightness / contrast /
saturation -> ImageEnhance.{Brightness, Contrast, Color}.enhance; adjust_hue -> HSV split, uint8 add on H, merge, convert;
rotate -> Image.rotate(angle, NEAREST, expand=False, center=None, fillcolor=0); resized_crop (RandomResizedCrop of
mae/main_pretrain.py:157) -> Image.crop(box).resize(size, BICUBIC).  Each is called here exactly like that and the
uint8 result stored beside its input and parameters (tests/golden/augment.npz, ~250 KB).  GaussianBlur is torchvision tensor
code (no Pillow routine behind it) and therefore has no vector here: that stage stays parity-unpinned.
"""
import os

import numpy as np
from PIL import Image, ImageEnhance
import PIL

HERE = os.path.dirname(os.path.abspath(__file__))


def frame(h, w, seed, noise):
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 9.0 + c) * np.cos(yy / 13.0 - c) for c in range(3)], -1)
    return np.clip(base + rng.normal(0, noise, (h, w, 3)), 0, 255).astype(np.uint8)


def main():
    out = {"pillow_version": np.array(PIL.__version__)}
    imgs = {"smooth": frame(48, 64, 1, 12.0), "noise": np.random.Generator(np.random.PCG64(2)).integers(0, 256, (40, 40, 3), dtype=np.uint8)}
    for name, a in imgs.items():
        im = Image.fromarray(a)
        out[f"img/{name}"] = a
        for f in (0.6, 0.93, 1.0, 1.17, 1.4):
            out[f"brightness/{name}/{f}"] = np.asarray(ImageEnhance.Brightness(im).enhance(f))
        for f in (0.5, 0.88, 1.31, 1.5):
            out[f"contrast/{name}/{f}"] = np.asarray(ImageEnhance.Contrast(im).enhance(f))
        for f in (0.75, 1.06, 1.25):
            out[f"saturation/{name}/{f}"] = np.asarray(ImageEnhance.Color(im).enhance(f))
        for f in (-0.01, -0.0041, 0.0, 0.0039, 0.0079, 0.01):
            h, s, v = im.convert("HSV").split()
            nh = np.array(h, dtype=np.uint8)
            with np.errstate(over="ignore"):
                nh += np.uint8(np.int64(f * 255) & 0xFF)   # functional_pil.adjust_hue: np_h += np.uint8(hue_factor * 255)
            out[f"hue/{name}/{f}"] = np.asarray(Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB"))
        out[f"hsv/{name}"] = np.asarray(im.convert("HSV"))
        for ang in (0.0, 13.7, -77.3, 90.0, 123.456, 179.9, 180.0, -180.0, 45.0, 270.0, -0.4):
            out[f"rotate/{name}/{ang}"] = np.asarray(im.rotate(ang, Image.NEAREST, expand=False, center=None, fillcolor=0))
    for (h, w, oh, ow) in ((180, 240, 56, 56), (40, 50, 56, 56), (97, 56, 56, 56), (56, 133, 56, 56)):
        a = frame(h, w, 10 + h, 15.0)
        out[f"resize_in/{h}x{w}"] = a
        out[f"resize/{h}x{w}->{oh}x{ow}"] = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
    a = frame(150, 200, 77, 15.0)
    out["rrc_in"] = a
    for (t, l, h, w) in ((0, 0, 150, 200), (20, 30, 100, 120), (100, 150, 40, 37), (3, 5, 56, 56)):
        out[f"rrc/{t},{l},{h},{w}"] = np.asarray(Image.fromarray(a).crop((l, t, l + w, t + h)).resize((56, 56), Image.BICUBIC))
    path = os.path.join(HERE, "augment.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB (Pillow {PIL.__version__})")


if __name__ == "__main__":
    main()
