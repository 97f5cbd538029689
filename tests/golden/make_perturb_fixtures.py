#!/usr/bin/env python3
"""Golden vectors for the eval-time perturbations by RUNNING THE REFERENCE's own PerRowPerturbations
(src/ssl4polyp/classification/data/transforms.py:143-203; Pillow 12.2 underneath):

    python tests/golden/make_perturb_fixtures.py [--reference /root/reference]        (build container only)

transforms.py imports torchvision at module level (`from torchvision import transforms as T`, used by ClassificationTransforms
only); torchvision is not installed and cannot be, so an EMPTY module of that name is put into sys.modules for the import --
PerRowPerturbations itself touches Pillow, hmac and random only.  Each row below is fed with one of two frames; stored: the
frames, the rows (JSON) and what the reference returned.  Rows cover the spellings the parser accepts and the ones it silently
ignores ("blur_s1p5", "jpeg_q30": no number it can read -> the frame comes back unchanged), explicit metadata fields that win over
the name, HMAC-seeded and rng_seed-given occlusions, and the render_in_pipeline switch.
"""
import argparse
import importlib.util
import json
import os
import sys
import types

import numpy as np
from PIL import Image
import PIL

HERE = os.path.dirname(os.path.abspath(__file__))


def frame(h, w, seed, noise):
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 9.0 + c) * np.cos(yy / 13.0 - c) for c in range(3)], -1)
    return np.clip(base + rng.normal(0, noise, (h, w, 3)), 0, 255).astype(np.uint8)


ID = {"frame_path": "sun/case12/frame_000345.jpg", "frame_id": "frame_000345", "case_id": "case12"}
ROWS = [dict(ID, variant=v, perturbation_id=v) for v in (
    "clean", "blur_0p5", "blur_1", "blur_1p5", "blur_sigma_2", "blur_3", "blur_6p5", "blur_s1p5", "blur_0p001",
    "jpeg_30", "jpeg_q30", "jpeg_29p6", "jpeg_95",
    "bc_b1p2_c0p8", "bc_b0p6", "bc_c1p5", "bc_b1p4_c1p5", "bc_bminus1_c0p5", "bc_b0_c0",
    "occ_a0p15", "occ_0p05", "occ_a0p6", "occ_a1", "occ_a0p0001", "occ_aneg1", "weird_3")]
ROWS += [
    dict(ID, variant="blur_1", perturbation_id="x", blur_sigma=2.25),            # the field wins over the name
    dict(ID, variant="jpeg_90", perturbation_id="x", jpeg_q=41.6),
    dict(ID, variant="bc_b1p2_c0p8", perturbation_id="x", brightness=0.7, contrast=-1),
    dict(ID, variant="occ_a0p5", perturbation_id="x", bbox_area_frac="0.1", rng_seed=7),
    dict(ID, variant="occ_a0p1", perturbation_id="x", rng_seed="12345"),
    dict(ID, variant="", perturbation_id="blur_2"),                               # variant empty: perturbation_id names it
    dict(ID, variant="blur_2", perturbation_id="blur_2", render_in_pipeline="no"),
    dict(ID, variant="BLUR_1P25", perturbation_id="b"),
    dict(frame_path="other/img.png", frame_id=17, case_id=None, variant="occ_a0p2", perturbation_id="occ_a0p2"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    path = os.path.join(args.reference, "src", "ssl4polyp", "classification", "data", "transforms.py")
    spec = importlib.util.spec_from_file_location("ref_classification_transforms", path)
    ref = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = ref
    spec.loader.exec_module(ref)
    perturb = ref.PerRowPerturbations()
    frames = {"smooth": frame(64, 80, 3, 10.0), "noise": np.random.Generator(np.random.PCG64(4)).integers(0, 256, (56, 72, 3), dtype=np.uint8),
              "odd": frame(37, 53, 5, 25.0)}     # sides that are not multiples of 8 (JPEG pads its blocks by edge replication)
    out = {"pillow_version": np.array(PIL.__version__), "rows": np.array(json.dumps(ROWS))}
    for name, a in frames.items():
        out[f"img/{name}"] = a
        for i, row in enumerate(ROWS):
            if name == "noise" and i % 2:      # the second frame: every other row
                continue
            if name == "odd" and not str(row.get("variant", "")).lower().startswith(("jpeg", "blur_1", "occ_a0p15")):
                continue
            res = np.asarray(perturb(Image.fromarray(a), dict(row)))
            if np.array_equal(res, a):
                out[f"same/{name}/{i}"] = np.array(1, dtype=np.uint8)     # returned unchanged
            else:
                out[f"out/{name}/{i}"] = res
            out[f"seed/{i}"] = np.array(ref._row_hmac_seed(row, ref.DEFAULT_HMAC_KEY), dtype=np.uint64)
    dst = os.path.join(HERE, "perturb.npz")
    np.savez_compressed(dst, **out)
    print(f"wrote {dst}: {len(out)} arrays, {os.path.getsize(dst) / 1024:.0f} KiB (Pillow {PIL.__version__})")


if __name__ == "__main__":
    main()
