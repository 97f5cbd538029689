"""The augmentation oracle (oracle/augment_ref.py) against vectors produced by Pillow itself (tests/golden/augment.npz, made
by tests/golden/make_augment_fixtures.py): resize, brightness, contrast, saturation, hue and rotation bit for bit.  Plus the host
side of ssl4polyp_amd.data.DeviceAugmenter that needs no GPU: Pillow's resample taps, the blur taps and the rotation records
equal the oracle's, and the parameter draws follow the reference's ranges (transforms.py:238-245)."""
import numpy as np
import pytest
import torch


def test_oracle_equals_pillow_vectors(golden):
    from oracle import augment_ref as R
    fx = golden("augment.npz")
    fns = {"brightness": R.adjust_brightness, "contrast": R.adjust_contrast, "saturation": R.adjust_saturation, "hue": R.adjust_hue}
    n = 0
    for key, want in fx.items():
        parts = key.split("/")
        if parts[0] in fns:
            got = fns[parts[0]](fx["img/" + parts[1]][None], float(parts[2]))[0]
        elif parts[0] == "rotate":
            got = R.rotate_nearest(fx["img/" + parts[1]][None], float(parts[2]))[0]
        elif parts[0] == "hsv":
            got = R.rgb_to_hsv(fx["img/" + parts[1]])
        elif parts[0] == "resize":
            src, dst = parts[1].split("->")
            oh, ow = (int(v) for v in dst.split("x"))
            got = R.resize_bilinear(fx["resize_in/" + src][None], oh, ow)[0]
        elif parts[0] == "rrc":
            box = tuple(int(v) for v in parts[1].split(","))
            got = R.resized_crop(fx["rrc_in"][None], [box], 56, 56, "bicubic")[0]
        else:
            continue
        assert got.dtype == np.uint8 and np.array_equal(got, want), key
        n += 1
    assert n >= 60, n


def test_blur_restatement_against_a_direct_2d_convolution():
    """The blur is the one stage without a Pillow routine behind it (torchvision tensor code; torchvision is absent: parity
    unpinned).  What can be checked: the separable f32 restatement equals a direct float64 2-D convolution with the outer-product
    kernel (what torchvision's conv2d evaluates in some f32 order) to <= 1 grey level, on > 99 % of the pixels exactly."""
    from oracle import augment_ref as R
    rng = np.random.Generator(np.random.PCG64(5))
    img = rng.integers(0, 256, (2, 40, 44, 3), dtype=np.uint8)
    for sigma in (0.001, 0.4, 1.3, 2.0):
        got = R.gaussian_blur(img, 25, sigma)
        k = R.gaussian_kernel1d(25, sigma).astype(np.float64)
        xp = np.pad(img.astype(np.float64), ((0, 0), (12, 12), (12, 12), (0, 0)), mode="reflect")
        want = np.zeros(img.shape)
        for i in range(25):
            for j in range(25):
                want += k[i] * k[j] * xp[:, i:i + 40, j:j + 44, :]
        want = np.clip(np.rint(want), 0, 255)
        d = np.abs(got.astype(np.float64) - want)
        assert d.max() <= 1 and (d == 0).mean() > 0.99, (sigma, d.max(), (d == 0).mean())
    assert np.array_equal(R.gaussian_blur(img, 25, 0.001), img)   # sigma -> 0: the kernel is a delta


def test_host_builders_equal_the_oracle_and_draws_follow_the_reference_ranges():
    from oracle import augment_ref as R
    from ssl4polyp_amd import data as D
    for a, b in ((720, 224), (576, 224), (224, 224), (1350, 224), (130, 224)):
        bo, ik, ks = R._resample_coeffs(a, b)
        b2, t2, k2 = D._resample_coeffs(a, b)
        assert ks == k2 and np.array_equal(bo, b2) and np.array_equal(ik, t2), (a, b)
    assert np.array_equal(R.gaussian_kernel1d(25, 1.234), D._gaussian_taps(25, [1.234])[0])
    for ang in (0.0, 13.7, -77.3, 123.456, 179.9):
        g = D._rotation_geom(ang, 224, 224, 3)
        m = R.rotate_matrix(ang, 224, 224)
        if g[0] == 0:
            fix = lambda v: int(np.floor(v * 65536.0 + 0.5))
            assert g[1:7] == (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]),
                              fix(m[5] + m[3] * 0.5 + m[4] * 0.5))
        assert g[7] == 3
    assert D._rotation_geom(180.0, 224, 224, 0)[0] == 2 and D._rotation_geom(-90.0, 224, 224, 0)[0] == 4
    p = D.draw_train_params(4096, torch.Generator().manual_seed(1))
    assert all(sorted(r) == [0, 1, 2, 3] for r in p["order"])
    for key, lo, hi in (("brightness", 0.6, 1.4), ("contrast", 0.5, 1.5), ("saturation", 0.75, 1.25), ("hue", -0.01, 0.01),
                        ("sigma", 0.001, 2.0), ("angle", -180.0, 180.0)):
        v = p[key]
        assert v.min() >= lo and v.max() <= hi and abs(v.mean() - (lo + hi) / 2) < 0.05 * (hi - lo), key
    assert 0.4 < p["hflip"].mean() < 0.6 and 0.4 < p["vflip"].mean() < 0.6


def test_device_augmenter_is_gpu_only():
    from ssl4polyp_amd._lib import PolypMaeError
    from ssl4polyp_amd.data import DeviceAugmenter
    with pytest.raises(PolypMaeError):
        DeviceAugmenter("cpu")(torch.zeros(1, 224, 224, 3, dtype=torch.uint8))


def test_device_perturber_is_gpu_only_and_its_plan_is_pure_host_logic():
    """No CPU rendering path behind DevicePerturber (a CPU tensor is refused, not quietly handed to Pillow); the per-row plan is
    plain Python and does not need a device: spot values of the variant grammar (transforms.py:30-75, 149-203)."""
    from ssl4polyp_amd._lib import PolypMaeError
    from ssl4polyp_amd import data as D
    with pytest.raises(PolypMaeError):
        D.DevicePerturber("cpu")(torch.zeros(1, 8, 8, 3, dtype=torch.uint8), [{"variant": "blur_1"}])
    plan = D.perturbation_plan
    assert plan({"variant": "blur_1p5"}) == ("blur", 1.5) and plan({"variant": "blur_s1p5"}) == ("none",)
    assert plan({"variant": "blur_1", "blur_sigma": "2.25"}) == ("blur", 2.25) and plan({"variant": "blur_1", "blur_sigma": -1}) == ("blur", 1.0)
    assert plan({"variant": "jpeg_29p6"}) == ("jpeg", 30) and plan({"variant": "jpeg_90", "jpeg_q": 41.6}) == ("jpeg", 42)
    assert plan({"variant": "jpeg_q30"}) == ("none",) and plan({"variant": "jpeg_0"}) == ("jpeg", 1) and plan({"variant": "jpeg_250"}) == ("jpeg", 100)
    assert plan({"variant": "bc_b1p2_c0p8"}) == ("bc", 1.2, 0.8) and plan({"variant": "bc_c1p5"}) == ("bc", None, 1.5)
    assert plan({"variant": "bc_bminus1", "contrast": 0.5}) == ("bc", -1.0, 0.5)
    assert plan({"variant": "clean"}) == ("none",) and plan({}) == ("none",) and plan(None) == ("none",)
    assert plan({"variant": "", "perturbation_id": "blur_2"}) == ("blur", 2.0)
    assert plan({"variant": "blur_2", "render_in_pipeline": "no"}) == ("none",) and plan({"variant": "blur_2", "render_in_pipeline": "Yes"}) == ("blur", 2.0)
    occ = plan({"variant": "occ_a0p15", "rng_seed": "12"})
    assert occ == ("occ", 0.15, 12) and plan({"variant": "occ_aneg1"}) == ("none",)
    assert D.occlusion_rect(0.0, 1, 224, 224) is None
    fx0, fy0, fx1, fy1 = D.occlusion_rect(1.0, 1, 224, 224)   # area 1 at an aspect != 1: each side is clipped on its own, as there
    assert (fx1 - fx0 == 224 or fy1 - fy0 == 224) and 0 <= fx0 <= fx1 <= 224 and 0 <= fy0 <= fy1 <= 224
    x0, y0, x1, y1 = D.occlusion_rect(0.15, 12, 224, 224)
    assert 0 <= x0 < x1 <= 224 and 0 <= y0 < y1 <= 224 and 0.10 < (x1 - x0) * (y1 - y0) / 224 ** 2 < 0.20
