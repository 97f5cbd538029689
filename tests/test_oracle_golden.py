"""The CPU oracle against the golden vectors produced by running the reference
(tests/golden/make_fixtures.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import vit_mae_ref as O
from oracle import schedules_ref as S

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _sd(fx, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(prefix)}


def _grads(fn, sd, names):
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd2 = dict(sd)
    sd2.update(leaves)
    loss = fn(sd2)
    loss.backward()
    return {n: leaves[n].grad for n in names}


def test_meta_crosscheck_recorded():
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    cc = meta["transformers_crosscheck"]
    assert cc["mask_equal"] and cc["pred_max_abs_diff"] < 1e-5 and abs(cc["loss_ref"] - cc["loss_hf"]) < 1e-6
    assert meta["mae_vit_base_params"] == 111907840


def test_tiny_mae_forward_and_grads(golden):
    fx = golden("tiny_mae.npz")
    sd = _sd(fx, "w/")
    imgs, noise = torch.from_numpy(fx["imgs"]), torch.from_numpy(fx["noise"])
    loss, pred, mask = O.mae_forward(sd, imgs, noise, O.VIT_TINY)
    np.testing.assert_array_equal(mask.numpy(), fx["mask"])
    np.testing.assert_allclose(pred.numpy(), fx["pred"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(loss.item(), fx["loss"], rtol=1e-6)
    loss_np, pred_np, _ = O.mae_forward(sd, imgs, noise, O.VIT_TINY, norm_pix_loss=True)
    np.testing.assert_allclose(loss_np.item(), fx["loss_normpix"], rtol=1e-6)
    names = [k[2:] for k in fx if k.startswith("g/")]
    assert "pos_embed" not in names and "decoder_pos_embed" not in names  # frozen (models_mae.py:37,51)
    g = _grads(lambda s: O.mae_forward(s, imgs, noise, O.VIT_TINY)[0], sd, names)
    for n in names:
        ref = fx["g/" + n]
        np.testing.assert_allclose(g[n].numpy(), ref, rtol=0, atol=1e-6 + 2e-5 * np.abs(ref).max(), err_msg=n)


def test_tiny_classifiers(golden):
    fx = golden("tiny_cls.npz")
    cfg = O.ViTConfig(embed_dim=64, depth=2, num_heads=2)
    imgs, labels = torch.from_numpy(fx["imgs"]), torch.from_numpy(fx["labels"])
    sd = _sd(fx, "mae/w/")
    assert "decoder_pos_embed" in sd  # survives `del` of the decoder (models.py:171-175)
    logits = O.vit_classify(sd, imgs, cfg, learned_pos=False)
    np.testing.assert_allclose(logits.numpy(), fx["mae/logits"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(O.vit_classify(sd, imgs, cfg, out_token="spatial").numpy(),
                               fx["mae/logits_spatial"], rtol=0, atol=2e-6)
    pw = float(fx["pos_weight"])
    np.testing.assert_allclose(O.supervised_loss(logits, labels, pw).item(), fx["mae/loss"], rtol=1e-6)
    names = [k[len("mae/g/"):] for k in fx if k.startswith("mae/g/")]
    g = _grads(lambda s: O.supervised_loss(O.vit_classify(s, imgs, cfg), labels, pw), sd, names)
    for n in names:
        ref = fx["mae/g/" + n]
        np.testing.assert_allclose(g[n].numpy(), ref, rtol=0, atol=1e-7 + 2e-5 * np.abs(ref).max(), err_msg=n)
    sda = _sd(fx, "any/w/")
    np.testing.assert_allclose(O.vit_classify(sda, imgs, cfg, learned_pos=True).numpy(), fx["any/logits"],
                               rtol=0, atol=2e-6)


@pytest.mark.timeout(600)
def test_vith_mae_generated_weights(golden):
    """The oracle at the geometry of `mae_vit_huge_patch14` (models_mae.py:239-244: patch 14, D = 1280, 32 blocks, 16 heads of 80)
    against what the reference itself produced (tests/golden/make_mae_huge_grads.py): this pins the checker of the ViT-H GPU tests."""
    fx = golden("vith_mae_grads.npz")
    cfg = O.VIT_HUGE
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    sd = O.generated_state_dict(cfg, int(fx["weight_seed"]), decoder=True, n_class=None)
    imgs, _, noise = O.generated_batch(cfg, int(fx["batch"]), int(fx["batch_seed"]))
    names = [str(n) for n in fx["grad_names"]]
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd2 = dict(sd)
    sd2.update(leaves)
    loss, pred, mask = O.mae_forward(sd2, imgs, noise, cfg)
    loss.backward()
    assert pred.shape == (int(fx["batch"]), 256, 588)
    np.testing.assert_array_equal(mask.numpy().astype(np.uint8), fx["mask"])
    np.testing.assert_allclose(loss.item(), fx["loss"], rtol=2e-6)
    pred = pred.detach()
    np.testing.assert_allclose(pred[:, :8, :40].numpy(), fx["pred_slice"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(pred[:, -4:, -24:].numpy(), fx["pred_tail_slice"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(pred.abs().double().sum(dim=(1, 2)).numpy(), fx["pred_abs_sum_per_sample"], rtol=1e-5)
    norms = np.array([leaves[n].grad.double().norm().item() for n in names])
    np.testing.assert_allclose(norms, fx["grad_norms"], rtol=5e-4)
    for k in fx:
        if k.startswith("g/"):
            ref = fx[k]
            np.testing.assert_allclose(leaves[k[2:]].grad.numpy(), ref, rtol=0, atol=1e-7 + 2e-4 * np.abs(ref).max(), err_msg=k)
        elif k.startswith("g_corner/") or k.startswith("g_corner_last/"):
            n = k.split("/", 1)[1]
            g = leaves[n].grad.reshape(leaves[n].shape[0], -1)
            got = g[:32, :32] if k.startswith("g_corner/") else g[-32:, -32:]
            np.testing.assert_allclose(got.numpy(), fx[k], rtol=0, atol=1e-7 + 2e-4 * np.abs(fx[k]).max(), err_msg=k)


@pytest.mark.timeout(600)
def test_vitb_mae_generated_weights(golden):
    fx = golden("vitb_mae.npz")
    cfg = O.VIT_BASE
    sd = O.generated_state_dict(cfg, int(fx["weight_seed"]), decoder=True, n_class=None)
    imgs, _, noise = O.generated_batch(cfg, int(fx["batch"]), int(fx["batch_seed"]))
    names = [str(n) for n in fx["grad_names"]]
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd2 = dict(sd)
    sd2.update(leaves)
    loss, pred, mask = O.mae_forward(sd2, imgs, noise, cfg)
    loss.backward()
    np.testing.assert_array_equal(mask.numpy(), fx["mask"])
    np.testing.assert_allclose(loss.item(), fx["loss"], rtol=2e-6)
    np.testing.assert_allclose(pred[:, :8, :32].detach().numpy(), fx["pred_slice"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(pred.abs().sum(dim=(1, 2)).detach().numpy(), fx["pred_abs_sum_per_sample"], rtol=1e-5)
    norms = np.array([leaves[n].grad.norm().item() for n in names])
    np.testing.assert_allclose(norms, fx["grad_norms"], rtol=2e-4)
    for k in fx:
        if k.startswith("g/"):
            ref = fx[k]
            np.testing.assert_allclose(leaves[k[2:]].grad.numpy(), ref, rtol=0, atol=1e-7 + 5e-5 * np.abs(ref).max(), err_msg=k)
    np.testing.assert_allclose(leaves["blocks.5.mlp.fc1.weight"].grad[:16, :16].numpy(),
                               fx["g_slice/blocks.5.mlp.fc1.weight"], rtol=0, atol=1e-7)


@pytest.mark.timeout(600)
def test_vitb_classifiers_generated_weights(golden):
    fx = golden("vitb_cls.npz")
    cfg = O.VIT_BASE
    sd = O.generated_state_dict(cfg, int(fx["weight_seed"]), decoder=False, n_class=2)
    imgs, labels, _ = O.generated_batch(cfg, int(fx["batch"]), int(fx["batch_seed"]))
    np.testing.assert_array_equal(labels.numpy(), fx["labels"])
    names = [str(n) for n in fx["mae/grad_names"]]
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd2 = dict(sd)
    sd2.update(leaves)
    logits = O.vit_classify(sd2, imgs, cfg)
    loss = O.supervised_loss(logits, labels, float(fx["pos_weight"]))
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), fx["mae/logits"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(loss.item(), fx["mae/loss"], rtol=2e-6)
    norms = np.array([leaves[n].grad.norm().item() for n in names])
    np.testing.assert_allclose(norms, fx["mae/grad_norms"], rtol=2e-4)
    for k in fx:
        if k.startswith("mae/g/"):
            ref = fx[k]
            np.testing.assert_allclose(leaves[k[6:]].grad.numpy(), ref, rtol=0, atol=1e-8 + 5e-5 * np.abs(ref).max(), err_msg=k)
    sda = dict(sd)
    rng = np.random.Generator(np.random.PCG64(int(fx["any/pos_embed_seed"])))
    sda["pos_embed"] = torch.from_numpy(0.02 * rng.standard_normal((1, 197, 768))).float()
    np.testing.assert_allclose(O.vit_classify(sda, imgs, cfg, learned_pos=True).numpy(), fx["any/logits"],
                               rtol=0, atol=5e-6)


def test_tables(golden):
    fx = golden("tables.npz")
    for D in (768, 512, 64, 32):
        for gs in (14, 4):
            np.testing.assert_array_equal(O.sincos_2d(D, gs, cls_token=True), fx[f"sincos/{D}/{gs}"])
    ramp = torch.arange(2 * 3 * 32 * 32, dtype=torch.float32).reshape(2, 3, 32, 32)
    np.testing.assert_array_equal(O.patchify(ramp, 8).numpy(), fx["patchify_ramp"])
    np.testing.assert_array_equal(fx["unpatchify_ramp"], ramp.numpy())
    lr, min_lr, wu, ep = fx["mae_lr/args"]
    got = np.array([[S.mae_lr(e, lr, min_lr, wu, ep), 0.5 * S.mae_lr(e, lr, min_lr, wu, ep)] for e in fx["mae_lr/epochs"]])
    np.testing.assert_allclose(got, fx["mae_lr/lrs"], rtol=1e-15, atol=0)
    np.testing.assert_allclose([S.cls_cosine_lambda(e, 5, 100) for e in range(101)], fx["cls_lr/lambda_w5_e100"],
                               rtol=1e-15)
    z, y = torch.from_numpy(fx["bce/logits"]), torch.from_numpy(fx["bce/targets"])
    for pw in (1.0, 0.37, 2.5):
        np.testing.assert_allclose(O.supervised_loss(z, y, pw).item(), fx[f"bce/pw{pw}"], rtol=1e-6)


@pytest.mark.timeout(600)
def test_bf16_operand_rounding_alone_explains_the_bf16_tolerance(golden):
    """No kernel involved: rounding the contraction operands to bf16 (f32 accumulate) already moves ViT-B/16 logits by
    several 1e-3 relative and the MAE loss by < 1e-3 -- the bounds tests/test_gpu_models.py uses for the bf16 mode."""
    from oracle import vit_bf16_sim as Sim
    cfg = O.VIT_BASE
    fx = golden("vitb_cls.npz")
    sd = O.generated_state_dict(cfg, int(fx["weight_seed"]), decoder=False, n_class=2)
    imgs, _, noise = O.generated_batch(cfg, int(fx["batch"]), int(fx["batch_seed"]))
    with torch.no_grad():
        sim = Sim.vit_classify(sd, imgs, cfg)
    ref = torch.from_numpy(fx["mae/logits"])
    err = ((sim - ref).abs().max() / ref.abs().max()).item()
    assert 1e-4 < err < 2e-2, err
    fxm = golden("vitb_mae.npz")
    sdm = O.generated_state_dict(cfg, int(fxm["weight_seed"]), decoder=True, n_class=None)
    with torch.no_grad():
        loss, _, mask = Sim.mae_forward(sdm, imgs, noise, cfg)
    assert abs(loss.item() - float(fxm["loss"])) / float(fxm["loss"]) < 1e-3
    np.testing.assert_array_equal(mask.numpy(), fxm["mask"])


def _perturb_cases(golden):
    fx = golden("perturb.npz")
    rows = json.loads(str(fx["rows"]))
    for name in ("smooth", "noise", "odd"):
        img = fx[f"img/{name}"]
        for i, row in enumerate(rows):
            if f"out/{name}/{i}" in fx:
                yield name, i, row, img, fx[f"out/{name}/{i}"]
            elif f"same/{name}/{i}" in fx:
                yield name, i, row, img, img


def test_perturbation_plan_and_oracle_vs_reference(golden):
    """Eval-time perturbations (classification/data/transforms.py:143-203): the product's host-side plan (which perturbation, which
    parameters, which HMAC / rng_seed-seeded rectangle) followed by the oracle's pixel arithmetic (Pillow's box-blur GaussianBlur,
    the ImageEnhance blends, the inclusive rectangle) equals what the REFERENCE's PerRowPerturbations returned, bit for bit -- for
    every row of tests/golden/perturb.npz, the silently ignored spellings included -- the JPEG rows too: the oracle restates libjpeg's
    integer pipeline (colour conversion, islow DCT, quality-scaled quantisation and back) and never builds a bitstream."""
    from oracle import augment_ref as R
    from ssl4polyp_amd import data as D
    fx = golden("perturb.npz")
    rows = json.loads(str(fx["rows"]))
    for i, row in enumerate(rows):
        assert D._row_seed(row, D.DEFAULT_HMAC_KEY) == int(fx[f"seed/{i}"]), row
    kinds = set()
    for name, i, row, img, want in _perturb_cases(golden):
        plan = D.perturbation_plan(row)
        kinds.add(plan[0])
        if plan[0] == "none":
            got = img
        elif plan[0] == "blur":
            got = R.pil_gaussian_blur(img, plan[1])
        elif plan[0] == "bc":
            got = R.brightness_contrast(img, plan[1], plan[2])
        elif plan[0] == "occ":
            got = R.occlude(img, D.occlusion_rect(plan[1], plan[2], img.shape[1], img.shape[0]))
        else:   # the codec round trip restated without a bitstream (libjpeg's integer pipeline)
            got = R.jpeg_roundtrip(img, plan[1])
        assert np.array_equal(got, want), (name, i, row, plan)
    assert kinds == {"none", "blur", "bc", "occ", "jpeg"}
    # the box-blur weights the device kernel receives are the oracle's
    for sigma in (0.001, 0.5, 1.0, 1.5, 2.25, 3.0, 6.5, 10.0):
        r = R.pil_gaussian_box_radius(sigma)
        radius, ww, fw = D.pil_box_blur_params(sigma)
        assert radius == int(r) and ww == int(np.uint32(np.float32(16777216.0) / np.float32(r * np.float32(2) + np.float32(1))))
        assert fw == ((1 << 24) - (2 * radius + 1) * ww) // 2
