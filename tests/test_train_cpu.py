"""Host logic of the training-loop counterparts: LR schedules against the reference-generated tables, checkpoint
dictionary layout (misc.py:306-335, tc.py:7036-7067), timm-style weight-decay grouping."""
import os
from types import SimpleNamespace

import numpy as np
import torch

from ssl4polyp_amd import train as T
from ssl4polyp_amd.optim import add_weight_decay


def test_schedules_match_reference_tables(golden):
    fx = golden("tables.npz")
    lr, min_lr, wu, ep = fx["mae_lr/args"]
    args = SimpleNamespace(lr=lr, min_lr=min_lr, warmup_epochs=wu, epochs=ep)
    opt = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=0.1)
    opt.add_param_group({"params": [torch.zeros(1, requires_grad=True)], "lr_scale": 0.5})
    got = []
    for e in fx["mae_lr/epochs"]:
        T.adjust_learning_rate(opt, float(e), args)
        got.append([g["lr"] for g in opt.param_groups])
    np.testing.assert_allclose(np.array(got), fx["mae_lr/lrs"], rtol=1e-15)
    np.testing.assert_allclose([T.cls_cosine_lambda(e, 5, 100) for e in range(101)], fx["cls_lr/lambda_w5_e100"], rtol=1e-15)


def test_supervised_loss_is_hip_only():
    """The loss is a HIP op (pm_supervised_loss_fwd); the CPU restatement lives in oracle/ (checked against the
    reference-generated values in test_oracle_golden.py).  On a CPU tensor the product function must refuse."""
    import pytest
    from ssl4polyp_amd._lib import PolypMaeError
    with pytest.raises(PolypMaeError):
        T.supervised_loss(torch.zeros(4, 2), torch.zeros(4, dtype=torch.long), pos_weight=1.0)


def test_checkpoint_dict_layout(tmp_path):
    import ssl4polyp_amd as A
    m = A.MaskedAutoencoderViT(img_size=32, patch_size=8, embed_dim=64, depth=1, num_heads=2, decoder_embed_dim=32,
                               decoder_depth=1, decoder_num_heads=1)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    args = SimpleNamespace(lr=1e-3, epochs=3)
    p = T.save_mae_checkpoint(tmp_path / "ckpts", 7, m, opt, args)
    assert p.name == "checkpoint-7.pth" and (tmp_path / "ckpts" / "last.pth").is_symlink()
    ck = torch.load(p, map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch", "scaler", "args"} and ck["epoch"] == 7
    assert set(ck["model"]) == set(m.state_dict())
    m2 = A.MaskedAutoencoderViT(img_size=32, patch_size=8, embed_dim=64, depth=1, num_heads=2, decoder_embed_dim=32,
                                decoder_depth=1, decoder_num_heads=1)
    start = T.load_mae_checkpoint(p, m2, torch.optim.AdamW(m2.parameters(), lr=1e-3), args)
    assert start == 8 and args.start_epoch == 8
    for (n, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), n
    # GradScaler-shaped `scaler` entry: the reference's load_model feeds it to an ENABLED GradScaler (misc.py:349-350)
    assert set(ck["scaler"]) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
    if hasattr(torch.amp, "GradScaler"):
        torch.amp.GradScaler("cpu", enabled=True).load_state_dict(ck["scaler"])


def test_mae_checkpoint_feeds_the_finetune_loader(tmp_path):
    """Pre-train -> fine-tune hand-off (models.py:168-170,186-194): `get_MAE_backbone(weight_path=...)`-style loading of a
    file written by save_mae_checkpoint (which pickles an argparse.Namespace under "args": needs weights_only=False on
    torch >= 2.6).  Encoder toy-sized, decoder at the fixed 512/8/16 geometry ViT_from_MAE instantiates."""
    import argparse
    import ssl4polyp_amd as A
    torch.manual_seed(4)
    m = A.MaskedAutoencoderViT(embed_dim=64, depth=1, num_heads=2, decoder_embed_dim=512, decoder_depth=8,
                               decoder_num_heads=16)
    with torch.no_grad():
        for prm in m.parameters():
            if prm.requires_grad and prm.ndim == 1:
                prm.normal_(std=0.1)  # biases / LN affine away from their init values
    args = argparse.Namespace(lr=1e-3, epochs=3, model="mae_vit_base_patch16")
    p = T.save_mae_checkpoint(tmp_path, 0, m, torch.optim.AdamW(m.parameters(), lr=1e-3), args)
    vm = A.ViT_from_MAE(str(p), True, 2, False, None, embed_dim=64, depth=1, num_heads=2, out_token="cls")
    sd_m, sd_v = m.state_dict(), vm.state_dict()
    copied = [k for k in sd_v if k in sd_m]
    assert "blocks.0.attn.qkv.weight" in copied and "decoder_pos_embed" in copied and "lin_head.weight" not in copied
    for k in copied:
        assert torch.equal(sd_v[k], sd_m[k]), k
    assert not any(k.startswith(("decoder_blocks", "decoder_embed", "mask_token")) for k in sd_v)


def test_cls_checkpoint_layout_resume_and_pointer(tmp_path):
    """tc.py:7036-7111 (save) / 5667-5714 + 5976-5980 (resume) / 3914-3940 (pointer): key names the reference's resume
    code reads, RNG restore, pointer repair, parent-checkpoint start, asynchronous writer."""
    import random
    import ssl4polyp_amd as A

    def make():
        torch.manual_seed(8)
        return A.ViT_from_MAE(None, True, 2, False, None, embed_dim=64, depth=1, num_heads=2, out_token="cls")

    m = make()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: T.cls_cosine_lambda(e, 1, 10))
    stem = tmp_path / "run" / "ViTB_SUN_s13"
    name = T.cls_checkpoint_name(stem.name, 3, "AUPRC")
    assert name == "ViTB_SUN_s13_e03_AUPRC.pth"
    random.seed(5); np.random.seed(6); torch.manual_seed(7)
    writer = T.AsyncCheckpointWriter()
    q = T.save_cls_checkpoint(stem.parent / name, 3, m, opt, sched, 0.5,
                              {"val_auprc": 0.9, "monitor_value": 0.9, "monitor_metric": "val_auprc", "thresholds": {"t": 0.4}},
                              pointer=stem.with_suffix(".pth"), writer=writer)
    expect = (random.random(), float(np.random.rand()), float(torch.rand(1)))
    writer.wait()
    ck = torch.load(q, map_location="cpu", weights_only=False)
    for key in ("epoch", "model_state_dict", "optimizer_state_dict", "scaler_state_dict", "scheduler_state_dict", "loss",
                "py_state", "np_state", "torch_state", "val_auprc"):  # what tc.py:5672-5685,5977-5980 index
        assert key in ck, key
    assert (stem.with_suffix(".pth")).is_symlink() and os.readlink(stem.with_suffix(".pth")) == name
    # no scheduler -> the key is ABSENT (tc.py:7065-7066), not None
    q2 = T.save_cls_checkpoint(tmp_path / "other" / "x_e01_best.pth", 1, m, opt, None, 0.1)
    assert "scheduler_state_dict" not in torch.load(q2, map_location="cpu", weights_only=False)
    # resume: weights, optimizer step counters, scheduler epoch, the three RNG streams
    m2 = make()
    with torch.no_grad():
        m2.lin_head.weight.add_(1.0)
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-3)
    sched2 = torch.optim.lr_scheduler.LambdaLR(opt2, lambda e: T.cls_cosine_lambda(e, 1, 10))
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    info = T.load_cls_checkpoint(stem, m2, opt2, sched2)
    assert info.start_epoch == 4 and info.best_val_perf == 0.9 and info.resume_monitor_available
    assert info.thresholds == {"t": 0.4} and not info.from_parent
    assert (random.random(), float(np.random.rand()), float(torch.rand(1))) == expect
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    # pointer missing -> newest <stem>_e*_*.pth is found and the pointer is repaired (tc.py:3914-3926, 5689-5690)
    stem.with_suffix(".pth").unlink()
    found, valid = T.find_existing_checkpoint(stem)
    assert found.name == name and not valid
    T.load_cls_checkpoint(stem, make())
    assert stem.with_suffix(".pth").is_symlink()
    # parent start (tc.py:5692-5714): weights only, epoch 1, no optimizer state
    m3 = make()
    with torch.no_grad():
        m3.lin_head.bias.fill_(3.0)
    info3 = T.load_cls_checkpoint(tmp_path / "child" / "ViTB_SUN_fromX_s13", m3, parent_checkpoint=q)
    assert info3.from_parent and info3.start_epoch == 1 and info3.best_val_perf is None and info3.thresholds == {"t": 0.4}
    assert torch.equal(m3.lin_head.bias, m.lin_head.bias)
    writer.close()


def test_augreg_npz_loader(tmp_path):
    """SUP-imnet initialisation (models.py:51-55,68-77 -> timm 0.4.12 _load_weights): a JAX-layout .npz built here from
    known torch weights must come back as those weights (Dense kernels [in,out], q/k/v kernels [D,H,dh], out kernel
    [H,dh,D], HWIO patch embedding).  Parity unpinned: neither timm nor the published file is in the container."""
    import ssl4polyp_amd as A
    torch.manual_seed(12)
    D, H, depth = 64, 2, 2
    src = A.VisionTransformer_from_Any(True, 2, False, None, D, depth, H, "cls", False)
    with torch.no_grad():
        for prm in src.parameters():
            prm.normal_(std=0.05)
    sd = {k: v.numpy() for k, v in src.state_dict().items()}
    w = {"embedding/kernel": sd["patch_embed.proj.weight"].transpose(2, 3, 1, 0), "embedding/bias": sd["patch_embed.proj.bias"],
         "cls": sd["cls_token"], "Transformer/posembed_input/pos_embedding": sd["pos_embed"],
         "Transformer/encoder_norm/scale": sd["norm.weight"], "Transformer/encoder_norm/bias": sd["norm.bias"]}
    for i in range(depth):
        bp, tp = f"Transformer/encoderblock_{i}/", f"blocks.{i}."
        qkv_w, qkv_b = sd[tp + "attn.qkv.weight"], sd[tp + "attn.qkv.bias"]
        for j, n in enumerate(("query", "key", "value")):
            w[f"{bp}MultiHeadDotProductAttention_1/{n}/kernel"] = qkv_w[j * D:(j + 1) * D].T.reshape(D, H, D // H)
            w[f"{bp}MultiHeadDotProductAttention_1/{n}/bias"] = qkv_b[j * D:(j + 1) * D].reshape(H, D // H)
        w[bp + "MultiHeadDotProductAttention_1/out/kernel"] = sd[tp + "attn.proj.weight"].T.reshape(H, D // H, D)
        w[bp + "MultiHeadDotProductAttention_1/out/bias"] = sd[tp + "attn.proj.bias"]
        for r in range(2):
            w[f"{bp}MlpBlock_3/Dense_{r}/kernel"] = sd[tp + f"mlp.fc{r + 1}.weight"].T
            w[f"{bp}MlpBlock_3/Dense_{r}/bias"] = sd[tp + f"mlp.fc{r + 1}.bias"]
        w[bp + "LayerNorm_0/scale"], w[bp + "LayerNorm_0/bias"] = sd[tp + "norm1.weight"], sd[tp + "norm1.bias"]
        w[bp + "LayerNorm_2/scale"], w[bp + "LayerNorm_2/bias"] = sd[tp + "norm2.weight"], sd[tp + "norm2.bias"]
    path = tmp_path / "B_16-toy.npz"
    np.savez(path, **w)
    dst = A.VisionTransformer_from_Any(True, 2, False, None, D, depth, H, "cls", ImageNet_weights=str(path))
    for k, v in dst.state_dict().items():
        if not k.startswith("lin_head"):  # the classifier head is created after the load and stays at its init (models.py:56-60)
            assert torch.equal(v, src.state_dict()[k]), k


def test_add_weight_decay_groups():
    import ssl4polyp_amd as A
    m = A.MaskedAutoencoderViT(img_size=32, patch_size=8, embed_dim=64, depth=1, num_heads=2, decoder_embed_dim=32,
                               decoder_depth=1, decoder_num_heads=1)
    no_decay, decay = add_weight_decay(m, 0.05)
    assert no_decay["weight_decay"] == 0.0 and decay["weight_decay"] == 0.05
    ids = {id(p) for p in no_decay["params"]}
    for n, p in m.named_parameters():
        if not p.requires_grad:
            assert id(p) not in ids and all(id(p) != id(q) for q in decay["params"])  # frozen sincos tables
        elif p.ndim == 1 or n.endswith(".bias"):
            assert id(p) in ids, n
    # cls_token / mask_token are 3-D -> decayed, exactly as timm's add_weight_decay does for the reference
    assert any(p is m.cls_token for p in decay["params"])
