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


def test_supervised_loss_matches_reference_values(golden):
    fx = golden("tables.npz")
    z, y = torch.from_numpy(fx["bce/logits"]), torch.from_numpy(fx["bce/targets"])
    for pw in (1.0, 0.37, 2.5):
        got = T.supervised_loss(z, y, pos_weight=torch.tensor(pw))
        np.testing.assert_allclose(got.item(), fx[f"bce/pw{pw}"], rtol=1e-6)


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
    # the fine-tune loader of the reference reads ["model"] and copies name-matched tensors (models.py:168-170,186-194)
    vm = A.ViT_from_MAE(str(p), True, 2, False, None, embed_dim=64, depth=1, num_heads=2, out_token="cls") \
        if False else None  # (ViT_from_MAE fixes img 224 / patch 16: shapes differ from this toy MAE)
    m2 = A.MaskedAutoencoderViT(img_size=32, patch_size=8, embed_dim=64, depth=1, num_heads=2, decoder_embed_dim=32,
                                decoder_depth=1, decoder_num_heads=1)
    start = T.load_mae_checkpoint(p, m2, torch.optim.AdamW(m2.parameters(), lr=1e-3), args)
    assert start == 8 and args.start_epoch == 8
    for (n, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), n
    q = T.save_cls_checkpoint(tmp_path / "run" / "stem_e03_best.pth", 3, m, opt, None, 0.5, {"val_auprc": 0.9})
    ck = torch.load(q, map_location="cpu", weights_only=False)
    for key in ("epoch", "model_state_dict", "optimizer_state_dict", "scaler_state_dict", "scheduler_state_dict", "loss",
                "python_random_state", "numpy_random_state", "torch_rng_state", "val_auprc"):
        assert key in ck


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
