#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (the reference checkout does not travel):

    python tests/golden/make_fixtures.py [--reference /root/reference]

What it imports from the reference (unchanged, from its checkout):
    ssl4polyp.models.mae.models_mae      MaskedAutoencoderViT / mae_vit_base_patch16
    ssl4polyp.models.models              ViT_from_MAE, VisionTransformer_from_Any
    ssl4polyp.models.mae.util.pos_embed  get_2d_sincos_pos_embed
    ssl4polyp.models.mae.util.lr_sched   adjust_learning_rate

``timm==0.4.12`` (requirements.txt:5) is not installed and cannot be (no network),
so an in-process stand-in restating the four timm 0.4.12 classes the reference
uses (PatchEmbed, Attention/Mlp/Block, VisionTransformer) is installed in
``sys.modules`` first; ``np.float`` (removed in numpy>=1.24, used at
pos_embed.py:56) is aliased.  The stand-in is itself cross-checked against the
independent ViT-MAE implementation of the installed ``transformers`` package;
the result is recorded in ``meta.json``.

Outputs (all small; weights for ViT-B/16 come from oracle.generated_state_dict
so only inputs' seeds and outputs are stored):
    tiny_mae.npz      tiny MAE: reference-initialised weights + inputs + outputs + grads
    tiny_cls.npz      tiny ViT_from_MAE / VisionTransformer_from_Any classifiers
    vitb_mae.npz      ViT-B/16 MAE, generated weights (seed in file), B=2
    vitb_cls.npz      ViT-B/16 classifier (both variants), B=2
    tables.npz        sincos tables, patchify ramp, LR schedules, BCE values
    meta.json         provenance + transformers cross-check
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


# --------------------------------------------------------------------------
# stand-in for the absent timm==0.4.12 (algorithm as published in that release)
# --------------------------------------------------------------------------
def install_timm_standin():
    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
            super().__init__()
            self.img_size = (img_size, img_size)
            self.patch_size = (patch_size, patch_size)
            self.grid_size = (img_size // patch_size, img_size // patch_size)
            self.num_patches = self.grid_size[0] * self.grid_size[1]
            self.flatten = flatten
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
            self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

        def forward(self, x):
            B, C, H, W = x.shape
            assert H == self.img_size[0] and W == self.img_size[1]
            x = self.proj(x)
            if self.flatten:
                x = x.flatten(2).transpose(1, 2)
            return self.norm(x)

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, out_features)
            self.drop = nn.Dropout(drop)

        def forward(self, x):
            return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))

    class Attention(nn.Module):
        def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
            super().__init__()
            self.num_heads = num_heads
            self.scale = (dim // num_heads) ** -0.5
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.attn_drop = nn.Dropout(attn_drop)
            self.proj = nn.Linear(dim, dim)
            self.proj_drop = nn.Dropout(proj_drop)

        def forward(self, x):
            B, N, C = x.shape
            qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
            q, k, v = qkv[0], qkv[1], qkv[2]
            attn = (q @ k.transpose(-2, -1)) * self.scale
            attn = self.attn_drop(attn.softmax(dim=-1))
            x = (attn @ v).transpose(1, 2).reshape(B, N, C)
            return self.proj_drop(self.proj(x))

    class Block(nn.Module):
        def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0,
                     drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
            super().__init__()
            self.norm1 = norm_layer(dim)
            self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
            self.drop_path = nn.Identity()
            self.norm2 = norm_layer(dim)
            self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

        def forward(self, x):
            x = x + self.drop_path(self.attn(self.norm1(x)))
            x = x + self.drop_path(self.mlp(self.norm2(x)))
            return x

    class VisionTransformer(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                     num_heads=12, mlp_ratio=4.0, qkv_bias=True):
            super().__init__()
            norm_layer = partial(nn.LayerNorm, eps=1e-6)
            self.num_classes = num_classes
            self.num_features = self.embed_dim = embed_dim
            self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
            self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
            self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
            self.pos_drop = nn.Dropout(p=0.0)
            self.blocks = nn.Sequential(*[
                Block(embed_dim, num_heads, mlp_ratio, qkv_bias=qkv_bias, norm_layer=norm_layer)
                for _ in range(depth)])
            self.norm = norm_layer(embed_dim)
            self.pre_logits = nn.Identity()
            self.head = nn.Linear(embed_dim, num_classes)
            nn.init.trunc_normal_(self.pos_embed, std=0.02)
            nn.init.trunc_normal_(self.cls_token, std=0.02)
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    nn.init.trunc_normal_(m.weight, std=0.02)
                    nn.init.zeros_(m.bias)
                elif isinstance(m, nn.LayerNorm):
                    nn.init.zeros_(m.bias)
                    nn.init.ones_(m.weight)

    import importlib.machinery
    timm = types.ModuleType("timm")
    timm.__version__ = "0.4.12"
    models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    for mod in (timm, models, vt):
        mod.__spec__ = importlib.machinery.ModuleSpec(mod.__name__, None)
    vt.PatchEmbed, vt.Block, vt.Attention, vt.Mlp, vt.VisionTransformer = PatchEmbed, Block, Attention, Mlp, VisionTransformer
    timm.models = models
    models.vision_transformer = vt
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.vision_transformer": vt})
    if not hasattr(np, "float"):
        np.float = float  # pos_embed.py:56


def t2n(t):
    return t.detach().cpu().numpy()


def sd2npz(sd, prefix="w/"):
    return {prefix + k: t2n(v) for k, v in sd.items()}


def load_generated(model, sd):
    own = model.state_dict()
    assert set(own.keys()) == set(sd.keys()), (set(own) ^ set(sd))
    model.load_state_dict(sd, strict=True)


def run_ref_mae(model, imgs, noise, mask_ratio=0.75):
    """Run reference forward with `noise` substituted for torch.rand (models_mae.py:132)."""
    real_rand = torch.rand
    calls = []

    def fake_rand(*shape, **kw):
        calls.append(shape)
        assert tuple(shape) == tuple(noise.shape), shape
        return noise.clone()

    torch.rand = fake_rand
    try:
        out = model(imgs, mask_ratio=mask_ratio)
    finally:
        torch.rand = real_rand
    assert len(calls) == 1
    return out


def named_grads(model, names):
    params = dict(model.named_parameters())
    return {n: t2n(params[n].grad) for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(args.reference, "src"))
    install_timm_standin()
    torch.set_num_threads(8)
    torch.manual_seed(0)

    from ssl4polyp.models.mae import models_mae
    from ssl4polyp.models import models as ref_models
    from ssl4polyp.models.mae.util import pos_embed as ref_pos
    from ssl4polyp.models.mae.util import lr_sched as ref_lr
    from oracle import vit_mae_ref as O

    meta = {"reference": "irconde/SSL4POLYP", "torch": torch.__version__, "numpy": np.__version__,
            "timm": "stand-in for 0.4.12 (absent)"}

    # ---------------- tiny MAE (reference init, weights stored) ----------------
    cfg = O.VIT_TINY
    for norm_pix in (False, True):
        torch.manual_seed(11)
        m = models_mae.MaskedAutoencoderViT(
            img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth,
            num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
            decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=4,
            norm_layer=partial(nn.LayerNorm, eps=1e-6), norm_pix_loss=norm_pix)
        # perturb biases / LN affine / tokens so zero-initialised terms are exercised
        g = torch.Generator().manual_seed(12)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.requires_grad and (p.ndim == 1 or n in ("cls_token", "mask_token")):
                    p.add_(0.05 * torch.randn(p.shape, generator=g))
        imgs, _, noise = O.generated_batch(cfg, 4, seed=21)
        loss, pred, mask = run_ref_mae(m, imgs, noise)
        loss.backward()
        if not norm_pix:
            out = sd2npz(m.state_dict())
            out.update(imgs=t2n(imgs), noise=t2n(noise), loss=t2n(loss), pred=t2n(pred), mask=t2n(mask))
            for n, gnp in named_grads(m, [n for n, p in m.named_parameters() if p.grad is not None]).items():
                out["g/" + n] = gnp
            tiny_mae = out
        else:
            tiny_mae.update(loss_normpix=t2n(loss), pred_normpix=t2n(pred))
            tiny_mae["g_normpix/decoder_pred.bias"] = t2n(dict(m.named_parameters())["decoder_pred.bias"].grad)
    np.savez_compressed(os.path.join(HERE, "tiny_mae.npz"), **tiny_mae)

    # ---------------- tiny classifiers ----------------
    tiny_cls = {}
    ccfg = O.ViTConfig(embed_dim=64, depth=2, num_heads=2)  # img 224 / patch 16 are fixed by models.py:155-165
    torch.manual_seed(13)
    vm = ref_models.ViT_from_MAE(None, True, 2, False, None, embed_dim=64, depth=2, num_heads=2, out_token="cls")
    g = torch.Generator().manual_seed(14)
    with torch.no_grad():
        for n, p in vm.named_parameters():
            if p.requires_grad and (p.ndim == 1 or n == "cls_token"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    imgs, labels, _ = O.generated_batch(ccfg, 2, seed=22)
    logits = vm(imgs)
    lossf = nn.BCEWithLogitsLoss(pos_weight=torch.tensor(1.7))
    z = logits[:, 1] - logits[:, 0]                      # tc.py:3347-3359
    loss = lossf(z, labels.to(z.dtype))                  # tc.py:3370-3373
    loss.backward()
    tiny_cls.update(sd2npz(vm.state_dict(), "mae/w/"))
    tiny_cls.update({"imgs": t2n(imgs), "labels": t2n(labels), "mae/logits": t2n(logits), "mae/loss": t2n(loss),
                     "pos_weight": np.float32(1.7)})
    for n, gnp in named_grads(vm, [n for n, p in vm.named_parameters() if p.grad is not None]).items():
        tiny_cls["mae/g/" + n] = gnp
    vm.out_token = "spatial"
    tiny_cls["mae/logits_spatial"] = t2n(vm(imgs))
    assert vm.head is True and not isinstance(vm.head, nn.Module)  # models.py:177 quirk

    torch.manual_seed(15)
    va = ref_models.VisionTransformer_from_Any(True, 2, False, None, 64, 2, 2, "cls", False)
    with torch.no_grad():
        for n, p in va.named_parameters():
            if p.ndim == 1:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    logits = va(imgs)
    tiny_cls.update(sd2npz(va.state_dict(), "any/w/"))
    tiny_cls["any/logits"] = t2n(logits)
    np.savez_compressed(os.path.join(HERE, "tiny_cls.npz"), **tiny_cls)

    # ---------------- ViT-B/16, generated weights ----------------
    cfg = O.VIT_BASE
    B = 2
    sd = O.generated_state_dict(cfg, seed=101, decoder=True, n_class=None)
    m = models_mae.mae_vit_base_patch16(norm_pix_loss=False)
    load_generated(m, sd)
    nparams = sum(p.numel() for p in m.parameters())
    meta["mae_vit_base_params"] = nparams
    imgs, labels, noise = O.generated_batch(cfg, B, seed=202)
    loss, pred, mask = run_ref_mae(m, imgs, noise)
    loss.backward()
    gn = {n: float(p.grad.norm()) for n, p in m.named_parameters() if p.grad is not None}
    vitb_mae = dict(weight_seed=101, batch_seed=202, batch=B, loss=t2n(loss), mask=t2n(mask),
                    pred_slice=t2n(pred[:, :8, :32]), pred_mean=t2n(pred.mean()), pred_std=t2n(pred.std()),
                    pred_abs_sum_per_sample=t2n(pred.abs().sum(dim=(1, 2))),
                    grad_names=np.array(list(gn.keys())), grad_norms=np.array(list(gn.values()), dtype=np.float64))
    gsel = ["decoder_pred.bias", "blocks.0.attn.qkv.bias", "blocks.11.mlp.fc2.bias", "cls_token", "mask_token",
            "decoder_blocks.7.norm2.weight", "patch_embed.proj.bias"]
    for n, gnp in named_grads(m, gsel).items():
        vitb_mae["g/" + n] = gnp
    vitb_mae["g_slice/blocks.5.mlp.fc1.weight"] = t2n(dict(m.named_parameters())["blocks.5.mlp.fc1.weight"].grad[:16, :16])
    np.savez_compressed(os.path.join(HERE, "vitb_mae.npz"), **vitb_mae)

    # transformers ViT-MAE cross-check of the timm stand-in (independent implementation)
    try:
        from transformers import ViTMAEConfig, ViTMAEForPreTraining
        hcfg = ViTMAEConfig(layer_norm_eps=1e-6, norm_pix_loss=False, attn_implementation="eager")
        hm = ViTMAEForPreTraining(hcfg).eval()
        hsd = {}
        hsd["vit.embeddings.cls_token"] = sd["cls_token"]
        hsd["vit.embeddings.position_embeddings"] = sd["pos_embed"]
        hsd["vit.embeddings.patch_embeddings.projection.weight"] = sd["patch_embed.proj.weight"]
        hsd["vit.embeddings.patch_embeddings.projection.bias"] = sd["patch_embed.proj.bias"]

        def map_block(src, dst, D):
            qw, qb = sd[src + "attn.qkv.weight"], sd[src + "attn.qkv.bias"]
            for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                hsd[dst + f"attention.{nm}.weight"] = qw[j * D:(j + 1) * D]
                hsd[dst + f"attention.{nm}.bias"] = qb[j * D:(j + 1) * D]
            for a, b in (("attn.proj", "attention.o_proj"), ("norm1", "layernorm_before"),
                         ("norm2", "layernorm_after"), ("mlp.fc1", "mlp.fc1"), ("mlp.fc2", "mlp.fc2")):
                hsd[dst + b + ".weight"] = sd[src + a + ".weight"]
                hsd[dst + b + ".bias"] = sd[src + a + ".bias"]

        for i in range(12):
            map_block(f"blocks.{i}.", f"vit.layers.{i}.", 768)
        for i in range(8):
            map_block(f"decoder_blocks.{i}.", f"decoder.decoder_layers.{i}.", 512)
        for a, b in (("norm", "vit.layernorm"), ("decoder_embed", "decoder.decoder_embed"),
                     ("decoder_norm", "decoder.decoder_norm"), ("decoder_pred", "decoder.decoder_pred")):
            hsd[b + ".weight"] = sd[a + ".weight"]
            hsd[b + ".bias"] = sd[a + ".bias"]
        hsd["decoder.mask_token"] = sd["mask_token"]
        hsd["decoder.decoder_pos_embed"] = sd["decoder_pos_embed"]
        missing, unexpected = hm.load_state_dict(hsd, strict=False)
        with torch.no_grad():
            ho = hm(pixel_values=imgs, noise=noise)
        meta["transformers_crosscheck"] = {
            "missing": list(missing), "unexpected": list(unexpected),
            "loss_ref": float(loss.detach()), "loss_hf": float(ho.loss),
            "pred_max_abs_diff": float((ho.logits - pred).abs().max()),
            "mask_equal": bool(torch.equal(ho.mask, mask))}
    except Exception as exc:  # pragma: no cover
        meta["transformers_crosscheck"] = {"error": repr(exc)}
    print("transformers cross-check:", meta["transformers_crosscheck"])

    # classifier variants on ViT-B/16
    vitb_cls = dict(weight_seed=103, batch_seed=202, batch=B, pos_weight=np.float32(1.0))
    sdc = O.generated_state_dict(cfg, seed=103, decoder=False, n_class=2)
    vm = ref_models.ViT_from_MAE(None, True, 2, False, None, embed_dim=768, depth=12, num_heads=12, out_token="cls")
    own = vm.state_dict()
    sdc_mae = dict(sdc)
    sdc_mae["decoder_pos_embed"] = own["decoder_pos_embed"]  # left behind by `del` of the decoder (models.py:171-175)
    load_generated(vm, sdc_mae)
    logits = vm(imgs)
    z = logits[:, 1] - logits[:, 0]
    loss = nn.BCEWithLogitsLoss(pos_weight=torch.tensor(1.0))(z, labels.to(z.dtype))
    loss.backward()
    gn = {n: float(p.grad.norm()) for n, p in vm.named_parameters() if p.grad is not None}
    vitb_cls.update({"mae/logits": t2n(logits), "mae/loss": t2n(loss), "labels": t2n(labels),
                     "mae/grad_names": np.array(list(gn.keys())),
                     "mae/grad_norms": np.array(list(gn.values()), dtype=np.float64)})
    for n, gnp in named_grads(vm, ["lin_head.weight", "lin_head.bias", "norm.weight", "blocks.0.norm1.bias",
                                   "blocks.11.attn.proj.bias", "patch_embed.proj.bias", "cls_token"]).items():
        vitb_cls["mae/g/" + n] = gnp
    meta["vit_from_mae_state_keys_extra"] = sorted(set(own.keys()) - set(sdc.keys()))

    va = ref_models.VisionTransformer_from_Any(True, 2, False, None, 768, 12, 12, "cls", False)
    sda = dict(sdc)
    rng = np.random.Generator(np.random.PCG64(104))
    sda["pos_embed"] = torch.from_numpy(0.02 * rng.standard_normal((1, 197, 768))).float()  # learnable in timm
    own = va.state_dict()
    extra = sorted(set(own.keys()) - set(sda.keys()))
    meta["vit_from_any_state_keys_extra"] = extra
    va.load_state_dict(sda, strict=False)
    vitb_cls["any/logits"] = t2n(va(imgs))
    vitb_cls["any/pos_embed_seed"] = 104
    np.savez_compressed(os.path.join(HERE, "vitb_cls.npz"), **vitb_cls)

    # ---------------- tables ----------------
    tables = {}
    for D in (768, 512, 64, 32):
        for gs in (14, 4):
            tables[f"sincos/{D}/{gs}"] = ref_pos.get_2d_sincos_pos_embed(D, gs, cls_token=True).astype(np.float64)
    ramp = torch.arange(2 * 3 * 32 * 32, dtype=torch.float32).reshape(2, 3, 32, 32)
    mt = models_mae.MaskedAutoencoderViT(img_size=32, patch_size=8, embed_dim=64, depth=1, num_heads=4,
                                         decoder_embed_dim=32, decoder_depth=1, decoder_num_heads=4)
    tables["patchify_ramp"] = t2n(mt.patchify(ramp))
    tables["unpatchify_ramp"] = t2n(mt.unpatchify(mt.patchify(ramp)))

    class A:  # lr_sched.py:9-21 argument bundle (run_hyperkvasir defaults: warmup 40, epochs 400)
        lr = 1e-3 * 64 / 256
        min_lr = 0.0
        warmup_epochs = 40
        epochs = 400

    opt = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=0.1)
    opt.add_param_group({"params": [torch.zeros(1, requires_grad=True)], "lr_scale": 0.5})
    eps = np.concatenate([np.linspace(0, 400, 81), np.array([0.25, 39.99, 40.0, 40.01, 399.5])])
    lrs = []
    for e in eps:
        ref_lr.adjust_learning_rate(opt, float(e), A)
        lrs.append([g["lr"] for g in opt.param_groups])
    tables["mae_lr/epochs"] = eps
    tables["mae_lr/lrs"] = np.array(lrs, dtype=np.float64)
    tables["mae_lr/args"] = np.array([A.lr, A.min_lr, A.warmup_epochs, A.epochs], dtype=np.float64)

    # tc.py:3952-3957 cosine lambda (tc.py itself needs torchvision and cannot be imported: restated here)
    def lr_lambda(epoch, warmup_epochs=5, total_epochs=100):
        if warmup_epochs > 0 and epoch < warmup_epochs:
            return float(epoch + 1) / float(max(1, warmup_epochs))
        progress = (epoch - warmup_epochs) / float(max(1, total_epochs - warmup_epochs))
        progress = min(max(progress, 0.0), 1.0)
        return 0.5 * (1.0 + math.cos(math.pi * progress))

    tables["cls_lr/lambda_w5_e100"] = np.array([lr_lambda(e) for e in range(101)], dtype=np.float64)

    z = torch.tensor([[0.3, -1.2], [2.0, 2.5], [-4.0, 6.0], [0.0, 0.0]])
    y = torch.tensor([1, 0, 1, 0])
    for pw in (1.0, 0.37, 2.5):
        zz = z[:, 1] - z[:, 0]
        tables[f"bce/pw{pw}"] = t2n(nn.BCEWithLogitsLoss(pos_weight=torch.tensor(pw))(zz, y.float()))
    tables["bce/logits"] = t2n(z)
    tables["bce/targets"] = t2n(y)
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **tables)

    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith((".npz", ".json")):
            print(f"{fn:16s} {os.path.getsize(os.path.join(HERE, fn)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
