"""Precision mode "fp16" end to end (round 4): the reference's own AMP arithmetic -- fp16 matmul operands, f32 accumulation,
dynamic loss scaling (train_classification.py:4527-4546, engine_pretrain.py:52-72) -- on v_mfma_f32_32x32x16_f16, and the
device-resident loss scaler that replaces GradScaler's host read-back.  `pytest -m gpu` on the GPU box.

Tolerances: SURVEY 8-d as written for loss / pred / gradients (1e-3 / 1e-2 / 1e-2); tiny-model bounds are the bf16 bounds of
tests/test_gpu_models.py divided by 8 (three more significant bits) x 1.5.
"""
import functools

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
S = 4096.0  # static loss scale of the parity comparisons (a power of two: exact)


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _sd(fx, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(prefix)}


def _tiny_cls(prec, seed=5):
    import ssl4polyp_amd as A
    torch.manual_seed(seed)
    return A.ViT_from_MAE(None, True, 2, False, None, embed_dim=64, depth=2, num_heads=2, out_token="cls", precision=prec).to(DEV)


def test_tiny_mae_and_classifier_vs_reference_fixture_fp16(golden):
    """The reference-generated tiny fixtures (tests/golden/tiny_mae.npz, tiny_cls.npz) in fp16 mode, loss scaled by 2^12."""
    import ssl4polyp_amd as A
    from oracle import vit_mae_ref as O
    fx = golden("tiny_mae.npz")
    cfg = O.VIT_TINY
    m = A.MaskedAutoencoderViT(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth,
                               num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
                               decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=4,
                               norm_layer=functools.partial(torch.nn.LayerNorm, eps=1e-6), precision="fp16")
    m.load_state_dict(_sd(fx, "w/"))
    m.to(DEV)
    imgs, noise = torch.from_numpy(fx["imgs"]).to(DEV), torch.from_numpy(fx["noise"]).to(DEV)
    loss, pred, mask = m(imgs, mask_ratio=0.75, noise=noise)
    assert torch.equal(mask.cpu(), torch.from_numpy(fx["mask"]))
    e_loss, e_pred = rel(loss, fx["loss"]), rel(pred, fx["pred"])
    (loss * S).backward()
    worst = max(rel_l2(p.grad / S, fx["g/" + n]) for n, p in m.named_parameters() if "g/" + n in fx)
    print(f"[measured] tiny MAE fp16: loss {e_loss:.3e}, pred {e_pred:.3e}, worst gradient rel-L2 {worst:.3e}")
    assert e_loss < 1e-3 and e_pred < 1.5e-3 and worst < 3e-3
    fx = golden("tiny_cls.npz")
    imgs, labels = torch.from_numpy(fx["imgs"]).to(DEV), torch.from_numpy(fx["labels"]).to(DEV)
    vm = A.ViT_from_MAE(None, True, 2, False, None, embed_dim=64, depth=2, num_heads=2, out_token="cls", precision="fp16")
    vm.load_state_dict(_sd(fx, "mae/w/"))
    vm.to(DEV)
    logits = vm(imgs)
    loss = A.supervised_loss(logits, labels, pos_weight=float(fx["pos_weight"]))
    (loss * S).backward()
    e_log, e_loss = rel(logits, fx["mae/logits"]), rel(loss, fx["mae/loss"])
    worst = max(rel_l2(p.grad / S, fx["mae/g/" + n]) for n, p in vm.named_parameters() if "mae/g/" + n in fx)
    print(f"[measured] tiny cls fp16: logits {e_log:.3e}, loss {e_loss:.3e}, worst gradient rel-L2 {worst:.3e}")
    assert e_log < 1.5e-3 and e_loss < 1e-3 and worst < 3e-3


def test_backward_is_linear_in_the_loss_scale_and_passes_inf_through():
    """What makes ANY loss scaler (ours or torch's GradScaler) correct on this path: gradients of (S x loss) are exactly S x the
    gradients of the loss for a power-of-two S while nothing over- or underflows, and an overflowing scale surfaces as inf / nan
    in the f32 weight gradients (so that a found_inf scan sees it) instead of being clamped."""
    imgs = torch.randn(4, 3, 224, 224, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    labels = torch.tensor([1, 0, 1, 1], device=DEV)
    import ssl4polyp_amd as A
    grads = {}
    for s in (256.0, 4096.0, 2.0 ** 40):
        m = _tiny_cls("fp16")
        (A.supervised_loss(m(imgs), labels, pos_weight=1.0) * s).backward()
        grads[s] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    for n in grads[256.0]:
        a, b = grads[256.0][n] * 16.0, grads[4096.0][n]
        assert rel_l2(a, b) < 2e-3, n   # (not bit-equal: values near fp16's subnormal range at S = 256 round differently)
    flat = torch.cat([g.flatten() for g in grads[2.0 ** 40].values()])
    assert not torch.isfinite(flat).all(), "an overflowing loss scale must be visible in the weight gradients"


def test_device_loss_scaler_skips_backs_off_and_grows_without_host_sync():
    """optim.LossScaler + FusedAdamW == GradScaler.step / update semantics (tc.py:4545-4546), decided on the device:
    a step whose gradients hold inf leaves parameters, moments and the step counter untouched and halves the scale; clean
    steps apply AdamW to the UNSCALED gradients (equal to torch.optim.AdamW fed grad / scale) and grow the scale after
    `growth_interval` of them."""
    import ssl4polyp_amd as A
    from ssl4polyp_amd.optim import FusedAdamW, LossScaler
    imgs = torch.randn(4, 3, 224, 224, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    labels = torch.tensor([1, 0, 0, 1], device=DEV)
    m = _tiny_cls("fp16", seed=7)
    ref = _tiny_cls("fp16", seed=7)
    opt = FusedAdamW(m, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    topt = torch.optim.AdamW([p for p in ref.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    sc = LossScaler(init_scale=2.0 ** 40, growth_interval=3)   # 2^40: the first steps overflow fp16 -> skipped, scale halves
    assert sc.state_dict()["scale"] == 2.0 ** 40
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    n_skipped = 0
    for it in range(40):
        opt.zero_grad(set_to_none=True)
        sc.scale(A.supervised_loss(m(imgs), labels, pos_weight=1.0)).backward()
        sc.unscale_(opt)
        sc.step(opt)
        sc.update()
        c = sc.counters()
        if c["found_inf_last"]:
            n_skipped += 1
            for n, p in m.named_parameters():
                assert torch.equal(p.detach(), before[n]), f"skipped step touched {n}"
            assert sc.get_scale() == 2.0 ** (40 - n_skipped)
        else:
            break
    assert 1 <= n_skipped < 40 and c["skipped"] == n_skipped and c["steps"] == n_skipped + 1
    scale_ok = 2.0 ** (40 - n_skipped)
    assert sc.get_scale() == scale_ok  # one clean step: tracker 1 of 3
    # the clean step equals torch AdamW on the unscaled gradients of the same weights
    topt.zero_grad(set_to_none=True)
    (A.supervised_loss(ref(imgs), labels, pos_weight=1.0) * scale_ok).backward()
    for p in ref.parameters():
        if p.grad is not None:
            p.grad.div_(scale_ok)
    topt.step()
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if p.requires_grad and not n.endswith("attn.qkv.bias"):
            assert rel_l2(p, q) < 2e-5, n
    assert opt.param_groups[0]["step"] == n_skipped + 1  # (host mirror counts calls; the device record counts updates)
    assert int(opt._hyper[0, 6]) == 1
    # growth: far below the overflow edge, growth_interval = 3 clean steps in a row double the scale
    sc.update(new_scale=1024.0)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        sc.scale(A.supervised_loss(m(imgs), labels, pos_weight=1.0)).backward()
        sc.step(opt)
    assert sc.get_scale() == 2048.0 and sc.counters()["skipped"] == n_skipped
    # state dict round trip in GradScaler's layout
    sd = sc.state_dict()
    assert set(sd) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"} and sd["_growth_tracker"] == 0
    sc2 = LossScaler()
    sc2.load_state_dict(sd)
    assert sc2.get_scale() == sd["scale"]
    # the unscaled gradient statistics the reference logs after scaler.unscale_
    opt.zero_grad(set_to_none=True)
    sc.scale(A.supervised_loss(m(imgs), labels, pos_weight=1.0)).backward()
    gs = sc.unscaled_grad_stats(opt)
    want = sum(float((p.grad.double() / sc.get_scale()).square().sum()) for p in m.parameters() if p.grad is not None)
    assert abs(float(gs[0]) - want) < 1e-3 * want and float(gs[1]) == 0 and float(gs[2]) == 0
    # dropping the scaler: the plain step must not inherit a stale skip flag / unscale factor
    opt.zero_grad(set_to_none=True)
    A.supervised_loss(m(imgs), labels, pos_weight=1.0).backward()
    w0 = m.lin_head.weight.detach().clone()
    opt.step()
    assert not torch.equal(m.lin_head.weight.detach(), w0) and float(opt._hyper[0, 9]) == 0 and float(opt._hyper[0, 10]) == 0


def test_torch_gradscaler_drives_the_fp16_path_too():
    """The reference's loop verbatim (tc.py:4533-4546) with torch's own GradScaler on the fp16 HIP models + FusedAdamW."""
    import ssl4polyp_amd as A
    from ssl4polyp_amd.optim import FusedAdamW
    imgs = torch.randn(4, 3, 224, 224, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    labels = torch.tensor([0, 1, 1, 0], device=DEV)
    m = _tiny_cls("fp16", seed=11)
    opt = FusedAdamW(m, lr=1e-3)
    scaler = torch.amp.GradScaler("cuda", init_scale=4096.0)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = A.supervised_loss(m(imgs), labels, pos_weight=1.0)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        scaler.step(opt)
        scaler.update()
        losses.append(float(loss))
    assert scaler.get_scale() == 4096.0 and losses[-1] < losses[0] and all(l == l for l in losses)


@pytest.mark.parametrize("kind", ["cls", "mae"])
def test_fp16_training_trajectory_follows_fp32_mode(kind):
    """30 optimizer steps of ViT-B/16 from one seed in fp16 (dynamic loss scaling, overlapped AdamW) and in fp32 mode: the loss
    curves stay together, nothing is skipped once the scale has settled."""
    import ssl4polyp_amd as A
    from ssl4polyp_amd.optim import FusedAdamW, LossScaler, add_weight_decay
    B = 16 if kind == "cls" else 32
    g = torch.Generator(device=DEV).manual_seed(21)
    imgs = torch.randn(B, 3, 224, 224, device=DEV, generator=g)
    labels = (torch.rand(B, device=DEV, generator=g) < 0.5).long()
    noises = [torch.rand(B, 196, device=DEV, generator=g) for _ in range(30)]
    curves, scalers = {}, {}
    for prec in ("fp32", "fp16"):
        torch.manual_seed(0)
        if kind == "cls":
            m = A.get_MAE_backbone(None, True, 2, False, None, precision=prec).to(DEV)
            opt = FusedAdamW(m, lr=1e-4, weight_decay=0.05, overlap_forward=True)
        else:
            m = A.mae_vit_base_patch16(precision=prec).to(DEV)
            opt = FusedAdamW(m, add_weight_decay(m, 0.05), lr=1.5e-4, betas=(0.9, 0.95), overlap_forward=True)
        sc = LossScaler(init_scale=65536.0) if prec == "fp16" else None
        out = []
        for it in range(30):
            opt.zero_grad(set_to_none=True)
            loss = A.supervised_loss(m(imgs), labels, pos_weight=1.0) if kind == "cls" else m(imgs, mask_ratio=0.75, noise=noises[it])[0]
            if sc is not None:
                sc.scale(loss).backward()
                sc.step(opt)
                sc.update()
            else:
                loss.backward()
                opt.step()
            out.append(loss.detach())
        curves[prec] = torch.stack(out).cpu()
        scalers[prec] = sc
        del m, opt
        torch.cuda.empty_cache()
    c = scalers["fp16"].counters()
    d = (curves["fp16"] - curves["fp32"]).abs() / curves["fp32"].abs().clamp_min(1e-6)
    print(f"[measured] {kind} fp16 vs fp32-mode trajectory: max rel loss distance {d.max():.3e} (step {int(d.argmax())}), last {d[-1]:.3e}; "
          f"loss scale {scalers['fp16'].get_scale():.0f}, skipped {c['skipped']} of {c['steps']}")
    # skipped steps shift the fp16 curve by that many steps against the fp32 one: compare only when none was skipped
    assert torch.isfinite(curves["fp16"]).all() and curves["fp16"][-1] < curves["fp16"][0]
    if c["skipped"] == 0:
        assert d.max() < (2e-2 if kind == "cls" else 2e-3)
    else:
        assert c["skipped"] <= 8


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_registered_linear_bias_gelu_backward_runs_pm_dgelu(dt):
    """torch.ops.polypmae.linear_bias_gelu with 16-bit activations: its backward is dgrad / wgrad GEMMs around ONE HIP
    elementwise pass (pm_dgelu on the saved pre-activation) -- no ATen GELU kernel on a public op of the library."""
    from ssl4polyp_amd import _lib, ops  # noqa: F401
    from ssl4polyp_amd.engine import _ptr, _stream
    g = torch.Generator().manual_seed(5)
    M, K, N = 200, 128, 256
    x = (torch.randn(M, K, generator=g) * 0.5).to(DEV).to(dt).requires_grad_(True)
    W = (torch.randn(N, K, generator=g) * 0.08).to(DEV).to(dt).requires_grad_(True)
    b = (torch.randn(N, generator=g) * 0.1).to(DEV).requires_grad_(True)
    y = torch.ops.polypmae.linear_bias_gelu(x, W, b)
    dy = (torch.randn(M, N, generator=g)).to(DEV).to(dt)
    gx, gW, gb = torch.autograd.grad(y, (x, W, b), dy)
    xr, Wr, br = x.detach().float().requires_grad_(True), W.detach().float().requires_grad_(True), b.detach().clone().requires_grad_(True)
    pre = F.linear(xr, Wr, br)
    yr = F.gelu(pre.detach().to(dt).float() + (pre - pre.detach()))  # gelu of the STORED pre-activation, gradient through pre
    rx, rW, rb = torch.autograd.grad(yr, (xr, Wr, br), dy.float())
    tol = 2e-2 if dt == torch.bfloat16 else 3e-3
    assert rel(y.float(), yr) < tol and rel(gx.float(), rx) < tol and rel(gW, rW) < tol and rel(gb, rb) < tol
    # and the entry point alone, elementwise exactness class
    lib = _lib.load()
    a, p = dy.contiguous(), pre.detach().to(dt).contiguous()
    out = torch.empty_like(a)
    _lib.check(lib.pm_dgelu(_ptr(a), _ptr(p), _ptr(out), _lib.dtype_code(dt), a.numel(), _stream()), "pm_dgelu")
    p32 = p.float().requires_grad_(True)
    F.gelu(p32).backward(a.float())
    assert rel(out.float(), p32.grad) < (1e-2 if dt == torch.bfloat16 else 1.5e-3)


def test_fp16_training_loops_checkpoint_the_loss_scaler(tmp_path):
    """train_one_epoch_mae / train_epoch_cls with `loss_scaler=` (engine_pretrain.py:65-72, tc.py:4533-4546): losses go down, the
    logged gradient statistics are those of the UNSCALED gradients, and both checkpoint formats carry the scaler's state
    (misc.py:311-318 `scaler`, tc.py:7049 `scaler_state_dict`) and restore it."""
    from types import SimpleNamespace
    import numpy as np
    import ssl4polyp_amd as A
    from ssl4polyp_amd import train as T
    from ssl4polyp_amd.optim import FusedAdamW, LossScaler, add_weight_decay
    dev = torch.device("cuda", 0)
    torch.manual_seed(21)
    m = A.MaskedAutoencoderViT(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2, decoder_embed_dim=32, decoder_depth=1,
                               decoder_num_heads=1, precision="fp16").to(dev)
    opt = FusedAdamW(m, add_weight_decay(m, 0.05), lr=2e-3, betas=(0.9, 0.95))
    sc = LossScaler(init_scale=1024.0, growth_interval=5)
    args = SimpleNamespace(lr=2e-3, min_lr=0.0, warmup_epochs=1, epochs=4, accum_iter=2, mask_ratio=0.75)
    loader = T.SyntheticLoader(8, 12, dev, img_size=32, seed=3)
    s0 = T.train_one_epoch_mae(m, loader, opt, dev, 0, args, log_every=3, printer=None, loss_scaler=sc)
    s1 = T.train_one_epoch_mae(m, loader, opt, dev, 1, args, log_every=3, printer=None, loss_scaler=sc)
    assert np.isfinite(s0.loss) and s1.loss < s0.loss and s1.grad_nan == 0 and s1.grad_inf == 0
    # 12 optimizer steps (24 micro-batches, accum 2), growth every 5 clean ones: 1024 -> 4096; the logged norm is unscaled
    assert sc.counters() == {"steps": 12, "skipped": 0, "found_inf_last": False} and sc.get_scale() == 4096.0
    assert 0 < s1.history[-1]["grad_norm"] < 100
    ck = T.save_mae_checkpoint(tmp_path, 1, m, opt, args, scaler_state=sc.state_dict())
    sc2 = LossScaler()
    m2 = A.MaskedAutoencoderViT(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2, decoder_embed_dim=32, decoder_depth=1,
                                decoder_num_heads=1, precision="fp16").to(dev)
    m2._rt.ensure(dev)
    opt2 = FusedAdamW(m2, add_weight_decay(m2, 0.05), lr=2e-3, betas=(0.9, 0.95))
    assert T.load_mae_checkpoint(ck, m2, opt2, args, loss_scaler=sc2) == 2
    assert sc2.get_scale() == 4096.0 and sc2.state_dict()["_growth_tracker"] == sc.state_dict()["_growth_tracker"]
    # classification loop + its checkpoint format
    vm = _tiny_cls("fp16")
    copt = FusedAdamW(vm, lr=1e-3, weight_decay=0.05)
    csc = LossScaler(init_scale=4096.0)
    cl = T.SyntheticLoader(4, 6, dev, seed=4)
    c0 = T.train_epoch_cls(vm, cl, copt, dev, pos_weight=torch.tensor(1.0, device=dev), log_every=2, printer=None, loss_scaler=csc)
    c1 = T.train_epoch_cls(vm, cl, copt, dev, pos_weight=torch.tensor(1.0, device=dev), log_every=2, printer=None, loss_scaler=csc)
    assert c0.steps == 6 and c1.loss < c0.loss and csc.counters()["skipped"] == 0
    path = T.save_cls_checkpoint(tmp_path / "run_e01_best.pth", 1, vm, copt, loss=c1.loss, extra={"val_loss": 0.5}, loss_scaler=csc)
    assert torch.load(str(path), weights_only=False)["scaler_state_dict"]["scale"] == 4096.0
    csc2 = LossScaler()
    vm2 = _tiny_cls("fp16", seed=9)
    vm2._rt.ensure(dev)
    info = T.load_cls_checkpoint(tmp_path / "run", vm2, FusedAdamW(vm2, lr=1e-3), loss_scaler=csc2)
    assert info.start_epoch == 2 and csc2.get_scale() == 4096.0
