#!/usr/bin/env python3
"""Golden loss / prediction / GRADIENTS of `mae_vit_huge_patch14` (models_mae.py:239-244) by RUNNING THE REFERENCE.

Run in the build container only (the reference checkout does not travel):

    python tests/golden/make_mae_huge_grads.py [--reference /root/reference]

The reference's own `models_mae.mae_vit_huge_patch14` (imported from its checkout, with the timm 0.4.12 stand-in of
make_fixtures.py) is loaded with the PCG64 weights of oracle.generated_state_dict(VIT_HUGE, seed 61) and fed the PCG64 batch of
oracle.generated_batch(VIT_HUGE, 4, seed 62) -- the tensors tests/test_gpu_parity_large.py rebuilds on the GPU box.  ViT-H/14 is
the one factory whose shapes differ in kind from the benchmarked ViT-B/16: 80-wide heads, 257 tokens, a 588-element patch.

Output: tests/golden/vith_mae_grads.npz -- loss, mask, a slice and per-sample sums of pred, the L2 norm of EVERY parameter
gradient, whole small gradients and 32 x 32 corners of a few weight-matrix gradients (the patch-embedding and decoder_pred
corners sit on the padded reduction dimension of the HIP path).
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

WEIGHT_SEED, BATCH_SEED, B = 61, 62, 4
FULL = ["cls_token", "mask_token", "patch_embed.proj.bias", "blocks.0.norm1.weight", "blocks.0.attn.qkv.bias",
        "blocks.15.mlp.fc1.bias", "blocks.31.mlp.fc2.bias", "blocks.31.attn.proj.bias", "norm.weight", "norm.bias",
        "decoder_embed.bias", "decoder_blocks.0.attn.qkv.bias", "decoder_blocks.7.mlp.fc2.bias", "decoder_norm.weight",
        "decoder_pred.bias"]
CORNER = ["patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.0.attn.proj.weight", "blocks.15.mlp.fc1.weight",
          "blocks.31.mlp.fc2.weight", "decoder_embed.weight", "decoder_blocks.0.attn.qkv.weight",
          "decoder_blocks.7.mlp.fc2.weight", "decoder_pred.weight"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(args.reference, "src"))
    import make_fixtures as MF
    MF.install_timm_standin()
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    from ssl4polyp.models.mae import models_mae
    from oracle import vit_mae_ref as O

    cfg = O.VIT_HUGE
    t0 = time.perf_counter()
    sd = O.generated_state_dict(cfg, WEIGHT_SEED, decoder=True, n_class=None)
    m = models_mae.mae_vit_huge_patch14(norm_pix_loss=False)
    MF.load_generated(m, sd)
    imgs, _, noise = O.generated_batch(cfg, B, BATCH_SEED)
    loss, pred, mask = MF.run_ref_mae(m, imgs, noise)
    loss.backward()
    print(f"reference forward + backward: loss {float(loss):.6f} ({time.perf_counter() - t0:.0f} s)", flush=True)
    grads = {n: p.grad.detach() for n, p in m.named_parameters() if p.grad is not None}
    names = list(grads)
    pred = pred.detach()
    out = dict(weight_seed=WEIGHT_SEED, batch_seed=BATCH_SEED, batch=B, loss=np.float64(float(loss)),
               mask=mask.numpy().astype(np.uint8), pred_slice=pred[:, :8, :40].numpy(),
               pred_tail_slice=pred[:, -4:, -24:].numpy(),   # the last of the 588 outputs
               pred_abs_sum_per_sample=pred.abs().double().sum(dim=(1, 2)).numpy(),
               pred_norm=np.float64(float(pred.double().norm())),
               grad_names=np.array(names), grad_norms=np.array([float(grads[n].double().norm()) for n in names], dtype=np.float64))
    for n in FULL:
        out["g/" + n] = grads[n].numpy()
    for n in CORNER:
        g = grads[n].reshape(grads[n].shape[0], -1)
        out["g_corner/" + n] = g[:32, :32].numpy()
        out["g_corner_last/" + n] = g[-32:, -32:].numpy()   # rows / columns 556..587 of the 588-wide matrices
    path = os.path.join(HERE, "vith_mae_grads.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(names)} gradient norms, {os.path.getsize(path) / 1024:.0f} KiB, {time.perf_counter() - t0:.0f} s")


if __name__ == "__main__":
    main()
