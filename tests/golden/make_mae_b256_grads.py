#!/usr/bin/env python3
"""Golden GRADIENTS of the benchmarked MAE configuration (C3: ViT-B/16, mask 0.75, B = 256) by RUNNING THE REFERENCE.

Run in the build container only (the reference checkout does not travel):

    python tests/golden/make_mae_b256_grads.py [--reference /root/reference]

The reference's own `models_mae.mae_vit_base_patch16` (imported from its checkout, with the timm 0.4.12 stand-in of
make_fixtures.py) is loaded with the PCG64 weights of oracle.generated_state_dict(seed 41) and fed the PCG64 batch of
oracle.generated_batch(256, seed 42) -- the tensors tests/test_gpu_parity_large.py rebuilds on the GPU box -- in 8 chunks
of 32 samples: the loss is sum(l * mask) / sum(mask) with exactly 147 masked patches per sample, so the full-batch loss is
the mean of the chunk losses and the full-batch gradient the mean of the chunk gradients (fp32 summation order aside).
A backward at B = 256 takes minutes of CPU: too long for the GPU box's test run, which is why it is a fixture.

Output: tests/golden/vitb_mae_b256_grads.npz -- loss, mask row sums, the L2 norm of EVERY parameter gradient, whole small
gradients (vectors / tokens of a few layers) and 32x32 corners of a few weight-matrix gradients.
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

WEIGHT_SEED, BATCH_SEED, B, CHUNK = 41, 42, 256, 32
FULL = ["cls_token", "mask_token", "patch_embed.proj.bias", "blocks.0.norm1.weight", "blocks.0.attn.qkv.bias",
        "blocks.5.mlp.fc1.bias", "blocks.11.mlp.fc2.bias", "norm.weight", "norm.bias", "decoder_embed.bias",
        "decoder_blocks.0.attn.qkv.bias", "decoder_blocks.3.norm2.bias", "decoder_blocks.7.mlp.fc1.bias",
        "decoder_blocks.7.mlp.fc2.bias", "decoder_norm.weight", "decoder_pred.bias"]
CORNER = ["patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.5.mlp.fc1.weight", "blocks.11.attn.proj.weight",
          "blocks.11.mlp.fc2.weight", "decoder_embed.weight", "decoder_blocks.0.mlp.fc1.weight",
          "decoder_blocks.4.attn.qkv.weight", "decoder_blocks.7.attn.proj.weight", "decoder_blocks.7.mlp.fc2.weight",
          "decoder_pred.weight"]


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

    cfg = O.VIT_BASE
    sd = O.generated_state_dict(cfg, WEIGHT_SEED, decoder=True, n_class=None)
    m = models_mae.mae_vit_base_patch16(norm_pix_loss=False)
    MF.load_generated(m, sd)
    imgs, _, noise = O.generated_batch(cfg, B, BATCH_SEED)
    t0 = time.perf_counter()
    total = 0.0
    masks = []
    for c0 in range(0, B, CHUNK):
        loss, _, mask = MF.run_ref_mae(m, imgs[c0:c0 + CHUNK], noise[c0:c0 + CHUNK])
        assert float(mask.sum()) == CHUNK * 147
        (loss * (CHUNK / B)).backward()   # gradients accumulate in .grad: mean over the chunks
        total += float(loss) * (CHUNK / B)
        masks.append(mask)
        print(f"chunk {c0 // CHUNK}: loss {float(loss):.6f}  ({time.perf_counter() - t0:.0f} s)", flush=True)
    mask = torch.cat(masks)
    grads = {n: p.grad.detach() for n, p in m.named_parameters() if p.grad is not None}
    names = list(grads)
    out = dict(weight_seed=WEIGHT_SEED, batch_seed=BATCH_SEED, batch=B, chunk=CHUNK, loss=np.float64(total),
               mask_rowsum=mask.sum(1).numpy().astype(np.int32), mask_first_rows=mask[:4].numpy().astype(np.uint8),
               grad_names=np.array(names), grad_norms=np.array([float(grads[n].double().norm()) for n in names], dtype=np.float64))
    for n in FULL:
        out["g/" + n] = grads[n].numpy()
    for n in CORNER:
        g = grads[n]
        out["g_corner/" + n] = g.reshape(g.shape[0], -1)[:32, :32].numpy()
    path = os.path.join(HERE, "vitb_mae_b256_grads.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: loss {total:.6f}, {len(names)} gradient norms, {os.path.getsize(path) / 1024:.0f} KiB, "
          f"{time.perf_counter() - t0:.0f} s")


if __name__ == "__main__":
    main()
