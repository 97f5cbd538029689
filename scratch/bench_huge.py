"""MAE pre-train step of the larger factories (models_mae.py:231-244) on one GPU: ms / step, img / s, algorithmic TFLOP / s.
usage: python scratch/bench_huge.py [--model mae_vit_huge_patch14] [--batch 64] [--precision bf16] [--steps 20]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssl4polyp_amd as A
from ssl4polyp_amd.optim import FusedAdamW, LossScaler, add_weight_decay


def gflop_per_img(m, mask_ratio=0.75):
    """contractions only, step = 3 x forward (BASELINE.md section 3's convention)"""
    pe = m.patch_embed
    L, p = pe.num_patches, pe.patch_size[0]
    keep = int(L * (1 - mask_ratio))
    De, Dd = m.cls_token.shape[-1], m.mask_token.shape[-1]
    def stack(D, N, depth):  # MACs: qkv + proj + mlp = 12 D^2 per token, attention 2 N D per token
        return depth * N * (12 * D * D + 2 * N * D)
    macs = keep * 3 * p * p * De + stack(De, keep + 1, len(m.blocks)) + (keep + 1) * De * Dd + \
        stack(Dd, L + 1, len(m.decoder_blocks)) + (L + 1) * Dd * 3 * p * p
    return 3 * 2 * macs / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mae_vit_huge_patch14")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    A.reserve_streams(dev)
    torch.manual_seed(0)
    m = getattr(A, a.model)(norm_pix_loss=False, precision=a.precision).to(dev)
    opt = FusedAdamW(m, add_weight_decay(m, 0.05), lr=1e-5 * a.batch / 256, betas=(0.9, 0.95), overlap_forward=True)  # (warm-up-sized lr)
    scaler = LossScaler() if a.precision == "fp16" else None
    g = torch.Generator(device=dev).manual_seed(1234)
    imgs = [torch.randn(a.batch, 3, 224, 224, generator=g, device=dev) for _ in range(4)]

    def step(i):
        opt.zero_grad(set_to_none=True)
        loss, _, _ = m(imgs[i % 4], mask_ratio=0.75)
        if scaler is not None:
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
        else:
            loss.backward()
            opt.step()
        return loss

    for i in range(a.warmup):
        l = step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        l = step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    gf = gflop_per_img(m)
    print(f"{a.model} bs={a.batch} {a.precision}: {dt * 1e3:.2f} ms/step, {a.batch / dt:.1f} img/s, {gf:.1f} GFLOP/img/step -> "
          f"{a.batch / dt * gf / 1e3:.1f} TFLOP/s ({a.batch / dt * gf / 1e3 / 2500:.3f} of 2.5 PFLOP/s), loss {float(l):.4f}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
