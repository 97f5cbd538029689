#!/usr/bin/env python3
"""What each bf16 rounding point of the HIP path costs against the fp32 oracle (CPU only, no kernel involved).

    python scratch/rounding_cost.py [--batch 16] [--init generated|fresh] [--workload cls|mae] [--out profiles/...json]

Runs oracle/vit_bf16_grad_sim.py once per variant of `VARIANTS` (one rounding point switched off at a time) plus PyTorch's own
torch.autocast(cpu, bf16) of the oracle, and prints logits / loss / worst-gradient errors -- the table of DESIGN.md section 2.
--init fresh = the product's own initialisation (torch.manual_seed(0), what bench.py trains from); generated = the PCG64
weights of the parity tests.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vit_bf16_grad_sim as S  # noqa: E402
from oracle import vit_mae_ref as O  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--init", choices=["generated", "fresh"], default="generated")
    ap.add_argument("--workload", choices=["cls", "mae"], default="cls")
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--variants", default="all")
    ap.add_argument("--family", choices=["bf16", "fp16"], default="bf16",
                    help="bf16: one bf16 rounding point off at a time (round 3); fp16: the 16-bit type of the forward / backward "
                         "operands and the loss scale (round 4: which 16-bit mode meets SURVEY 8-d as written)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    cfg = O.VIT_BASE
    cls = args.workload == "cls"
    if args.init == "generated":
        sd = O.generated_state_dict(cfg, 31 if cls else 41, decoder=not cls, n_class=2 if cls else None)
    else:
        import ssl4polyp_amd as A
        torch.manual_seed(0)
        m = A.get_MAE_backbone(None, True, 2, False, None) if cls else A.mae_vit_base_patch16()
        sd = {k: v.detach().clone() for k, v in m.state_dict().items() if cls is False or k != "decoder_pos_embed"}
    imgs, labels, noise = O.generated_batch(cfg, args.batch, 32 if cls else 42)
    pw = 1.7 if args.init == "generated" else 1.0

    def run(fn):
        leaves = {n: v.clone().requires_grad_("pos_embed" not in n) for n, v in sd.items()}
        t0 = time.perf_counter()
        if cls:
            out = fn(leaves)
            loss = O.supervised_loss(out.float(), labels, pw)
        else:
            loss, out, _ = fn(leaves)
        loss.backward()
        return out.detach().float(), loss.detach(), {n: v.grad for n, v in leaves.items() if v.grad is not None}, time.perf_counter() - t0

    if cls:
        ref = run(lambda p: O.vit_classify(p, imgs, cfg))
    else:
        ref = run(lambda p: O.mae_forward(p, imgs, noise, cfg))
    print(f"fp32 oracle: {ref[3]:.1f} s", flush=True)

    def report(name, got):
        out, loss, grads, dt = got
        errs = {n: rel_l2(g, ref[2][n]) for n, g in grads.items() if not n.endswith("attn.qkv.bias")}
        mats = {n: e for n, e in errs.items() if sd[n].ndim >= 2 and sd[n].shape[0] > 1}
        vecs = {n: e for n, e in errs.items() if n not in mats}
        wm, wv = max(mats, key=mats.get), max(vecs, key=vecs.get)
        # common scalar factor: g ~ alpha * g_ref for every parameter?
        alphas = sorted(float((grads[n].double() * ref[2][n].double()).sum() / (ref[2][n].double() ** 2).sum()) for n in mats)
        alpha = alphas[len(alphas) // 2]
        resid = max(rel_l2(grads[n] / alpha, ref[2][n]) for n in mats)
        rec = {"variant": name, ("logits_max_rel" if cls else "pred_rel_l2"): (rel(out, ref[0]) if cls else rel_l2(out, ref[0])),
               "loss_rel": rel(loss, ref[1]), "matrix_grad_worst": mats[wm], "matrix_grad_worst_name": wm,
               "matrix_grad_median": sorted(mats.values())[len(mats) // 2], "vector_grad_worst": vecs[wv],
               "vector_grad_worst_name": wv, "common_factor_median": alpha, "matrix_grad_worst_after_common_factor": resid,
               "seconds": round(dt, 1)}
        print(json.dumps(rec), flush=True)
        return rec

    recs = []
    table = S.VARIANTS if args.family == "bf16" else S.FP16_VARIANTS
    names = list(table) if args.variants == "all" else [n for n in table if any(t in n for t in args.variants.split(","))]
    for name in names:
        rnd = table[name]
        if cls:
            recs.append(report(name, run(lambda p: S.vit_classify(p, imgs, cfg, rnd))))
        else:
            recs.append(report(name, run(lambda p: S.mae_forward(p, imgs, noise, cfg, rnd=rnd))))

    def autocast(p):
        with torch.autocast("cpu", dtype=torch.bfloat16):
            return O.vit_classify(p, imgs, cfg) if cls else O.mae_forward(p, imgs, noise, cfg)
    if args.family == "bf16":
        recs.append(report("torch.autocast(cpu, bf16) of the oracle (yardstick)", run(autocast)))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump({"workload": args.workload, "batch": args.batch, "init": args.init, "pos_weight": pw, "records": recs}, fh, indent=1)


if __name__ == "__main__":
    main()
