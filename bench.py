#!/usr/bin/env python3
"""bench.py -- training-step throughput of the MI355X hot path on synthetic Hyperkvasir-shaped batches.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cls|mae] [--precision bf16|fp32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = zero_grad -> forward -> loss -> backward (-> RCCL gradient all-reduce) -> fused AdamW, inputs
resident in HBM.  Default workload = BASELINE.json configs[1]: ViT-B/16 classification
fine-tune, bf16, bs=64/GPU, 224^2.  Rank 0 prints ONE JSON line (contract in the task statement) carrying
`roofline` (achieved algorithmic TFLOP/s vs the 2.5 PFLOP/s dense bf16 MFMA peak, plus the dominant
kernel's own live-measured average duration) and `cpu_baseline` (the CPU oracle timed on this host).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# algorithmic GFLOP per image per training step (contractions only, step = 3 x forward)
GFLOP_PER_IMG = {"cls": 105.38, "mae": 58.16}
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # MI355X_MICROARCH.md: dense bf16 MFMA / f32 MFMA
PEAK_HBM_GBPS = 8000.0                         # MI355X_MICROARCH.md: HBM3E
HBM_KERNELS = {}
OVERLAP_ADAMW = os.environ.get("PM_OVERLAP_ADAMW", "1") != "0"  # A/B switch: AdamW beside the next forward


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["cls", "mae"], default="cls")
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 64 cls / 256 mae)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-stats", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--input", choices=["resident", "host"], default="resident",
                    help="resident (the contract: batch already in HBM) or host: uint8 HWC frames in pinned host memory, "
                         "copied and normalised on a side stream each step (PCIe-inclusive rate, DESIGN.md)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the step as one hipGraph (auto = off: eager launch keeps up and overlaps the two streams better)")
    return ap.parse_args()


def build(workload, precision, device, world, batch):
    import ssl4polyp_amd as A
    from ssl4polyp_amd.optim import FusedAdamW, add_weight_decay
    from ssl4polyp_amd.parallel import DataParallel
    torch.manual_seed(0)  # identical init on every rank (then broadcast from rank 0 anyway)
    if workload == "cls":
        model = A.get_MAE_backbone(None, True, 2, False, None, precision=precision)
    else:
        model = A.mae_vit_base_patch16(norm_pix_loss=False, precision=precision)
    ddp = DataParallel(model, device)
    if workload == "cls":
        # tc.py:5751-5768: AdamW(lr 1e-3, wd 0.05) over two groups head / backbone (config/base.yaml:1-4)
        head = list(model.lin_head.parameters())
        hid = {id(p) for p in head}
        groups = [{"params": head, "name": "head"},
                  {"params": [p for p in model.parameters() if id(p) not in hid and p.requires_grad], "name": "backbone"}]
        opt = FusedAdamW(model, groups, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.05, overlap_forward=OVERLAP_ADAMW)
    else:
        # main_pretrain.py:201-218: lr = blr * eff_batch / 256, betas (0.9, 0.95), no decay on 1-D params
        lr = 1e-3 * batch * world / 256
        opt = FusedAdamW(model, add_weight_decay(model, 0.05), lr=lr, betas=(0.9, 0.95), overlap_forward=OVERLAP_ADAMW)
    opt.grad_sync = ddp.sync
    opt.grad_scale = 1.0 / world
    return model, ddp, opt


def make_batch(workload, batch, device, rank):
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    imgs = torch.randn(batch, 3, 224, 224, generator=g, device=device)
    labels = (torch.rand(batch, generator=g, device=device) < 0.5).long()
    return imgs, labels


def make_step(workload, ddp, opt, imgs, labels):
    pos_weight = torch.tensor(1.0, device=imgs.device)
    if workload == "cls":
        def step(imgs=imgs, labels=labels):
            opt.zero_grad(set_to_none=True)
            logits = ddp(imgs)
            z = logits[:, 1] - logits[:, 0]                       # tc.py:3347-3359
            loss = F.binary_cross_entropy_with_logits(z, labels.float(), pos_weight=pos_weight)  # tc.py:6090-6102
            loss.backward()
            opt.step()
            return loss
    else:
        def step(imgs=imgs, labels=labels):
            opt.zero_grad(set_to_none=True)
            loss, _, _ = ddp(imgs, mask_ratio=0.75)
            loss.backward()
            opt.step()
            return loss
    return step


def kernel_stats(model, step):
    """Per-launch HIP-event timing of every GEMM of one step (events on the stream the kernels are launched on)."""
    k = model._rt.k
    orig = k.gemm
    rec = []

    def timed(A, lda, akm, B, ldb, bkm, bias, C, ldc, epi, M, N, K, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(A, lda, akm, B, ldb, bkm, bias, C, ldc, epi, M, N, K, **kw)
        e1.record()
        rec.append((("tn" if akm else "n") + ("n" if bkm else "t"), M, N, K, e0, e1))

    # the HBM-bound kernels of the path, timed the same way: algorithmic bytes / launch time against the HBM peak
    hbm = []
    orig_lnf, orig_lnb = k.layernorm_fwd, k.layernorm_bwd
    act_b = 2 if model._rt.k.precision == "bf16" else 4

    def ln_fwd(x, gamma, beta, y, mean, rstd, M, D):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig_lnf(x, gamma, beta, y, mean, rstd, M, D); e1.record()
        hbm.append(("layernorm_fwd", M * D * (4 + act_b), e0, e1))

    def ln_bwd(dy, x, gamma, mean, rstd, dres, dx, dx_act, dgamma, dbeta, dcolsum, M, D):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig_lnb(dy, x, gamma, mean, rstd, dres, dx, dx_act, dgamma, dbeta, dcolsum, M, D); e1.record()
        hbm.append(("layernorm_bwd", M * D * (act_b + 4 + 4 + 4 + act_b), e0, e1))

    k.gemm, k.layernorm_fwd, k.layernorm_bwd = timed, ln_fwd, ln_bwd
    try:
        for _ in range(3):
            rec.clear()
            hbm.clear()
            step()
        torch.cuda.synchronize()
    finally:
        k.gemm, k.layernorm_fwd, k.layernorm_bwd = orig, orig_lnf, orig_lnb
    hb = {}
    for name, nbytes, e0, e1 in hbm:
        d = hb.setdefault(name, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1) * 1e-3
        d[2] += nbytes
    global HBM_KERNELS
    HBM_KERNELS = {n: {"launches": c, "avg_us": round(t / c * 1e6, 2), "GBps": round(b / t / 1e9, 1),
                       "frac_of_hbm_peak": round(b / t / 1e9 / PEAK_HBM_GBPS, 3)} for n, (c, t, b) in hb.items()}
    by = {}
    for lay, M, N, K, e0, e1 in rec:
        d = by.setdefault(lay, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1) * 1e-3
        d[2] += 2.0 * M * N * K
    out = {}
    for lay, (n, t, fl) in by.items():
        out[lay] = {"launches": n, "avg_us": round(t / n * 1e6, 2), "tflops": round(fl / t / 1e12, 1)}
    tot_t = sum(v[1] for v in by.values())
    tot_f = sum(v[2] for v in by.values())
    return out, tot_t, tot_f


def cpu_baseline(workload, steps):
    """The CPU oracle (oracle/vit_mae_ref.py, fp32, torch CPU threads) on a bounded sample of the same workload:
    bs=8 synthetic batch, reference step order (zero_grad -> fwd -> loss -> bwd -> AdamW)."""
    from oracle import vit_mae_ref as O
    cfg = O.VIT_BASE
    B = 8
    sd = O.generated_state_dict(cfg, seed=1, decoder=(workload == "mae"), n_class=2 if workload == "cls" else None)
    params = {n: v.clone().requires_grad_("pos_embed" not in n) for n, v in sd.items()}
    leaves = [p for p in params.values() if p.requires_grad]
    opt = torch.optim.AdamW(leaves, lr=1e-3, weight_decay=0.05)
    imgs, labels, noise = O.generated_batch(cfg, B, seed=2)

    def one():
        opt.zero_grad()
        if workload == "cls":
            loss = O.supervised_loss(O.vit_classify(params, imgs, cfg), labels, 1.0)
        else:
            loss = O.mae_forward(params, imgs, noise, cfg)[0]
        loss.backward()
        opt.step()

    one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = time.perf_counter() - t0
    return {"value": round(B * steps / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed steps (+1 warm-up) of the fp32 CPU oracle, {workload} ViT-B/16 224^2, bs={B}"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # rehearsal hook (one-GPU box): BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo, so the N>1 code path
    # (rendezvous, broadcast, bucketed overlap, max-over-ranks timing) can be exercised without N GPUs
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    batch = args.batch or (64 if args.workload == "cls" else 256)
    model, ddp, opt = build(args.workload, args.precision, device, world, batch)
    imgs, labels = make_batch(args.workload, batch, device, rank)
    eager_step = make_step(args.workload, ddp, opt, imgs, labels)
    use_graph = args.graph == "on"  # auto: eager (measured faster: graph replay serialises the wgrad side stream)
    if use_graph:
        from ssl4polyp_amd.graph import GraphedStep
        graphed = GraphedStep(eager_step, opt, warmup=3)  # capture failures are fatal: no silent eager fallback
        step = graphed.replay
    else:
        step = eager_step

    host_pool, host_gen = [], torch.Generator().manual_seed(1234 + rank)
    if args.input == "host":  # four decoded batches in pinned host memory, built outside the timed region
        host_pool = [(torch.randint(0, 256, (batch, 224, 224, 3), dtype=torch.uint8, generator=host_gen).pin_memory(),
                      (torch.rand(batch, generator=host_gen) < 0.5).long().pin_memory()) for _ in range(4)]

    def host_feed(n):
        # uint8 HWC frames + labels in pinned host memory -> DevicePrefetcher (H2D + flips + ToTensor + Normalize)
        from ssl4polyp_amd.data import DevicePrefetcher
        return DevicePrefetcher([host_pool[i % 4] for i in range(n)], device, flip_p=0.5, generator=host_gen)

    if args.input == "host":
        if use_graph:
            sys.exit("--input host feeds a new batch every step: use eager launch")
        for im, lb in host_feed(args.warmup):
            step(im, lb)
    else:
        for _ in range(args.warmup):
            step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.input == "host":
        for im, lb in host_feed(args.steps):
            loss = step(im, lb)
    else:
        for _ in range(args.steps):
            loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss_val = float(loss.detach())
    if not (loss_val == loss_val):
        sys.exit("non-finite loss in the timed region")

    if rank == 0:
        ips = batch * world * args.steps / dt
        per_gpu_tflops = ips / world * GFLOP_PER_IMG[args.workload] / 1e3
        peak = PEAK_TFLOPS[args.precision]
        roof = {"bound": "mfma", "achieved": round(per_gpu_tflops, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(per_gpu_tflops / peak, 4), "traffic": None,
                "basis": f"{GFLOP_PER_IMG[args.workload]} algorithmic GFLOP/img/step x img/s/GPU (BASELINE.md §3)"}
        if not args.no_kernel_stats:
            ks, gt, gf = kernel_stats(model, eager_step)
            dom = max(ks.items(), key=lambda kv: kv[1]["launches"] * kv[1]["avg_us"])
            roof["kernel"] = {"name": f"gemm_kernel<{args.precision},{dom[0]}>", **dom[1],
                              "frac": round(dom[1]["tflops"] / peak, 4),
                              "note": "in-step launch times: launches of the two forward chains / of the dgrad and "
                                      "weight-gradient streams overlap, so each shares the CUs (stand-alone rates: DESIGN.md)"}
            try:  # PMC traffic of the dominant kernel, from the committed rocprofv3 --pmc passes (profiles/)
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as fh:
                    t = json.load(fh).get(args.workload, {}).get(dom[0])
                if t and args.precision == "bf16" and batch == 64:
                    roof["traffic"] = {"MB_per_launch": t["MB_per_launch"], "algorithmic_MB_per_launch": t["algorithmic_MB_per_launch"],
                                       "source": t["source"]}
            except (OSError, ValueError):
                pass
            roof["gemm_by_layout"] = ks
            roof["hbm_kernels"] = HBM_KERNELS  # in-step (beside the weight-gradient stream), algorithmic bytes / time
            roof["gemm_share_of_step"] = round(gt / (dt / args.steps), 3)
        out = {
            "metric": "training-step images/sec/node, ViT-B/16 224^2 (" + ("cls fine-tune" if args.workload == "cls" else "MAE pre-train") + ")",
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic" if args.input == "resident" else "synthetic uint8 frames in pinned host memory (PCIe-inclusive)",
            "config": {"workload": ("ViT-B/16 classification fine-tune" if args.workload == "cls" else
                                    "MAE pre-train ViT-B/16 mask 0.75") + f", bs={batch}/GPU, 224^2, AdamW, random init",
                       "global_batch": batch * world, "parallelism": f"dp{world}", "final_loss": round(loss_val, 5),
                       "launch": "hipGraph replay" if use_graph else "eager"},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_steps)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
