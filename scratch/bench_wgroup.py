"""Grouped weight gradients of one transformer block stand-alone: time per launch for the ViT-B block (108 full-K tiles) and
the MAE decoder block (48 tiles x k-slices).  usage: python scratch/bench_wgroup.py   (PM_GROUP_ORDER=0 = row-by-row baseline)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
k = Kernels("bf16"); dev = "cuda"; bf = torch.bfloat16
def t(*s, dt=bf): return (torch.randn(*s, device=dev) * 0.5).to(dt)
for M, D in ((12608, 768), (12800, 768), (50432, 512)):
    H = 4 * D
    dx, dh, dm, dq = t(M, D), t(M, H), t(M, D), t(M, 3 * D)
    hact, ln2, attn, ln1 = t(M, H), t(M, D), t(M, D), t(M, D)
    g2, g1, gp, gq = (torch.zeros(o, i, device=dev) for o, i in ((D, H), (H, D), (D, D), (3 * D, D)))
    b1, bq = torch.zeros(H, device=dev), torch.zeros(3 * D, device=dev)
    items = [(dx, hact, g2, False), (dh, ln2, g1, False, b1), (dm, attn, gp, False), (dq, ln1, gq, False, bq)]
    assert k.wgrad_group(items, M)
    ref = dq.float().t() @ ln1.float()
    err = ((gq - ref).abs().max() / ref.abs().max()).item()
    ref2 = dx.float().t() @ hact.float()
    err2 = ((g2 - ref2).abs().max() / ref2.abs().max()).item()
    fl = 2.0 * M * (D * H * 2 + D * D + 3 * D * D)
    res = []
    for rnd in range(3):
        for _ in range(3): k.wgrad_group(items, M)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): k.wgrad_group(items, M)
        e1.record(); torch.cuda.synchronize(); us = e0.elapsed_time(e1) * 1e3 / 20
        res.append(f"{us:7.1f}us {fl / us / 1e6:4.0f}TF")
    print(f"M={M} D={D}: rel err qkv {err:.2e} fc2 {err2:.2e}  " + "  ".join(res))
