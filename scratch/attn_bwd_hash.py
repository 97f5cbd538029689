"""Hash of the attention backward's output on a seeded ViT-B-sized problem (A/B builds / switches must agree bit for bit)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
k = Kernels("bf16"); dev = "cuda"; bf = torch.bfloat16
torch.manual_seed(7)
for (B, N, H, dh) in ((64, 197, 12, 64), (30, 100, 12, 64), (23, 224, 12, 64)):
    M, D = B * N, H * dh
    qkv = (torch.randn(M, 3 * D, device=dev) * 0.7).to(bf)
    out = torch.empty(M, D, dtype=bf, device=dev); lse = torch.empty(B * H * N, device=dev)
    dout = (torch.randn(M, D, device=dev) * 0.5).to(bf); delta = torch.empty(B * H * N, device=dev)
    dqkv = torch.full((M, 3 * D), float("nan"), dtype=bf, device=dev)
    k.attention_fwd(qkv, out, lse, B, N, H, dh)
    k.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H, dh)
    torch.cuda.synchronize()
    assert torch.isfinite(dqkv.float()).all()
    print(B, N, H, dh, hashlib.sha256(dqkv.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16])
