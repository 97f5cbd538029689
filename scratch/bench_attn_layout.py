"""Attention forward / fused backward at ViT-B (B = 64, N = 197, H = 12, dh = 64) reading qkv token-major ([B, N, 3, H, dh], what the
qkv Linear writes today) or head-major ([3, B, H, N, dh], PM_ATTN_HEADMAJOR=1: what a head-scattering qkv epilogue would write).
Warm = the same buffer every launch (operands in L2 / Infinity Cache, as right behind the qkv GEMM); cold = 8 buffers in rotation
(464 MB > the 256 MB Infinity Cache).  Checks the result against plain torch.   usage: [PM_ATTN_HEADMAJOR=1] python scratch/bench_attn_layout.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
hm = os.environ.get("PM_ATTN_HEADMAJOR") == "1"
k = Kernels("bf16"); dev = "cuda"; bf = torch.bfloat16
for (B, N, H, dh) in ((64, 197, 12, 64), (32, 197, 12, 64)):
    M, D = B * N, H * dh
    def mk(seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        tok = (torch.randn(B, N, 3, H, dh, device=dev, generator=g) * 0.5).to(bf)
        return tok, (tok.permute(2, 0, 3, 1, 4).contiguous() if hm else tok)
    bufs = [mk(s) for s in range(8)]
    out = torch.empty(M, D, dtype=bf, device=dev); lse = torch.empty(B * H * N, device=dev)
    dout = (torch.randn(M, D, device=dev) * 0.5).to(bf); delta = torch.empty(B * H * N, device=dev); dqkv = torch.empty(M, 3 * D, dtype=bf, device=dev)
    tok, buf = bufs[0]
    k.attention_fwd(buf, out, lse, B, N, H, dh)
    q, kk, v = tok.float().permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ kk.transpose(-2, -1) * dh ** -0.5, -1) @ v).transpose(1, 2).reshape(M, D)
    err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
    k.attention_bwd(buf, out, dout, lse, delta, dqkv, B, N, H, dh)
    qr = tok.float().requires_grad_(True)
    q, kk, v = qr.permute(2, 0, 3, 1, 4)
    (torch.softmax(q @ kk.transpose(-2, -1) * dh ** -0.5, -1) @ v).transpose(1, 2).reshape(M, D).backward(dout.float())
    errb = ((dqkv.float().view(B, N, 3, H, dh) - qr.grad).abs().max() / qr.grad.abs().max()).item()
    def run(fn, n=40):
        for i in range(4): fn(i)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
    fw = run(lambda i: k.attention_fwd(bufs[0][1], out, lse, B, N, H, dh))
    fc = run(lambda i: k.attention_fwd(bufs[i % 8][1], out, lse, B, N, H, dh))
    bw = run(lambda i: k.attention_bwd(bufs[0][1], out, dout, lse, delta, dqkv, B, N, H, dh))
    bc = run(lambda i: k.attention_bwd(bufs[i % 8][1], out, dout, lse, delta, dqkv, B, N, H, dh))
    print(f"{'head-major' if hm else 'token-major'} B={B}: fwd warm {fw:5.1f} cold {fc:5.1f} us | bwd warm {bw:5.1f} cold {bc:5.1f} us | rel err fwd {err:.1e} bwd {errb:.1e}")
