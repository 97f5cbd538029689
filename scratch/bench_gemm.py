"""Micro-benchmark of the GEMM shapes of one ViT-B/16 block at B=64 (M = 12608 tokens)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
from ssl4polyp_amd._lib import EPI_STORE, EPI_GELU, EPI_RESIDUAL, EPI_DGELU, EPI_ACCUM
k = Kernels("bf16")
dev = "cuda"
M = int(os.environ.get("M", 12608)); D = 768; H = 3072
bf = torch.bfloat16
def t(*s, dt=bf): return (torch.randn(*s, device=dev) * 0.5).to(dt)
x, x3, xh = t(M, D), t(M, 3 * D), t(M, H)
Wqkv, Wproj, W1, W2 = t(3 * D, D), t(D, D), t(H, D), t(D, H)
b3, b1, bH = t(3 * D, dt=torch.float32), t(D, dt=torch.float32), t(H, dt=torch.float32)
o3, oD, oH, aux = torch.empty(M, 3 * D, dtype=bf, device=dev), torch.empty(M, D, dtype=bf, device=dev), torch.empty(M, H, dtype=bf, device=dev), torch.empty(M, H, dtype=bf, device=dev)
rD, fD = t(M, D, dt=torch.float32), torch.empty(M, D, device=dev)
gW3, gWp, gW1, gW2 = torch.empty(3 * D, D, device=dev), torch.empty(D, D, device=dev), torch.empty(H, D, device=dev), torch.empty(D, H, device=dev)
cases = [
 ("fwd qkv   NT store", lambda: k.linear_fwd(x, Wqkv, b3, o3, M, 3 * D, D), 2 * M * 3 * D * D),
 ("fwd proj  NT resid", lambda: k.linear_fwd(x, Wproj, b1, fD, M, D, D, EPI_RESIDUAL, resid=rD), 2 * M * D * D),
 ("fwd fc1   NT gelu ", lambda: k.linear_fwd(x, W1, bH, oH, M, H, D, EPI_GELU, aux=aux), 2 * M * H * D),
 ("fwd fc2   NT resid", lambda: k.linear_fwd(xh, W2, b1, fD, M, D, H, EPI_RESIDUAL, resid=rD), 2 * M * H * D),
 ("dgrad fc2 NN dgelu", lambda: k.linear_dgrad(x, W2, oH, M, D, H, EPI_DGELU, aux=aux), 2 * M * H * D),
 ("dgrad fc1 NN store", lambda: k.linear_dgrad(xh, W1, oD, M, H, D), 2 * M * H * D),
 ("dgrad prj NN store", lambda: k.linear_dgrad(x, Wproj, oD, M, D, D), 2 * M * D * D),
 ("dgrad qkv NN store", lambda: k.linear_dgrad(x3, Wqkv, oD, M, 3 * D, D), 2 * M * 3 * D * D),
 ("wgrad fc2 TN      ", lambda: k.linear_wgrad(x, xh, gW2, M, D, H, False), 2 * M * H * D),
 ("wgrad fc1 TN      ", lambda: k.linear_wgrad(xh, x, gW1, M, H, D, False), 2 * M * H * D),
 ("wgrad prj TN      ", lambda: k.linear_wgrad(x, x, gWp, M, D, D, False), 2 * M * D * D),
 ("wgrad qkv TN      ", lambda: k.linear_wgrad(x3, x, gW3, M, 3 * D, D, False), 2 * M * 3 * D * D),
]
tot_t = tot_f = 0
for name, fn, fl in cases:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    tot_t += us; tot_f += fl
    print(f"{name}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
print(f"block GEMM total: {tot_t:.1f} us  {tot_f / tot_t / 1e6:.1f} TFLOP/s   (x12 blocks = {tot_t * 12 / 1e3:.2f} ms)")
