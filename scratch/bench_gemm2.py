"""GEMM shape sweep over the tuning configs (fwd + dgrad shapes)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
from ssl4polyp_amd._lib import EPI_STORE, EPI_GELU, EPI_RESIDUAL, EPI_DGELU
k = Kernels("bf16")
k.lib.pm_debug_gemm_config.argtypes = [ctypes.c_int]
dev = "cuda"
M = int(os.environ.get("M", 12608)); D = 768; H = 3072
bf = torch.bfloat16
def t(*s, dt=bf): return (torch.randn(*s, device=dev) * 0.5).to(dt)
x, x3, xh = t(M, D), t(M, 3 * D), t(M, H)
Wqkv, Wproj, W1, W2 = t(3 * D, D), t(D, D), t(H, D), t(D, H)
b3, b1, bH = t(3 * D, dt=torch.float32), t(D, dt=torch.float32), t(H, dt=torch.float32)
o3, oD, oH, aux = (torch.empty(M, n, dtype=bf, device=dev) for n in (3 * D, D, H, H))
rD, fD = t(M, D, dt=torch.float32), torch.empty(M, D, device=dev)
cases = [
 ("fwd qkv   NT store", lambda: k.linear_fwd(x, Wqkv, b3, o3, M, 3 * D, D), 2 * M * 3 * D * D),
 ("fwd proj  NT resid", lambda: k.linear_fwd(x, Wproj, b1, fD, M, D, D, EPI_RESIDUAL, resid=rD), 2 * M * D * D),
 ("fwd fc1   NT gelu ", lambda: k.linear_fwd(x, W1, bH, oH, M, H, D, EPI_GELU, aux=aux), 2 * M * H * D),
 ("fwd fc2   NT resid", lambda: k.linear_fwd(xh, W2, b1, fD, M, D, H, EPI_RESIDUAL, resid=rD), 2 * M * H * D),
 ("dgrad fc2 NN dgelu", lambda: k.linear_dgrad(x, W2, oH, M, D, H, EPI_DGELU, aux=aux), 2 * M * H * D),
 ("dgrad fc1 NN store", lambda: k.linear_dgrad(xh, W1, oD, M, H, D), 2 * M * H * D),
 ("dgrad prj NN store", lambda: k.linear_dgrad(x, Wproj, oD, M, D, D), 2 * M * D * D),
 ("dgrad qkv NN store", lambda: k.linear_dgrad(x3, Wqkv, oD, M, 3 * D, D), 2 * M * 3 * D * D),
]
# correctness spot check of every config against torch on one shape each layout
ref_nt = (x.float() @ Wqkv.float().t() + b3)
ref_nn = (x3.float() @ Wqkv.float())
for cfg in (3, 8, 9):
    k.lib.pm_debug_gemm_config(cfg)
    k.linear_fwd(x, Wqkv, b3, o3, M, 3 * D, D); k.linear_dgrad(x3, Wqkv, oD, M, 3 * D, D)
    e1 = ((o3.float() - ref_nt).abs().max() / ref_nt.abs().max()).item()
    e2 = ((oD.float() - ref_nn).abs().max() / ref_nn.abs().max()).item()
    print(f"cfg {cfg}: NT rel err {e1:.2e}  NN rel err {e2:.2e}")
for cfg in (3, 6, 8, 9):
    k.lib.pm_debug_gemm_config(cfg)
    tot_t = tot_f = 0
    line = []
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
        line.append(f"{name.split()[0]+' '+name.split()[1]}:{us:6.1f}us/{fl / us / 1e6:5.0f}TF")
    print(f"cfg {cfg}: " + "  ".join(line))
    print(f"   total {tot_t:.1f} us  {tot_f / tot_t / 1e6:.1f} TFLOP/s")
