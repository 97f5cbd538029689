"""Forward / dgrad GEMMs at small M (MAE encoder at bs = 64/GPU: M = 3 200, or 1 600 per forward chain; half-batch cls: 6 304):
the 256-wide ring kernel (default dispatch for M >= 1024) against the 128 x 128 LDS-DMA kernel (pm_gemm_opts.variant = 1).
usage: python scratch/bench_gemm_smallm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
from ssl4polyp_amd._lib import EPI_STORE, EPI_GELU, EPI_RESIDUAL, EPI_DGELU
dev = "cuda"; bf = torch.bfloat16
def t(*s, dt=bf): return (torch.randn(*s, device=dev) * 0.5).to(dt)
def run(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
ks = {0: Kernels("bf16"), 1: Kernels("bf16")}
ks[1].gemm_variant = 1
D = int(os.environ.get("D", "768"))
for M in [int(m) for m in os.environ.get("MS", "1600,3200,6304,12608").split(",")]:
    rows = []
    for name, N, K, kind in (("qkv", 3 * D, D, "store"), ("proj", D, D, "resid"), ("fc1+gelu", 4 * D, D, "gelu"), ("fc2", D, 4 * D, "resid"),
                             ("dfc2+dgelu", 4 * D, D, "dgelu"), ("dfc1", D, 4 * D, "dgrad"), ("dqkv", D, 3 * D, "dgrad")):
        x, W = t(M, K), t(N, K)
        bias = torch.zeros(N, device=dev)
        out16, aux = torch.empty(M, N, dtype=bf, device=dev), t(M, N)
        out32, res = torch.empty(M, N, device=dev), t(M, N, dt=torch.float32)
        Wk = t(K, N)  # dgrad: W stored [K_out?]: dx[M,N] = dy[M,K] @ W[K,N] (k-major B)
        res_t = {}
        for v, k in ks.items():
            if kind == "store": fn = lambda: k.linear_fwd(x, W, bias, out16, M, N, K)
            elif kind == "resid": fn = lambda: k.linear_fwd(x, W, bias, out32, M, N, K, EPI_RESIDUAL, resid=res)
            elif kind == "gelu": fn = lambda: k.linear_fwd(x, W, bias, out16, M, N, K, EPI_GELU, aux=aux)
            elif kind == "dgelu": fn = lambda: k.linear_dgrad(x, Wk, out16, M, K, N, EPI_DGELU, aux=aux)
            else: fn = lambda: k.linear_dgrad(x, Wk, out16, M, K, N)
            res_t[v] = run(fn)
        rows.append(f"{name} {res_t[0]:.1f}/{res_t[1]:.1f}")
    print(f"M={M}: ring/128x128 us: " + "  ".join(rows))
