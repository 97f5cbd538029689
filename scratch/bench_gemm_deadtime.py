"""What a ring-GEMM tile costs besides its k-loop: the same [M, N] problem at K = 64 (two k-steps: launch + prologue + epilogue + drain)
and at K = 768 / 3072, per epilogue.  usage: python scratch/bench_gemm_deadtime.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
from ssl4polyp_amd._lib import EPI_STORE, EPI_GELU, EPI_RESIDUAL
dev = "cuda"; bf = torch.bfloat16
def t(*s, dt=bf): return (torch.randn(*s, device=dev) * 0.5).to(dt)
def run(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
k = Kernels("bf16")
# an empty kernel's launch-to-launch cost on this box, for scale
z = torch.zeros(64, device=dev)
print(f"back-to-back tiny launches (torch fill of 64 floats): {run(lambda: z.zero_()):.1f} us")
for M in (6304, 12608):
    for N, epi_name in ((768, "resid"), (768, "store"), (2304, "store"), (3072, "gelu")):
        row = []
        for K in (64, 256, 768, 3072):
            x, W = t(M, K), t(N, K)
            bias = torch.zeros(N, device=dev)
            out16, aux = torch.empty(M, N, dtype=bf, device=dev), torch.empty(M, N, dtype=bf, device=dev)
            out32, res = torch.empty(M, N, device=dev), t(M, N, dt=torch.float32)
            if epi_name == "resid": fn = lambda: k.linear_fwd(x, W, bias, out32, M, N, K, EPI_RESIDUAL, resid=res)
            elif epi_name == "gelu": fn = lambda: k.linear_fwd(x, W, bias, out16, M, N, K, EPI_GELU, aux=aux)
            else: fn = lambda: k.linear_fwd(x, W, bias, out16, M, N, K)
            row.append(f"K={K}: {run(fn):.1f}")
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        print(f"M={M} N={N} {epi_name:5s} ({tiles} tiles of 256x256): " + "  ".join(row) + " us")
