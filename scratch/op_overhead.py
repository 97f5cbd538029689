"""Host cost per launch: the C-ABI through ctypes (what the engine does) vs the same kernel through its registered torch op."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd import ops
from ssl4polyp_amd.engine import Kernels
k = Kernels("bf16"); dev = "cuda"
M, D = 64, 768
x = torch.randn(M, D, device=dev); g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
y = torch.empty(M, D, dtype=torch.bfloat16, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t0; torch.cuda.synchronize(); return dt / n * 1e6
a = t(lambda: k.layernorm_fwd(x, g, b, y, mean, rstd, M, D))
with torch.no_grad():
    c = t(lambda: torch.ops.polypmae.layernorm(x, g, b, 1e-6, True))
xr = x.clone().requires_grad_(True)
d = t(lambda: torch.ops.polypmae.layernorm(xr, g, b, 1e-6, True))
e = t(lambda: torch.nn.functional.layer_norm(x, (D,), g, b, 1e-6))
print(f"host us per launch: ctypes C-ABI {a:.1f} | registered op (no grad) {c:.1f} | registered op (autograd node) {d:.1f} | aten layer_norm {e:.1f}")
