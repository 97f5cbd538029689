import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
from ssl4polyp_amd._lib import EPI_STORE, EPI_RESIDUAL, EPI_GELU, EPI_DGELU
k = Kernels("bf16")
dev="cuda"; M=12608; D=768; H=3072; bf=torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else "qkv"; cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
def t(*s, dt=bf): return (torch.randn(*s, device=dev) * 0.5).to(dt)
k.gemm_variant = cfg
if which == "qkv":
    x, W, b, o = t(M, D), t(3*D, D), t(3*D, dt=torch.float32), torch.empty(M, 3*D, dtype=bf, device=dev)
    fn = lambda: k.linear_fwd(x, W, b, o, M, 3*D, D)
elif which == "dqkv":
    x3, W, o = t(M, 3*D), t(3*D, D), torch.empty(M, D, dtype=bf, device=dev)
    fn = lambda: k.linear_dgrad(x3, W, o, M, 3*D, D)
elif which == "wqkv":
    x3, x, o = t(M, 3*D), t(M, D), torch.empty(3*D, D, device=dev)
    fn = lambda: k.linear_wgrad(x3, x, o, M, 3*D, D, False)
for _ in range(20): fn()
torch.cuda.synchronize()
