import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
from ssl4polyp_amd._lib import EPI_STORE, EPI_RESIDUAL
k = Kernels("bf16"); k.lib.pm_debug_gemm_config.argtypes = [ctypes.c_int]
dev="cuda"; M=12608; D=768; H=3072; bf=torch.bfloat16
def t(*s, dt=bf): return (torch.randn(*s, device=dev) * 0.5).to(dt)
x, x3, xh = t(M, D), t(M, 3*D), t(M, H); Wqkv, Wproj, W1, W2 = t(3*D, D), t(D, D), t(H, D), t(D, H)
b1 = t(D, dt=torch.float32); oD = torch.empty(M, D, dtype=bf, device=dev); rD, fD = t(M, D, dt=torch.float32), torch.empty(M, D, device=dev)
cases = [("fwd proj", lambda: k.linear_fwd(x, Wproj, b1, fD, M, D, D, EPI_RESIDUAL, resid=rD), 2*M*D*D),
         ("fwd fc2 ", lambda: k.linear_fwd(xh, W2, b1, fD, M, D, H, EPI_RESIDUAL, resid=rD), 2*M*H*D),
         ("dgr fc1 ", lambda: k.linear_dgrad(xh, W1, oD, M, H, D), 2*M*H*D),
         ("dgr proj", lambda: k.linear_dgrad(x, Wproj, oD, M, D, D), 2*M*D*D),
         ("dgr qkv ", lambda: k.linear_dgrad(x3, Wqkv, oD, M, 3*D, D), 2*M*3*D*D)]
ref = (xh.float() @ W1.float())
for cfg in (9, 10):
    k.lib.pm_debug_gemm_config(cfg); k.linear_dgrad(xh, W1, oD, M, H, D)
    print(cfg, 'rel err', ((oD.float()-ref).abs().max()/ref.abs().max()).item())
for cfg in (9, 10, 0):
    k.lib.pm_debug_gemm_config(cfg); out=[]; tot=0
    for name, fn, fl in cases:
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize(); us=e0.elapsed_time(e1)*1e3/20; tot+=us
        out.append(f"{name} {us:6.1f}us/{fl/us/1e6:4.0f}TF")
    print(f"cfg {cfg}: "+"  ".join(out)+f"  total {tot:.1f}")
k.lib.pm_debug_gemm_config(0)
