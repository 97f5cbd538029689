import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
from ssl4polyp_amd._lib import EPI_STORE, EPI_RESIDUAL, EPI_GELU, EPI_DGELU
k = Kernels("bf16")
dev="cuda"; M=int(os.environ.get("M", 12608)); D=int(os.environ.get("D", 768)); H=4*D; bf=torch.bfloat16
CFGS = [int(c) for c in os.environ.get("CFGS", "0,16,17,18,19,20,21").split(",")]
def t(*s, dt=bf): return (torch.randn(*s, device=dev) * 0.5).to(dt)
x, x3, xh = t(M, D), t(M, 3*D), t(M, H); Wqkv, Wproj, W1, W2 = t(3*D, D), t(D, D), t(H, D), t(D, H)
b3, b1, bH = t(3*D, dt=torch.float32), t(D, dt=torch.float32), t(H, dt=torch.float32)
o3, oD, oH, aux = (torch.empty(M, n, dtype=bf, device=dev) for n in (3*D, D, H, H)); rD, fD = t(M, D, dt=torch.float32), torch.empty(M, D, device=dev)
cases = [("qkv", lambda: k.linear_fwd(x, Wqkv, b3, o3, M, 3*D, D), 2*M*3*D*D),
         ("proj", lambda: k.linear_fwd(x, Wproj, b1, fD, M, D, D, EPI_RESIDUAL, resid=rD), 2*M*D*D),
         ("fc1g", lambda: k.linear_fwd(x, W1, bH, oH, M, H, D, EPI_GELU, aux=aux), 2*M*H*D),
         ("fc2", lambda: k.linear_fwd(xh, W2, b1, fD, M, D, H, EPI_RESIDUAL, resid=rD), 2*M*H*D),
         ("dfc2", lambda: k.linear_dgrad(x, W2, oH, M, D, H, EPI_DGELU, aux=aux), 2*M*H*D),
         ("dfc1", lambda: k.linear_dgrad(xh, W1, oD, M, H, D), 2*M*H*D),
         ("dproj", lambda: k.linear_dgrad(x, Wproj, oD, M, D, D), 2*M*D*D),
         ("dqkv", lambda: k.linear_dgrad(x3, Wqkv, oD, M, 3*D, D), 2*M*3*D*D)]
ref = x3.float() @ Wqkv.float(); ref2 = x.float() @ Wqkv.float().t() + b3
for cfg in CFGS:
    k.gemm_variant = cfg; oD.zero_(); o3.zero_()
    k.linear_dgrad(x3, Wqkv, oD, M, 3*D, D); k.linear_fwd(x, Wqkv, b3, o3, M, 3*D, D)
    print(cfg, "rel err dgrad", ((oD.float()-ref).abs().max()/ref.abs().max()).item(), "fwd", ((o3.float()-ref2).abs().max()/ref2.abs().max()).item())
for rnd in range(2):
  for name, fn, fl in cases:
    out=[]
    for cfg in CFGS:
        k.gemm_variant = cfg
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize(); us=e0.elapsed_time(e1)*1e3/20
        out.append(f"c{cfg}:{us:6.1f}us {fl/us/1e6:4.0f}TF")
    print(f"{name:5s} "+"  ".join(out))
k.gemm_variant = 0
