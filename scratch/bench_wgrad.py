import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
import ctypes
k = Kernels("bf16"); k.gemm_variant = (int(os.environ.get("WV", 0)) << 6) | (int(os.environ.get("HDM", 0)) << 8); k.WGRAD_BLOCKS = int(os.environ.get("WB", 256)); dev="cuda"; M=12608; D=768; H=3072; bf=torch.bfloat16
def t(*s, dt=bf): return (torch.randn(*s, device=dev) * 0.5).to(dt)
xD, x3, xH = t(M, D), t(M, 3*D), t(M, H)
gq, gp, g1, g2 = (torch.empty(n, kk, device=dev) for n, kk in ((3*D, D), (D, D), (H, D), (D, H)))
cases = [("wqkv", lambda: k.linear_wgrad(x3, xD, gq, M, 3*D, D, False), 2*M*3*D*D),
         ("wproj", lambda: k.linear_wgrad(xD, xD, gp, M, D, D, False), 2*M*D*D),
         ("wfc1", lambda: k.linear_wgrad(xH, xD, g1, M, H, D, False), 2*M*H*D),
         ("wfc2", lambda: k.linear_wgrad(xD, xH, g2, M, D, H, False), 2*M*H*D),
         ("wfc2acc", lambda: k.linear_wgrad(xD, xH, g2, M, D, H, True), 2*M*H*D)]
ref = x3.float().t() @ xD.float(); k.linear_wgrad(x3, xD, gq, M, 3*D, D, False)
print("wqkv rel err", ((gq-ref).abs().max()/ref.abs().max()).item())
for rnd in range(2):
    out = []
    for name, fn, fl in cases:
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize(); us=e0.elapsed_time(e1)*1e3/20
        out.append(f"{name}:{us:6.1f}us {fl/us/1e6:4.0f}TF")
    print("  ".join(out))
