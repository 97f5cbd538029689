import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
k = Kernels("bf16"); k.lib.pm_debug_gemm_config.argtypes = [ctypes.c_int]
dev="cuda"; M=12608; D=768; H=3072; bf=torch.bfloat16
def t(*s, dt=bf): return (torch.randn(*s, device=dev) * 0.5).to(dt)
x, xh = t(M, D), t(M, H); Wqkv, W2 = t(3*D, D), t(D, H); b3=t(3*D, dt=torch.float32)
o3 = torch.empty(M, 3*D, dtype=bf, device=dev); oD = torch.empty(M, D, dtype=bf, device=dev)
def run(fn, fl, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); us=e0.elapsed_time(e1)*1e3/n
    return us
for base in (3, 2):
  for dbg, nm in ((0,'full'),(1,'no-dma'),(2,'no-mfma'),(4,'no-epi'),(3,'no-dma no-mfma'),(7,'barriers only')):
    k.lib.pm_debug_gemm_config(base | (dbg<<4))
    a = run(lambda: k.linear_fwd(x, Wqkv, b3, o3, M, 3*D, D), 0)
    b = run(lambda: k.linear_fwd(xh, W2, None, oD, M, D, H), 0)
    print(f"cfg{base} {nm:16s} qkv(N2304,K768) {a:7.1f} us   fc2-like(N768,K3072) {b:7.1f} us")
k.lib.pm_debug_gemm_config(0)
