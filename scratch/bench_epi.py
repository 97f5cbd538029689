import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
k = Kernels("bf16"); k.lib.pm_debug_gemm_config.argtypes = [ctypes.c_int]
dev="cuda"; M=12608; bf=torch.bfloat16
def run(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/n
for N in (2304, 768, 3072):
  for K in (32, 64, 256, 768):
    x=(torch.randn(M,K,device=dev)*.5).to(bf); w=(torch.randn(N,K,device=dev)*.5).to(bf); o=torch.empty(M,N,dtype=bf,device=dev)
    b=torch.randn(N,device=dev)
    res=[]
    for cfg in (1,3,4,6):
        k.lib.pm_debug_gemm_config(cfg)
        res.append(run(lambda: k.linear_fwd(x,w,b,o,M,N,K)))
    tb=run(lambda: torch.matmul(x,w.t()))
    cp=run(lambda: o.copy_(o))  # pure read+write of the output size
    print(f"N={N} K={K}: cfg1 {res[0]:6.1f} cfg3 {res[1]:6.1f} cfg4 {res[2]:6.1f} cfg6 {res[3]:6.1f}  blas {tb:6.1f}  copy(out) {cp:6.1f} us   out={M*N*2/1e6:.0f}MB")
k.lib.pm_debug_gemm_config(0)
