import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
k = Kernels("bf16"); dev="cuda"; bf=torch.bfloat16
for (B, N, H, dh) in ((64, 197, 12, 64), (256, 50, 12, 64), (256, 197, 16, 32)):
    M = B*N; D = H*dh
    qkv = (torch.randn(M, 3*D, device=dev)*0.5).to(bf); out = torch.empty(M, D, dtype=bf, device=dev); lse = torch.empty(B*H*N, device=dev)
    dout = (torch.randn(M, D, device=dev)*0.5).to(bf); delta = torch.empty(B*H*N, device=dev); dqkv = torch.empty(M, 3*D, dtype=bf, device=dev)
    def run(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/n
    tf = run(lambda: k.attention_fwd(qkv, out, lse, B, N, H, dh))
    tb = run(lambda: k.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H, dh))
    fl = 4*N*N*dh*B*H
    print(f"B={B} N={N} H={H} dh={dh}: fwd {tf:6.1f} us ({fl/tf/1e6:4.0f} TF)  bwd {tb:6.1f} us ({2.5*fl/tb/1e6:4.0f} TF)")
