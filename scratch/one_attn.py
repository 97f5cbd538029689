import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
k = Kernels("bf16"); dev="cuda"; bf=torch.bfloat16
B, N, H, dh = 64, 197, 12, 64
M = B*N; D = H*dh
qkv = (torch.randn(M, 3*D, device=dev)*0.5).to(bf); out = torch.empty(M, D, dtype=bf, device=dev); lse = torch.empty(B*H*N, device=dev)
dout = (torch.randn(M, D, device=dev)*0.5).to(bf); delta = torch.empty(B*H*N, device=dev); dqkv = torch.empty(M, 3*D, dtype=bf, device=dev)
for _ in range(12):
    k.attention_fwd(qkv, out, lse, B, N, H, dh); k.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H, dh)
torch.cuda.synchronize()
