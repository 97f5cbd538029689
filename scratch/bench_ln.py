import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
k = Kernels("bf16"); dev="cuda"
for M, D in ((12608, 768), (50432, 512), (12800, 768)):
    x = torch.randn(M, D, device=dev); g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    y = torch.empty(M, D, dtype=torch.bfloat16, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    dy = torch.randn(M, D, device=dev).bfloat16(); dres = torch.randn(M, D, device=dev); dx = torch.empty(M, D, device=dev); dxa = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    dg, db, dc = (torch.zeros(D, device=dev) for _ in range(3))
    k.layernorm_fwd(x, g, b, y, mean, rstd, M, D)
    def run(fn, n=30):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/n
    tf = run(lambda: k.layernorm_fwd(x, g, b, y, mean, rstd, M, D))
    tb = run(lambda: k.layernorm_bwd(dy, x, g, mean, rstd, dres, dx, dxa, dg, db, dc, M, D))
    print(f"M={M} D={D}: fwd {tf:6.1f} us ({M*D*6/tf/1e6:.2f} TB/s)   bwd+reduce {tb:6.1f} us ({M*D*18/tb/1e6:.2f} TB/s)")
