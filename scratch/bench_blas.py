import torch, time
dev="cuda"; bf=torch.bfloat16; M=12608
def run(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1e3/n
for name,(m,n,k) in {"qkv":(M,2304,768),"proj":(M,768,768),"fc1":(M,3072,768),"fc2":(M,768,3072),"wgrad_fc1":(3072,768,M),"wgrad_qkv":(2304,768,M), "sq4096":(4096,4096,4096), "sq8192":(8192,8192,8192)}.items():
    a=(torch.randn(m,k,device=dev)*0.5).to(bf); b=(torch.randn(n,k,device=dev)*0.5).to(bf)
    us=run(lambda: torch.matmul(a,b.t()))
    print(f"hipBLASLt/rocBLAS {name:10s} {m}x{n}x{k}: {us:8.1f} us  {2*m*n*k/us/1e6:7.1f} TFLOP/s")
