import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
torch.manual_seed(0)
B,N,H,dh=2,197,12,64
qkv=(torch.randn(B,N,3*H*dh)*1.5).bfloat16().float()
ref64=None
q,k,v=qkv.double().reshape(B,N,3,H,dh).permute(2,0,3,1,4)
s=(q@k.transpose(-2,-1))*dh**-0.5
o64=(s.softmax(-1)@v).transpose(1,2).reshape(B,N,H*dh)
for prec in ("fp32","bf16"):
    kk=Kernels(prec)
    x=qkv.to(kk.act_dtype).cuda(); out=torch.empty(B,N,H*dh,dtype=kk.act_dtype,device="cuda"); lse=torch.empty(B,H,N,device="cuda")
    kk.attention_fwd(x,out,lse,B,N,H,dh); torch.cuda.synchronize()
    d=(out.double().cpu()-o64).abs()
    print(prec,"max abs err",d.max().item(),"median rel err",(d/o64.abs().clamp_min(1e-9)).median().item(),"max|o|",o64.abs().max().item())
    lse64=torch.logsumexp(s,-1)
    print("   lse max abs err",(lse.double().cpu()-lse64).abs().max().item())
