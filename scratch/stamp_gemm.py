"""Diagnostic: per-segment cycle breakdown of the ping-pong GEMM k-loop (needs the -DPM_GEMM_STAMP build:
   scratch/build_stamp.sh -> ssl4polyp_amd/lib/libpolypmae_stamp.so; run with POLYPMAE_LIB pointing at it)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
from ssl4polyp_amd._lib import EPI_STORE, EPI_RESIDUAL, EPI_GELU, EPI_DGELU
k = Kernels("bf16"); k.lib.pm_debug_gemm_stamps.argtypes = [ctypes.c_void_p]
dev="cuda"; M=int(os.environ.get("M", 12608)); D=768; H=3072; bf=torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else "qkv"; cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
def t(*s, dt=bf): return (torch.randn(*s, device=dev) * 0.5).to(dt)
k.gemm_variant = cfg
if which == "qkv":
    x, W, b, o = t(M, D), t(3*D, D), t(3*D, dt=torch.float32), torch.empty(M, 3*D, dtype=bf, device=dev)
    fn = lambda: k.linear_fwd(x, W, b, o, M, 3*D, D)
elif which == "fc2":
    xh, W, b, r, o = t(M, H), t(D, H), t(D, dt=torch.float32), t(M, D, dt=torch.float32), torch.empty(M, D, device=dev)
    fn = lambda: k.linear_fwd(xh, W, b, o, M, D, H, EPI_RESIDUAL, resid=r)
elif which == "dqkv":
    x3, W, o = t(M, 3*D), t(3*D, D), torch.empty(M, D, dtype=bf, device=dev)
    fn = lambda: k.linear_dgrad(x3, W, o, M, 3*D, D)
elif which == "wqkv":
    x3, x, o = t(M, 3*D), t(M, D), torch.empty(3*D, D, device=dev)
    fn = lambda: k.linear_wgrad(x3, x, o, M, 3*D, D, False)
for _ in range(200): fn()   # settle the clock
buf = torch.zeros(4096 * 8 * 16, dtype=torch.int64, device=dev)
k.lib.pm_debug_gemm_stamps(buf.data_ptr())
fn(); torch.cuda.synchronize()
k.lib.pm_debug_gemm_stamps(None)
s = buf.view(-1, 8, 16).cpu().double()
used = s[:, 0, 7] > 0
s = s[used]
names = ["vmcnt wait", "barrier", "MFMA 0-7 (+reads, DMA)", "MFMA 8-15 (+reads, DMA)", "-", "-", "-"] if cfg in (24, 25, 26) else ["barrier R", "ds_reads issue", "DMA issue", "vmcnt wait", "lgkmcnt wait", "barrier M", "MFMA issue"]
for g, sl in (("waves 0-3 (lead)", slice(0, 4)), ("waves 4-7 (lag)", slice(4, 8))):
    per = s[:, sl, :7] / s[:, sl, 7:8]
    m = per.mean(dim=(0, 1)); tot = m.sum().item()
    print(f"{which} cfg{cfg} {g}: blocks {s.shape[0]}, k-steps/block {int(s[0,0,7])}, cycles per k-step {tot:.0f}")
    for n, v in zip(names, m.tolist()): print(f"    {n:16s} {v:7.1f}  ({100*v/tot:4.1f} %)")

ent, lend, iss, ack = s[:, :, 8], s[:, :, 9], s[:, :, 10], s[:, :, 11]
loop = s[:, :, :7].sum(dim=2)
t0 = ent.min()
print(f"whole kernel ({s.shape[0]} blocks): first entry -> last ack {(ack.max()-t0):.0f} cyc")
print(f"  per wave means: entry->loop start {(lend-ent-loop).mean():.0f}   k-loop {loop.mean():.0f}   loop end->stores issued {(iss-lend).mean():.0f}   stores issued->acked {(ack-iss).mean():.0f}   lifetime {(ack-ent).mean():.0f}")
e = (ent[:, 0] - t0).sort().values
print("  block entry times (cyc after first): p10 %.0f p50 %.0f p90 %.0f max %.0f" % tuple(e[int(len(e)*q)] if q < 1 else e[-1] for q in (0.1, 0.5, 0.9, 1)))
x = (ack.max(dim=1).values - t0).sort().values
print("  block finish times: p10 %.0f p50 %.0f p90 %.0f max %.0f" % tuple(x[int(len(x)*q)] if q < 1 else x[-1] for q in (0.1, 0.5, 0.9, 1)))
