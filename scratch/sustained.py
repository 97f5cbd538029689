import os, sys, time, subprocess, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.engine import Kernels
k = Kernels("bf16"); dev="cuda"; M=12608; D=768; bf=torch.bfloat16
def t(*s, dt=bf): return (torch.randn(*s, device=dev) * 0.5).to(dt)
x, W, b, o = t(M, D), t(3*D, D), t(3*D, dt=torch.float32), torch.empty(M, 3*D, dtype=bf, device=dev)
fn = lambda: k.linear_fwd(x, W, b, o, M, 3*D, D)
fl = 2*M*3*D*D
samples = []
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), out.strip().splitlines()[-1] if out.strip() else "?"))
        except Exception as e:
            samples.append((time.time(), repr(e)))
        time.sleep(0.4)
th = threading.Thread(target=poll); th.start()
time.sleep(1.0)
for n in (20, 200, 2000, 20000, 20000):
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); us=e0.elapsed_time(e1)*1e3/n
    print(f"n={n:6d}: {us:6.1f} us/GEMM  {fl/us/1e6:5.0f} TF", flush=True)
stop = True; th.join()
hdr = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()
print(hdr[0] if hdr else "no header")
t0 = samples[0][0]
for ts, s in samples[::2]: print(f"{ts-t0:5.1f}s {s}")
