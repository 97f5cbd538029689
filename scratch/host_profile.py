"""cProfile of the host side of the benched step (which Python / ctypes calls the 3-5 ms of enqueue time per step are made of).
usage: python scratch/host_profile.py [cls|mae] [steps]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
wl = sys.argv[1] if len(sys.argv) > 1 else "cls"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
batch = 64 if wl == "cls" else 256
model, ddp, opt = bench.build(wl, "bf16", dev, 1, batch)
imgs, labels = bench.make_batch(wl, batch, dev, 0)
step = bench.make_step(wl, ddp, opt, imgs, labels)
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
import time
t0 = time.perf_counter()
pr.enable()
for _ in range(n): step()
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"{wl}: host time per step with the queue filling: {(t1 - t0) / n * 1e3:.3f} ms")
st = pstats.Stats(pr); st.sort_stats("tottime")
import io
buf = io.StringIO(); st.stream = buf; st.print_stats(28)
for ln in buf.getvalue().splitlines():
    if ln.strip(): print(ln[:170])
