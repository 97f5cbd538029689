import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "cls"; bs = 64 if wl == "cls" else 256
model, ddp, opt = bench.build(wl, "bf16", dev, 1, bs)
imgs, labels = bench.make_batch(wl, bs, dev, 0)
step = bench.make_step(wl, ddp, opt, imgs, labels)
for _ in range(5): step()
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    for _ in range(10): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{wl}: host enqueue {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
pstats.Stats(pr).sort_stats("tottime").print_stats(35)
