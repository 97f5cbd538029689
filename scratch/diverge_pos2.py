"""Async variant: no per-step synchronisation (as bench.py's loops); loss every 5 steps.  usage: diverge_pos2.py steps overlap(0|1) split(1|2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ssl4polyp_amd.engine import Kernels
dev = torch.device("cuda", 0)
steps = int(sys.argv[1])
bench.OVERLAP_ADAMW["cls"] = sys.argv[2] == "1"
Kernels.SPLIT_FORWARD = int(sys.argv[3])
model, ddp, opt = bench.build("cls", "bf16", dev, 1, 64)
imgs, labels = bench.make_batch("cls", 64, dev, 0)
step = bench.make_step("cls", ddp, opt, imgs, labels)
losses = []
for it in range(steps):
    losses.append(step().detach())
torch.cuda.synchronize()
vals = [float(x) for x in losses]
first_bad = next((i for i, v in enumerate(vals) if v != v), None)
print(f"overlap={sys.argv[2]} split={sys.argv[3]}: first non-finite step {first_bad}; losses[::10] = {[round(v, 4) for v in vals[::10]]}")
