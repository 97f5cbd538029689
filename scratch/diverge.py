"""Where does the benched cls step (lr 1e-3 constant, one fixed batch, random labels) stop being finite?  Loss and a weight /
activation scale every 20 steps, bf16 and fp32 modes of the HIP path, to tell an honest optimisation blow-up from a kernel
overflow.  usage: python scratch/diverge.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for prec in ("bf16", "fp32"):
    model, ddp, opt = bench.build("cls", prec, dev, 1, 64)
    imgs, labels = bench.make_batch("cls", 64, dev, 0)
    step = bench.make_step("cls", ddp, opt, imgs, labels)
    bad = None
    for it in range(steps):
        loss = step()
        if it % 20 == 0 or it == steps - 1:
            torch.cuda.synchronize()
            lv = float(loss.detach())
            wmax = max(float(p.detach().abs().max()) for p in model.parameters())
            gfin = all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
            print(f"{prec} step {it:4d} loss {lv:.6f} max|w| {wmax:.3f} grads finite {gfin}", flush=True)
            if not (lv == lv) or not gfin:
                bad = it
                break
    print(prec, "first non-finite (20-step grid):", bad)
    del model, ddp, opt
    torch.cuda.empty_cache()
