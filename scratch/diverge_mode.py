"""Per-step loss / first non-finite tensor of a fine-tune regime (finetune.py:49-91) on the benched step.
usage: python scratch/diverge_mode.py MODE [steps] [sync 0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
mode = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sync = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
for prec in ("bf16", "fp32"):
    model, ddp, opt = bench.build("cls", prec, dev, 1, 64, mode)
    imgs, labels = bench.make_batch("cls", 64, dev, 0)
    step = bench.make_step("cls", ddp, opt, imgs, labels)
    hist = []
    for it in range(steps):
        loss = step()
        hist.append(loss.detach())
        if not sync:
            continue
        torch.cuda.synchronize()
        model._rt.wait_updates()
        torch.cuda.synchronize()
        lv = float(loss.detach())
        bad_g = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
        bad_w = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p.detach()).all())]
        if it % 10 == 0 or bad_g or bad_w or lv != lv:
            gn = max((float(p.grad.abs().max()) for p in model.parameters() if p.grad is not None), default=0.0)
            print(f"{prec} {mode} step {it:4d} loss {lv:.6f} max|g| {gn:.3e} max|w| {max(float(p.detach().abs().max()) for p in model.parameters()):.3f}", flush=True)
        if bad_g or bad_w or lv != lv:
            print("non-finite grads:", bad_g[:8], "weights:", bad_w[:8])
            break
    if not sync:
        torch.cuda.synchronize()
        vals = [float(x) for x in hist]
        print(prec, mode, "async: first non-finite", next((i for i, v in enumerate(vals) if v != v), None), [round(v, 4) for v in vals[::10]])
    del model, ddp, opt
    torch.cuda.empty_cache()
