"""Full fine-tune with the positional table trainable (finetune.py:52-55): per-step loss, first non-finite tensor.
usage: python scratch/diverge_pos.py [steps] [overlap 0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if len(sys.argv) > 2:
    bench.OVERLAP_ADAMW["cls"] = sys.argv[2] == "1"
for prec in ("bf16", "fp32"):
    model, ddp, opt = bench.build("cls", prec, dev, 1, 64)
    imgs, labels = bench.make_batch("cls", 64, dev, 0)
    step = bench.make_step("cls", ddp, opt, imgs, labels)
    for it in range(steps):
        loss = step()
        torch.cuda.synchronize()
        model._rt.wait_updates()
        torch.cuda.synchronize()
        lv = float(loss.detach())
        bad_g = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
        bad_w = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p.detach()).all())]
        if it % 10 == 0 or bad_g or bad_w or lv != lv:
            pe = model.pos_embed.detach()
            print(f"{prec} step {it:4d} loss {lv:.6f} |pos|max {float(pe.abs().max()):.3f} |dpos|max "
                  f"{float(model.pos_embed.grad.abs().max()) if model.pos_embed.grad is not None else -1:.3e} "
                  f"max|w| {max(float(p.detach().abs().max()) for p in model.parameters()):.3f}", flush=True)
        if bad_g or bad_w or lv != lv:
            print("non-finite grads:", bad_g[:6], "weights:", bad_w[:6])
            break
    del model, ddp, opt
    torch.cuda.empty_cache()
