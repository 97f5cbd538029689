"""Why is the bf16 parity block of bench.py (fresh init, B=64) at 2e-2 on the logits?  fp32 mode, bf16 mode and the CPU
bf16-operand emulation (oracle/vit_bf16_sim.py) on the same weights / batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from oracle import vit_mae_ref as O
from oracle import vit_bf16_sim as S
torch.set_num_threads(16)
dev = torch.device("cuda", 0)
cfg = O.VIT_BASE
imgs, labels = bench.make_batch("cls", 64, dev, 0)
def rel(a, b): a, b = a.double().cpu(), b.double().cpu(); return float((a - b).abs().max() / b.abs().max())
ref = None
for prec in ("fp32", "bf16"):
    model, ddp, opt = bench.build("cls", prec, dev, 1, 64)
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        got = model(imgs)
        if ref is None:
            ref = O.vit_classify(sd, imgs.cpu(), cfg)
            sim = S.vit_classify(sd, imgs.cpu(), cfg)
            print("ref logits: max |z| %.4f  mean |z| %.4f ; emulation vs fp32 oracle: %.3e" % (ref.abs().max(), ref.abs().mean(), rel(sim, ref)))
    print(prec, "HIP vs fp32 oracle: logits max-rel %.3e" % rel(got, ref), " vs emulation %.3e" % rel(got, sim))
    # per-block growth of the residual stream at init
x = O.patch_embed(imgs.cpu(), sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], 16)
print("patch-embed output rms %.3f" % x.pow(2).mean().sqrt())
feats = O.vit_features(sd, imgs.cpu(), cfg, False)
print("final features rms %.3f, cls row rms %.3f" % (feats.pow(2).mean().sqrt(), feats[:, 0].pow(2).mean().sqrt()))
