"""Determinism probe: the same MAE (or cls) step N times in one process; every step's gradient hash must be the same.
usage: python scratch/det_mae.py [mae|cls] [steps]     (env PM_TWO_GROUPS etc. apply)"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssl4polyp_amd as A

wl = sys.argv[1] if len(sys.argv) > 1 else "mae"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
torch.manual_seed(0)
if wl == "mae":
    m = A.mae_vit_base_patch16(norm_pix_loss=True).to("cuda"); B = 256
else:
    m = A.get_ImageNet_or_random_ViT(True, 3, False, False, False).to("cuda"); B = 64
x = torch.randn(B, 3, 224, 224, device="cuda"); y = torch.randint(0, 3, (B,), device="cuda")
seen = {}
for s in range(steps):
    torch.cuda.manual_seed(1234)
    for p in m.parameters():
        p.grad = None
    loss = m(x, 0.75)[0] if wl == "mae" else A.supervised_loss(m(x), y)
    loss.backward()
    torch.cuda.synchronize()
    h = hashlib.sha256()
    bad = []
    for n, p in m.named_parameters():
        if p.grad is not None:
            b = p.grad.detach().cpu().numpy().tobytes()
            h.update(b)
            d = hashlib.sha256(b).hexdigest()
            if s and seen.get(n) != d:
                bad.append(n)
            seen.setdefault(n, d)
    print(s, h.hexdigest()[:16], float(loss.detach()), "DIFF: " + ",".join(bad[:8]) if bad else "", flush=True)
