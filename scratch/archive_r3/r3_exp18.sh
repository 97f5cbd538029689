#!/bin/bash
# round-3 experiment 18: a ViT-B block's weight gradients as TWO launches on two side streams -- (fc2, fc1) behind dfc2 and
# (proj, qkv) behind the attention backward (pm_vit_block_bwd.two_groups) -- so that dfc2 shares the chip with 36 workgroups
# instead of 108.  Also: bit-identity of the gradients against the single launch.
F="--no-parity --no-cpu-baseline --no-torch-baseline --no-mae --no-c5 --no-kernel-stats --steps 40"
sel='import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"], r["config"]["final_loss"])'
run() { echo -n "$2 $1: "; env $1 python bench.py $F --workload $2 2>/dev/null | python -c "$sel"; }
python - <<'PY' || exit 1
import os, subprocess, sys, json
code = r'''
import os, torch, hashlib
import ssl4polyp_amd as A
torch.manual_seed(0)
m = A.get_ImageNet_or_random_ViT(True, 3, False, False, False).to("cuda")
x = torch.randn(64, 3, 224, 224, device="cuda"); y = torch.randint(0, 3, (64,), device="cuda")
for _ in range(2):
    for p in m.parameters(): p.grad = None
    loss = A.supervised_loss(m(x), y); loss.backward()
torch.cuda.synchronize()
h = hashlib.sha256()
for n, p in m.named_parameters():
    if p.grad is not None: h.update(p.grad.detach().cpu().numpy().tobytes())
print(h.hexdigest(), float(loss))
'''
out = []
for v in ("0", "1"):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PM_TWO_GROUPS=v), capture_output=True, text=True)
    print("two_groups", v, r.stdout.strip(), r.stderr.strip()[-300:])
    out.append(r.stdout.strip())
sys.exit(0 if out[0] == out[1] and out[0] else 1)
PY
for i in 1 2; do
run "PM_TWO_GROUPS=0" cls
run "PM_TWO_GROUPS=1" cls
run "PM_TWO_GROUPS=0" mae
run "PM_TWO_GROUPS=1" mae
done
