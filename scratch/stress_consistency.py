"""Stress: N training steps of the full ViT-B/16 cls workload from the same seed under different scheduling switches;
the final parameters must agree to round-off (a cross-stream race would show as a large difference)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch, bench
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    wl = os.environ.get("WL", "cls"); bs = 64 if wl == "cls" else 128
    model, ddp, opt = bench.build(wl, "bf16", dev, 1, bs)
    imgs, labels = bench.make_batch(wl, bs, dev, 0)
    g = torch.Generator(device=dev).manual_seed(77)
    step = bench.make_step(wl, ddp, opt, imgs, labels)
    if wl == "mae":
        noise = torch.rand(bs, 196, device=dev, generator=g)
        def step():
            opt.zero_grad(set_to_none=True)
            loss, _, _ = model(imgs, 0.75, noise=noise)
            loss.backward(); opt.step(); return loss
    losses = []
    for it in range(int(os.environ.get("STEPS", 25))):
        losses.append(float(step().detach()))
    torch.cuda.synchronize()
    sd = {k: v.float().cpu() for k, v in model.state_dict().items()}
    torch.save({"losses": losses, "sd": sd}, sys.argv[2])
    sys.exit(0)
import torch
runs = {"base": {}, "base2": {}, "nosplit": {"PM_SPLIT_FWD": "1"}, "nooverlap": {"PM_OVERLAP_ADAMW": "0"},
        "wgrad256": {"PM_WGRAD_BLOCKS": "256"}, "nogroup": {"PM_GROUP_WGRAD": "0"}, "nosplitk": {"PM_GROUP_SPLIT": "0"},
        "tail0": {"PM_UNGROUP_TAIL": "0"}, "dgradpp": {"PM_DGRAD_PP": "1"}, "noblockcalls": {"PM_BLOCK_CALLS": "0"},
        # every kernel serialised by the runtime: if the asynchronous multi-stream schedule had a race, this run would differ
        "serialized": {"AMD_SERIALIZE_KERNEL": "3", "AMD_SERIALIZE_COPY": "3"}}
out = {}
for name, env in runs.items():
    path = f"/tmp/stress_{name}.pt"
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, __file__, "child", path], check=True, env=e)
    out[name] = torch.load(path)
ref = out["base"]
for name in runs:
    if name == "base": continue
    o = out[name]
    worst, wn = 0.0, ""
    for k in ref["sd"]:
        if k.endswith("attn.qkv.bias") or "pos_embed" in k: continue
        a, b = ref["sd"][k], o["sd"][k]
        d = ((a - b).norm() / a.norm().clamp_min(1e-30)).item()
        if d > worst: worst, wn = d, k
    print(f"{name:10s} final loss {o['losses'][-1]:.6f} (base {ref['losses'][-1]:.6f})  max |dloss| {max(abs(x-y) for x,y in zip(ref['losses'],o['losses'])):.2e}  worst param rel-L2 {worst:.2e} ({wn})")
