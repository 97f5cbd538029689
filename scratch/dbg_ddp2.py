import os, sys, socket
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch, torch.distributed as dist, torch.multiprocessing as mp, numpy as np
from test_gpu_parallel import _model, _batch
def run(model, ddp, opt, imgs, labels, steps):
    out = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        z = ddp(imgs); loss = torch.nn.functional.binary_cross_entropy_with_logits(z[:,1]-z[:,0], labels); loss.backward()
        g = {n: p.grad.detach().float().cpu().numpy().copy() for n,p in model.named_parameters() if p.grad is not None}
        opt.step(); torch.cuda.synchronize()
        out.append((g, {n: p.detach().float().cpu().numpy().copy() for n,p in model.named_parameters()}, opt._hyper.cpu().numpy().copy()))
    return out
def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssl4polyp_amd.parallel import DataParallel
    from ssl4polyp_amd.optim import FusedAdamW
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    model = _model("fp32"); ddp = DataParallel(model, dev, bucket_mb=0.05)
    mode = os.environ.get("DBG_MODE", "")
    if mode == "presync":
        orig = ddp.sync._launch
        def patched(t):
            torch.cuda.current_stream().synchronize(); return orig(t)
        ddp.sync._launch = patched
    if mode == "postsync":
        origw = ddp.sync.wait
        def pw():
            origw(); torch.cuda.synchronize()
        ddp.sync.wait = pw
    opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05); opt.grad_sync, opt.grad_scale = ddp.sync, 0.5
    imgs, labels = _batch(rank)
    q.put((rank, run(model, ddp, opt, imgs.to(dev), labels.to(dev), 2))); dist.destroy_process_group()
if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue(); s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ps = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]; res = {}
    for _ in ps:
        r, o = q.get(timeout=300); res[r] = o
    [p.join() for p in ps]
    from ssl4polyp_amd.parallel import DataParallel
    from ssl4polyp_amd.optim import FusedAdamW
    dev = torch.device("cuda", 0); model = _model("fp32"); ddp = DataParallel(model, dev)
    opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05)
    imgs, labels = _batch("all"); single = run(model, ddp, opt, imgs.to(dev), labels.to(dev), 2)
    e2 = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
    bad = sorted(((e2(res[0][0][1][n], single[0][1][n]), n) for n in single[0][1]), reverse=True)[:8]
    print("worst params after step 0:", bad)
    badg = sorted(((e2(res[0][0][0][n]*0.5, single[0][0][n]), n) for n in single[0][0]), reverse=True)[:5]
    print("worst grads at step 0:", badg)
    for step in range(2):
        print("step", step, "hyper ddp", res[0][step][2][0][:9], " single", single[step][2][0][:9])
        for n in ["lin_head.weight", "blocks.2.mlp.fc2.bias", "blocks.0.norm1.bias", "cls_token"]:
            e = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
            if step == 1:
                print(f"   {n:24s} ddp1*0.5 vs single0 {e(res[0][1][0][n]*0.5, single[0][0][n]):.2e}  ddp1 vs ddp0 {e(res[0][1][0][n], res[0][0][0][n]):.2e}  ddp1*0.25 vs single1 {e(res[0][1][0][n]*0.25, single[1][0][n]):.2e}  single1 vs single0 {e(single[1][0][n], single[0][0][n]):.2e}")
            print(f"   {n:24s} grad(x0.5) err {e(res[0][step][0][n]*0.5, single[step][0][n]):.2e}   param err {e(res[0][step][1][n], single[step][1][n]):.2e}")
