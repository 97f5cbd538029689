import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parallel import _model, _batch
def grads(model, ddp, imgs, labels):
    z = ddp(imgs); loss = torch.nn.functional.binary_cross_entropy_with_logits(z[:,1]-z[:,0], labels); loss.backward(); torch.cuda.synchronize()
    return {n: p.grad.detach().float().cpu().numpy() for n,p in model.named_parameters() if p.grad is not None}, loss.item()
def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssl4polyp_amd.parallel import DataParallel
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    model = _model("fp32"); ddp = DataParallel(model, dev, bucket_mb=0.05)
    imgs, labels = _batch(rank)
    g, l = grads(model, ddp, imgs.to(dev), labels.to(dev))
    # also local-only grads (sync disabled)
    model.zero_grad(set_to_none=True); ddp.sync.enabled = False
    gl, _ = grads(model, ddp, imgs.to(dev), labels.to(dev))
    q.put((rank, g, gl, l)); dist.destroy_process_group()
if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue(); s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ps = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]; res = {}
    for _ in ps:
        r, g, gl, l = q.get(timeout=300); res[r] = (g, gl, l)
    [p.join() for p in ps]
    from ssl4polyp_amd.parallel import DataParallel
    dev = torch.device("cuda", 0); model = _model("fp32"); ddp = DataParallel(model, dev)
    imgs, labels = _batch("all"); gs, ls = grads(model, ddp, imgs.to(dev), labels.to(dev))
    import numpy as np
    print("losses", res[0][2], res[1][2], "single", ls, "mean", (res[0][2]+res[1][2])/2)
    for n in ["lin_head.weight", "blocks.2.mlp.fc2.weight", "blocks.2.mlp.fc2.bias", "blocks.0.attn.qkv.weight", "blocks.0.norm1.bias", "cls_token", "patch_embed.proj.weight"]:
        a = res[0][0][n] * 0.5; loc = (res[0][1][n] + res[1][1][n]) * 0.5; w = gs[n]
        e = lambda x: np.linalg.norm(x - w) / np.linalg.norm(w)
        print(f"{n:28s} synced*0.5 vs single {e(a):.2e}   mean(local0,local1) vs single {e(loc):.2e}   rank0==rank1 {np.array_equal(res[0][0][n], res[1][0][n])}")
