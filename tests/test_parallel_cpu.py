"""world_size-2 gloo rehearsal (CPU) of the N>1 path: the bucketed gradient all-reduce schedule of
parallel.GradSync over the flat gradient range, driven the way the engine drives it during backward."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bucket_mb, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ssl4polyp_amd as A
        from ssl4polyp_amd.flat import FlatParams
        from ssl4polyp_amd.parallel import GradSync
        torch.manual_seed(0)
        m = A.MaskedAutoencoderViT(img_size=32, patch_size=8, embed_dim=64, depth=3, num_heads=2, decoder_embed_dim=32,
                                   decoder_depth=2, decoder_num_heads=1)
        rt = m._rt
        rt.flat = FlatParams(m, torch.bfloat16)
        rt.flat.materialize(torch.device("cpu"))
        f = rt.flat
        g = torch.Generator().manual_seed(100 + rank)
        for r in ("vec", "mat"):
            f.G[r].copy_(torch.randn(f.G[r].shape, generator=g))
        want = {}
        for r in ("vec", "mat"):
            t = f.G[r].clone()
            dist.all_reduce(t)
            want[r] = t
        sync = GradSync(rt, None, bucket_mb=bucket_mb)
        launches = []
        orig = sync._launch
        sync._launch = lambda t: (launches.append(t.numel()), orig(t))[1]
        # the order in which the engine finishes gradients (models._MaeFn.backward)
        for i in reversed(range(2)):
            sync.block_done("decoder_blocks.", i)
        for i in reversed(range(3)):
            sync.block_done("blocks.", i)
        sync.backward_done(in_backward=False)
        sync.wait()
        ok = all(torch.equal(f.G[r], want[r]) for r in ("vec", "mat"))
        covered = sum(launches[:-1]) == f.G["mat"].numel() and launches[-1] == f.G["vec"].numel()
        # no_sync-style disable: nothing is launched
        sync.enabled = False
        before = f.G["mat"].clone()
        sync.block_done("blocks.", 0)
        sync.backward_done(in_backward=False)
        untouched = torch.equal(before, f.G["mat"])
        q.put((rank, ok, covered, untouched, len(launches)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb,min_launches", [(64.0, 2), (0.05, 4)])
def test_bucketed_allreduce_schedule_gloo(bucket_mb, min_launches):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_mb, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, covered, untouched, n in res:
        assert ok, f"rank {rank}: reduced gradients differ from a plain all_reduce"
        assert covered, f"rank {rank}: buckets do not tile the flat gradient range exactly once"
        assert untouched
        assert n >= min_launches


def _eval_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ssl4polyp_amd import train as T
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(12, 2))  # same weights on every rank (seeded)
        g = torch.Generator().manual_seed(5)
        loader = [(torch.randn(3 + (i % 2), 3, 2, 2, generator=g), torch.arange(3 + (i % 2)) + 100 * i) for i in range(5)]
        seen = []
        fwd = model.forward
        model.forward = lambda x: (seen.append(x.shape[0]), fwd(x))[1]
        lg, tg, pr = T.evaluate_cls(model, loader, "cpu", return_probs=True)
        q.put((rank, lg.numpy(), tg.numpy(), pr.numpy(), len(seen)))
    finally:
        dist.destroy_process_group()


def test_sharded_evaluation_gloo():
    """evaluate_cls under world 2: each rank runs only its share of the batches, all ranks return the full result in
    loader order, equal to the single-process evaluation."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r = q.get(timeout=120)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from ssl4polyp_amd import train as T
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(12, 2))
    g = torch.Generator().manual_seed(5)
    loader = [(torch.randn(3 + (i % 2), 3, 2, 2, generator=g), torch.arange(3 + (i % 2)) + 100 * i) for i in range(5)]
    lg, tg, pr = T.evaluate_cls(model, loader, "cpu", return_probs=True)
    assert torch.allclose(pr, torch.sigmoid(lg[:, 1] - lg[:, 0]))
    for rank in (0, 1):
        rl, rt, rp, n_fwd = res[rank]
        assert n_fwd == (3 if rank == 0 else 2)  # batches 0,2,4 / 1,3
        assert torch.equal(torch.from_numpy(rl), lg) and torch.equal(torch.from_numpy(rt), tg)
        assert torch.equal(torch.from_numpy(rp), pr)
    # multi-class: softmax
    assert torch.allclose(T.class_probabilities(torch.tensor([[1.0, 2.0, 3.0]])).sum(), torch.tensor(1.0))
