"""N>1 data-parallel path on real device memory: two ranks (both on cuda:0, gloo transport -- RCCL needs one GPU per
rank and the test box has one) drive the full engine + GradSync (side-stream bucket all-reduce behind event fences)
+ FusedAdamW, and must (i) stay bit-identical across ranks and (ii) reproduce a single-process run on the
concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(prec):
    import ssl4polyp_amd as A
    torch.manual_seed(7)
    return A.ViT_from_MAE(None, True, 2, False, None, embed_dim=64, depth=3, num_heads=2, out_token="cls", precision=prec)


def _batch(rank_or_all, n=4):
    g = torch.Generator().manual_seed(99)
    imgs = torch.randn(2 * n, 3, 224, 224, generator=g)
    labels = (torch.rand(2 * n, generator=g) < 0.5).float()
    if rank_or_all == "all":
        return imgs, labels
    return imgs[rank_or_all * n:(rank_or_all + 1) * n], labels[rank_or_all * n:(rank_or_all + 1) * n]


def _run_steps(model, ddp, opt, imgs, labels, steps, scale):
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        z = ddp(imgs)
        # per-rank mean loss; gradients are summed across ranks and scaled by 1/world in the optimizer
        loss = torch.nn.functional.binary_cross_entropy_with_logits(z[:, 1] - z[:, 0], labels)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    return {n: p.detach().float().cpu().clone() for n, p in model.named_parameters()}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ssl4polyp_amd.optim import FusedAdamW
        from ssl4polyp_amd.parallel import DataParallel
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        model = _model("fp32")
        if rank == 1:  # different initial weights on rank 1: the constructor's broadcast must fix that
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.01)
        ddp = DataParallel(model, dev, bucket_mb=0.05)  # small buckets -> several overlapped all-reduces per step
        assert ddp.sync is not None
        opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05)
        opt.grad_sync, opt.grad_scale = ddp.sync, 1.0 / world
        imgs, labels = _batch(rank)
        launches = []
        orig = ddp.sync._launch
        ddp.sync._launch = lambda r, lo, hi: (launches.append(hi - lo), orig(r, lo, hi))[1]
        params = _run_steps(model, ddp, opt, imgs.to(dev), labels.to(dev), 3, 1.0 / world)
        q.put((rank, {n: v.numpy() for n, v in params.items()}, len(launches)))  # by value (no fd passing)
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        rank, params, n_launch = q.get(timeout=300)
        res[rank] = ({n: torch.from_numpy(v) for n, v in params.items()}, n_launch)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] >= 3 * 3, "expected several bucket launches per step"
    for n in res[0][0]:
        assert torch.equal(res[0][0][n], res[1][0][n]), f"ranks diverged on {n}"
    # single process on the concatenated batch: mean over 8 samples == average of the two per-rank means
    from ssl4polyp_amd.optim import FusedAdamW
    from ssl4polyp_amd.parallel import DataParallel
    dev = torch.device("cuda", 0)
    model = _model("fp32")
    ddp = DataParallel(model, dev)
    assert ddp.sync is None
    opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05)
    imgs, labels = _batch("all")
    single = _run_steps(model, ddp, opt, imgs.to(dev), labels.to(dev), 3, 1.0)
    worst = 0.0
    for n, want in single.items():
        if n.endswith("attn.qkv.bias"):
            continue  # key-bias gradient is pure round-off noise, normalised by Adam (see test_gpu_models.py)
        got = res[0][0][n]
        err = ((got - want).norm() / want.norm().clamp_min(1e-30)).item()
        print(f"{n:40s} {err:.3e}")
        worst = max(worst, err)
    assert worst < 2e-5, worst


def _nccl_worker(port, q):
    """world size 1 on the RCCL backend with the synchroniser forced on: the collectives really run on the side stream
    (RCCL's Work.wait() is a stream-side wait, unlike gloo's host block) behind the autograd end-of-backward callback,
    through a staged freeze schedule (linear probe -> head+1 -> full), with the overlapped AdamW."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from ssl4polyp_amd.optim import FusedAdamW
        from ssl4polyp_amd.parallel import DataParallel
        import ssl4polyp_amd as A
        imgs, labels = _batch("all")
        imgs, labels = imgs.to(dev), labels.to(dev)

        def run(force):
            model = _model("bf16")
            ddp = DataParallel(model, dev, bucket_mb=0.05, force_sync=force)
            assert (ddp.sync is not None) == force
            opt = FusedAdamW(model, lr=1e-3, weight_decay=0.05, overlap_forward=True)
            if force:
                opt.grad_sync = ddp.sync
            sent = []
            for step in range(6):
                mode = ["none", "none", "head+1", "head+1", "full", "full"][step]
                for n, p in model.named_parameters():
                    if "pos_embed" not in n:
                        p.requires_grad_(mode == "full" or n.startswith("lin_head") or (mode == "head+1" and n.startswith("blocks.2.")))
                opt.zero_grad(set_to_none=True)
                loss = A.supervised_loss(ddp(imgs), labels.long(), pos_weight=1.0)
                loss.backward()
                opt.step()
                if force:
                    sent.append(4 * sum(hi - lo for _, lo, hi in ddp.sync.launched))
            torch.cuda.synchronize()
            return {n: p.detach().float().cpu().numpy() for n, p in model.named_parameters()}, sent, float(loss)

        a, sent, la = run(True)
        b, _, lb = run(False)
        same = all((a[n] == b[n]).all() for n in a)
        q.put((same, sent, la, lb))
    finally:
        dist.destroy_process_group()


def test_rccl_world1_forced_sync_matches_unsynchronised_run():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    same, sent, la, lb = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    # a one-rank SUM all-reduce is the identity: the forced-sync run must equal the plain run bit for bit -- any missing
    # fence between the RCCL stream, the wgrad side stream, the overlapped AdamW and the next forward would show here
    assert same and la == lb
    assert sent[0] == sent[1] == 4 * (128 + 64) and sent[1] < sent[2] == sent[3] < sent[4] == sent[5], sent


def _pm_comm_worker(q):
    """pm_comm_* (C-ABI over RCCL, include/polypmae.h) at world size 1 on cuda:0: id -> create -> bucketed in-place SUM on a side
    stream -> destroy.  A one-rank SUM is the identity; what is exercised is the binding (dlopen of RCCL, the handle, the group
    call, stream semantics)."""
    import ctypes
    try:
        from ssl4polyp_amd import _lib
        lib = _lib.load()
        torch.cuda.set_device(0)
        idb = ctypes.create_string_buffer(128)
        st = lib.pm_comm_unique_id(idb)
        if st != 0:
            q.put(("no-rccl", st))
            return
        comm = ctypes.c_void_p()
        _lib.check(lib.pm_comm_create(ctypes.byref(comm), idb, 0, 1), "pm_comm_create")
        rank, world = ctypes.c_int(-1), ctypes.c_int(-1)
        _lib.check(lib.pm_comm_world(comm, ctypes.byref(rank), ctypes.byref(world)), "pm_comm_world")
        g = torch.randn(1 << 20, device="cuda")
        want = g.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        lo = (ctypes.c_long * 3)(0, 4096, 600000)
        hi = (ctypes.c_long * 3)(4096, 600000, 1 << 20)
        _lib.check(lib.pm_comm_allreduce_f32(comm, g.data_ptr(), lo, hi, 3, side.cuda_stream), "pm_comm_allreduce_f32")
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        bad_args = lib.pm_comm_allreduce_f32(comm, g.data_ptr(), lo, hi, 0, side.cuda_stream)
        _lib.check(lib.pm_comm_destroy(comm), "pm_comm_destroy")
        q.put(("ok", bool(torch.equal(g, want)), rank.value, world.value, bad_args))
    except Exception as e:  # pragma: no cover
        q.put(("error", repr(e)))


def test_pm_comm_c_abi_world1():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_pm_comm_worker, args=(q,))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    if res[0] == "no-rccl":
        pytest.skip(f"no loadable RCCL on this box (status {res[1]})")
    assert res[0] == "ok", res
    _, same, rank, world, bad = res
    assert same and rank == 0 and world == 1 and bad != 0
