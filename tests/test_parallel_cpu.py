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
        # the order in which the engine finishes gradients (models._MaeFn.backward)
        for i in reversed(range(2)):
            sync.block_done("decoder_blocks.", i)
        for i in reversed(range(3)):
            sync.block_done("blocks.", i)
        sync.backward_done(in_backward=False)
        sync.wait()
        ok = all(torch.equal(f.G[r], want[r]) for r in ("vec", "mat"))
        launches = [hi - lo for _, lo, hi in sync.launched]
        # every matrix is trainable; the frozen positional tables sit between trainable vectors and travel with them
        covered = sum(launches[:-1]) == f.G["mat"].numel() and launches[-1] == f.G["vec"].numel() and \
            [r for r, _, _ in sync.launched][-1] == "vec"
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


def _freeze_worker(rank, world, port, q):
    """Staged fine-tune schedule (finetune.py:49-126 via tc.py:924-953): none -> head+1 -> full.  The synchroniser must
    follow the requires_grad flags: only trainable ranges are reduced, frozen ranges are left alone, the plan is rebuilt
    exactly when the set changes."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ssl4polyp_amd as A
        from ssl4polyp_amd.flat import FlatParams
        from ssl4polyp_amd.parallel import GradSync
        torch.manual_seed(0)
        m = A.ViT_from_MAE(None, True, 2, False, None, embed_dim=64, depth=3, num_heads=2, out_token="cls")
        rt = m._rt
        rt.flat = FlatParams(m, torch.bfloat16)
        rt.flat.materialize(torch.device("cpu"))
        f = rt.flat
        sync = GradSync(rt, None, bucket_mb=0.05, merge_gap=0)
        out = []

        def set_mode(mode):  # what configure_finetune_parameters does to the flags
            for n, p in m.named_parameters():
                if n in ("pos_embed", "decoder_pos_embed"):
                    continue  # fixed sincos tables: never trainable (models_mae.py:37,51)
                p.requires_grad_(mode == "full" or n.startswith("lin_head") or (mode == "head+1" and n.startswith("blocks.2.")))

        for step, mode in enumerate(["none", "none", "head+1", "full", "full"]):
            set_mode(mode)
            g = torch.Generator().manual_seed(1000 * step + rank)
            for r in ("vec", "mat"):
                f.G[r].copy_(torch.randn(f.G[r].shape, generator=g))
            mine = {r: f.G[r].clone() for r in ("vec", "mat")}
            want = {}
            for r in ("vec", "mat"):
                t = f.G[r].clone()
                dist.all_reduce(t)
                want[r] = t
            for i in reversed(range(3)):
                sync.block_done("blocks.", i)
            sync.backward_done(in_backward=False)
            sync.wait()
            good = True
            for i, p in enumerate(f.params):
                r, lo, n = f.region[i], f.offset[i], f.numel[i]
                ref = want[r] if p.requires_grad else mine[r]  # reduced iff trainable, else untouched
                good &= torch.equal(f.G[r][lo:lo + n], ref[lo:lo + n])
            out.append((mode, good, 4 * sum(hi - lo for _, lo, hi in sync.launched), sync.trainable_bytes(), sync.plan_builds))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_freeze_aware_gradient_sync_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_freeze_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        modes = [o[0] for o in out]
        assert modes == ["none", "none", "head+1", "full", "full"]
        for mode, good, sent, trainable, builds in out:
            assert good, f"rank {rank} {mode}: reduced / untouched ranges wrong"
            assert sent == trainable, f"rank {rank} {mode}: sent {sent} B for {trainable} B of trainable gradients"
        assert [o[4] for o in out] == [1, 1, 2, 3, 3]  # plan rebuilt only at the transitions
        # linear probe: lin_head.weight (2 x 64 -> one 128-element segment) + lin_head.bias (one 64-element aligned segment)
        assert out[0][2] == 4 * (128 + 64)
        assert out[0][2] < out[2][2] < out[3][2]


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
        n_list = len(seen)
        # a real DataLoader: the rank-local batch sampler must keep the dataset from ever producing foreign items
        class Items(torch.utils.data.Dataset):
            def __init__(self):
                self.touched = []
            def __len__(self):
                return 23
            def __getitem__(self, i):
                self.touched.append(i)
                gi = torch.Generator().manual_seed(i)
                return torch.randn(3, 2, 2, generator=gi), torch.tensor(i)
        ds = Items()
        dl = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, num_workers=0)
        lg2, tg2 = T.evaluate_cls(model, dl, "cpu")
        q.put((rank, lg.numpy(), tg.numpy(), pr.numpy(), n_list, sorted(ds.touched), tg2.numpy(), lg2.numpy()))
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
        rl, rt, rp, n_fwd, touched, tg2, lg2 = res[rank]
        assert n_fwd == (3 if rank == 0 else 2)  # batches 0,2,4 / 1,3
        # DataLoader path: 23 items in batches of 4 -> rank 0 decodes batches 0,2,4 (items 0-3, 8-11, 16-19), rank 1 the
        # others; nobody touches a foreign item, and both ranks hold all 23 results in dataset order
        mine = [i for i in range(23) if (i // 4) % 2 == rank]
        assert touched == mine, (rank, touched)
        assert tg2.tolist() == list(range(23)) and lg2.shape == (23, 2)
        assert torch.equal(torch.from_numpy(rl), lg) and torch.equal(torch.from_numpy(rt), tg)
        assert torch.equal(torch.from_numpy(rp), pr)
    # multi-class: softmax
    assert torch.allclose(T.class_probabilities(torch.tensor([[1.0, 2.0, 3.0]])).sum(), torch.tensor(1.0))


def test_deferred_pieces_wait_for_the_stream_their_block_finished_on():
    """ADVICE r3 (parallel.py): with a bucket larger than one block, block i's pieces are deferred and finally launched by a LATER
    block_done -- which the engine may issue from a different stream (the ungrouped tail block reports on `side`, block i+1's
    (proj, qkv) launch ran on `side2`).  GradSync records a ready-event per reported block and orders the launching stream behind
    every event it has not consumed yet.  CPU rehearsal with event doubles: every launch must first wait for ALL blocks reported
    since the previous launch, and a one-block bucket (nothing deferred) waits only for its own block."""
    import ssl4polyp_amd as A
    from ssl4polyp_amd.flat import FlatParams
    from ssl4polyp_amd.parallel import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        m = A.ViT_from_MAE(None, True, 2, False, None, embed_dim=64, depth=4, num_heads=2, out_token="cls")
        rt = m._rt
        rt.flat = FlatParams(m, torch.bfloat16)
        rt.flat.materialize(torch.device("cpu"))
        log = []

        class Ev:
            def __init__(self, block, stream):
                self.block, self.stream = block, stream

            def wait_on(self, cur):
                log.append(("wait", self.block, self.stream))

        for bucket_mb, expect_deferred in ((64.0, True), (0.01, False)):
            sync = GradSync(rt, None, bucket_mb=bucket_mb, force=True)
            cur = {}
            sync._record_ready = lambda: Ev(cur["block"], cur["stream"])
            orig_launch = sync._launch
            sync._launch = lambda r, lo, hi: (log.append(("launch", r, lo, hi)), orig_launch(r, lo, hi))[1]
            log.clear()
            # blocks 3, 2 report on "side2", the (ungrouped) tail blocks 1, 0 on "side" -- the engine's pattern
            for i, st in ((3, "side2"), (2, "side2"), (1, "side"), (0, "side")):
                cur.update(block=i, stream=st)
                sync.block_done("blocks.", i)
            sync.backward_done(in_backward=False)
            sync.wait()
            launches = [j for j, e in enumerate(log) if e[0] == "launch" and e[1] == "mat"]
            assert launches, log
            seen = set()
            for j in launches:      # every mat launch is preceded (since the previous launch) by waits for the blocks it carries
                k = j - 1
                while k >= 0 and log[k][0] == "wait":
                    seen.add(log[k][1])
                    k -= 1
            assert seen == {0, 1, 2, 3}, (bucket_mb, log)
            first = launches[0]
            waited_first = [e[1] for e in log[:first] if e[0] == "wait"]
            if expect_deferred:     # one launch at the end carries everything: it waited for the side2 blocks too
                assert sorted(waited_first) == [0, 1, 2, 3] and sync.waited_events == 4
            else:                   # a bucket per block: the first launch waited for block 3 alone
                assert waited_first == [3]
    finally:
        dist.destroy_process_group()
