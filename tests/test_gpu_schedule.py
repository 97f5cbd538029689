"""Race check of the asynchronous schedule (three streams: main chain, second forward chain, weight-gradient / AdamW stream,
all fenced by events): the same seeded ViT-B/16 training run is repeated in a process where the HIP runtime serialises every
kernel and copy (AMD_SERIALIZE_KERNEL=3, AMD_SERIALIZE_COPY=3 -- must be set before the runtime initialises, hence the child
processes).  With every fence in place the two runs produce the same bits; a missing fence shows as a difference, because the
serialised run cannot race."""
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
import bench
dev = torch.device("cuda", 0)
wl = sys.argv[3]
bs = 64 if wl == "cls" else 128
model, ddp, opt = bench.build(wl, "bf16", dev, 1, bs)
imgs, labels = bench.make_batch(wl, bs, dev, 0)
if wl == "cls":
    step = bench.make_step(wl, ddp, opt, imgs, labels)
else:
    noise = torch.rand(bs, 196, device=dev, generator=torch.Generator(device=dev).manual_seed(77))
    def step():
        opt.zero_grad(set_to_none=True)
        loss, _, _ = model(imgs, 0.75, noise=noise)
        loss.backward(); opt.step(); return loss
losses = [float(step().detach()) for _ in range(int(sys.argv[4]))]
torch.cuda.synchronize()
torch.save({"losses": losses, "sd": {k: v.float().cpu() for k, v in model.state_dict().items()}}, sys.argv[2])
"""


def _run(tmp_path, name, workload, steps, extra_env):
    out = tmp_path / f"{name}.pt"
    env = dict(os.environ)
    env.update(extra_env)
    subprocess.run([sys.executable, "-c", CHILD, REPO, str(out), workload, str(steps)], check=True, env=env, timeout=600)
    return torch.load(out)


@pytest.mark.gpu
@pytest.mark.parametrize("workload,steps", [("cls", 6), ("mae", 4)])
def test_async_schedule_equals_serialised_runtime(tmp_path, workload, steps):
    a = _run(tmp_path, "async", workload, steps, {})
    b = _run(tmp_path, "serial", workload, steps, {"AMD_SERIALIZE_KERNEL": "3", "AMD_SERIALIZE_COPY": "3"})
    assert a["losses"] == b["losses"]
    assert a["sd"].keys() == b["sd"].keys()
    for k in a["sd"]:
        assert torch.equal(a["sd"][k], b["sd"][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("workload,steps", [("cls", 4), ("mae", 3)])
def test_auto_vmcnt_build_equals_shipped_library(tmp_path, workload, steps):
    """The PM_LDS_IMAGE assumption as a gate (pm_common.h): the shipped k-loops read MFMA fragments with ds_read_b64_tr_b16 through a
    `__restrict__` LDS pointer, which keeps this compiler from guarding each read with `s_waitcnt vmcnt(0)` while LDS-DMA loads are
    in flight; the hand-placed counted waits + barriers are then the only synchronisation.  `-DPM_AUTO_VMCNT` (built by
    __graft_entry__.build() into a side library) restores the compiler's guard -- slower, but correct whatever the waitcnt pass
    does.  The same seeded ViT-B steps through both libraries must give the same bits; the compiler that built them is on record."""
    import json
    import __graft_entry__ as G
    alt = G.alt_lib_path("autovmcnt")
    assert os.path.exists(alt), f"{alt} missing: __graft_entry__.build() makes it"
    info = json.load(open(G.BUILD_INFO))
    assert "autovmcnt" in info["alt_libs"] and info["hipcc"] not in ("", "unknown"), info
    here = G.hipcc_version()
    if here != "unknown":
        assert here == info["hipcc"], f"libraries built by `{info['hipcc']}`, this box has `{here}`: rebuild and re-run the gate"
    a = _run(tmp_path, "shipped", workload, steps, {})
    b = _run(tmp_path, "autovmcnt", workload, steps, {"POLYPMAE_LIB": alt})
    assert a["losses"] == b["losses"]
    for k in a["sd"]:
        assert torch.equal(a["sd"][k], b["sd"][k]), k


@pytest.mark.gpu
def test_side_streams_are_shared_by_every_model_of_the_process():
    """One set of side streams per device (engine._shared_stream): the HIP runtime multiplexes streams onto a few hardware queues, and
    a set per model put a later model's weight-gradient stream onto the main stream's queue (the MAE step of the multi-model bench
    line lost 11 %).  reserve_streams() creates them up front and is idempotent."""
    import ssl4polyp_amd as A
    from ssl4polyp_amd import engine
    DEV = torch.device("cuda", 0)
    A.reserve_streams(DEV)
    first = {role: engine._shared_stream(DEV, role) for role in ("aux0", "side")}
    A.reserve_streams(DEV)
    k1, k2 = engine.Kernels("bf16"), engine.Kernels("fp32")
    for k in (k1, k2):
        assert k.aux_stream(DEV, 0) is first["aux0"] and k.side_stream(DEV) is first["side"]
        # the second weight-gradient stream (backward only) is the second forward chain's stream (forward only): main + two
        # side streams are busy at any time, one hardware queue each, one left for RCCL
        assert k.side_stream2(DEV) is (first["aux0"] if engine.MERGE_AUX_SIDE2 else engine._shared_stream(DEV, "side2"))
    assert len({s.cuda_stream for s in first.values()} | {torch.cuda.current_stream().cuda_stream}) == 3
