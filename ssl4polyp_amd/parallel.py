"""Data-parallel replication: one process per MI355X, gradients all-reduced with RCCL over xGMI.

Replaces the reference's ``torch.nn.parallel.DistributedDataParallel`` wrap (mae/main_pretrain.py:212-214,
train_classification.py:5746-5750) and its per-step scalar collectives / barriers
(train_classification.py:4548-4550,4631-4632; engine_pretrain.py:81).  The engine writes every gradient
into one flat f32 range in forward order and finishes it back-to-front during backward, so:

  * a bucket is a contiguous slice of that range -- no packing / unpacking copies;
  * buckets are launched from the END of the range as soon as the blocks they cover are done
    (``block_done`` is called by the engine after each block's wgrad GEMMs are enqueued);
  * each ``all_reduce`` is issued on a side HIP stream fenced by an event recorded on the compute
    stream, so RCCL runs under the remaining backward; the compute stream only waits (stream-side, no host
    block) right before the optimizer reads the gradients;
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are large but no larger than one transformer block
    (default 24 MiB < the 28 MiB of a ViT-B block's matrices, so a bucket closes at every block boundary: 12-13
    collectives per step, >= 3.5 MiB per link per ring step) -- the last bucket is the only one the backward cannot
    hide, so its size is the exposed tail.
Gradients are summed; the mean is folded into the fused AdamW (``grad_scale = 1/world``).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


class GradSync:
    def __init__(self, rt, process_group=None, bucket_mb: float = 24.0):
        self.rt = rt
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self.enabled = True  # False inside no_sync() (gradient accumulation micro-steps)
        self.comm_stream: Optional[torch.cuda.Stream] = None
        self._works: List = []
        self._hi = None       # everything in mat[_hi:] has been handed to RCCL
        self._armed = False

    # ------------------------------------------------------------------------------------------
    def _begin(self):
        f = self.rt.flat
        if self.comm_stream is None and f.device.type == "cuda":
            self.comm_stream = torch.cuda.Stream(device=f.device)
        self._hi = f.G["mat"].numel()
        self._works = []
        self._armed = True

    def _launch(self, tensor):
        if not tensor.is_cuda:  # gloo rehearsal of the bucket schedule on CPU (tests)
            self._works.append(dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.comm_stream.wait_event(ev)
        with torch.cuda.stream(self.comm_stream):
            self._works.append(dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _ready_down_to(self, lo: int, force: bool):
        """mat[lo:] is final.  Hand [lo, _hi) to RCCL if it is a full bucket (or `force`)."""
        if not self._armed:
            self._begin()
        if lo < self._hi and (force or self._hi - lo >= self.bucket_elems):
            self._launch(self.rt.flat.G["mat"][lo:self._hi])
            self._hi = lo

    # -- called by the engine ----------------------------------------------------------------------
    def block_done(self, prefix: str, i: int):
        if not self.enabled or self.world == 1:
            return
        f = self.rt.flat
        self._ready_down_to(f.offset[f.index[f"{prefix}{i}.attn.qkv.weight"]], force=False)

    def backward_done(self, in_backward: bool = True):
        if not self.enabled or self.world == 1:
            return
        self._ready_down_to(0, force=True)
        self._launch(self.rt.flat.G["vec"])
        self._armed = False
        if in_backward:
            # make the compute stream wait for RCCL at the end of this backward pass (before any optimizer)
            torch.autograd.Variable._execution_engine.queue_callback(self.wait)

    def wait(self):
        for w in self._works:
            w.wait()  # stream-side wait on the current (compute) stream
        self._works = []


class DataParallel(nn.Module):
    """Thin wrapper: broadcast parameters from rank 0, attach the gradient synchroniser to the module's runtime."""

    def __init__(self, module: nn.Module, device: torch.device, process_group=None, bucket_mb: float = 24.0):
        super().__init__()
        self.module = module
        module.to(device)
        rt = module._rt
        rt.ensure(device)
        if dist.is_initialized() and dist.get_world_size(process_group) > 1:
            for r in ("vec", "mat"):
                dist.broadcast(rt.flat.P[r], src=0, group=process_group)
            rt.flat._shadow_versions = None  # force a shadow refresh from the broadcast weights
            self.sync = GradSync(rt, process_group, bucket_mb)
            rt.grad_sync = self.sync
        else:
            self.sync = None

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    class _NoSync:
        def __init__(self, sync):
            self.sync = sync

        def __enter__(self):
            if self.sync is not None:
                self.sync.enabled = False

        def __exit__(self, *a):
            if self.sync is not None:
                self.sync.enabled = True

    def no_sync(self):
        """Skip the all-reduce for gradient-accumulation micro-steps (the reference all-reduces on every
        micro-step, engine_pretrain.py:64-72 -- SURVEY appendix A; this is the cheaper equivalent)."""
        return DataParallel._NoSync(self.sync)
