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
    hide, so its size is the exposed tail;
  * only TRAINABLE ranges travel: the plan is the set of flat segments whose parameters have ``requires_grad=True``
    (merged across gaps below ``merge_gap`` elements -- a frozen positional table between two trainable vectors costs
    less on the wire than a second collective), rebuilt whenever the set changes.  The staged fine-tune schedule of the
    reference (finetune.py:49-126 applied by train_classification.py:924-953: none -> head+1 -> head+2 -> full)
    flips ``requires_grad`` between epochs; torch DDP fixes its reducer at construction, this one follows the flags:
    linear probe ("none") all-reduces lin_head only (6 KB instead of 343 MB).
Gradients are summed; the mean is folded into the fused AdamW (``grad_scale = 1/world``).
"""
from __future__ import annotations

import warnings
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn


class GradSync:
    def __init__(self, rt, process_group=None, bucket_mb: float = 24.0, merge_gap: int = 1 << 18, force: bool = False):
        self.rt = rt
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self.merge_gap = int(merge_gap)
        self.force = bool(force)  # run the collectives at world size 1 too (tests: exercises the RCCL stream semantics)
        self.enabled = True       # False inside no_sync() (gradient accumulation micro-steps)
        self._works: List = []
        self._hi = None           # everything in mat[_hi:] has been handed to RCCL
        self._armed = False
        self._plan_key = None
        self._ranges: Dict[str, List[Tuple[int, int]]] = {"mat": [], "vec": []}
        self.launched: List[Tuple[str, int, int]] = []  # (region, lo, hi) of the last backward's collectives (tests / logs)
        self.plan_builds = 0
        # "ready" events of blocks whose pieces were DEFERRED (bucket not full yet): the engine calls block_done(i) on the stream
        # behind which block i's gradients are final -- the weight-gradient side stream, or the second one of a two-launch block,
        # or the main stream for a frozen block -- and a later block_done may run on ANOTHER of those streams (the ungrouped tail
        # block issues on `side`, block i+1's (proj, qkv) launch ran on `side2`).  The collective that finally carries a deferred
        # piece is therefore ordered behind the event recorded when its block was reported, not just behind the issuing stream.
        # (With the default 24-MiB bucket every ViT-B block flushes in its own call and nothing is ever deferred; any bucket_mb
        # above one block's 28 MiB needs this.)
        self._ready_events: List = []
        self.waited_events = 0    # (tests: deferred ready-events a launch had to wait for)

    # ------------------------------------------------------------------------------------------
    @property
    def active(self) -> bool:
        return self.enabled and (self.world > 1 or self.force)

    def _plan(self):
        """Trainable flat segments per region, merged across small gaps; rebuilt when requires_grad flags change."""
        f = self.rt.flat
        key = (id(f.P.get("mat")), tuple(p.requires_grad for p in f.params))
        if key == self._plan_key:
            return
        A = f.ALIGN
        segs: Dict[str, List[Tuple[int, int]]] = {"mat": [], "vec": []}
        for i, p in enumerate(f.params):
            if p.requires_grad:
                segs[f.region[i]].append((f.offset[i], f.offset[i] + (f.numel[i] + A - 1) // A * A))
        for r, lst in segs.items():
            lst.sort()
            merged: List[Tuple[int, int]] = []
            for lo, hi in lst:
                if merged and lo - merged[-1][1] <= self.merge_gap:
                    merged[-1] = (merged[-1][0], max(hi, merged[-1][1]))
                else:
                    merged.append((lo, hi))
            self._ranges[r] = merged
        self._plan_key = key
        self.plan_builds += 1

    def trainable_bytes(self) -> int:
        self._plan()
        return 4 * sum(hi - lo for r in self._ranges.values() for lo, hi in r)

    def _begin(self):
        f = self.rt.flat
        self._plan()
        self._hi = f.G["mat"].numel()
        self._works = []
        self.launched = []
        self._ready_events = []
        self._armed = True

    def _record_ready(self):
        """An event on the CURRENT stream (gradients of the block just reported are final behind it), or None on the CPU."""
        if not self.rt.flat.G["mat"].is_cuda:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.rt.flat.G["mat"].device))
        return ev

    def _wait_ready(self):
        """Order the current stream behind every block reported since the last launch (their pieces ride in this one)."""
        cur = torch.cuda.current_stream(self.rt.flat.G["mat"].device) if self.rt.flat.G["mat"].is_cuda else None
        for ev in self._ready_events:
            if ev is not None:
                if hasattr(ev, "wait_on"):
                    ev.wait_on(cur)        # (test double)
                else:
                    cur.wait_event(ev)
                self.waited_events += 1
        self._ready_events = []

    def _launch(self, region: str, lo: int, hi: int):
        tensor = self.rt.flat.G[region][lo:hi]
        self.launched.append((region, lo, hi))
        if not tensor.is_cuda:  # gloo rehearsal of the bucket schedule on CPU (tests)
            self._works.append(dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            return
        # Issued from the weight-gradient stream the engine calls us on (the gradients are final in ITS order): the process
        # group's own RCCL stream waits for that stream's work so far, the collective runs there, and wait() orders the
        # optimizer's stream behind it.  No stream of our own in between: every extra stream is one more candidate for
        # sharing a hardware queue with the main stream (engine.reserve_streams).
        self._works.append(dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _pending(self, lo: int, hi: int) -> List[Tuple[int, int]]:
        """Trainable pieces of mat[lo:hi), back to front."""
        out = []
        for a, b in reversed(self._ranges["mat"]):
            a2, b2 = max(a, lo), min(b, hi)
            if a2 < b2:
                out.append((a2, b2))
        return out

    def _ready_down_to(self, lo: int, force: bool):
        """mat[lo:] is final.  Hand the trainable part of [lo, _hi) to RCCL if it fills a bucket (or `force`)."""
        if not self._armed:
            self._begin()
        if lo >= self._hi:
            return
        pieces = self._pending(lo, self._hi)
        if not pieces:
            self._hi = lo
            return
        if force or sum(b - a for a, b in pieces) >= self.bucket_elems:
            self._wait_ready()
            for a, b in pieces:
                self._launch("mat", a, b)
            self._hi = lo

    # -- called by the engine ----------------------------------------------------------------------
    def block_done(self, prefix: str, i: int):
        if not self.active:
            return
        f = self.rt.flat
        if not self._armed:
            self._begin()
        self._ready_events.append(self._record_ready())
        self._ready_down_to(f.offset[f.index[f"{prefix}{i}.attn.qkv.weight"]], force=False)

    def backward_done(self, in_backward: bool = True):
        if not self.active:
            return
        self._ready_down_to(0, force=True)
        for a, b in self._ranges["vec"]:
            self._launch("vec", a, b)
        self._armed = False
        if in_backward:
            # make the compute stream wait for RCCL at the end of this backward pass (before any optimizer)
            torch.autograd.Variable._execution_engine.queue_callback(self.wait)

    def wait(self):
        """Order the CURRENT stream after every collective of the last backward (RCCL: a stream-side event wait, no host
        block; gloo: blocks the host).  Idempotent: the autograd end-of-backward callback and the optimizer both call it,
        whichever stream / thread they run on; the list is dropped when the next backward arms the synchroniser."""
        for w in self._works:
            w.wait()


class DataParallel(nn.Module):
    """Thin wrapper: broadcast parameters from rank 0, attach the gradient synchroniser to the module's runtime."""

    def __init__(self, module: nn.Module, device: torch.device, process_group=None, bucket_mb: float = 24.0,
                 force_sync: bool = False):
        super().__init__()
        self.module = module
        module.to(device)
        rt = module._rt
        rt.ensure(device)
        if dist.is_initialized() and (dist.get_world_size(process_group) > 1 or force_sync):
            from . import engine
            dev = torch.device(device)
            if dev.type == "cuda" and not any(k[1] == "side" for k in engine._STREAMS):
                warnings.warn("ssl4polyp_amd.reserve_streams(device) was not called before torch.distributed initialised: the "
                              "engine's side streams may share a hardware queue with RCCL's or with the main stream and then "
                              "run one after the other (up to 1.4x step time measured).  Call it right after "
                              "torch.cuda.set_device().", stacklevel=2)
            for r in ("vec", "mat"):
                dist.broadcast(rt.flat.P[r], src=0, group=process_group)
            rt.flat._shadow_versions = None  # force a shadow refresh from the broadcast weights
            self.sync = GradSync(rt, process_group, bucket_mb, force=force_sync)
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
