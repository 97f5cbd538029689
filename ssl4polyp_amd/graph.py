"""hipGraph capture of a whole training step (zero_grad -> forward -> loss -> backward -> fused AdamW).

A ViT-B/16 step is ~330 kernel launches of 10-150 us each; launched eagerly from Python the host needs
~15 ms per step, about what the GPU needs, so any host jitter shows up as idle GPU time (and as
stragglers under data parallelism).  Capturing the step once and replaying it removes the host from the loop;
the reference's per-step host syncs (loss.item(), per-parameter grad-norm .item()s, torch.cuda.synchronize():
engine_pretrain.py:55,74; tc.py:1437-1454,4551) have no counterpart here.

Everything the step reads must live at fixed addresses: inputs are copied into static tensors, the optimizer's
hyper-parameters live in device memory (optim.FusedAdamW / pm_adamw_dev), the MAE masking noise comes from
torch's graph-safe Philox generator.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


class GraphedStep:
    def __init__(self, step_fn: Callable[[], torch.Tensor], optimizer=None, warmup: int = 3):
        """`step_fn()` runs ONE full step on static input tensors and returns the (static) loss tensor."""
        self.optimizer = optimizer
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step_fn()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        if optimizer is not None:
            optimizer.sync_hyper(force=True)
        host_steps = [int(g.get("step", 0)) for g in optimizer.param_groups] if optimizer is not None else []
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):  # records the launches; nothing executes during capture
            self.loss = step_fn()
        if optimizer is not None:
            for g, s in zip(optimizer.param_groups, host_steps):
                g["step"] = s
        self.steps = 0

    def replay(self) -> torch.Tensor:
        opt = self.optimizer
        if opt is not None:
            opt.sync_hyper()  # picks up lr changes made by a scheduler since the last replay
        self.graph.replay()
        if opt is not None:
            for g in opt.param_groups:  # keep the host mirror of the step counter honest (state_dict)
                g["step"] = int(g.get("step", 0)) + 1
        self.steps += 1
        return self.loss
