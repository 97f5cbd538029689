"""Training-loop counterparts of the reference's two hot loops, built on the MI355X engine.

  mae/engine_pretrain.py:22-100   train_one_epoch   -> train_one_epoch_mae
  classification/train_classification.py:4446-4650  train_epoch -> train_epoch_cls
  mae/util/lr_sched.py:9-21 / tc.py:3943-3971       LR schedules (same formulas)
  mae/util/misc.py:306-352 / tc.py:7036-7111        checkpoint dictionaries (same keys, torch.save format)

Same step order as the reference (zero_grad -> forward -> loss -> backward -> [grad stats] -> optimizer), but
nothing in the step reads a device value back to the host: the reference's per-step `loss.item()`,
`torch.cuda.synchronize()`, per-parameter `isnan().any()` / `float(norm)` and the per-step loss all-reduce +
barrier (engine_pretrain.py:55,67,74,81; tc.py:4536-4551,4631-4632) are replaced by device-side accumulators
that are read once per `log_every` steps (one tiny all-reduce there under data parallelism).
"""
from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass, field
from pathlib import Path
from typing import Callable, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------------------
# schedules
# ---------------------------------------------------------------------------------------------------
def mae_lr(epoch: float, lr: float, min_lr: float, warmup_epochs: float, epochs: float) -> float:
    """lr_sched.py:11-15 -- half-cycle cosine after a linear warm-up from 0; `epoch` is fractional."""
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))


def adjust_learning_rate(optimizer, epoch: float, args) -> float:
    """lr_sched.py:9-21 (honours per-group `lr_scale`)."""
    lr = mae_lr(epoch, args.lr, args.min_lr, args.warmup_epochs, args.epochs)
    for group in optimizer.param_groups:
        group["lr"] = lr * group["lr_scale"] if "lr_scale" in group else lr
    return lr


def cls_cosine_lambda(epoch: int, warmup_epochs: int, total_epochs: int) -> float:
    """tc.py:3952-3957 -- per-EPOCH cosine factor with (epoch+1)/warmup warm-up (use with LambdaLR)."""
    if warmup_epochs > 0 and epoch < warmup_epochs:
        return float(epoch + 1) / float(max(1, warmup_epochs))
    progress = (epoch - warmup_epochs) / float(max(1, total_epochs - warmup_epochs))
    progress = min(max(progress, 0.0), 1.0)
    return 0.5 * (1.0 + math.cos(math.pi * progress))


# ---------------------------------------------------------------------------------------------------
# fine-tune regimes (classification/finetune.py:49-91, applied by tc.py:5740 and, per stage, tc.py:924-953)
# ---------------------------------------------------------------------------------------------------
FINETUNE_MODES = ("none", "head+1", "head+2", "full")


def configure_finetune_parameters(model, mode: str) -> None:
    """finetune.py:49-91 on a model of this package (the reference's own function works on it unchanged -- it only touches
    `.parameters()`, `.lin_head`, `.head`, `.blocks`, `.frozen`; this restatement serves bench.py and the tests, which cannot
    import the reference on the GPU box).  "full": EVERY parameter trainable (the sincos pos_embed of the MAE-derived
    classifier included, see models._VitClsFn.backward); otherwise everything frozen except lin_head (and `.head` when it is
    a module with parameters) and the last 1 / 2 blocks.  Gradients are cleared; `.frozen` follows the mode."""
    import torch.nn as nn
    mode = str(mode).strip().lower()
    if mode not in FINETUNE_MODES:
        raise ValueError(f"Unsupported fine-tuning mode '{mode}'. Expected one of {sorted(FINETUNE_MODES)}.")
    model = _unwrap(model)
    for p in model.parameters():
        p.requires_grad_(mode == "full")
        p.grad = None
    if mode != "full":
        heads = [m for m in (getattr(model, "lin_head", None), getattr(model, "head", None)) if isinstance(m, nn.Module)]
        for h in heads:
            for p in h.parameters():
                p.requires_grad_(True)
        blocks = getattr(model, "blocks", None)
        tail = {"head+1": 1, "head+2": 2}.get(mode, 0)
        if tail and blocks is not None and len(blocks) > 0:
            for blk in list(blocks)[-tail:]:
                for p in blk.parameters():
                    p.requires_grad_(True)
    if hasattr(model, "frozen"):
        model.frozen = mode == "none"


# ---------------------------------------------------------------------------------------------------
# loss of the fine-tune path (tc.py:3347-3374, 6086-6104)
# ---------------------------------------------------------------------------------------------------
def supervised_loss(logits: torch.Tensor, targets: torch.Tensor, pos_weight: Optional[torch.Tensor] = None,
                    class_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One HIP launch for the loss and its gradient (pm_supervised_loss_fwd) instead of ~10 small ATen kernels; runs on
    the GPU only, like the rest of the product path."""
    from .models import supervised_loss as _hip_loss
    return _hip_loss(logits, targets, pos_weight, class_weights)


# ---------------------------------------------------------------------------------------------------
# epoch loops
# ---------------------------------------------------------------------------------------------------
@dataclass
class EpochStats:
    steps: int = 0
    samples: int = 0
    loss: float = float("nan")          # mean over the epoch (all ranks)
    grad_norm: float = float("nan")     # last logged global gradient norm
    grad_nan: int = 0
    grad_inf: int = 0
    seconds: float = 0.0
    lr: float = 0.0
    history: List[Dict[str, float]] = field(default_factory=list)

    @property
    def samples_per_sec(self) -> float:
        return self.samples / self.seconds if self.seconds > 0 else 0.0


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _flush(stats: EpochStats, acc: torch.Tensor, gstats: Optional[torch.Tensor], step: int, lr: float, printer):
    """One host read-back (+ one tiny all-reduce) per log interval: acc = [sum(loss), n_steps]."""
    buf = torch.cat([acc, gstats if gstats is not None else acc.new_zeros(3)])
    if _world() > 1:
        dist.all_reduce(buf)
    vals = buf.tolist()
    mean_loss = vals[0] / max(vals[1], 1.0)
    if not math.isfinite(mean_loss):
        raise FloatingPointError(f"Loss is {mean_loss}, stopping training")  # engine_pretrain.py:59-62
    rec = {"step": step, "loss": mean_loss, "lr": lr}
    if gstats is not None:
        w = _world()  # every rank holds the SUM of the per-rank gradients: |mean grad| = sqrt(sum_sq / w) / w after the reduce
        rec.update(grad_norm=math.sqrt(max(vals[2], 0.0) / w) / w, grad_nan=int(vals[3]), grad_inf=int(vals[4]))
        stats.grad_norm, stats.grad_nan, stats.grad_inf = rec["grad_norm"], stats.grad_nan + rec["grad_nan"], \
            stats.grad_inf + rec["grad_inf"]
    stats.history.append(rec)
    if printer is not None:
        printer(rec)
    return vals[0], vals[1]


def train_one_epoch_mae(model, data_loader: Iterable, optimizer, device, epoch: int, args, log_every: int = 20,
                        printer: Optional[Callable] = print, grad_stats: bool = True, loss_scaler=None) -> EpochStats:
    """engine_pretrain.py:22-100.  `model(samples, mask_ratio=...) -> (loss, pred, mask)`; `args` needs lr, min_lr,
    warmup_epochs, epochs, accum_iter, mask_ratio.  `model` may be a parallel.DataParallel wrapper.
    loss_scaler: optim.LossScaler for precision mode fp16 (engine_pretrain.py:65-72: loss_scaler.scale(loss).backward();
    loss_scaler.step(optimizer); loss_scaler.update()) -- None for bf16 / fp32."""
    scale = loss_scaler.scale if loss_scaler is not None else (lambda x: x)
    opt_step = (lambda: (loss_scaler.step(optimizer), loss_scaler.update())) if loss_scaler is not None else optimizer.step
    opt_stats = (lambda: loss_scaler.unscaled_grad_stats(optimizer)) if loss_scaler is not None else \
        getattr(optimizer, "grad_stats", None)
    model.train(True)
    accum = max(int(getattr(args, "accum_iter", 1)), 1)
    optimizer.zero_grad(set_to_none=True)
    n_iter = len(data_loader)
    stats = EpochStats()
    acc = torch.zeros(2, dtype=torch.float32, device=device)   # [sum(loss), count] since the last flush
    total, count = 0.0, 0.0
    gs = None
    lr = 0.0
    t0 = time.perf_counter()
    no_sync = getattr(model, "no_sync", None)
    for it, batch in enumerate(data_loader):
        samples = batch[0] if isinstance(batch, (tuple, list)) else batch
        if it % accum == 0:
            lr = adjust_learning_rate(optimizer, it / n_iter + epoch, args)  # per-iteration schedule
        samples = samples.to(device, non_blocking=True)
        last_micro = (it + 1) % accum == 0
        if no_sync is not None and not last_micro:
            with no_sync():
                loss, _, _ = model(samples, mask_ratio=args.mask_ratio)
                scale(loss / accum).backward()
        else:
            loss, _, _ = model(samples, mask_ratio=args.mask_ratio)
            scale(loss / accum).backward()
        acc[0] += loss.detach()
        acc[1] += 1.0
        if last_micro:
            if grad_stats and opt_stats is not None and (it // accum + 1) % log_every == 0:
                gs = opt_stats()
            opt_step()
            optimizer.zero_grad(set_to_none=True)
        stats.steps += 1
        stats.samples += samples.shape[0] * _world()
        if (it + 1) % (log_every * accum) == 0 or it + 1 == n_iter:
            s, c = _flush(stats, acc, gs, it + 1, lr, printer)
            total, count = total + s, count + c
            acc.zero_()
            gs = None
    torch.cuda.synchronize(device) if torch.device(device).type == "cuda" else None
    stats.seconds = time.perf_counter() - t0
    stats.loss = total / max(count, 1.0)
    stats.lr = lr
    return stats


def train_epoch_cls(model, train_loader: Iterable, optimizer, device, pos_weight: Optional[torch.Tensor] = None,
                    class_weights: Optional[torch.Tensor] = None, max_batches: Optional[int] = None, log_every: int = 20,
                    printer: Optional[Callable] = print, grad_stats: bool = True, loss_scaler=None) -> EpochStats:
    """tc.py:4446-4650 hot loop: zero_grad -> model(data) -> BCE/CE -> backward -> grad norm -> optimizer.step().
    Batches are (data, target) or (data, target, meta) as produced by the reference's pack_collate.
    loss_scaler: optim.LossScaler for precision mode fp16 (tc.py:4533-4546: scaler.scale(loss).backward(); scaler.unscale_;
    grad norms; scaler.step(optimizer); scaler.update()) -- None for bf16 / fp32."""
    model.train(True)
    scale = loss_scaler.scale if loss_scaler is not None else (lambda x: x)
    opt_step = (lambda: (loss_scaler.step(optimizer), loss_scaler.update())) if loss_scaler is not None else optimizer.step
    opt_stats = (lambda: loss_scaler.unscaled_grad_stats(optimizer)) if loss_scaler is not None else \
        getattr(optimizer, "grad_stats", None)
    stats = EpochStats()
    acc = torch.zeros(2, dtype=torch.float32, device=device)
    total, count = 0.0, 0.0
    gs = None
    t0 = time.perf_counter()
    n = 0
    for it, batch in enumerate(train_loader):
        if max_batches is not None and it >= max_batches:
            break
        data, target = batch[0], batch[1]
        data = data.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        optimizer.zero_grad(set_to_none=True)
        output = model(data)
        loss = supervised_loss(output, target, pos_weight, class_weights)
        scale(loss).backward()
        if grad_stats and opt_stats is not None and (it + 1) % log_every == 0:
            gs = opt_stats()
        opt_step()
        acc[0] += loss.detach()
        acc[1] += 1.0
        stats.steps += 1
        stats.samples += data.shape[0] * _world()
        n = it + 1
        if n % log_every == 0:
            s, c = _flush(stats, acc, gs, n, optimizer.param_groups[0]["lr"], printer)
            total, count = total + s, count + c
            acc.zero_()
            gs = None
    if n % log_every != 0 and n > 0:
        s, c = _flush(stats, acc, gs, n, optimizer.param_groups[0]["lr"], printer)
        total, count = total + s, count + c
    torch.cuda.synchronize(device) if torch.device(device).type == "cuda" else None
    stats.seconds = time.perf_counter() - t0
    stats.loss = total / max(count, 1.0)
    stats.lr = optimizer.param_groups[0]["lr"]
    return stats


def class_probabilities(logits: torch.Tensor) -> torch.Tensor:
    """tc.py:4700-4730: binary packs (n_class = 2) score with sigmoid(l1 - l0), multi-class with softmax -- computed on
    the device the logits live on.  Returns [B] (binary: P(class 1)) or [B, n_class]."""
    logits = logits.float()
    if logits.shape[-1] == 2:
        return torch.sigmoid(logits[:, 1] - logits[:, 0])
    return torch.softmax(logits, dim=-1)


class _RankBatches:
    """Batch sampler view: batches rank, rank + world, ... of an existing batch sampler (loader order preserved)."""

    def __init__(self, batch_sampler, rank: int, world: int, max_batches: Optional[int] = None):
        self.bs, self.rank, self.world, self.max_batches = batch_sampler, rank, world, max_batches

    def __iter__(self):
        for i, b in enumerate(self.bs):
            if self.max_batches is not None and i >= self.max_batches:
                return
            if i % self.world == self.rank:
                yield b

    def __len__(self):
        n = len(self.bs) if self.max_batches is None else min(len(self.bs), self.max_batches)
        return (n - self.rank + self.world - 1) // self.world if n > self.rank else 0


_RANK_LOADERS: "weakref.WeakKeyDictionary" = None  # base DataLoader -> {(rank, world, max_batches): rank-local DataLoader}


def _rank_local_loader(loader, rank: int, world: int, max_batches: Optional[int]):
    """DataLoader around this rank's batches of `loader` (same dataset, collate, workers, contexts), built once per
    (loader, rank, world, max_batches) and kept: with persistent_workers the workers survive from one validation pass to the
    next instead of respawning every epoch."""
    import weakref
    from torch.utils.data import DataLoader, RandomSampler
    global _RANK_LOADERS
    if _RANK_LOADERS is None:
        _RANK_LOADERS = weakref.WeakKeyDictionary()
    per = _RANK_LOADERS.setdefault(loader, {})
    key = (rank, world, max_batches)
    hit = per.get(key)
    if hit is not None:
        return hit
    # every rank walks ITS OWN copy of the base batch sampler and keeps batches rank, rank + world, ...: the copies must
    # produce the same order, i.e. the sampler is sequential or seeded identically on every rank.  An unseeded RandomSampler
    # would silently duplicate some samples and miss others in the gathered result.
    inner = getattr(loader.batch_sampler, "sampler", None)
    if isinstance(inner, RandomSampler) and getattr(inner, "generator", None) is None:
        raise ValueError("sharded evaluation needs a loader whose order is the same on every rank: the reference validates "
                         "with shuffle=False (tc.py:6660-6728); got an unseeded RandomSampler")
    kw = dict(num_workers=loader.num_workers, collate_fn=loader.collate_fn, pin_memory=loader.pin_memory,
              worker_init_fn=loader.worker_init_fn, timeout=loader.timeout, generator=loader.generator)
    if getattr(loader, "pin_memory_device", ""):
        kw["pin_memory_device"] = loader.pin_memory_device
    if loader.num_workers > 0:
        kw.update(prefetch_factor=loader.prefetch_factor, persistent_workers=loader.persistent_workers,
                  multiprocessing_context=loader.multiprocessing_context)
    mine = DataLoader(loader.dataset, batch_sampler=_RankBatches(loader.batch_sampler, rank, world, max_batches), **kw)
    per[key] = mine
    return mine


def shard_eval_loader(loader, rank: int, world: int, max_batches: Optional[int] = None):
    """This rank's share of an evaluation loader as (global batch index, batch) pairs.  A torch DataLoader is rebuilt
    (once, see _rank_local_loader) around a rank-local batch sampler, so its workers only open / decode / transform this
    rank's frames (PackDataset does PIL open + transform per item, data/packs.py:70-80); any other iterable is walked with
    the foreign batches skipped."""
    from torch.utils.data import DataLoader
    if isinstance(loader, DataLoader) and loader.batch_sampler is not None:
        for j, batch in enumerate(_rank_local_loader(loader, rank, world, max_batches)):
            yield rank + j * world, batch
        return
    for it, batch in enumerate(loader):
        if max_batches is not None and it >= max_batches:
            break
        if it % world == rank:
            yield it, batch


@torch.no_grad()
def evaluate_cls(model, loader: Iterable, device, max_batches: Optional[int] = None, shard: Optional[bool] = None,
                 return_probs: bool = False):
    """tc.py:4652-4812 forward part: forward-only kernels in eval mode; logits (and optionally probabilities) of the whole
    pass stay ON THE DEVICE and cross to the host once at the end -- the reference does `logits.detach().cpu()` + `torch.cat`
    per batch (tc.py:4790-4812), a host sync per batch that drains the queue.  Targets never leave the host.  Metrics stay in
    the reference's code.
    shard (default: whenever torch.distributed is initialised with world > 1): rank r evaluates batches r, r + world, ...
    through a rank-local batch sampler (it never decodes another rank's frames); the pieces are all-gathered (one padded
    device collective) and returned in loader order on every rank -- the reference validates on rank 0 only while the other
    GPUs idle (tc.py:6660-6728)."""
    model.eval()
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    if shard is None:
        shard = world > 1
    if not shard:
        world, rank = 1, 0
    dev = torch.device(device)
    idx, sizes, lgs, tgs = [], [], [], []
    for it, batch in shard_eval_loader(loader, rank, world, max_batches):
        lg = model(batch[0].to(dev, non_blocking=True)).float()   # stays on the device
        idx.append(it)
        sizes.append(lg.shape[0])
        lgs.append(lg)
        tg = torch.as_tensor(batch[1]).detach().cpu()   # targets stay on the host, in their own dtype and trailing shape
        if tg.ndim == 0 or tg.shape[0] != lg.shape[0]:
            raise ValueError(f"evaluate_cls: batch {it} has {lg.shape[0]} samples but targets of shape {tuple(tg.shape)}")
        tgs.append(tg)
    n_class = lgs[0].shape[1] if lgs else 0
    if world > 1:
        # the bookkeeping (batch indices, sizes, host-side targets) is tiny and goes as one object gather; the logits travel
        # as ONE padded device collective (ranks may hold different row counts: ragged last batch, odd batch count)
        meta = [None] * world
        dist.all_gather_object(meta, (idx, sizes, n_class, tgs))   # (per-batch host tensors: sliced on dim 0 only, never flattened)
        n_class = max(m[2] for m in meta)
        rows = [sum(m[1]) for m in meta]
        pad = max(rows) if rows else 0
        mine = torch.zeros(pad, n_class, dtype=torch.float32, device=dev)
        if lgs:
            mine[:rows[rank]] = torch.cat(lgs)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        pieces = []  # (global batch index, logits rows on the device, targets)
        for r, (r_idx, r_sizes, _, r_tgs) in enumerate(meta):
            o = 0
            for b, n, tg in zip(r_idx, r_sizes, r_tgs):
                assert tg.shape[0] == n, (r, b, n, tuple(tg.shape))
                pieces.append((b, parts[r][o:o + n], tg))
                o += n
        pieces.sort(key=lambda x: x[0])
        logits_dev = torch.cat([x[1] for x in pieces]) if pieces else mine[:0]
        targets = torch.cat([x[2] for x in pieces]) if pieces else torch.zeros(0, dtype=torch.int64)
    else:
        logits_dev = torch.cat(lgs) if lgs else torch.zeros(0, 0, device=dev)
        targets = torch.cat(tgs) if tgs else torch.zeros(0, dtype=torch.int64)
    if return_probs:
        probs_dev = class_probabilities(logits_dev) if logits_dev.numel() else logits_dev.new_zeros(0)
        return logits_dev.cpu(), targets, probs_dev.cpu()   # the pass's only device -> host transfers
    return logits_dev.cpu(), targets


# ---------------------------------------------------------------------------------------------------
# checkpoints
# ---------------------------------------------------------------------------------------------------
def _unwrap(model):
    return getattr(model, "module", model)


def _is_main() -> bool:
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def grad_scaler_state() -> dict:
    """What an enabled ``torch.cuda.amp.GradScaler`` serialises (its initial values).  bf16 needs no loss scaling, but
    the reference's resume paths call ``scaler.load_state_dict(ckpt[...])`` on an ENABLED scaler (misc.py:349-350,
    tc.py:5978), which rejects an empty dict -- so the checkpoints carry a well-formed one."""
    return {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 0}


def _host_state_dict(model) -> Dict[str, torch.Tensor]:
    return {k: v.detach().cpu() for k, v in _unwrap(model).state_dict().items()}


def _host_optimizer_state(optimizer) -> dict:
    """Optimizer state with every tensor already in host memory, built on the CALLING thread (its current stream orders the
    copies after the last update): the checkpoint writer thread must never touch device memory -- its own current stream is
    the default stream, which nothing orders against a training loop that runs under torch.cuda.stream(...)."""
    import inspect
    sd_fn = optimizer.state_dict
    try:
        takes_host = "host" in inspect.signature(sd_fn).parameters
    except (TypeError, ValueError):
        takes_host = False
    if takes_host:
        return sd_fn(host=True)

    def to_host(o):
        if torch.is_tensor(o):
            return o.detach().cpu()
        if isinstance(o, dict):
            return {k: to_host(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return type(o)(to_host(v) for v in o)
        return o
    return to_host(sd_fn())


def _symlink_pointer(pointer: Path, target: Path) -> None:
    """tc.py:3929-3940 _update_checkpoint_pointer / misc.py:323-333: relative symlink, copy when links are refused."""
    import shutil
    pointer.parent.mkdir(parents=True, exist_ok=True)
    try:
        if pointer.is_symlink() or pointer.exists():
            pointer.unlink()
        pointer.symlink_to(target.name)
    except OSError:
        shutil.copy2(target, pointer)


class AsyncCheckpointWriter:
    """Takes ``torch.save`` (pickling + disk) off the training thread: the caller hands over a payload whose tensors are
    already host copies; one background thread writes `<path>.tmp`, renames it and then moves the pointer.  `wait()`
    joins outstanding writes (call it before reading a checkpoint back or at exit).  The reference saves synchronously
    on rank 0 while every other rank sits in the barrier behind it (tc.py:7036-7111 + 7246; misc.py:306-335)."""

    def __init__(self):
        import concurrent.futures
        self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
        self._pending: List = []

    def submit(self, payload: dict, path: Path, pointer: Optional[Path] = None):
        def job():
            tmp = path.with_name(path.name + ".tmp")
            torch.save(payload, tmp)
            os.replace(tmp, path)
            if pointer is not None:
                _symlink_pointer(pointer, path)
            return path
        fut = self._pool.submit(job)
        self._pending.append(fut)
        return fut

    def wait(self) -> None:
        pend, self._pending = self._pending, []
        for f in pend:
            f.result()

    def close(self) -> None:
        self.wait()
        self._pool.shutdown(wait=True)


def _write(payload: dict, path: Path, pointer: Optional[Path], writer: Optional[AsyncCheckpointWriter]) -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    if writer is not None:
        writer.submit(payload, path, pointer)
        return
    torch.save(payload, path)
    if pointer is not None:
        _symlink_pointer(pointer, path)


def save_mae_checkpoint(output_dir, epoch: int, model, optimizer, args, scaler_state: Optional[dict] = None,
                        writer: Optional[AsyncCheckpointWriter] = None) -> Path:
    """misc.py:306-335: <out>/checkpoint-<epoch>.pth = {model, optimizer, epoch, scaler, args} + last.pth symlink.
    `scaler` holds a GradScaler-shaped dict (see grad_scaler_state) so that the reference's load_model accepts the file."""
    out = Path(output_dir)
    path = out / f"checkpoint-{epoch}.pth"
    if _is_main():
        to_save = {"model": _host_state_dict(model), "optimizer": _host_optimizer_state(optimizer), "epoch": epoch,
                   "scaler": scaler_state if scaler_state else grad_scaler_state(), "args": args}
        _write(to_save, path, out / "last.pth", writer)
    return path


def load_mae_checkpoint(path, model, optimizer=None, args=None, loss_scaler=None) -> int:
    """misc.py:338-352: restores model (+ optimizer, loss scaler and start epoch).  Returns the next epoch to run."""
    ckpt = torch.load(str(path), map_location="cpu", weights_only=False)
    _unwrap(model).load_state_dict(ckpt["model"])
    start = 0
    if optimizer is not None and "optimizer" in ckpt and "epoch" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
        start = int(ckpt["epoch"]) + 1
        if args is not None:
            args.start_epoch = start
        if loss_scaler is not None and "scaler" in ckpt:  # misc.py:349-350
            loss_scaler.load_state_dict(ckpt["scaler"])
    return start


def cls_checkpoint_name(stem: str, epoch: int, selection_tag: str = "best") -> str:
    """tc.py:7089-7091: `<stem>_e<EE>_<tag>.pth` (the pointer is `<stem>.pth`)."""
    return f"{stem}_e{epoch:02d}_{selection_tag}.pth"


def find_existing_checkpoint(stem_path):
    """tc.py:3914-3926 _find_existing_checkpoint: the `<stem>.pth` pointer if present, else the newest `<stem>_e*_*.pth`.
    Returns (path | None, pointer_valid)."""
    stem_path = Path(stem_path)
    pointer = stem_path.with_suffix(".pth") if stem_path.suffix != ".pth" else stem_path
    stem_path = pointer.with_suffix("")
    if pointer.exists() or pointer.is_symlink():
        return pointer, True
    if not stem_path.parent.exists():
        return None, False
    cands = sorted(stem_path.parent.glob(f"{stem_path.name}_e*_*.pth"))
    return (cands[-1], False) if cands else (None, False)


def save_cls_checkpoint(path, epoch: int, model, optimizer, scheduler=None, loss: Optional[float] = None,
                        extra: Optional[dict] = None, pointer=None, writer: Optional[AsyncCheckpointWriter] = None,
                        loss_scaler=None) -> Path:
    """tc.py:7036-7111 payload: epoch / model_state_dict / optimizer_state_dict / scaler_state_dict / loss /
    py_state / np_state / torch_state (+ scheduler_state_dict only when a scheduler exists, tc.py:7065-7066); `extra`
    carries the reference's val_* / monitor_* / threshold fields unchanged.  `pointer`: the `<stem>.pth` link to move
    onto this file (tc.py:7111).  `loss_scaler`: the optim.LossScaler (or torch GradScaler) of a precision-mode-fp16 run; its state
    goes where the reference puts `scaler.state_dict()` (tc.py:7049)."""
    import random

    import numpy as np
    path = Path(path)
    if _is_main():
        scaler_sd = loss_scaler.state_dict() if loss_scaler is not None else None
        payload = {"epoch": epoch, "model_state_dict": _host_state_dict(model),
                   "optimizer_state_dict": _host_optimizer_state(optimizer), "scaler_state_dict": scaler_sd or grad_scaler_state(),
                   "loss": loss,
                   "py_state": random.getstate(), "np_state": np.random.get_state(), "torch_state": torch.get_rng_state()}
        if scheduler is not None:
            payload["scheduler_state_dict"] = scheduler.state_dict()
        payload.update(extra or {})
        _write(payload, path, Path(pointer) if pointer is not None else None, writer)
    return path


@dataclass
class ClsResume:
    start_epoch: int = 1
    best_val_perf: Optional[float] = None
    resume_monitor_available: bool = False
    thresholds: Dict = field(default_factory=dict)
    threshold_records: Dict = field(default_factory=dict)
    source: Optional[Path] = None
    from_parent: bool = False


def load_cls_checkpoint(stem_path, model, optimizer=None, scheduler=None, parent_checkpoint=None,
                        restore_rng: bool = True, loss_scaler=None) -> ClsResume:
    """tc.py:5667-5714 + 5976-5980: resume from `<stem>.pth` / the newest `<stem>_e*_*.pth` (model, optimizer, scheduler,
    Python / NumPy / torch RNG streams, monitor value, thresholds; repairs a missing pointer), else start from a parent
    run's weights (`parent_checkpoint`: payload with model_state_dict or a bare state dict), else a fresh start."""
    import random

    import numpy as np
    info = ClsResume()
    existing, pointer_valid = find_existing_checkpoint(stem_path)
    if existing is not None:
        main = torch.load(str(existing), map_location="cpu", weights_only=False)
        _unwrap(model).load_state_dict(main["model_state_dict"])
        info.start_epoch = int(main["epoch"]) + 1
        monitor = main.get("monitor_value")
        if monitor is None and "val_loss" in main:
            monitor = main.get("val_loss")
        if monitor is None:
            monitor = main.get("val_perf")
        else:
            info.resume_monitor_available = True
        info.best_val_perf = monitor
        if restore_rng:
            random.setstate(main["py_state"])
            np.random.set_state(main["np_state"])
            torch.set_rng_state(main["torch_state"])
        info.thresholds = dict(main.get("thresholds", {}) or {})
        info.threshold_records = dict(main.get("threshold_records", {}) or {})
        info.source = Path(existing)
        if not pointer_valid:
            stem = Path(stem_path)
            _symlink_pointer(stem if stem.suffix == ".pth" else stem.with_suffix(".pth"), Path(existing))
        if info.best_val_perf is not None:  # tc.py:5976-5980
            if optimizer is not None and "optimizer_state_dict" in main:
                optimizer.load_state_dict(main["optimizer_state_dict"])
            if scheduler is not None and "scheduler_state_dict" in main:
                scheduler.load_state_dict(main["scheduler_state_dict"])
            if loss_scaler is not None and main.get("scaler_state_dict"):  # tc.py:5978
                loss_scaler.load_state_dict(main["scaler_state_dict"])
        return info
    if parent_checkpoint:
        parent = Path(parent_checkpoint).expanduser()
        if not parent.exists():
            raise FileNotFoundError(f"Parent checkpoint '{parent_checkpoint}' does not exist.")
        state = torch.load(str(parent), map_location="cpu", weights_only=False)
        if isinstance(state, dict) and "model_state_dict" in state:
            info.thresholds = dict(state.get("thresholds", {}) or {})
            info.threshold_records = dict(state.get("threshold_records", {}) or {})
            state = state["model_state_dict"]
        _unwrap(model).load_state_dict(state)
        info.source, info.from_parent = parent, True
    return info


# ---------------------------------------------------------------------------------------------------
# synthetic loader (the reference's PIL / torchvision pipeline is out of scope: SURVEY §8-f rank 2)
# ---------------------------------------------------------------------------------------------------
class SyntheticLoader:
    """Post-transform Hyperkvasir/SUN-shaped batches resident on the device: imgs ~ N(0,1) [B,3,S,S], labels ~ Bern(.5)."""

    def __init__(self, batch_size: int, n_batches: int, device, img_size: int = 224, seed: int = 1234, fresh: bool = False):
        self.batch_size, self.n, self.device, self.fresh = batch_size, n_batches, device, fresh
        g = torch.Generator(device=device).manual_seed(seed)
        self.gen = g
        self.imgs = torch.randn(batch_size, 3, img_size, img_size, generator=g, device=device)
        self.labels = (torch.rand(batch_size, generator=g, device=device) < 0.5).long()

    def __len__(self):
        return self.n

    def __iter__(self):
        for _ in range(self.n):
            if self.fresh:
                self.imgs.normal_(generator=self.gen)
                self.labels = (torch.rand(self.batch_size, generator=self.gen, device=self.device) < 0.5).long()
            yield self.imgs, self.labels
