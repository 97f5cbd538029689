"""Device-side tail of the input pipeline.

The reference decodes, resizes and augments with PIL on DataLoader workers and hands float32 NCHW batches to the GPU
(classification/data/transforms.py:225-253, mae/main_pretrain.py:156-191).  Here the workers stop at decoded uint8
HWC frames; `DevicePrefetcher` stages them through pinned host buffers, copies them on a dedicated stream while the
previous step computes (a uint8 batch is a quarter of the float32 bytes on PCIe) and runs the last three transform
stages on the device in one HBM-bound kernel (`pm_preprocess_u8`): optional horizontal / vertical flip, ToTensor,
Normalize -- bit-exact with torchvision's float32 arithmetic.
"""
from __future__ import annotations

import concurrent.futures
from typing import Iterable, Iterator, Optional, Sequence, Tuple

import torch

from . import _lib

IMAGENET_MEAN: Sequence[float] = (0.485, 0.456, 0.406)   # transforms.py:16
IMAGENET_STD: Sequence[float] = (0.229, 0.224, 0.225)    # transforms.py:17


def preprocess_u8(frames: torch.Tensor, flips: Optional[torch.Tensor] = None, mean: Sequence[float] = IMAGENET_MEAN,
                  std: Sequence[float] = IMAGENET_STD, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """frames: uint8 [B, H, W, 3] on the GPU; flips: uint8 [B] (bit 0 horizontal, bit 1 vertical) or None.
    Returns float32 [B, 3, H, W] = Normalize(mean, std)(ToTensor(frame)) of the (flipped) frames."""
    if frames.dtype != torch.uint8 or frames.ndim != 4 or frames.shape[-1] != 3 or not frames.is_contiguous():
        raise ValueError("frames must be a contiguous uint8 [B, H, W, 3] tensor")
    if not frames.is_cuda:
        raise _lib.PolypMaeError("preprocess_u8 runs on the GPU only (no CPU fallback)")
    B, H, W, _ = frames.shape
    if out is None:
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=frames.device)
    if flips is not None and (flips.dtype != torch.uint8 or flips.numel() != B or flips.device != frames.device):
        raise ValueError("flips must be uint8 [B] on the frames' device")
    lib = _lib.load()
    st = lib.pm_preprocess_u8(frames.data_ptr(), flips.data_ptr() if flips is not None else None, out.data_ptr(), B, H, W,
                              float(mean[0]), float(mean[1]), float(mean[2]), float(std[0]), float(std[1]), float(std[2]),
                              torch.cuda.current_stream(frames.device).cuda_stream)
    _lib.check(st, "pm_preprocess_u8")
    return out


class DevicePrefetcher:
    """Wraps a loader that yields (frames uint8 [B,H,W,3] on the host, *rest): copies batch i+1 to the device on a
    side stream (pinned staging, two slots) while batch i is consumed, and yields (imgs float32 [B,3,H,W] on the
    device, *rest on the device).  The yielded image tensor is a per-slot buffer that is refilled two batches later
    (consume it within the step, as a training loop does).  `flip_p > 0` draws per-sample horizontal / vertical flips (RandomHorizontalFlip /
    RandomVerticalFlip of the reference's train transform) from `generator`."""

    def __init__(self, loader: Iterable, device, mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD,
                 flip_p: float = 0.0, generator: Optional[torch.Generator] = None):
        self.loader, self.device = loader, torch.device(device)
        self.mean, self.std, self.flip_p, self.generator = mean, std, float(flip_p), generator
        self._pinned = [None, None]
        self._flip_pin = [None, None]   # per slot: pinned flip flags
        self._dev = [None, None]        # per slot: (uint8 frames, float32 images) on the device
        self._consumed = [None, None]   # per slot: event recorded on the consumer's stream after it used the batch
        self._slot_copied = [None, None]  # per slot: event after the slot's host-to-device copies were enqueued
        self._stream: Optional[torch.cuda.Stream] = None

    def __len__(self):
        return len(self.loader)

    def _stage(self, slot: int, batch) -> Tuple:
        frames, rest = batch[0], tuple(batch[1:])
        if frames.dtype != torch.uint8:
            raise ValueError("DevicePrefetcher expects uint8 HWC frames from the loader")
        if frames.is_pinned():  # e.g. DataLoader(pin_memory=True): no staging copy
            pin = frames
        else:
            pin = self._pinned[slot]
            if pin is None or pin.shape != frames.shape:
                pin = torch.empty(frames.shape, dtype=torch.uint8).pin_memory()
                self._pinned[slot] = pin
            pin.copy_(frames)
        flips = None
        if self.flip_p > 0:
            r = torch.rand(2, frames.shape[0], generator=self.generator)
            f8 = ((r[0] < self.flip_p).to(torch.uint8) | ((r[1] < self.flip_p).to(torch.uint8) << 1))
            # pinned too: a pageable host-to-device copy is synchronous and would stall the enqueue of the step
            flips = self._flip_pin[slot]
            if flips is None or flips.numel() != f8.numel():
                flips = torch.empty(f8.numel(), dtype=torch.uint8).pin_memory()
                self._flip_pin[slot] = flips
            flips.copy_(f8)
        # device buffers are owned per slot and reused (no allocator traffic on the copy stream): the copy stream first
        # waits until the consumer's work on the batch that last used this slot has been enqueued AND executed
        bufs = self._dev[slot]
        if bufs is None or bufs[0].shape != frames.shape:
            B, H, W, _ = frames.shape
            bufs = (torch.empty(frames.shape, dtype=torch.uint8, device=self.device),
                    torch.empty(B, 3, H, W, dtype=torch.float32, device=self.device))
            self._dev[slot] = bufs
        with torch.cuda.stream(self._stream):
            if self._consumed[slot] is not None:
                self._stream.wait_event(self._consumed[slot])
            bufs[0].copy_(pin, non_blocking=True)
            fl = flips.to(self.device, non_blocking=True) if flips is not None else None
            imgs = preprocess_u8(bufs[0], fl, self.mean, self.std, out=bufs[1])
            rest_dev = tuple(t.to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in rest)
            ev = torch.cuda.Event()
            ev.record(self._stream)
            self._slot_copied[slot] = ev
        return imgs, rest_dev, ev

    def __iter__(self) -> Iterator:
        if self.device.type != "cuda":
            raise _lib.PolypMaeError("DevicePrefetcher needs a GPU (the transform tail is a HIP kernel)")
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        it = iter(self.loader)
        # Staging runs on a worker thread: the pinned-memory copy and the host-to-device enqueue block their caller for
        # about as long as a CPU memcpy of the batch (1 ms per 64 frames measured), and the main thread must spend that
        # time enqueueing the training step instead.
        pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)

        def job(slot, batch):
            torch.cuda.set_device(self.device)
            # this slot's pinned buffers (frames and / or flip flags) were last read by the copies issued two batches ago,
            # which wait on the consumer's event and may not have executed yet: drain them before the host rewrites them
            if self._slot_copied[slot] is not None:
                self._slot_copied[slot].synchronize()
            return self._stage(slot, batch)

        try:
            slot = 0
            try:
                fut = pool.submit(job, slot, next(it))
            except StopIteration:
                return
            while fut is not None:
                imgs, rest, ev = fut.result()
                cur, slot = slot, slot ^ 1
                try:
                    fut = pool.submit(job, slot, next(it))
                except StopIteration:
                    fut = None
                main = torch.cuda.current_stream(self.device)
                main.wait_event(ev)
                for t in rest:
                    if torch.is_tensor(t):
                        t.record_stream(main)
                yield (imgs,) + rest
                # the consumer has enqueued its work on this batch: the slot's buffers may be refilled once that work ran
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(self.device))
                self._consumed[cur] = done
        finally:
            pool.shutdown(wait=True)
