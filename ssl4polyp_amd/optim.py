"""Fused AdamW + gradient statistics over the engine's flat parameter storage.

Same update rule and hyper-parameter surface as ``torch.optim.AdamW`` as the reference uses it
(train_classification.py:5766-5768: two groups head/backbone, default betas; mae/main_pretrain.py:217-218:
decay / no-decay groups from timm ``add_weight_decay``, betas (0.9, 0.95)), but one HIP launch per
contiguous flat segment instead of ~150 tensors, and the same launch refreshes the bf16 shadow weights the
MFMA GEMMs read.  ``grad_stats`` replaces the per-parameter host-synchronising loops of
train_classification.py:1437-1454 (_compute_grad_norm) and mae/util/misc.py:387-400 (detect_grad_anomalies)
with one pass that stays on the device.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import contextlib

import torch

from . import _lib
from .engine import _ptr, _stream


def add_weight_decay(model, weight_decay=1e-5, skip_list=()):
    """timm 0.4.12 optim_factory.add_weight_decay (main_pretrain.py:217): 1-D params and biases get no decay."""
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if len(param.shape) == 1 or name.endswith(".bias") or name in skip_list:
            no_decay.append(param)
        else:
            decay.append(param)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, params=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 overlap_forward: bool = False):
        """`model`: an ssl4polyp_amd module (owns the flat storage).  `params`: iterable of parameters or of
        torch-style param-group dicts (default: all parameters of the model).
        overlap_forward: run the update on the engine's side stream, block by block in forward order, and let the next
        forward wait per block (the update is HBM-bound, the forward GEMMs MFMA-bound: ~0.5 ms of the step hides).
        The model's forward / state_dict and this optimizer's state_dict / grad_stats wait for the update by
        themselves; code that reads parameter tensors directly right after step() must call `sync()` first."""
        self._rt = model._rt
        self.overlap_forward = bool(overlap_forward)
        if params is None:
            params = list(model.parameters())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._M: Dict[str, torch.Tensor] = {}
        self._V: Dict[str, torch.Tensor] = {}
        self._plan = None
        self._plan_key = None
        self.grad_scale = 1.0  # e.g. 1/world_size when gradients were all-reduced with SUM
        self.grad_sync = None  # parallel.GradSync: waited on before the update

    # -- planning: contiguous flat segments per group --------------------------------------------
    def _ensure_state(self):
        f = self._rt.flat
        if f is None or not f.P:
            raise _lib.PolypMaeError("FusedAdamW: run a forward pass (or model._rt.ensure(device)) before step()")
        key = (id(f.P["mat"]), id(f.P["vec"]))
        if self._plan_key != key:
            old_M, old_V = self._M, self._V
            self._M = {r: torch.zeros_like(t) for r, t in f.P.items()}
            self._V = {r: torch.zeros_like(t) for r, t in f.P.items()}
            for r in old_M:  # re-materialised storage: carry the moments over
                if old_M[r].numel() == self._M[r].numel():
                    self._M[r].copy_(old_M[r])
                    self._V[r].copy_(old_V[r])
            self._plan_key = key
            self._plan = None
        return f

    def _segments(self, f, group):
        """[(region, lo, hi)] covering the group's parameters that currently have a gradient."""
        idx = {id(p): i for i, p in enumerate(f.params)}
        items = []
        for p in group["params"]:
            if p.grad is None:
                continue
            i = idx.get(id(p))
            if i is None:
                raise _lib.PolypMaeError("FusedAdamW: parameter does not belong to the model's flat storage")
            if not f.grad_is_flat(f.names[i]):  # foreign gradient tensor: stage it into the flat range
                f.grad_view(f.names[i]).copy_(p.grad)
            A = f.ALIGN
            items.append((f.region[i], f.offset[i], f.offset[i] + (f.numel[i] + A - 1) // A * A))
        items.sort()
        segs = []
        for r, lo, hi in items:
            if segs and segs[-1][0] == r and segs[-1][2] == lo:
                segs[-1] = (r, segs[-1][1], hi)
            else:
                segs.append((r, lo, hi))
        return segs

    # -- device-resident hyper-parameters (so that a captured step can be replayed as a hipGraph) ---------------
    HYPER_STRIDE = 16  # floats per group: lr, beta1, beta2, eps, wd, grad_scale, step, bc1, rsqrt_bc2 (pm_adamw_dev)

    def _host_hyper(self):
        rows = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            rows.append((float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                         float(self.grad_scale)))
        return rows

    def sync_hyper(self, force: bool = False):
        """Push lr / betas / eps / weight_decay / grad_scale to the device records when they changed on the host.
        Called automatically by step() outside stream capture; call it yourself between graph replays."""
        f = self._ensure_state()
        rows = self._host_hyper()
        if getattr(self, "_hyper", None) is None or self._hyper.device != f.device:
            self._hyper = torch.zeros(len(rows), self.HYPER_STRIDE, dtype=torch.float32, device=f.device)
            for gi, group in enumerate(self.param_groups):  # resume: restore the step counter
                self._hyper[gi, 6] = float(group.get("step", 0))
            self._hyper_host = None
        if force or rows != self._hyper_host:
            # The MAE loop changes lr EVERY iteration (engine_pretrain.py:47-48): the upload must not block the host, or
            # each step would wait for the whole backward to drain before AdamW and the next forward can be enqueued.
            # Pinned staging slots, asynchronous copy on the current stream, one event per slot (a slot is rewritten
            # only after its previous copy has executed: four steps of run-ahead before the host could ever wait here).
            if f.device.type == "cuda":
                if getattr(self, "_hyper_pin", None) is None or self._hyper_pin[0].shape[0] != len(rows):
                    self._hyper_pin = [torch.empty(len(rows), 6, dtype=torch.float32).pin_memory() for _ in range(4)]
                    self._hyper_pin_ev = [None] * 4
                    self._hyper_slot = 0
                i = self._hyper_slot
                self._hyper_slot = (i + 1) % 4
                if self._hyper_pin_ev[i] is not None:
                    self._hyper_pin_ev[i].synchronize()
                self._hyper_pin[i].copy_(torch.tensor(rows, dtype=torch.float32))
                self._hyper[:, :6].copy_(self._hyper_pin[i], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(f.device))
                self._hyper_pin_ev[i] = ev
            else:
                self._hyper[:, :6].copy_(torch.tensor(rows, dtype=torch.float32))
            self._hyper_host = rows

    @torch.no_grad()
    def step(self, closure=None, loss_scaler: "Optional[LossScaler]" = None):
        """loss_scaler: the gradients carry its loss scale (precision mode fp16).  The update then starts with one statistics
        pass over them and `pm_loss_scale_update` -- found_inf, the skip flag and 1/scale go into the device-side hyper records,
        the scale moves (GradScaler.step + update, train_classification.py:4545-4546) -- and `pm_adamw_dev` unscales inside
        the update or returns untouched on a non-finite step: no host read-back anywhere."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        f = self._ensure_state()
        rt = self._rt
        rt.wait_updates()  # a previous overlapped update (reads the hyper records and the gradients)
        if self.grad_sync is not None:
            self.grad_sync.wait()
        capturing = f.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if not capturing or getattr(self, "_hyper", None) is None:
            self.sync_hyper()
        lib = self._rt.k.lib
        overlap = self.overlap_forward and not capturing and f.device.type == "cuda"
        work = []  # (group index, region, lo, hi)
        for gi, group in enumerate(self.param_groups):
            group["step"] = int(group.get("step", 0)) + 1  # host mirror (the device record is authoritative)
            for r, lo, hi in self._segments(f, group):
                work.append((gi, r, lo, hi))
        scaled = loss_scaler is not None and loss_scaler.enabled
        stat_segs = [(r, lo, hi) for _, r, lo, hi in work] if scaled else []
        if not scaled and getattr(self, "_scaler_used", False):  # a scaler was dropped: clear its skip flag / unscale factor
            self._hyper[:, 9:11] = 0.0
            self._scaler_used = False
        if overlap:
            # forward order: vectors first, then the matrices by offset, cut at the block boundaries so that the next
            # forward can start on block 0 while the later blocks are still being updated
            cuts = sorted(f.offset[i] for i, n in enumerate(f.names) if n.endswith("attn.qkv.weight"))
            split = []
            for gi, r, lo, hi in work:
                if r != "mat":
                    split.append((gi, r, lo, hi))
                    continue
                pts = [lo] + [c for c in cuts if lo < c < hi] + [hi]
                split += [(gi, r, a, b) for a, b in zip(pts[:-1], pts[1:])]
            work = sorted(split, key=lambda w: (w[1] != "vec", w[2]))
            main = torch.cuda.current_stream(f.device)
            side = rt.k.side_stream(f.device)
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            stream_ctx = torch.cuda.stream(side)
        else:
            stream_ctx = contextlib.nullcontext()
        with stream_ctx:
            if scaled:
                state, stats = loss_scaler._device_state(f.device)
                stats.zero_()
                for r, lo, hi in stat_segs:
                    _lib.check(lib.pm_grad_stats(_ptr(f.G[r][lo:hi]), hi - lo, _ptr(stats), _stream()), "pm_grad_stats")
                _lib.check(lib.pm_loss_scale_update(_ptr(state), _ptr(stats), _ptr(self._hyper), len(self.param_groups),
                                                    loss_scaler.growth_factor, loss_scaler.backoff_factor,
                                                    loss_scaler.growth_interval, _stream()), "pm_loss_scale_update")
                self._scaler_used = True
            _lib.check(lib.pm_adamw_tick(_ptr(self._hyper), len(self.param_groups), _stream()), "pm_adamw_tick")
            for gi, r, lo, hi in work:
                shadow = f.S[lo:hi] if (r == "mat" and f.S is not None) else None
                _lib.check(lib.pm_adamw_dev(_ptr(f.P[r][lo:hi]), _ptr(f.G[r][lo:hi]), _ptr(self._M[r][lo:hi]),
                                            _ptr(self._V[r][lo:hi]), _ptr(shadow),
                                            _lib.dtype_code(shadow.dtype) if shadow is not None else 0, hi - lo,
                                            _ptr(self._hyper[gi]), _stream()), "pm_adamw_dev")
                if overlap:
                    done = torch.cuda.Event()
                    done.record(side)
                    rt.pending_updates.append((r, lo, hi, done))
        return loss

    def sync(self) -> None:
        """Make the current stream wait for an overlapped update (no-op otherwise)."""
        self._rt.wait_updates()

    def grad_stats(self) -> torch.Tensor:
        """Device tensor [sum(g^2), #NaN, #Inf] over every gradient of every group (no host sync)."""
        f = self._ensure_state()
        self._rt.wait_updates()
        out = torch.zeros(3, dtype=torch.float32, device=f.device)
        lib = self._rt.k.lib
        for group in self.param_groups:
            for r, lo, hi in self._segments(f, group):
                _lib.check(lib.pm_grad_stats(_ptr(f.G[r][lo:hi]), hi - lo, _ptr(out), _stream()), "pm_grad_stats")
        return out

    # -- torch.optim.AdamW-compatible (de)serialisation --------------------------------------------
    def state_dict(self, host: bool = False):
        """torch.optim.AdamW's layout.  host=True: the moments are copied to host memory HERE, on the calling thread's current
        stream (two D2H copies of the flat ranges, then per-parameter views) -- what a checkpoint writer thread needs: nothing
        device-side outlives the call and no device clone of the 2 x f32 moments is made."""
        self._rt.wait_updates()
        f = self._rt.flat
        idx = {id(p): i for i, p in enumerate(f.params)} if f is not None else {}
        state, groups, k = {}, [], 0
        M, V = self._M, self._V
        if host and M:
            M = {r: t.detach().cpu() for r, t in M.items()}
            V = {r: t.detach().cpu() for r, t in V.items()}
        for group in self.param_groups:
            ids = []
            for p in group["params"]:
                i = idx.get(id(p))
                if i is not None and M:
                    r, lo, n = f.region[i], f.offset[i], f.numel[i]
                    state[k] = {"step": torch.tensor(float(group.get("step", 0))),
                                "exp_avg": M[r][lo:lo + n].view(p.shape).clone(),
                                "exp_avg_sq": V[r][lo:lo + n].view(p.shape).clone()}
                ids.append(k)
                k += 1
            g = {kk: v for kk, v in group.items() if kk != "params"}
            g["params"] = ids
            groups.append(g)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        f = self._ensure_state()
        self._rt.wait_updates()
        idx = {id(p): i for i, p in enumerate(f.params)}
        for group, saved in zip(self.param_groups, sd["param_groups"]):
            for kk, v in saved.items():
                if kk != "params":
                    group[kk] = v
            for p, k in zip(group["params"], saved["params"]):
                st = sd["state"].get(k)
                if st is None:
                    continue
                i = idx[id(p)]
                r, lo, n = f.region[i], f.offset[i], f.numel[i]
                self._M[r][lo:lo + n].copy_(st["exp_avg"].reshape(-1))
                self._V[r][lo:lo + n].copy_(st["exp_avg_sq"].reshape(-1))
                group["step"] = int(float(st["step"]))
        self._hyper = None  # rebuild the device records (incl. the step counters) at the next step()


class _ScaleLossFn(torch.autograd.Function):
    """loss * scale and d loss * scale, the scale read from DEVICE memory by the kernel (pm_scale): no host value is baked in."""

    @staticmethod
    def forward(ctx, loss, scale):
        lib = _lib.load()
        x = loss.reshape(1).contiguous().float()
        out = torch.empty_like(x)
        _lib.check(lib.pm_scale(_ptr(x), _ptr(scale), _ptr(out), 1, _stream()), "pm_scale")
        ctx.scale = scale
        return out.reshape(())

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        d = dout.reshape(1).contiguous().float()
        out = torch.empty_like(d)
        _lib.check(lib.pm_scale(_ptr(d), _ptr(ctx.scale), _ptr(out), 1, _stream()), "pm_scale")
        return out.reshape(()), None


class LossScaler:
    """Dynamic loss scaling for precision mode "fp16", resident on the device.

    Same surface and arithmetic as the ``torch.cuda.amp.GradScaler`` of the reference's AMP loops
    (train_classification.py:4533-4546: ``scaler.scale(loss).backward(); scaler.unscale_(opt); scaler.step(opt); scaler.update()``;
    engine_pretrain.py:65-72 through mae/util/misc.py:252-282), same defaults (65536, x2 after 2000 clean steps, x0.5 after a
    non-finite one).  Different mechanics: GradScaler.step reads ``found_inf`` back to the host before it decides whether to call
    ``optimizer.step()`` -- one pipeline drain per step; here the decision, the unscaling and the scale update are three device
    kernels inside ``FusedAdamW.step`` (one statistics pass, ``pm_loss_scale_update``, then an AdamW that multiplies by
    1/scale or returns untouched), so the host never waits.  ``unscale_`` is therefore a no-op kept for call-compatibility:
    the flat gradients stay scaled, ``unscaled_grad_stats`` gives the reference's logged norms.
    torch's own GradScaler also works with these models and FusedAdamW (the backward is linear in the loss gradient)."""

    def __init__(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000, enabled: bool = True):
        self.init_scale, self.growth_factor, self.backoff_factor = float(init_scale), float(growth_factor), float(backoff_factor)
        self.growth_interval, self.enabled = int(growth_interval), bool(enabled)
        self._state: Optional[torch.Tensor] = None   # f32[8] on the device: pm_loss_scale_update's record
        self._stats: Optional[torch.Tensor] = None   # f32[3]: sum g^2, #NaN, #Inf of the last step's scaled gradients
        self._pending: Optional[dict] = None         # a state dict loaded before the device is known

    def _device_state(self, device):
        if self._state is None or self._state.device != torch.device(device):
            st = torch.zeros(8, dtype=torch.float32)
            st[0] = self.init_scale
            if self._pending is not None:
                st[0], st[1] = float(self._pending["scale"]), float(self._pending.get("_growth_tracker", 0))
                self._pending = None
            self._state = st.to(device)
            self._stats = torch.zeros(3, dtype=torch.float32, device=device)
        return self._state, self._stats

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        if not self.enabled:
            return loss
        state, _ = self._device_state(loss.device)
        return _ScaleLossFn.apply(loss, state[0:1])

    def unscale_(self, optimizer) -> None:
        """No-op: FusedAdamW unscales inside its update (see the class comment)."""

    def step(self, optimizer, *args, **kwargs):
        if not isinstance(optimizer, FusedAdamW):
            raise _lib.PolypMaeError("LossScaler.step needs ssl4polyp_amd.optim.FusedAdamW (the skip / unscale decision runs inside "
                                     "its device kernels); use torch.cuda.amp.GradScaler with other optimizers")
        return optimizer.step(*args, loss_scaler=self if self.enabled else None, **kwargs)

    def update(self, new_scale: Optional[float] = None) -> None:
        """The scale already moved inside step() (device side).  new_scale: set it explicitly, as GradScaler.update does."""
        if new_scale is not None and self._state is not None:
            self._state[0] = float(new_scale)
            self._state[1] = 0.0

    def unscaled_grad_stats(self, optimizer) -> torch.Tensor:
        """[sum(g^2), #NaN, #Inf] of the UNSCALED gradients (device tensor, no host sync): what the reference logs after
        scaler.unscale_ (train_classification.py:4535-4544)."""
        gs = optimizer.grad_stats()
        if self.enabled and self._state is not None:
            gs[0] = gs[0] / (self._state[0] * self._state[0])
        return gs

    # -- host-side views (each is one device read-back: logging / checkpoints only) -------------------------------------------
    def get_scale(self) -> float:
        if not self.enabled:
            return 1.0
        if self._state is None:
            return float(self._pending["scale"]) if self._pending is not None else self.init_scale
        return float(self._state[0])

    def counters(self) -> dict:
        """{"steps", "skipped", "found_inf_last"} since construction."""
        if self._state is None:
            return {"steps": 0, "skipped": 0, "found_inf_last": False}
        v = self._state.tolist()
        return {"steps": int(v[4]), "skipped": int(v[3]), "found_inf_last": bool(v[2])}

    def state_dict(self) -> dict:
        """torch.cuda.amp.GradScaler's layout (what the reference's checkpoints store: misc.py:311-318, tc.py:7036-7067)."""
        if not self.enabled:
            return {}
        tracker = int(self._state[1]) if self._state is not None else int((self._pending or {}).get("_growth_tracker", 0))
        return {"scale": self.get_scale(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": tracker}

    def load_state_dict(self, sd: dict) -> None:
        if not sd:
            return
        self.growth_factor = float(sd.get("growth_factor", self.growth_factor))
        self.backoff_factor = float(sd.get("backoff_factor", self.backoff_factor))
        self.growth_interval = int(sd.get("growth_interval", self.growth_interval))
        if self._state is not None:
            self._state[0] = float(sd["scale"])
            self._state[1] = float(sd.get("_growth_tracker", 0))
        else:
            self._pending = dict(sd)
