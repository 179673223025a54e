"""Fused optimizer step over the flat parameter arena.

Replaces, in one HBM pass (od_adamw_ema): Lightning's `gradient_clip_val: 1.0` global-norm clip
(osu_dreamer/models/diffusion/model.yml:39), `torch.optim.AdamW(self.parameters(), **opt_args)`
(train.py:110-118, torch defaults betas=(0.9,0.999), eps=1e-8, decoupled decay on every tensor)
and `AveragedModel.update_parameters` (train.py:125-126; first update copies, then
lerp(ema, p, 1-0.99)).  It is a `torch.optim.Optimizer` so `LambdaLR` and Lightning drive it
unchanged: the learning rate is read from `param_groups[0]["lr"]`.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


class FusedAdamWEMA(torch.optim.Optimizer):
    def __init__(self, model, ema=None, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, max_grad_norm: Optional[float] = None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(list(model.parameters()), defaults)
        self.model, self.ema = model, ema
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        self._alloc()

    def _alloc(self):
        d = self.model.arena.data
        self.exp_avg = torch.zeros_like(d)
        self.exp_avg_sq = torch.zeros_like(d)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=d.device)

    def zero_grad(self, set_to_none: bool = False):
        """Keeps every .grad aliased to the arena (set_to_none would orphan the views)."""
        self.model.attach_grads().zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        model = self.model
        d = model.arena.data
        if self.exp_avg.device != d.device:
            self._alloc()
        reducer = getattr(model, "_reducer", None)
        if reducer is not None:
            reducer.wait()                       # gradient all-reduce must have landed
        g = model.adopt_grads()                  # = the arena's gradient buffer (foreign .grad tensors, e.g. DDP bucket views, are copied in)
        grp = self.param_groups[0]
        self.step_count += 1
        clip = float(self.max_grad_norm) if self.max_grad_norm else 0.0
        # the fused attention backward's sticky error word, folded into the step ON THE DEVICE, every step: a failed launch (incomplete
        # or NaN dq) turns the norm into NaN and the update below into a no-op; the trainer raises at its next logging point
        # (DenoiserEngine.check_attn_status) — no host synchronisation here
        eng = getattr(model, "_engine", None)
        status = eng.attn_status_ptr() if eng is not None else 0
        if status and reducer is not None and reducer.world > 1:
            # data parallel: the word every rank acts on is the OR over the ranks (GradBucketReducer.global_status) — a failed rank's gradients
            # are in everybody's all-reduced arena already
            self._global_status = reducer.global_status(eng.attn_status_view())
            status = self._global_status.data_ptr()
        if clip > 0 or status:
            self.gnorm_sq.zero_()
            ops.sqnorm(g, self.gnorm_sq, status)
        ema_buf, ema_mode, decay = None, 0, 0.0
        if self.ema is not None:
            ema_buf = self.ema.module.arena.data
            ema_mode = 1 if (self.ema.count + self.ema.fused_pending) == 0 else 2
            decay = self.ema.decay
            self.ema.fused_pending += 1
        ops.adamw_ema(d, g, self.exp_avg, self.exp_avg_sq, ema_buf, grp["lr"], grp["betas"][0], grp["betas"][1],
                      grp["eps"], grp["weight_decay"], self.step_count, decay, ema_mode, self.gnorm_sq, clip, status)
        return loss

    def check_device_status(self):
        """Raise if a kernel of an earlier step reported an error on the device (one small stream-ordered read: call it where the host
        synchronises anyway — logging, validation, checkpoints)."""
        eng = getattr(self.model, "_engine", None)
        if eng is not None:
            eng.check_attn_status()
        gs = getattr(self, "_global_status", None)
        if gs is not None and int(gs.item()) != 0:
            raise RuntimeError("od_flash_attn_bwd_fused failed on another rank of this data-parallel job (status 4): the optimizer steps since then "
                               "were skipped on every rank")

    # flat-state checkpointing (resume via --ckpt-path)
    def state_dict(self):
        return {"step_count": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step_count"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
