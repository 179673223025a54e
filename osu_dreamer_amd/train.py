"""`DiffusionTrainer` — drop-in for osu_dreamer/models/diffusion/train.py:33-139 on the HIP path.

Same constructor kwargs (they are the YAML keys under `model:` and the checkpoint's
`hyper_parameters`), same LightningModule hooks (`forward`, `configure_optimizers`,
`training_step`, `on_train_batch_end`, `validation_step`) and the same state-dict layout
(`diffusion.*`, `diffusion_ema.module.*`, `diffusion_ema.n_averaged`).  If
`pytorch_lightning` is importable the class derives from `pl.LightningModule`; otherwise from
`nn.Module` and `osu_dreamer_amd.fit.Trainer` drives the same hooks.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
from torch import nn

from . import ops
from .lr_schedule import LRScheduleArgs, make_lr_schedule
from .model import DiffusionModel, DiffusionModelArgs, _coerce_args
from .optim import FusedAdamWEMA

try:  # optional: the GPU image does not ship Lightning
    import pytorch_lightning as pl
    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # pragma: no cover
    _Base = nn.Module
    HAVE_LIGHTNING = False

NUM_LABELS = 5   # osu_dreamer/data/beatmap/encode.py:50


def frame_dist_sq(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Squared distance in the per-frame metric (train.py:22-31): sum over channels, mean over
    frames.  Host-side convenience for callers; the training step computes it in od_make_xt /
    od_loss_grad."""
    return (a - b).square().sum(1).mean(1)


class EMAModel(nn.Module):
    """Stand-in for torch.optim.swa_utils.AveragedModel(multi_avg_fn=get_ema_multi_avg_fn(.99))
    (train.py:67): `.module` is a DiffusionModel holding the averaged weights, `n_averaged` a long
    buffer.  The average itself is produced by the fused optimizer pass (od_adamw_ema); calling
    `update_parameters` outside that pass runs the same kernel in EMA-only form."""

    def __init__(self, model: DiffusionModel, decay: float = 0.99):
        super().__init__()
        self.module = DiffusionModel(model.emb_dim, model.a_dim, model.style_dim, model.args)
        self.module.load_state_dict(model.state_dict())
        self.module.requires_grad_(False)
        self.decay = decay
        self.register_buffer("n_averaged", torch.tensor(0, dtype=torch.long))
        self.fused_pending = 0
        self.count = 0            # host mirror of n_averaged (no device sync in the step)

    def update_parameters(self, model: DiffusionModel):
        if self.fused_pending > 0:          # already averaged inside the optimizer's pass
            self.fused_pending -= 1
        else:
            ops.ema_update(self.module.arena.data, model.arena.data, self.decay, 1 if self.count == 0 else 2)
        self.count += 1
        self.n_averaged += 1

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        if prefix + "n_averaged" in state_dict:
            self.count = int(state_dict[prefix + "n_averaged"])


class DiffusionTrainer(_Base):
    def __init__(
        self,
        # validation parameters
        val_batches: int,
        # training parameters
        opt_args: Dict[str, Any],
        schedule_args: LRScheduleArgs,
        osl_weight: float,
        del_weight: float,
        # model hparams
        emb_dim: int,
        a_dim: int,
        style_dim: int,
        diffusion_args: DiffusionModelArgs,
    ):
        super().__init__()
        if HAVE_LIGHTNING:
            self.save_hyperparameters()
        self.hparams_dict = dict(val_batches=val_batches, opt_args=opt_args, schedule_args=schedule_args,
                                 osl_weight=osl_weight, del_weight=del_weight, emb_dim=emb_dim, a_dim=a_dim,
                                 style_dim=style_dim, diffusion_args=diffusion_args)
        self.val_batches = val_batches
        self.opt_args = dict(opt_args)
        self.lr_schedule = make_lr_schedule(schedule_args)
        self.osl_weight = float(osl_weight)
        self.del_weight = float(del_weight)
        diffusion_args = _coerce_args(diffusion_args)
        self.diffusion = DiffusionModel(emb_dim, a_dim, style_dim, diffusion_args)
        self.diffusion_ema = EMAModel(self.diffusion, decay=0.99)
        self.gradient_clip_val: Optional[float] = None     # set by the trainer shell (model.yml:39)
        self._logged: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ loss (train.py:69-108)
    def forward(self, model: DiffusionModel, h, x1, s, _labels=None, *, t=None, x0=None):
        """Distance-marching loss.  `t` / `x0` may be passed to pin the noise (parity tests);
        by default they are drawn as the reference does: stratified logit-normal t, x0 ~ N(0,I)."""
        B = x1.size(0)
        dev = x1.device
        if t is None:
            u01 = (torch.randperm(B, device=dev) + torch.rand(B, device=dev)) / B
            t = torch.special.ndtri(u01.clamp(1e-6, 1 - 1e-6)).sigmoid().to(torch.float32)
        if x0 is None:
            x0 = torch.randn_like(x1, dtype=torch.float32)
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())
        out = _TrainLossFn.apply(model, self.osl_weight, self.del_weight, needs_grad, h, x1, s, t, x0,
                                 *(model.parameters() if needs_grad else ()))
        loss = out[0]
        logs = {"loss": out[0].detach(), "osl": out[1].detach(), "del": out[2].detach(), "u_mape": out[3].detach()}
        return loss, logs

    # ------------------------------------------------------------------ Lightning protocol
    def configure_optimizers(self):
        opt = FusedAdamWEMA(self.diffusion, ema=self.diffusion_ema, **self.opt_args)
        opt.max_grad_norm = self.gradient_clip_val
        return {
            "optimizer": opt,
            "lr_scheduler": {
                "scheduler": torch.optim.lr_scheduler.LambdaLR(opt, self.lr_schedule),
                "interval": "step",
            },
        }

    def _log(self, d: Dict[str, torch.Tensor]):
        self._logged.update(d)
        if HAVE_LIGHTNING and getattr(self, "_trainer", None) is not None:
            self.log_dict(d)

    def training_step(self, batch, batch_idx):
        loss, log_dict = self(self.diffusion, *batch)
        self._log({f"train/{k}": v for k, v in log_dict.items()})
        return loss

    def on_train_batch_end(self, *args, **kwargs):
        self.diffusion_ema.update_parameters(self.diffusion)

    def validation_step(self, batch, batch_idx, *args, t=None, x0=None, **kwargs):
        """train.py:128-139.  `t` / `x0` (keyword-only, not in the reference's signature) pin the noise for parity tests."""
        h, z, s, l = batch
        with torch.no_grad():
            vb = self.val_batches
            seg = z.size(-1) // vb
            bl = vb * seg
            # (1, C, vb*seg) -> (vb, C, seg)   [train.py:132-137]
            h = h[..., :bl].reshape(h.size(1), vb, seg).permute(1, 0, 2).contiguous()
            z = z[..., :bl].reshape(z.size(1), vb, seg).permute(1, 0, 2).contiguous()
            s = s.expand(vb, -1).contiguous()
            l = l.expand(vb, -1).contiguous()
            _, log_dict = self(self.diffusion_ema.module, h, z, s, l, t=t, x0=x0)
        self._log({f"val/{k}": v for k, v in log_dict.items()})
        return log_dict


class _TrainLossFn(torch.autograd.Function):
    """forward: xt = lerp(x0,x1,t) -> denoiser forward -> loss and its gradient wrt (u, v)
    (all HIP kernels).  backward: the denoiser backward, scaled by the incoming grad."""

    @staticmethod
    def forward(ctx, model: DiffusionModel, osl_w, del_w, needs_grad, h, x1, s, t, x0, *params):
        f = model._f32c
        h, x1, s, t, x0 = f(h), f(x1), f(s), f(t), f(x0)
        B, E, L = x1.shape
        dev = x1.device
        if h.shape[0] == 1 and B > 1:
            h = h.expand(B, -1, -1).contiguous()
        eng, dt = model.engine, model._dtype()
        eng.pack_weights(dt, train=needs_grad)
        eng.plan(B, L, h.shape[0], dt, train=needs_grad)
        xt = eng.buf("loss.xt", (B, E, L), torch.float32)
        dsq = eng.buf("loss.dsq", (B,), torch.float32)
        sums = eng.buf("loss.sums", (B, 3), torch.float32)
        u = eng.buf("loss.u", (B,), torch.float32)
        v = eng.buf("loss.v", (B, E, L), torch.float32)
        dv = eng.buf("loss.dv", (B, E, L), torch.float32)
        du = eng.buf("loss.du", (B,), torch.float32)
        dsq.zero_()
        sums.zero_()
        ops.make_xt(x0, x1, t, xt, dsq)
        eng.conditioning(h, s)
        eng.pred(xt, u, v)
        out = torch.empty(4, dtype=torch.float32, device=dev)
        eng._det_flush(dsq)                              # (OD_DETERMINISTIC: make_xt's per-sample sums, read by loss_grad)
        ops.loss_grad(xt, x1, u, v, dsq, dv, sums, model.c0, osl_w, del_w)
        eng._det_flush(sums)
        ops.loss_finalize(sums, dsq, u, out, du, model.c0, osl_w, del_w)
        ctx.model, ctx.style, ctx.nparams = model, s, len(params)
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def backward(ctx, g_loss, *_):
        model = ctx.model
        eng = model.engine
        model.attach_grads()
        t = eng.ws.t
        du, dv = t["loss.du"], t["loss.dv"]
        if g_loss is not None:          # autograd glue: scale the seed gradient (1.0 for loss.backward())
            du.mul_(g_loss)
            dv.mul_(g_loss)
        eng.backward(t["loss.xt"], ctx.style, du, dv, reducer=getattr(model, "_reducer", None))
        return (None,) * 9 + (None,) * ctx.nparams
