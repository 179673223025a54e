"""`fit-denoiser` — the training shell around `DiffusionTrainer`.

The reference drives its LightningModule with `LightningCLI` + `pytorch_lightning.Trainer`
(osu_dreamer/scripts/fit_denoiser.py:17-32) configured by models/diffusion/model.yml:3-40.
Lightning is not installed on the MI355X image, so this module implements exactly the subset
that config uses, calling the same hooks in the same order:

  seed_everything · precision bf16-mixed (autocast) · gradient_clip_val (global norm, fused into
  the optimizer pass) · AdamW + LambdaLR stepped per batch · on_train_batch_end (EMA) ·
  validation over full maps with the EMA weights · ModelCheckpoint(monitor=val/loss, mode=min,
  save_top_k=1) · resume from --ckpt-path · log_every_n_steps · devices N (one process per GPU,
  gradient exchange by osu_dreamer_amd.ddp.GradBucketReducer over RCCL).

Checkpoints keep Lightning's layout (`state_dict`, `hyper_parameters`, `optimizer_states`,
`lr_schedulers`, `global_step`, `epoch`) so `export-inference` (models/inference/artifact.py)
reads them unchanged.
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import os
import random
import time
from typing import Any, Dict, Optional

import numpy as np
import torch
import yaml

from . import _lib, launch
from .data import LatentDataModule
from .lr_schedule import LRScheduleArgs
from .model import BackboneArgs, DiffusionModelArgs
from .train import DiffusionTrainer

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model.yml")


def seed_everything(seed) -> int:
    seed = 0 if seed is True else int(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed


def _plain(o):
    if dataclasses.is_dataclass(o):
        return {k: _plain(v) for k, v in dataclasses.asdict(o).items()}
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    return o


class Trainer:
    def __init__(self, max_epochs: int = -1, max_steps: int = -1, precision: str = "bf16-mixed",
                 gradient_clip_val: Optional[float] = None, log_every_n_steps: int = 5,
                 val_check_interval: Optional[int] = None, limit_val_batches: Optional[int] = None,
                 default_root_dir: str = "runs/denoiser", accelerator: str = "gpu", devices: int = 1,
                 enable_checkpointing: bool = True, **_ignored):
        self.max_epochs, self.max_steps = max_epochs, max_steps
        self.precision = str(precision)
        self.gradient_clip_val = gradient_clip_val
        self.log_every_n_steps = log_every_n_steps
        self.val_check_interval = val_check_interval
        self.limit_val_batches = limit_val_batches
        self.root = default_root_dir
        self.enable_checkpointing = enable_checkpointing
        self.global_step, self.epoch = 0, 0
        self.best_val = float("inf")
        # `devices: N` = N ranks, one per GPU (model.yml:11).  The ranks are started by the CLI (`python -m osu_dreamer_amd
        # fit-denoiser` -> launch.spawn_ranks_if_needed) or by the caller's own torchrun, before anything touches the GPU; inside a rank
        # WORLD_SIZE and `devices` must agree in BOTH directions (devices: 1 under a 4-rank torchrun is a mistake, not a 4-GPU run).
        self.devices = launch.parse_devices(devices)
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != self.devices:
            raise RuntimeError(f"trainer.devices={self.devices} but WORLD_SIZE={self.world}: start the run with "
                               "`python -m osu_dreamer_amd fit-denoiser` (it spawns one rank per device) or under "
                               f"`torchrun --nproc-per-node {self.devices}`, and keep trainer.devices equal to the rank count")
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.history = []

    # ------------------------------------------------------------------
    def _autocast(self, device):
        # "16-mixed" (Lightning: fp16 autocast + GradScaler): here bf16 compute with the attention core on IEEE-half operands
        # (module.diffusion.attn_dtype = float16, set in _fit): accumulation is fp32 everywhere and dO is re-scaled by a power of two
        # inside od_flash_attn_bwd_fused, so no loss scaling is needed
        if self.precision.startswith("bf16") or self.precision.startswith("16"):
            return torch.autocast(device.type, dtype=torch.bfloat16)
        return torch.autocast(device.type, enabled=False)

    def _to(self, batch, device):
        return tuple(t.to(device, non_blocking=True) for t in batch)

    def save_checkpoint(self, path, module, opt, sched):
        ckpt = {
            "state_dict": module.state_dict(),
            "hyper_parameters": _plain(module.hparams_dict),
            "optimizer_states": [opt.state_dict()],
            "lr_schedulers": [sched.state_dict()],
            "global_step": self.global_step, "epoch": self.epoch, "best_val": self.best_val,
        }
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + ".tmp"
        torch.save(ckpt, tmp)
        os.replace(tmp, path)

    def validate(self, module, datamodule, device):
        sums: Dict[str, float] = {}
        n = 0
        was_training = module.training
        module.eval()                                 # Lightning runs validation in eval mode (Dropout1d off)
        for i, batch in enumerate(datamodule.val_dataloader()):
            if self.limit_val_batches is not None and i >= self.limit_val_batches:
                break
            with self._autocast(device):
                logs = module.validation_step(self._to(batch, device), i)
            for k, v in logs.items():
                sums[k] = sums.get(k, 0.0) + float(v)
            n += 1
        module.train(was_training)
        return {f"val/{k}": v / max(n, 1) for k, v in sums.items()}

    def fit(self, module: DiffusionTrainer, datamodule, ckpt_path: Optional[str] = None):
        if torch.cuda.is_available():
            torch.cuda.set_device(self.local_rank)
            device = torch.device("cuda", self.local_rank)
        elif _lib.loaded_path() not in (None, _lib.DEFAULT_SO):
            device = torch.device("cpu")     # test-suite only: kernels bound to the SIMT emulator build
        else:
            raise RuntimeError("fit-denoiser needs an MI355X: osu_dreamer_amd has no CPU path")
        reducer, own_group = None, False
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group("nccl" if device.type == "cuda" else "gloo",
                                        **({"device_id": device} if device.type == "cuda" else {}))
                own_group = True
        try:
            out = self._fit(module, datamodule, ckpt_path, device)
        except BaseException:
            # failure (possibly a dead peer): nothing that waits for the device or for other ranks — abort the RCCL communicator and let
            # the exception end the process; the launcher (torchrun agent) then stops the remaining ranks and returns non-zero
            reducer = getattr(module.diffusion, "_reducer", None)
            if reducer is not None:
                try:
                    reducer.close(abort=True)
                except Exception:
                    pass
                module.diffusion._reducer = None
            raise
        # orderly teardown: drain the device, destroy the communicator the C ABI created, then the process group — but only a group this
        # call initialised (left to interpreter exit the order is arbitrary and RCCL can hang or abort there)
        if device.type == "cuda":
            torch.cuda.synchronize()
        reducer = getattr(module.diffusion, "_reducer", None)
        if reducer is not None:
            reducer.close()
            module.diffusion._reducer = None
        if own_group:
            import torch.distributed as dist
            dist.destroy_process_group()
        return out

    def _fit(self, module, datamodule, ckpt_path, device):
        reducer = None
        module.gradient_clip_val = self.gradient_clip_val
        module.to(device)
        if self.precision.startswith("16"):
            for m in (module.diffusion, module.diffusion_ema.module):
                m.attn_dtype = torch.float16
        cfg = module.configure_optimizers()
        opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
        if ckpt_path:
            ck = torch.load(ckpt_path, map_location=device, weights_only=False)
            module.load_state_dict(ck["state_dict"])
            opt.load_state_dict(ck["optimizer_states"][0])
            sched.load_state_dict(ck["lr_schedulers"][0])
            self.global_step, self.epoch = ck["global_step"], ck["epoch"]
            self.best_val = ck.get("best_val", float("inf"))
        from .ddp import StepAgreement
        if self.world > 1:
            from .ddp import GradBucketReducer
            reducer = GradBucketReducer(module.diffusion)
            reducer.broadcast_state(opt, module.diffusion_ema, src=0)     # weights, AdamW moments, EMA: rank 0's
        agree = StepAgreement(self.world)
        log_path = os.path.join(self.root, "metrics.jsonl")
        if self.rank == 0:
            os.makedirs(self.root, exist_ok=True)
        t_last = time.time()
        done = False
        while not done:
            it = iter(datamodule.train_dataloader())
            batch_idx = -1
            while True:
                batch = next(it, None)
                # uneven shards: the epoch ends for every rank as soon as one rank has no batch left
                if not agree.all_have(batch is not None):
                    break
                batch_idx += 1
                batch = self._to(batch, device)
                opt.zero_grad()
                with self._autocast(device):
                    loss = module.training_step(batch, batch_idx)
                loss.backward()
                opt.step()
                sched.step()
                module.on_train_batch_end()
                self.global_step += 1
                if self.global_step % self.log_every_n_steps == 0:
                    opt.check_device_status()          # EVERY rank: a failed launch on one rank must end the job, not train on
                if self.global_step % self.log_every_n_steps == 0 and self.rank == 0:
                    rec = {"step": self.global_step, "epoch": self.epoch, "lr": opt.param_groups[0]["lr"],
                           "s_per_step": (time.time() - t_last) / self.log_every_n_steps,
                           **{k: float(v) for k, v in module._logged.items() if k.startswith("train/")}}
                    t_last = time.time()
                    self.history.append(rec)
                    with open(log_path, "a") as f:
                        f.write(json.dumps(rec) + "\n")
                if self.val_check_interval and self.global_step % self.val_check_interval == 0:
                    self._validate_and_checkpoint(module, datamodule, device, opt, sched)
                if 0 < self.max_steps <= self.global_step:
                    done = True
                    break
            self.epoch += 1
            if not self.val_check_interval:
                self._validate_and_checkpoint(module, datamodule, device, opt, sched)
            if 0 < self.max_epochs <= self.epoch:
                done = True
        return self.history

    def _validate_and_checkpoint(self, module, datamodule, device, opt, sched):
        opt.check_device_status()                      # never checkpoint past a failed launch
        val = self.validate(module, datamodule, device)
        if self.rank != 0:
            return
        self.history.append({"step": self.global_step, **val})
        with open(os.path.join(self.root, "metrics.jsonl"), "a") as f:
            f.write(json.dumps({"step": self.global_step, **val}) + "\n")
        if self.enable_checkpointing and val.get("val/loss", float("inf")) < self.best_val:
            self.best_val = val["val/loss"]
            self.save_checkpoint(os.path.join(self.root, "checkpoints", "best.ckpt"), module, opt, sched)


def build_from_config(cfg: Dict[str, Any]):
    m = dict(cfg["model"])
    da = dict(m["diffusion_args"])
    da["backbone_args"] = BackboneArgs(**da["backbone_args"])
    m["diffusion_args"] = DiffusionModelArgs(**da)
    m["schedule_args"] = LRScheduleArgs(**m["schedule_args"])
    module = DiffusionTrainer(**m)
    t = dict(cfg.get("trainer", {}))
    for k in ("callbacks", "logger", "accumulate_grad_batches", "enable_progress_bar", "enable_model_summary", "benchmark"):
        t.pop(k, None)
    return module, Trainer(**t)


def fit_denoiser(config: str = DEFAULT_CONFIG, ckpt_path: Optional[str] = None, **overrides):
    """begin a training run for the diffusion model (reference: scripts/fit_denoiser.py:17-32)."""
    with open(config) as f:
        cfg = yaml.safe_load(f)
    for k, v in overrides.items():          # e.g. data__data_path="..."
        sec, key = k.split("__")
        cfg.setdefault(sec, {})[key] = v
    if cfg.get("seed_everything") not in (None, False):
        seed_everything(cfg["seed_everything"])
    devices = launch.parse_devices(cfg.get("trainer", {}).get("devices", 1))
    if devices > 1 and launch.world_from_env() is None:
        # this function runs INSIDE a rank; only the CLI (main(), below) starts ranks, because that has to happen in a process
        # that never touches the GPU
        raise RuntimeError(f"fit_denoiser() with trainer.devices={devices} must run inside a rank: use `python -m osu_dreamer_amd "
                           f"fit-denoiser` (it starts the {devices} ranks) or `torchrun --nproc-per-node {devices}`")
    module, trainer = build_from_config(cfg)
    data = LatentDataModule(**cfg["data"], rank=trainer.rank, world_size=trainer.world)
    trainer.fit(module, data, ckpt_path=ckpt_path)
    return module, trainer


def main(argv=None):
    ap = argparse.ArgumentParser(prog="osu_dreamer_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)
    f = sub.add_parser("fit-denoiser", help="begin a training run for the diffusion model")
    f.add_argument("-c", "--config", default=DEFAULT_CONFIG)
    f.add_argument("--ckpt-path", default=None)
    from . import predict as predict_cmd
    predict_cmd.add_parser(sub)
    a = ap.parse_args(argv)
    if a.cmd == "predict":
        return predict_cmd.run(a)
    if a.cmd == "fit-denoiser":
        with open(a.config) as fh:
            devices = launch.parse_devices((yaml.safe_load(fh).get("trainer") or {}).get("devices", 1))
        # trainer.devices N > 1: this process only starts the N ranks (children; nothing here has touched the GPU)
        rc = launch.spawn_ranks_if_needed(devices, ["fit-denoiser", "-c", a.config] + (["--ckpt-path", a.ckpt_path] if a.ckpt_path else []),
                                          module="osu_dreamer_amd")
        if rc is not None:
            raise SystemExit(rc)
        fit_denoiser(a.config, a.ckpt_path)


if __name__ == "__main__":
    main()
