"""Loading trained denoiser weights onto the HIP path.

Two artefacts exist upstream:
  * a Lightning `.ckpt` written by fit-denoiser: `state_dict` keys `diffusion.*` (raw weights) and
    `diffusion_ema.module.*` (EMA weights), `hyper_parameters.diffusion_args` etc.;
  * `inference.pt` written by `export-inference` (osu_dreamer/models/inference/artifact.py:9-42):
    `hparams` {emb_dim, style_dim, latent_args{h_dim...}, diffusion_args, ...} and a `state_dict` whose
    denoiser entries are the EMA weights re-keyed to `diffusion.*`.
Both load into `osu_dreamer_amd.model.DiffusionModel` unchanged because parameter names/shapes are the
reference's.  Only the denoiser is constructed; the latent and style models are out of scope.
"""
from __future__ import annotations

from dataclasses import fields, is_dataclass
from typing import Any, Dict, Type, TypeVar

import torch

from .model import BackboneArgs, DiffusionModel, DiffusionModelArgs

T = TypeVar("T")


def dataclass_from_dict(cls: Type[T], data: Dict[str, Any]) -> T:
    """Nested dict -> dataclass, ignoring unknown keys (same contract as artifact.py:52-71)."""
    if not is_dataclass(cls):
        raise TypeError(f"{cls} is not a dataclass")
    types = {f.name: f.type for f in fields(cls)}
    kw = {}
    for k, v in data.items():
        if k not in types:
            continue
        t = types[k]
        if isinstance(t, str):
            t = {"BackboneArgs": BackboneArgs, "DiffusionModelArgs": DiffusionModelArgs}.get(t, t)
        kw[k] = dataclass_from_dict(t, v) if (is_dataclass(t) and isinstance(t, type) and isinstance(v, dict)) else v
    return cls(**kw)


def _args(d) -> DiffusionModelArgs:
    return d if isinstance(d, DiffusionModelArgs) else dataclass_from_dict(DiffusionModelArgs, dict(d))


def denoiser_from_checkpoint(path: str, use_ema: bool = True, device="cuda") -> DiffusionModel:
    """Denoiser from a fit-denoiser `.ckpt` (reference's or ours)."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    hp = ck["hyper_parameters"]
    model = DiffusionModel(hp["emb_dim"], hp["a_dim"], hp["style_dim"], _args(hp["diffusion_args"]))
    prefix = "diffusion_ema.module." if use_ema else "diffusion."
    sd = {k[len(prefix):]: v for k, v in ck["state_dict"].items() if k.startswith(prefix)}
    model.load_state_dict(sd)
    return model.to(device).eval()


def denoiser_from_inference_artifact(path: str, device="cuda") -> DiffusionModel:
    """Denoiser part of an `inference.pt` bundle."""
    art = torch.load(path, map_location="cpu", weights_only=False)
    hp = art["hparams"]
    la = hp["latent_args"]
    a_dim = la["h_dim"] if isinstance(la, dict) else la.h_dim          # inference/model.py:32
    model = DiffusionModel(hp["emb_dim"], a_dim, hp["style_dim"], _args(hp["diffusion_args"]))
    sd = {k[len("diffusion."):]: v for k, v in art["state_dict"].items() if k.startswith("diffusion.")}
    model.load_state_dict(sd)
    return model.to(device).eval()
