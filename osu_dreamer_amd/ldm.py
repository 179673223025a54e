"""`LDM`: the whole inference pipeline on the HIP path (osu_dreamer/models/inference/model.py:16-51,
artifact.py:9-49) — audio encoder -> style sampler -> denoiser sampler -> chart decoder, one device, no host
round trips between the stages.

Same `LDMArgs`, same sub-module names (`latent`, `style`, `diffusion`), hence the same `state_dict()` keys as the
reference's `LDM`: `load_inference(path)` takes an `inference.pt` written by the reference's `export-inference`,
and `save_inference(...)` writes one from three fit checkpoints with the reference's re-keying.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass, is_dataclass
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .inference import dataclass_from_dict
from .latent import A_DIM, LatentModel, LatentModelArgs, LayerArgs
from .model import BackboneArgs, DiffusionModel, DiffusionModelArgs
from .style import StyleModel, StyleModelArgs


@dataclass
class LDMArgs:                        # inference/model.py:16-24
    emb_dim: int
    style_dim: int
    n_downs: int
    stride: int
    latent_args: LatentModelArgs
    style_args: StyleModelArgs
    diffusion_args: DiffusionModelArgs


_NESTED = {"latent_args": LatentModelArgs, "style_args": StyleModelArgs, "diffusion_args": DiffusionModelArgs,
           "ae_args": LayerArgs, "backbone_args": BackboneArgs}


def ldm_args_from_dict(hp: dict) -> LDMArgs:
    """`hparams` of an inference artifact (nested plain dicts or dataclasses) -> LDMArgs."""
    def conv(key, v):
        cls = _NESTED.get(key)
        if is_dataclass(v) and not isinstance(v, type):
            v = asdict(v)
        if cls is None or not isinstance(v, dict):
            return v
        return dataclass_from_dict(cls, {k: conv(k, x) for k, x in v.items()})
    return LDMArgs(**{k: conv(k, hp[k]) for k in LDMArgs.__dataclass_fields__})


def pad_to_multiple(x: torch.Tensor, chunk_size: int) -> torch.Tensor:
    """Right-pad the time axis to a multiple of `chunk_size` by replication (data/modules/beatmap.py:26-30)."""
    pad = (chunk_size - x.size(-1) % chunk_size) % chunk_size
    return F.pad(x, (0, pad), mode="replicate") if pad > 0 else x


class LDM(nn.Module):
    def __init__(self, args: LDMArgs):
        super().__init__()
        if isinstance(args, dict):
            args = ldm_args_from_dict(args)
        self.args = args
        self.latent = LatentModel(args.emb_dim, args.style_dim, args.n_downs, args.stride, args.latent_args)
        self.style = StyleModel(args.style_dim, args.style_args)
        self.diffusion = DiffusionModel(args.emb_dim, args.latent_args.h_dim, args.style_dim, args.diffusion_args)
        self.requires_grad_(False)

    def set_precision(self, compute_dtype: Optional[torch.dtype] = None, f32_matmul: str = "f32"):
        """Compute dtype of the two big models (None = fp32, torch.bfloat16) and the fp32 product mode
        ("f32" exact, "bf16x3" three bf16 MFMAs per product).  The style MLP always runs fp32."""
        for m in (self.latent, self.diffusion):
            m.compute_dtype, m.f32_matmul = compute_dtype, f32_matmul

    @torch.no_grad()
    def sample(self, audio: torch.Tensor, labels: torch.Tensor, num_steps: int, show_progress: bool = False,
               *, s_init: Optional[torch.Tensor] = None, x_init: Optional[torch.Tensor] = None
               ) -> Tuple[torch.Tensor, torch.Tensor]:
        """audio (72, L) spectrogram, labels (B, 5) -> chart (B, 9, L), labels (B, 5)   (inference/model.py:34-51).
        `s_init` / `x_init` (not in the reference's signature) pin the two samplers' starting noise for tests."""
        if audio.dim() != 2 or audio.size(0) != A_DIM:
            raise ValueError(f"audio must be ({A_DIM}, L), got {tuple(audio.shape)}")
        L = audio.size(-1)
        audio = pad_to_multiple(audio.to(torch.float32), self.latent.chunk_size)
        skips, h = self.latent.audio_encoder(audio[None])
        s = self.style.sample(labels) if s_init is None else self.style.sample(labels, s_init=s_init)
        z = self.diffusion.sample(h, s, num_steps, show_progress=show_progress, x_init=x_init)
        chart, out_labels = self.latent.decode(z, s, skips=skips)
        return chart[..., :L], out_labels


def load_inference(model_path: str, device="cuda") -> LDM:
    """An `inference.pt` bundle -> LDM on `device`, eval mode (artifact.py:44-49)."""
    art = torch.load(model_path, map_location="cpu", weights_only=False)
    model = LDM(ldm_args_from_dict(art["hparams"]))
    model.load_state_dict(art["state_dict"])
    return model.to(device).eval()


def save_inference(latent_ckpt_path: str, denoiser_ckpt_path: str, style_ckpt_path: str, output_path: str):
    """Three fit checkpoints -> one inference artifact, EMA weights re-keyed exactly as artifact.py:9-42."""
    lat = torch.load(latent_ckpt_path, map_location="cpu", weights_only=False)
    den = torch.load(denoiser_ckpt_path, map_location="cpu", weights_only=False)
    sty = torch.load(style_ckpt_path, map_location="cpu", weights_only=False)
    hparams = {**{k: lat["hyper_parameters"][k] for k in ("emb_dim", "style_dim", "n_downs", "stride", "latent_args")},
               "diffusion_args": den["hyper_parameters"]["diffusion_args"],
               "style_args": sty["hyper_parameters"]["style_args"]}
    sd = {k: v for k, v in lat["state_dict"].items() if k.startswith("latent.")}
    for src, prefix, dst in ((den, "diffusion_ema.module.", "diffusion."), (sty, "style_ema.module.", "style.")):
        for k, v in src["state_dict"].items():
            if k.startswith(prefix):
                sd[dst + k[len(prefix):]] = v
    with open(output_path, "wb") as f:
        torch.save({"hparams": hparams, "state_dict": sd}, f)
