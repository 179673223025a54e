"""Checkpoint / inference-artefact interchange (no GPU needed: weights only)."""
import dataclasses

import torch

from osu_dreamer_amd.inference import dataclass_from_dict, denoiser_from_checkpoint, denoiser_from_inference_artifact
from osu_dreamer_amd.model import BackboneArgs, DiffusionModel, DiffusionModelArgs


def _tiny():
    return DiffusionModelArgs(32, 64, BackboneArgs(2, 4, 32, 2, 2), 16)


def test_dataclass_from_dict_roundtrip():
    a = _tiny()
    b = dataclass_from_dict(DiffusionModelArgs, {**dataclasses.asdict(a), "unknown_key": 1})
    assert b == a and isinstance(b.backbone_args, BackboneArgs)


def test_ckpt_and_inference_pt_load(tmp_path):
    torch.manual_seed(0)
    src = DiffusionModel(6, 16, 8, _tiny())
    with torch.no_grad():
        for p in src.parameters():
            p.normal_(0, 0.1)
    ema = {f"diffusion_ema.module.{k}": v.clone() + 1 for k, v in src.state_dict().items()}
    raw = {f"diffusion.{k}": v.clone() for k, v in src.state_dict().items()}
    hp = dict(emb_dim=6, a_dim=16, style_dim=8, diffusion_args=dataclasses.asdict(_tiny()))
    torch.save({"state_dict": {**raw, **ema, "diffusion_ema.n_averaged": torch.tensor(3)}, "hyper_parameters": hp},
               tmp_path / "d.ckpt")
    m = denoiser_from_checkpoint(str(tmp_path / "d.ckpt"), use_ema=True, device="cpu")
    for (k, v), (_, w) in zip(m.state_dict().items(), src.state_dict().items()):
        assert torch.equal(v, w + 1), k
    m = denoiser_from_checkpoint(str(tmp_path / "d.ckpt"), use_ema=False, device="cpu")
    assert all(torch.equal(v, w) for v, w in zip(m.state_dict().values(), src.state_dict().values()))
    # export-inference layout: EMA weights re-keyed to diffusion.*, hparams with latent_args.h_dim
    art = {"hparams": dict(emb_dim=6, style_dim=8, n_downs=3, stride=3, latent_args=dict(h_dim=16),
                           style_args={}, diffusion_args=dataclasses.asdict(_tiny())),
           "state_dict": {**{f"diffusion.{k}": v for k, v in src.state_dict().items()},
                          "latent.something": torch.zeros(1), "style.other": torch.zeros(1)}}
    torch.save(art, tmp_path / "inference.pt")
    m = denoiser_from_inference_artifact(str(tmp_path / "inference.pt"), device="cpu")
    assert all(torch.equal(v, w) for v, w in zip(m.state_dict().values(), src.state_dict().values()))
    assert len(m.state_dict()) == 56 and m.a_dim == 16
