"""LDM.sample end to end on the HIP path (audio encoder -> style sampler -> denoiser sampler -> chart decoder)
against the reference's own LDM.sample outputs (tests/golden/ldm_*.npz from oracle/make_golden.py), plus the
inference-artifact round trip.  Tolerance: 1e-4 rel-L2 on the chart (the north_star sampler bound)."""
import os

import numpy as np
import pytest
import torch

from oracle import denoiser_oracle as O
from oracle import latent_oracle as LO
from oracle import style_oracle as SO
from osu_dreamer_amd.ldm import LDM, LDMArgs, ldm_args_from_dict, load_inference, save_inference
from osu_dreamer_amd.latent import LatentModelArgs, LayerArgs
from osu_dreamer_amd.model import BackboneArgs, DiffusionModelArgs
from osu_dreamer_amd.style import StyleModelArgs
from kernel_backend import dev, rel_l2  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    fx = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    lv, sv, dv = ([int(x) for x in fx[k].tolist()] for k in ("ldims", "sdims", "ddims"))
    ld = LO.LatentDims(emb_dim=lv[0], style_dim=lv[1], n_downs=lv[2], stride=lv[3], h_dim=lv[4], n_layers=lv[5], expand=lv[6], radius=lv[7])
    sd = SO.StyleDims(style_dim=sv[0], label_features=sv[1], h_dim=sv[2], depth=sv[3], expand=sv[4])
    dd = O.Dims(emb_dim=dv[0], a_dim=dv[1], style_dim=dv[2], global_cond_dim=dv[3], backbone_dim=dv[4], n_heads=dv[5],
                head_dim=dv[6], depth=dv[7], expand=dv[8], radius=dv[9], u_head_dim=dv[10])
    return fx, ld, sd, dd


def hparams(ld, sd, dd):
    """Plain nested dicts, the form an inference artifact carries."""
    return dict(emb_dim=ld.emb_dim, style_dim=ld.style_dim, n_downs=ld.n_downs, stride=ld.stride,
                latent_args=dict(h_dim=ld.h_dim, ae_args=dict(n_layers=ld.n_layers, expand=ld.expand, radius=ld.radius),
                                 style_head_dim=8, style_heads=2),
                style_args=dict(label_features=sd.label_features, h_dim=sd.h_dim, depth=sd.depth, expand=sd.expand),
                diffusion_args=dict(global_cond_dim=dd.global_cond_dim, backbone_dim=dd.backbone_dim, u_head_dim=dd.u_head_dim,
                                    backbone_args=dict(depth=dd.depth, expand=dd.expand, head_dim=dd.head_dim,
                                                       n_heads=dd.n_heads, radius=dd.radius)))


def weights(fx, ld, sd, dd):
    seed = int(fx["seed"])
    PL, PS, PD = LO.init_latent_params(ld, seed), SO.init_style_params(sd, seed + 1), O.init_params(dd, seed=seed + 2)
    return {**{"latent." + k: v for k, v in PL.items()}, **{"style." + k: v for k, v in PS.items()},
            **{"diffusion." + k: v for k, v in PD.items()}}


def make_ldm(fx, ld, sd, dd, dev):
    m = LDM(ldm_args_from_dict(hparams(ld, sd, dd)))
    assert isinstance(m.args, LDMArgs) and isinstance(m.args.latent_args, LatentModelArgs)
    assert isinstance(m.args.latent_args.ae_args, LayerArgs) and isinstance(m.args.style_args, StyleModelArgs)
    assert isinstance(m.args.diffusion_args, DiffusionModelArgs) and isinstance(m.args.diffusion_args.backbone_args, BackboneArgs)
    res = m.load_state_dict(weights(fx, ld, sd, dd), strict=False)
    assert not res.unexpected_keys
    assert all(k.startswith(("latent.chart_encoder.", "latent.style_head.", "latent.temporal_")) for k in res.missing_keys)
    return m.to(dev).eval()


@pytest.mark.parametrize("name", ["ldm_tiny", "ldm_full_d2"])
def test_ldm_sample_vs_reference(dev, name):
    fx, ld, sd, dd = load(name)
    if dev.type == "cpu" and ld.h_dim > 64:
        pytest.skip("full-width pipeline runs on the GPU only")
    m = make_ldm(fx, ld, sd, dd, dev)
    audio, labels = fx["audio"].to(dev), fx["labels"].to(dev)
    n = int(fx["num_steps"])
    chart, out_labels = m.sample(audio, labels, n, s_init=fx["s_init"].to(dev), x_init=fx["x_init"].to(dev))
    assert tuple(chart.shape) == tuple(fx["chart"].shape) == (labels.shape[0], 9, audio.shape[-1])
    assert rel_l2(chart, fx["chart"]) < 1e-4
    assert rel_l2(out_labels, fx["out_labels"]) < 1e-4
    assert float(chart[:, :7].min()) >= 0 and float(chart[:, :7].max()) <= 1
    assert float(out_labels.min()) >= 0 and float(out_labels.max()) <= 10
    # faster product mode stays inside the same bound; unpinned noise draws fresh samples of the right shape
    m.set_precision(None, "bf16x3")
    c3, _ = m.sample(audio, labels, n, s_init=fx["s_init"].to(dev), x_init=fx["x_init"].to(dev))
    assert rel_l2(c3, fx["chart"]) < 1e-4
    m.set_precision()
    c4, l4 = m.sample(audio, labels, n)
    assert tuple(c4.shape) == tuple(chart.shape) and torch.isfinite(c4).all() and not torch.equal(c4, chart)
    with pytest.raises(ValueError):
        m.sample(audio[None], labels, n)


def test_inference_artifact_roundtrip(dev, tmp_path):
    """save_inference re-keys three fit checkpoints exactly like artifact.py:9-42; load_inference reads it back."""
    fx, ld, sd, dd = load("ldm_tiny")
    hp, w = hparams(ld, sd, dd), weights(fx, ld, sd, dd)
    ref = LDM(ldm_args_from_dict(hp))
    ref.load_state_dict(w, strict=False)
    full = ref.state_dict()
    lat = {"hyper_parameters": {k: hp[k] for k in ("emb_dim", "style_dim", "n_downs", "stride", "latent_args")},
           "state_dict": {**{k: v for k, v in full.items() if k.startswith("latent.")}, "critic.w": torch.zeros(1)}}
    den = {"hyper_parameters": {"diffusion_args": hp["diffusion_args"]},
           "state_dict": {**{"diffusion_ema.module." + k[len("diffusion."):]: v for k, v in full.items() if k.startswith("diffusion.")},
                          **{k: torch.zeros_like(v) for k, v in full.items() if k.startswith("diffusion.")},
                          "diffusion_ema.n_averaged": torch.tensor(3)}}
    sty = {"hyper_parameters": {"style_args": hp["style_args"]},
           "state_dict": {"style_ema.module." + k[len("style."):]: v for k, v in full.items() if k.startswith("style.")}}
    paths = []
    for nm, ck in (("latent", lat), ("denoiser", den), ("style", sty)):
        paths.append(str(tmp_path / (nm + ".ckpt")))
        torch.save(ck, paths[-1])
    out = str(tmp_path / "inference.pt")
    save_inference(*paths, out)
    art = torch.load(out, map_location="cpu", weights_only=False)
    assert set(art) == {"hparams", "state_dict"} and set(art["hparams"]) == set(hp)
    assert "critic.w" not in art["state_dict"] and sorted(art["state_dict"]) == sorted(full)
    m = load_inference(out, device=dev)
    assert not m.training
    for k, v in m.state_dict().items():
        assert torch.equal(v.cpu(), full[k]), k
    chart, _ = m.sample(fx["audio"].to(dev), fx["labels"].to(dev), int(fx["num_steps"]), s_init=fx["s_init"].to(dev),
                        x_init=fx["x_init"].to(dev))
    assert rel_l2(chart, fx["chart"]) < 1e-4


def test_predict_entry(dev, tmp_path):
    """`predict` (scripts/predict.py:56-77 between the audio decode and the .osz packaging): artifact + spectrogram + `--diff` rows ->
    chart signals and labels, as a function and as `python -m osu_dreamer_amd predict`."""
    import numpy as np
    from osu_dreamer_amd import fit, predict as P
    fx, ld, sd, dd = load("ldm_tiny")
    hp, w = hparams(ld, sd, dd), weights(fx, ld, sd, dd)
    ref = LDM(ldm_args_from_dict(hp))
    ref.load_state_dict(w, strict=False)
    art = str(tmp_path / "inference.pt")
    torch.save({"hparams": hp, "state_dict": ref.state_dict()}, art)
    spec = fx["audio"].numpy()
    diffs = [[5.5, 9., 8., 4., 6.], [3.2, 7., 6., 4., 5.]]
    sig, lab = P.predict(art, spec, diffs, sample_steps=3, device=str(dev), seed=7)
    assert sig.shape == (2, 9, spec.shape[-1]) and lab.shape == (2, 5) and np.isfinite(sig).all()
    # the same through the CLI
    np.save(tmp_path / "song.spec.npy", spec)
    out = str(tmp_path / "pred.npz")
    argv = ["predict", "--model-path", art, "--spec", str(tmp_path / "song.spec.npy"), "--sample-steps", "3", "--seed", "7",
            "--device", str(dev), "--out", out]
    for d in diffs:
        argv += ["--diff"] + [str(x) for x in d]
    fit.main(argv)
    z = np.load(out)
    assert np.array_equal(z["pred_signals"], sig) and np.array_equal(z["pred_labels"], lab)
    with pytest.raises(ValueError):
        P.predict(art, spec, [[1., 2., 3.]], device=str(dev))
