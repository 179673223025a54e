"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference)
on deterministic inputs.  Build-container only: the GPU box has no /root/reference.

    python oracle/make_golden.py            # writes tests/golden/*.npz

Nothing from the reference is copied: the fixtures hold inputs and the reference's
outputs.  Import shim: the container has python 3.10, the reference
needs 3.11 for one expression in common/rms_norm.py and imports a few packages that
are not installed (jaxtyping, pytorch_lightning, ...) only for type annotations and
base classes; we register minimal stand-in modules in sys.modules and compile
rms_norm.py with that one expression rewritten in memory.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("OSU_DREAMER_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

from oracle import denoiser_oracle as O  # noqa: E402


def _install_shims():
    class _Ann:
        def __class_getitem__(cls, item):
            return cls
    jt = types.ModuleType("jaxtyping")
    jt.Float = _Ann
    jt.Int = _Ann
    jt.Bool = _Ann
    sys.modules["jaxtyping"] = jt

    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass
    pl.LightningModule = LightningModule
    pl.LightningDataModule = LightningDataModule
    sys.modules["pytorch_lightning"] = pl

    for name in ["rosu_pp_py", "torchcodec", "torchcodec.decoders",
                 "torchcodec.decoders._audio_decoder", "resonators", "tinytag"]:
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules["torchcodec.decoders._audio_decoder"].AudioDecoder = object
    sys.modules["torchcodec.decoders"].AudioDecoder = object
    sys.modules["resonators"].ResonatorBank = object

    sys.path.insert(0, REF)
    import osu_dreamer.common  # noqa: F401  (package import only)
    path = os.path.join(REF, "osu_dreamer", "common", "rms_norm.py")
    src = open(path).read()
    bad = "gamma[:,*((None,) * (x.ndim-2))]"
    assert bad in src, "reference rms_norm.py changed; update the shim"
    src = src.replace(bad, "gamma[(slice(None),) + (None,) * (x.ndim-2)]")
    spec = importlib.util.spec_from_loader("osu_dreamer.common.rms_norm", loader=None, origin=path)
    mod = importlib.util.module_from_spec(spec)
    exec(compile(src, path, "exec"), mod.__dict__)
    sys.modules["osu_dreamer.common.rms_norm"] = mod


def _ref_model(d: O.Dims, P: O.Params):
    from osu_dreamer.models.diffusion.model import DiffusionModel, DiffusionModelArgs
    from osu_dreamer.models.diffusion.backbone import BackboneArgs
    args = DiffusionModelArgs(
        global_cond_dim=d.global_cond_dim, backbone_dim=d.backbone_dim,
        backbone_args=BackboneArgs(depth=d.depth, expand=d.expand, head_dim=d.head_dim,
                                   n_heads=d.n_heads, radius=d.radius),
        u_head_dim=d.u_head_dim)
    m = DiffusionModel(d.emb_dim, d.a_dim, d.style_dim, args)
    sd = m.state_dict()
    assert list(sd.keys()) == list(P.keys()), "oracle param inventory differs from reference"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
    m.load_state_dict(P)
    return m


def _ref_trainer(d: O.Dims, P: O.Params, val_batches=2, warmup_steps=1000, decay_start=30000):
    from osu_dreamer.models.diffusion.train import DiffusionTrainer
    from osu_dreamer.models.diffusion.model import DiffusionModelArgs
    from osu_dreamer.models.diffusion.backbone import BackboneArgs
    from osu_dreamer.common.lr_schedule import LRScheduleArgs
    args = DiffusionModelArgs(
        global_cond_dim=d.global_cond_dim, backbone_dim=d.backbone_dim,
        backbone_args=BackboneArgs(depth=d.depth, expand=d.expand, head_dim=d.head_dim,
                                   n_heads=d.n_heads, radius=d.radius),
        u_head_dim=d.u_head_dim)
    tr = DiffusionTrainer(
        val_batches=val_batches, opt_args=dict(lr=3e-4, weight_decay=0.01),
        schedule_args=LRScheduleArgs(warmup_init=.3, warmup_steps=warmup_steps, decay_start=decay_start),
        osl_weight=1., del_weight=30., emb_dim=d.emb_dim, a_dim=d.a_dim, style_dim=d.style_dim,
        diffusion_args=args)
    torch.set_float32_matmul_precision("highest")   # train.py:53 flips it globally
    tr.diffusion.load_state_dict(P)
    return tr


class _FixedNoise:
    """Make the reference's DiffusionTrainer.forward draw the (t, x0) we recorded:
    patches th.randperm / th.rand / th.randn_like inside train.py's namespace."""

    def __init__(self, train_mod, u01, x0):
        self.m, self.u01, self.x0 = train_mod, u01, x0

    def __enter__(self):
        th = self.m.th
        B = self.u01.numel()
        self._saved = (th.randperm, th.rand, th.randn_like)
        u01, x0 = self.u01, self.x0
        # u = (randperm + rand)/B  ->  make randperm return zeros and rand return u01*B
        th.randperm = lambda n, device=None: torch.zeros(n)
        th.rand = lambda n, device=None: u01 * B
        th.randn_like = lambda x: x0.clone()
        return self

    def __exit__(self, *a):
        th = self.m.th
        th.randperm, th.rand, th.randn_like = self._saved


def np_dict(**k):
    return {a: (b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)) for a, b in k.items()}


def gen_ops(out_dir):
    """Per-op fixtures straight from the reference's common/ functions."""
    from osu_dreamer.common.rms_norm import rms_norm
    from osu_dreamer.common.attn import rope, SDPSA
    from osu_dreamer.common.swiglu import SwiGLU
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 24, 37, generator=g)
    xr = torch.randn(2, 3, 19, 32, generator=g)
    att = SDPSA(48, 2, 16, d_out=48)
    ffn = SwiGLU(48, 4, 0., 2)
    with torch.no_grad():
        for p in list(att.parameters()) + list(ffn.parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        att.q_norm.weight.add_(1.0)
        att.k_norm.weight.add_(1.0)
        xa = torch.randn(2, 48, 29, generator=g)
        fx = {
            "x": x, "rms": rms_norm(x), "xr": xr, "rope": rope(xr),
            "xa": xa, "sdpsa": att(xa), "swiglu": ffn(xa),
        }
    for k, v in att.state_dict().items():
        fx["att." + k] = v
    for k, v in ffn.state_dict().items():
        fx["ffn." + k] = v
    np.savez_compressed(os.path.join(out_dir, "ops.npz"), **np_dict(**fx))


def gen_model(out_dir, name, d: O.Dims, B, L, seed, store_weights, audio_batch=None,
              num_steps=6, with_bf16=False):
    P = O.init_params(d, seed=seed)
    data = O.synthetic_batch(d, B, L, seed=seed + 1, audio_batch=audio_batch)
    m = _ref_model(d, P).eval()
    import osu_dreamer.models.diffusion.train as train_mod
    fx = {"dims": np.array(list(d.to_dict().values())), "B": B, "L": L, "seed": seed}

    with torch.no_grad():
        xt = torch.lerp(data["x0"], data["z"], data["t"][:, None, None])
        u, v = m(data["h"], data["s"], xt)
        fx.update(fwd_u=u, fwd_v=v)
        a, cg = m._precompute_conditioning(data["h"], data["s"])
        hin = m.proj_in(xt)
        l0 = m.net.layers[0](hin, a, cg)
        fx.update(cond_a=a, cond_cg=cg, layer0=l0)
        if with_bf16:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                ub, vb = m(data["h"], data["s"], xt)
            fx.update(fwd_u_bf16=ub.float(), fwd_v_bf16=vb.float())

    # sampler: feed our x_init by patching randn inside model.py's namespace
    import osu_dreamer.models.diffusion.model as model_mod
    saved = model_mod.th.randn
    model_mod.th.randn = lambda *a, **k: data["x_init"].clone()
    try:
        xs = m.sample(data["h"][:1] if audio_batch == 1 else data["h"], data["s"], num_steps)
    finally:
        model_mod.th.randn = saved
    fx.update(sample_x=xs, num_steps=num_steps)

    # training loss + grads through the reference trainer
    tr = _ref_trainer(d, P)
    B_ = data["t"].numel()
    u01 = torch.special.ndtr(torch.logit(data["t"].double())).float()
    with _FixedNoise(train_mod, u01, data["x0"]):
        tr.zero_grad()
        loss, logs = tr(tr.diffusion, data["h"], data["z"], data["s"], torch.zeros(B_, 5))
    # the t the reference derived from our u01 (ndtri∘ndtr round trip is not exact) is
    # recomputed the same way here and stored, so the oracle/HIP side uses identical t.
    u_eff = (torch.zeros(B_) + u01 * B_) / B_          # exactly what train.py:79 computes
    t_used = torch.special.ndtri(u_eff.clamp(1e-6, 1 - 1e-6)).sigmoid()
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
             for k, p in tr.diffusion.named_parameters()}
    fx.update(t_used=t_used, loss=loss.detach(), **{"log_" + k: v for k, v in logs.items()})
    gnorm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    fx.update(grad_norm=gnorm)
    if store_weights:
        for k, g in grads.items():
            fx["grad." + k] = g
    else:
        # full-size: keep per-tensor norms and a strided sub-sample of each gradient
        for k, g in grads.items():
            fx["gradnorm." + k] = g.norm()
            fx["gradsub." + k] = g.flatten()[::max(1, g.numel() // 64)][:64]

    # one optimizer + EMA step with the reference's own optimizer / AveragedModel
    cfg = tr.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    torch.nn.utils.clip_grad_norm_(tr.parameters(), 1.0)
    opt.step()
    sched.step()
    tr.on_train_batch_end()
    # second step on the same grads to exercise the EMA lerp branch and bias correction
    opt.step()
    sched.step()
    tr.on_train_batch_end()
    p2 = dict(tr.diffusion.named_parameters())
    e2 = dict(tr.diffusion_ema.module.named_parameters())
    if store_weights:
        for k in p2:
            fx["p2." + k] = p2[k].detach()
            fx["ema2." + k] = e2[k].detach()
        for k, w in P.items():
            fx["w." + k] = w
        for k, t_ in data.items():
            fx["in." + k] = t_
    else:
        for k in p2:
            fx["p2sub." + k] = p2[k].detach().flatten()[::max(1, p2[k].numel() // 64)][:64]
            fx["ema2sub." + k] = e2[k].detach().flatten()[::max(1, e2[k].numel() // 64)][:64]
    fx["lr_after"] = np.array([g["lr"] for g in opt.param_groups])
    fx["n_averaged"] = tr.diffusion_ema.n_averaged
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **np_dict(**fx))
    print(name, "loss", float(loss.detach()), "u", u.tolist()[:3], "gnorm", float(gnorm))


def gen_lr(out_dir):
    from osu_dreamer.common.lr_schedule import LRScheduleArgs, make_lr_schedule
    f = make_lr_schedule(LRScheduleArgs(warmup_init=.3, warmup_steps=1000, decay_start=30000))
    steps = np.array([0, 1, 10, 500, 999, 1000, 1001, 29999, 30000, 30001, 60000, 120000])
    np.savez(os.path.join(out_dir, "lr_schedule.npz"), steps=steps,
             mult=np.array([f(int(s)) for s in steps]))


def gen_style(out_dir, name, d, B, seed):
    """Style-model fixtures from the reference's StyleModel (models/style/model.py)."""
    from oracle import style_oracle as SO
    from osu_dreamer.models.style.model import StyleModel, StyleModelArgs
    import osu_dreamer.models.style.model as style_mod
    P = SO.init_style_params(d, seed)
    m = StyleModel(d.style_dim, StyleModelArgs(label_features=d.label_features, h_dim=d.h_dim, depth=d.depth, expand=d.expand))
    sd = m.state_dict()
    assert sorted(sd.keys()) == sorted(P.keys()), (sorted(sd.keys()), sorted(P.keys()))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
    m.load_state_dict(P)
    m.eval()
    g = torch.Generator().manual_seed(seed + 1)
    labels = torch.rand(B, 5, generator=g) * 10
    labels[0, 1] = -1.0                      # a dropped label (null embedding path)
    if B > 2:
        labels[2, :] = -1.0
    st = torch.randn(B, d.style_dim, generator=g)
    s_init = torch.randn(B, d.style_dim, generator=g)
    with torch.no_grad():
        c = m.compute_conditioning(labels)
        u, v = m(st, labels)
    saved = style_mod.th.randn
    style_mod.th.randn = lambda *a, **k: s_init.clone()
    try:
        ss = m.sample(labels, 16)
    finally:
        style_mod.th.randn = saved
    fx = {"dims": np.array([d.style_dim, d.label_features, d.h_dim, d.depth, d.expand]), "seed": seed, "B": B,
          "labels": labels, "st": st, "s_init": s_init, "cond": c, "fwd_u": u, "fwd_v": v, "sample_s": ss}
    if d.h_dim <= 64:
        for k, w in P.items():
            fx["w." + k] = w
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **np_dict(**fx))
    print(name, "u", u.tolist()[:3])


def gen_latent(out_dir, name, d, B, Ba, L, seed):
    """Latent-model inference-path fixtures from the reference's LatentModel (models/latent/model.py):
    audio_encoder -> (skips, h) and decode(z, s, skips=...) -> (chart, labels)."""
    from oracle import latent_oracle as LO
    from osu_dreamer.models.latent.model import LatentModel, LatentModelArgs
    from osu_dreamer.models.latent.unet import LayerArgs
    P = LO.init_latent_params(d, seed)
    PE = LO.init_latent_encoder_params(d, seed, 8, 2)
    m = LatentModel(d.emb_dim, d.style_dim, d.n_downs, d.stride,
                    LatentModelArgs(h_dim=d.h_dim, ae_args=LayerArgs(n_layers=d.n_layers, expand=d.expand, radius=d.radius),
                                    style_head_dim=8, style_heads=2))
    sd = m.state_dict()
    for k, w in {**P, **PE}.items():
        assert k in sd and tuple(sd[k].shape) == tuple(w.shape), k
    on_path = ("audio_encoder.", "proj_emb.", "decoder.", "proj_out.", "label_predictor.")
    assert sorted(k for k in sd if k.startswith(on_path)) == sorted(P.keys())
    assert sorted(sd.keys()) == sorted({**P, **PE}.keys())       # the two oracle inventories are the whole state dict
    m.load_state_dict({**P, **PE})
    m.eval()
    assert L % d.chunk_size == 0
    g = torch.Generator().manual_seed(seed + 1)
    audio = torch.randn(Ba, LO.A_DIM, L, generator=g)
    z = torch.randn(B, d.emb_dim, L // d.chunk_size, generator=g)
    z = z * z.pow(2).mean(1, keepdim=True).add(1e-6).rsqrt()
    s = torch.randn(B, d.style_dim, generator=g)
    s = s * s.pow(2).mean(1, keepdim=True).add(1e-6).rsqrt()
    with torch.no_grad():
        skips, h = m.audio_encoder(audio)
        feat = m.audio_encoder[0](audio)
        chart, labels = m.decode(z, s, skips=list(skips))
        chart2, _ = m.decode(z, s, audio=audio)
        assert torch.equal(chart, chart2)
        # the dataset-encoding direction on a synthetic chart (hit signals in [0,1], cursor in [-1,1])
        chart_in = torch.cat([torch.rand(B, 7, L, generator=g), torch.rand(B, 2, L, generator=g) * 2 - 1], dim=1)
        enc_z, enc_s = m.encode_chart(chart_in)
        o_z, o_s = LO.encode_chart({**P, **PE}, chart_in, d, 2)
        # the restatement against the reference, on the spot
        o_skips, o_h = LO.audio_encoder(P, audio, d)
        o_chart, o_labels = LO.decode(P, z, s, o_skips, d)
    def rel(a, b):
        return float((a - b).norm() / b.norm())
    assert rel(o_h, h) < 2e-6 and rel(o_chart, chart) < 2e-6 and rel(o_labels, labels) < 2e-6, (rel(o_h, h), rel(o_chart, chart))
    assert rel(o_z, enc_z) < 2e-6 and rel(o_s, enc_s) < 2e-6, (rel(o_z, enc_z), rel(o_s, enc_s))
    fx = {"chart_in": chart_in, "enc_z": enc_z, "enc_s": enc_s,"dims": np.array([d.emb_dim, d.style_dim, d.n_downs, d.stride, d.h_dim, d.n_layers, d.expand, d.radius]),
          "seed": seed, "audio": audio, "z": z, "s": s, "feat": feat, "h": h, "chart": chart, "labels": labels}
    for i, sk in enumerate(skips):
        fx[f"skip{i}"] = sk
    if d.h_dim <= 64:
        for k, w in {**P, **PE}.items():
            fx["w." + k] = w
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **np_dict(**fx))
    print(name, "oracle-vs-reference", rel(o_h, h), rel(o_chart, chart), rel(o_z, enc_z), rel(o_s, enc_s))


def gen_ldm(out_dir, name, ld, sd_, dd, B, L, num_steps, seed):
    """End-to-end fixture from the reference's LDM.sample (models/inference/model.py:34-51): audio (72, L) and
    labels in, chart and labels out, with the two samplers' initial noise pinned (th.randn is one torch
    function shared by both model modules, so it is patched once and answers by requested shape)."""
    from oracle import latent_oracle as LO, style_oracle as SO
    from osu_dreamer.models.inference.model import LDM, LDMArgs
    from osu_dreamer.models.latent.model import LatentModelArgs
    from osu_dreamer.models.latent.unet import LayerArgs
    from osu_dreamer.models.style.model import StyleModelArgs
    from osu_dreamer.models.diffusion.model import DiffusionModelArgs
    from osu_dreamer.models.diffusion.backbone import BackboneArgs
    assert dd.a_dim == ld.h_dim and dd.style_dim == ld.style_dim == sd_.style_dim and dd.emb_dim == ld.emb_dim
    PL, PS, PD = LO.init_latent_params(ld, seed), SO.init_style_params(sd_, seed + 1), O.init_params(dd, seed=seed + 2)
    m = LDM(LDMArgs(
        emb_dim=ld.emb_dim, style_dim=ld.style_dim, n_downs=ld.n_downs, stride=ld.stride,
        latent_args=LatentModelArgs(h_dim=ld.h_dim, ae_args=LayerArgs(ld.n_layers, ld.expand, ld.radius), style_head_dim=8, style_heads=2),
        style_args=StyleModelArgs(label_features=sd_.label_features, h_dim=sd_.h_dim, depth=sd_.depth, expand=sd_.expand),
        diffusion_args=DiffusionModelArgs(global_cond_dim=dd.global_cond_dim, backbone_dim=dd.backbone_dim, u_head_dim=dd.u_head_dim,
                                          backbone_args=BackboneArgs(depth=dd.depth, expand=dd.expand, head_dim=dd.head_dim,
                                                                     n_heads=dd.n_heads, radius=dd.radius))))
    full = {**{"latent." + k: v for k, v in PL.items()}, **{"style." + k: v for k, v in PS.items()},
            **{"diffusion." + k: v for k, v in PD.items()}}
    res = m.load_state_dict(full, strict=False)
    assert not res.unexpected_keys
    assert all(k.startswith(("latent.chart_encoder.", "latent.style_head.", "latent.temporal_")) for k in res.missing_keys)
    m.eval()
    g = torch.Generator().manual_seed(seed + 3)
    audio = torch.randn(LO.A_DIM, L, generator=g)
    labels = torch.rand(B, 5, generator=g) * 10
    labels[0, 2] = -1.0
    Lp = (L + ld.chunk_size - 1) // ld.chunk_size * ld.chunk_size
    s_init = torch.randn(B, ld.style_dim, generator=g)
    x_init = torch.randn(B, ld.emb_dim, Lp // ld.chunk_size, generator=g)
    by_shape = {tuple(s_init.shape): s_init, tuple(x_init.shape): x_init}
    saved = torch.randn
    torch.randn = lambda *a, **k: by_shape[tuple(a[0]) if len(a) == 1 and not isinstance(a[0], int) else tuple(a)].clone()
    try:
        chart, out_labels = m.sample(audio, labels, num_steps)
    finally:
        torch.randn = saved
    fx = {"ldims": np.array([ld.emb_dim, ld.style_dim, ld.n_downs, ld.stride, ld.h_dim, ld.n_layers, ld.expand, ld.radius]),
          "sdims": np.array([sd_.style_dim, sd_.label_features, sd_.h_dim, sd_.depth, sd_.expand]),
          "ddims": np.array([dd.emb_dim, dd.a_dim, dd.style_dim, dd.global_cond_dim, dd.backbone_dim, dd.n_heads, dd.head_dim,
                             dd.depth, dd.expand, dd.radius, dd.u_head_dim]),
          "seed": seed, "num_steps": num_steps, "audio": audio, "labels": labels, "s_init": s_init, "x_init": x_init,
          "chart": chart, "out_labels": out_labels}
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **np_dict(**fx))
    print(name, "chart", tuple(chart.shape), float(chart.abs().mean()))



def _grad_store(fx, prefix, grads, full):
    for k, g in grads.items():
        if full:
            fx[f"{prefix}." + k] = g
        else:
            fx[f"{prefix}norm." + k] = g.norm()
            fx[f"{prefix}sub." + k] = g.flatten()[::max(1, g.numel() // 64)][:64]


def gen_train_bf16(out_dir, name, d: O.Dims, B, L, seed, store_full):
    """The reference's training loss and gradients under `precision: bf16-mixed` (model.yml:12), i.e.
    DiffusionTrainer.forward + backward inside torch.autocast(bfloat16), next to the same quantities in fp32
    on the same weights / batch / noise (same seeds as the fp32 fixture of that config).  The pair anchors the
    tolerance of the HIP bf16 training step: its error against fp32 must stay within a small multiple of the
    reference's own bf16-vs-fp32 error."""
    import osu_dreamer.models.diffusion.train as train_mod
    P = O.init_params(d, seed=seed)
    data = O.synthetic_batch(d, B, L, seed=seed + 1)
    fx = {"dims": np.array(list(d.to_dict().values())), "B": B, "L": L, "seed": seed}
    u01 = torch.special.ndtr(torch.logit(data["t"].double())).float()
    u_eff = (torch.zeros(B) + u01 * B) / B
    fx["t_used"] = torch.special.ndtri(u_eff.clamp(1e-6, 1 - 1e-6)).sigmoid()
    for tag, ctx in (("f32", torch.autocast("cpu", enabled=False)), ("bf16", torch.autocast("cpu", dtype=torch.bfloat16))):
        tr = _ref_trainer(d, P)
        with _FixedNoise(train_mod, u01, data["x0"]):
            tr.zero_grad()
            with ctx:
                loss, logs = tr(tr.diffusion, data["h"], data["z"], data["s"], torch.zeros(B, 5))
        loss.backward()
        grads = {k: (p.grad.detach().float().clone() if p.grad is not None else torch.zeros_like(p))
                 for k, p in tr.diffusion.named_parameters()}
        fx[f"{tag}.loss"] = loss.detach().float()
        for k, v in logs.items():
            fx[f"{tag}.log_{k}"] = v.float()
        fx[f"{tag}.grad_norm"] = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
        _grad_store(fx, f"{tag}.grad", grads, store_full)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **np_dict(**fx))
    print(name, "loss f32", float(fx["f32.loss"]), "bf16", float(fx["bf16.loss"]),
          "gnorm f32", float(fx["f32.grad_norm"]), "bf16", float(fx["bf16.grad_norm"]))


def gen_train_f16(out_dir, name, d: O.Dims, B, L, seed, store_full):
    """The reference's training loss and gradients under `precision: 16-mixed` (the fp16 option of the trainer key model.yml:12;
    BASELINE configs[4] "attention in fp16"): DiffusionTrainer.forward + backward inside torch.autocast(float16), the loss multiplied by a
    GradScaler-style power of two before backward (starting at 2^16 and halved until every gradient is finite, as Lightning's scaler
    does over its first steps) and the gradients divided by it again — next to the fp32 and bf16-autocast results of the same weights /
    batch / noise.  The HIP step with attn_dtype = float16 (bf16 compute around a half-operand attention core) is held to the fp16 run
    within the reference's own bf16 noise."""
    import osu_dreamer.models.diffusion.train as train_mod
    P = O.init_params(d, seed=seed)
    data = O.synthetic_batch(d, B, L, seed=seed + 1)
    fx = {"dims": np.array(list(d.to_dict().values())), "B": B, "L": L, "seed": seed}
    u01 = torch.special.ndtr(torch.logit(data["t"].double())).float()
    u_eff = (torch.zeros(B) + u01 * B) / B
    fx["t_used"] = torch.special.ndtri(u_eff.clamp(1e-6, 1 - 1e-6)).sigmoid()
    for tag, mk_ctx in (("f32", lambda: torch.autocast("cpu", enabled=False)), ("bf16", lambda: torch.autocast("cpu", dtype=torch.bfloat16)),
                        ("f16", lambda: torch.autocast("cpu", dtype=torch.float16))):
        scale = 65536.0 if tag == "f16" else 1.0
        while True:
            tr = _ref_trainer(d, P)
            with _FixedNoise(train_mod, u01, data["x0"]):
                tr.zero_grad()
                with mk_ctx():
                    loss, logs = tr(tr.diffusion, data["h"], data["z"], data["s"], torch.zeros(B, 5))
            (loss * scale).backward()
            grads = {k: (p.grad.detach().float().clone() / scale if p.grad is not None else torch.zeros_like(p))
                     for k, p in tr.diffusion.named_parameters()}
            if all(bool(torch.isfinite(g).all()) for g in grads.values()) or scale <= 1.0:
                break
            scale /= 2
        fx[f"{tag}.loss_scale"] = scale
        fx[f"{tag}.loss"] = loss.detach().float()
        for k, v in logs.items():
            fx[f"{tag}.log_{k}"] = v.float()
        fx[f"{tag}.grad_norm"] = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
        _grad_store(fx, f"{tag}.grad", grads, store_full)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **np_dict(**fx))
    print(name, "loss f32", float(fx["f32.loss"]), "bf16", float(fx["bf16.loss"]), "f16", float(fx["f16.loss"]), "loss scale", fx["f16.loss_scale"],
          "gnorm f32", float(fx["f32.grad_norm"]), "bf16", float(fx["bf16.grad_norm"]), "f16", float(fx["f16.grad_norm"]))


def gen_round5(out_dir):
    # head_dim 64 (the half-operand attention core exists for it only): a small two-head model and the full-width model at depth 2
    # (per tensor: the gradient's norm and a 64-element sub-sample, for the fp32, bf16-autocast and fp16-autocast runs)
    gen_train_f16(out_dir, "train_f16_small_hd64_b2_l230", O.Dims(global_cond_dim=64, backbone_dim=128, n_heads=2, head_dim=64, depth=2, expand=2,
                                                                  radius=1, u_head_dim=16), B=2, L=230, seed=1500, store_full=False)
    gen_train_f16(out_dir, "train_f16_full_d2_b2_l96", O.Dims(depth=2), B=2, L=96, seed=1600, store_full=False)


TRAJ_WARMUP, TRAJ_DECAY = 12, 24          # the trajectory fixtures' schedule: warm-up, plateau and the start of the decay all fall inside 40 steps


def gen_trajectory(out_dir, name, d: O.Dims, B, L, seed, steps=40):
    """A TRAINING TRAJECTORY of the reference (models/diffusion/train.py:69-126): `steps` consecutive optimizer steps of DiffusionTrainer —
    forward + loss (fresh batch, t and x0 per step, regenerated from seed + 10 i on both sides), backward, clip 1.0 (model.yml:39),
    AdamW (train.py:110-121: bias correction, weight decay), the LR schedule (common/lr_schedule.py:10-21; warm-up 12 steps, decay from
    step 24 so that all three branches are walked) and the EMA copy-then-lerp (train.py:125-126) — once in fp32 and once with the forward
    under torch.autocast(bfloat16) (`precision: bf16-mixed`, model.yml:12).  Stored: the loss terms and the learning rate of every step,
    and of the final weights and EMA weights each tensor's norm and a 64-element sub-sample, per run."""
    import osu_dreamer.models.diffusion.train as train_mod
    P = O.init_params(d, seed=seed)
    fx = {"dims": np.array(list(d.to_dict().values())), "B": B, "L": L, "seed": seed, "steps": steps,
          "warmup_steps": TRAJ_WARMUP, "decay_start": TRAJ_DECAY}
    t_used = []
    for tag, mk_ctx in (("f32", lambda: torch.autocast("cpu", enabled=False)), ("bf16", lambda: torch.autocast("cpu", dtype=torch.bfloat16))):
        tr = _ref_trainer(d, P, warmup_steps=TRAJ_WARMUP, decay_start=TRAJ_DECAY)
        cfg = tr.configure_optimizers()
        opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
        losses, logs_all, lrs, gnorms = [], {}, [], []
        for i in range(steps):
            data = O.synthetic_batch(d, B, L, seed=seed + 1 + 10 * i)
            u01 = torch.special.ndtr(torch.logit(data["t"].double())).float()
            if tag == "f32":
                u_eff = (torch.zeros(B) + u01 * B) / B
                t_used.append(torch.special.ndtri(u_eff.clamp(1e-6, 1 - 1e-6)).sigmoid())
            lrs.append(opt.param_groups[0]["lr"])          # the rate this step's update uses
            with _FixedNoise(train_mod, u01, data["x0"]):
                opt.zero_grad()
                with mk_ctx():
                    loss, logs = tr(tr.diffusion, data["h"], data["z"], data["s"], torch.zeros(B, 5))
            loss.backward()
            gnorms.append(float(torch.nn.utils.clip_grad_norm_(tr.parameters(), 1.0)))
            opt.step()
            sched.step()
            tr.on_train_batch_end()
            losses.append(float(loss.detach()))
            for k, v in logs.items():
                logs_all.setdefault(k, []).append(float(v))
        fx[f"{tag}.loss"] = np.array(losses, dtype=np.float64)
        fx[f"{tag}.grad_norm"] = np.array(gnorms, dtype=np.float64)
        fx[f"{tag}.lr"] = np.array(lrs, dtype=np.float64)
        for k, v in logs_all.items():
            fx[f"{tag}.log_{k}"] = np.array(v, dtype=np.float64)
        fx[f"{tag}.n_averaged"] = tr.diffusion_ema.n_averaged
        pw = dict(tr.diffusion.named_parameters())
        ew = dict(tr.diffusion_ema.module.named_parameters())
        for k in pw:
            for pre, w in (("p", pw[k]), ("ema", ew[k])):
                w = w.detach().float()
                fx[f"{tag}.{pre}norm." + k] = w.norm()
                fx[f"{tag}.{pre}sub." + k] = w.flatten()[::max(1, w.numel() // 64)][:64]
            fx[f"{tag}.dnorm." + k] = (pw[k].detach() - P[k]).norm()       # how far the tensor moved: the scale of the comparison
    fx["t_used"] = torch.stack(t_used)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **np_dict(**fx))
    print(name, "loss f32", fx["f32.loss"][[0, steps // 2, -1]], "bf16", fx["bf16.loss"][[0, steps // 2, -1]], "lr", fx["f32.lr"][[0, TRAJ_WARMUP, -1]])


def gen_round6(out_dir):
    gen_trajectory(out_dir, "traj40_tiny_b3_l40", O.TINY, B=3, L=40, seed=1700)
    # head_dim 64, two heads: the shape class whose bf16 step runs the MFMA attention kernels the bench runs
    gen_trajectory(out_dir, "traj40_small_hd64_b2_l130", O.Dims(global_cond_dim=64, backbone_dim=128, n_heads=2, head_dim=64, depth=2, expand=2,
                                                                radius=1, u_head_dim=16), B=2, L=130, seed=1800)


def gen_validation(out_dir, name, d: O.Dims, val_batches, l, seed):
    """The reference's `validation_step` (train.py:128-139): one full map (1, C, l) cut into `val_batches` segments,
    loss under no_grad with the EMA weights.  The EMA copy is given DIFFERENT weights from the live model so that a
    step evaluating the wrong one cannot match.  `l` is deliberately not a multiple of `val_batches`."""
    import osu_dreamer.models.diffusion.train as train_mod
    P, Pe = O.init_params(d, seed=seed), O.init_params(d, seed=seed + 7)
    g = torch.Generator().manual_seed(seed + 1)
    h = torch.randn(1, d.a_dim, l, generator=g)
    z = torch.randn(1, d.emb_dim, l, generator=g)
    z = z * z.pow(2).mean(1, keepdim=True).add(1e-6).rsqrt()
    s = torch.randn(1, d.style_dim, generator=g)
    labels = torch.rand(1, 5, generator=g) * 10
    seg = l // val_batches
    t = torch.rand(val_batches, generator=g) * 0.9 + 0.05
    x0 = torch.randn(val_batches, d.emb_dim, seg, generator=g)
    u01 = torch.special.ndtr(torch.logit(t.double())).float()
    u_eff = (torch.zeros(val_batches) + u01 * val_batches) / val_batches
    tr = _ref_trainer(d, P, val_batches=val_batches)
    tr.diffusion_ema.module.load_state_dict(Pe)
    logged = {}
    tr.log_dict = lambda dct, *a, **k: logged.update({kk: vv.detach().clone() for kk, vv in dct.items()})
    with _FixedNoise(train_mod, u01, x0):
        tr.validation_step((h, z, s, labels), 0)
    fx = {"dims": np.array(list(d.to_dict().values())), "seed": seed, "val_batches": val_batches, "l": l,
          "h": h, "z": z, "s": s, "labels": labels, "x0": x0,
          "t_used": torch.special.ndtri(u_eff.clamp(1e-6, 1 - 1e-6)).sigmoid()}
    for k, v in logged.items():
        fx["log." + k.replace("/", "_")] = v
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **np_dict(**fx))
    print(name, {k: float(v) for k, v in logged.items()})


def gen_rope_long(out_dir, N=32768, D=64, seed=900):
    """`rope()` (common/attn.py:12-29) at BASELINE configs[4]'s length: positions up to 32767, where one ulp of a
    high inverse frequency is already 2e-3 rad.  Rows are sub-sampled (input and output stored for the same rows)."""
    from osu_dreamer.common.attn import rope, _rope_cache
    g = torch.Generator().manual_seed(seed)
    pos = torch.cat([torch.arange(0, 8), torch.tensor([63, 64, 1000, 1115, 4095, 4096, 8191, 8192, 12345, 16384, 20000,
                                                       30000, 32766, 32767])])
    x = torch.zeros(1, 2, N, D)
    x[:, :, pos] = torch.randn(1, 2, len(pos), D, generator=g)
    with torch.no_grad():
        y = rope(x)
    freqs = _rope_cache[(D, x.device)][:N]
    np.savez_compressed(os.path.join(out_dir, "rope_long.npz"),
                        **np_dict(N=N, D=D, pos=pos, x=x[:, :, pos], y=y[:, :, pos], angle=freqs[pos],
                                  inv_freq=10000 ** (torch.arange(0, D, 2).float() / -D)))
    print("rope_long", tuple(y[:, :, pos].shape))


def gen_feeder(out_dir, seed=1000):
    """The sample stream of the reference's LatentDataset (data/modules/latent.py:83-149) on a synthetic dataset written
    by OUR writer (osu_dreamer_amd.data.write_synthetic_dataset: deterministic numpy, re-created by the test): window
    offsets, per-map permutation, max_per_map cut, shuffle buffer, and the train/val mapset split of
    data/modules/beatmap.py:33-71.  world_size 1, no workers, torch.manual_seed(seed) before each iteration."""
    import tempfile
    from pathlib import Path
    from osu_dreamer.data.modules.latent import LatentDataset as RefDataset
    from osu_dreamer.data.modules.beatmap import hold_out_mapsets as ref_split
    from osu_dreamer_amd.data import write_synthetic_dataset
    frames = [200, 96, 333, 64, 150, 47, 260]
    fx = {"seed": seed, "frames": np.array(frames), "a_dim": 4, "emb_dim": 3, "style_dim": 2}
    with tempfile.TemporaryDirectory() as tmp:
        write_synthetic_dataset(tmp, n_maps=len(frames), frames=frames, a_dim=4, emb_dim=3, style_dim=2, seed=seed)
        root = Path(tmp)
        mapsets = sorted(root.iterdir())
        full = {m.name: (np.load(m / "h.npy"), np.load(m / "0.latent.npz")["z"]) for m in mapsets}
        # the split: the reference walks data_dir.iterdir() (unsorted); record it by mapset name
        tr_sets, val_sets = ref_split(root, "*.latent.npz", 3, .3)
        fx["split_listing"] = np.array([m.name for m in root.iterdir()])
        fx["split_train"] = np.array([m.name for m in tr_sets])
        fx["split_val"] = np.array([m.name for m in val_sets])
        for tag, kw in (("plain", dict(seq_len=48, shuffle_buffer_size=1, max_per_map=-1)),
                        ("shuffled", dict(seq_len=32, shuffle_buffer_size=4, max_per_map=3)),
                        ("fullmaps", dict(seq_len=None))):
            torch.manual_seed(seed)
            ds = RefDataset(mapsets, **kw)
            recs, sums = [], []
            for smp in ds:
                name = next(n for n, (hh, zz) in full.items() if np.allclose(np.load(root / n / "0.latent.npz")["s"], smp.s.numpy()))
                hh, zz = full[name]
                l = smp.z.shape[-1]
                starts = [i for i in range(zz.shape[-1] - l + 1) if np.array_equal(zz[:, i:i + l], smp.z.numpy())]
                assert len(starts) == 1 and np.array_equal(hh[:, starts[0]:starts[0] + l], smp.h.numpy())
                recs.append((int(name), starts[0], l))
                sums.append(float(smp.h.double().sum() + smp.z.double().sum()))
            fx[f"{tag}.stream"] = np.array(recs)
            fx[f"{tag}.sums"] = np.array(sums)
    np.savez_compressed(os.path.join(out_dir, "feeder_stream.npz"), **fx)
    print("feeder", {k: v.shape for k, v in fx.items() if hasattr(v, "shape") and k.endswith("stream")})


def gen_round2(out_dir):
    gen_rope_long(out_dir)
    gen_feeder(out_dir)
    gen_train_bf16(out_dir, "train_bf16_tiny_b3_l40", O.TINY, B=3, L=40, seed=100, store_full=True)
    gen_train_bf16(out_dir, "train_bf16_full_d2_b2_l96", O.Dims(depth=2), B=2, L=96, seed=300, store_full=False)
    gen_validation(out_dir, "val_tiny", O.TINY, val_batches=3, l=50, seed=1100)
    gen_validation(out_dir, "val_full_d2", O.Dims(depth=2), val_batches=8, l=8 * 24 + 5, seed=1200)


def gen_sampler50(out_dir, name, d: O.Dims, B, L, seed, num_steps=50, with_bf16=True):
    """BASELINE configs[3] where north_star claims it: the reference's DiffusionModel.sample (models/diffusion/model.py:117-138)
    for `num_steps` steps (num_steps + 1 network evaluations) with batch-1 audio broadcast against B styles
    ('#B A l', model.py:120).  Weights and inputs are regenerated from the seed on both sides; the fixture holds the
    whole (B, E, L) output (~100 KB), the reference's u0 / eta (recomputed the way model.py:131-132 does), the first
    evaluation's (u, v), and -- as the anchor of the bf16 tolerance -- the same call under torch.autocast(bfloat16)."""
    import osu_dreamer.models.diffusion.model as model_mod
    P = O.init_params(d, seed=seed)
    data = O.synthetic_batch(d, B, L, seed=seed + 1, audio_batch=1)
    m = _ref_model(d, P).eval()
    fx = {"dims": np.array(list(d.to_dict().values())), "B": B, "L": L, "seed": seed, "num_steps": num_steps}
    saved = model_mod.th.randn
    model_mod.th.randn = lambda *a, **k: data["x_init"].clone()
    try:
        import time
        t0 = time.time()
        xs = m.sample(data["h"], data["s"], num_steps)
        fx["ref_cpu_seconds"] = time.time() - t0
        with torch.no_grad():
            u0, v0 = m(data["h"], data["s"], data["x_init"])
        c0 = float(m.c0)
        eta = 1 - (c0 ** .5 / max(float(u0.mean()), c0 ** .5 + 1e-6)) ** (1 / num_steps)
        fx.update(sample_x=xs, u0=u0, v0=v0, eta=eta)
        if with_bf16:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                xb = m.sample(data["h"], data["s"], num_steps)
            fx["sample_x_bf16"] = xb.float()
    finally:
        model_mod.th.randn = saved
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **np_dict(**fx))
    msg = f"{name}: |x| {float(xs.norm()):.4f} eta {eta:.6f} ref cpu {fx['ref_cpu_seconds']:.1f}s"
    if with_bf16:
        msg += f" bf16-vs-f32 rel-L2 {float((xb.float() - xs).norm() / xs.norm()):.3e}"
    print(msg, flush=True)


def gen_round3(out_dir):
    # 50 steps of the full 46.9 M-parameter model: at a short length, and at configs[3] (B=4, L=1115: a 3-minute song)
    gen_sampler50(out_dir, "sample50_full_d8_b2_l64", O.FULL, B=2, L=64, seed=1300)
    gen_sampler50(out_dir, "sample50_full_d8_b4_l1115", O.FULL, B=4, L=1115, seed=1400)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _install_shims()
    out_dir = os.path.join(REPO, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    if os.environ.get("GOLDEN_ONLY") == "round2":
        gen_round2(out_dir)
        return
    if os.environ.get("GOLDEN_ONLY") == "round3":
        gen_round3(out_dir)
        return
    if os.environ.get("GOLDEN_ONLY") == "round5":
        gen_round5(out_dir)
        return
    if os.environ.get("GOLDEN_ONLY") == "round6":
        gen_round6(out_dir)
        return
    gen_lr(out_dir)
    gen_ops(out_dir)
    from oracle import style_oracle as SO
    if os.environ.get("GOLDEN_ONLY") != "latent":
        gen_style(out_dir, "style_tiny", SO.STYLE_TINY, B=3, seed=500)
        gen_style(out_dir, "style_full", SO.STYLE_FULL, B=4, seed=600)
    if os.environ.get("GOLDEN_ONLY") == "style":
        return
    from oracle import latent_oracle as LO
    gen_latent(out_dir, "latent_tiny", LO.LATENT_TINY, B=3, Ba=1, L=9 * 7, seed=700)
    gen_latent(out_dir, "latent_tiny_b2", LO.LATENT_TINY, B=2, Ba=2, L=9 * 5, seed=710)
    gen_latent(out_dir, "latent_full", LO.LATENT_FULL, B=2, Ba=1, L=27 * 6, seed=720)
    # whole pipeline, LDM.sample: tiny widths (weights regenerated from the seed on both sides) and the default
    # model.yml widths with a depth-2 denoiser, on a spectrogram whose length needs padding
    gen_ldm(out_dir, "ldm_tiny", LO.LatentDims(style_dim=8, n_downs=2, h_dim=32, n_layers=2), SO.STYLE_TINY,
            O.Dims(emb_dim=6, a_dim=32, style_dim=8, global_cond_dim=32, backbone_dim=64, n_heads=2, head_dim=32, depth=2,
                   expand=4, radius=2, u_head_dim=16), B=3, L=9 * 6 + 4, num_steps=6, seed=800)
    gen_ldm(out_dir, "ldm_full_d2", LO.LATENT_FULL, SO.STYLE_FULL, O.Dims(depth=2), B=2, L=27 * 5 + 11, num_steps=4, seed=810)
    if os.environ.get("GOLDEN_ONLY") == "latent":
        return
    # tiny config: weights + every gradient stored (a few hundred KB)
    gen_model(out_dir, "tiny_b3_l40", O.TINY, B=3, L=40, seed=100, store_weights=True, with_bf16=True)
    # tiny config, broadcast audio (sampler semantics '#B A l', model.py:120) and ragged L
    gen_model(out_dir, "tiny_b2_l77_bcast", O.TINY, B=2, L=77, seed=200, store_weights=True,
              audio_batch=1)
    # full-width config (depth 2): weights regenerated from the seed on both sides
    full2 = O.Dims(depth=2)
    gen_model(out_dir, "full_d2_b2_l96", full2, B=2, L=96, seed=300, store_weights=False,
              with_bf16=True)
    # full default config (46.9 M params) at a short length
    gen_model(out_dir, "full_d8_b2_l64", O.FULL, B=2, L=64, seed=400, store_weights=False,
              num_steps=4)
    gen_round2(out_dir)
    gen_round3(out_dir)
    gen_round5(out_dir)
    gen_round6(out_dir)


if __name__ == "__main__":
    main()
