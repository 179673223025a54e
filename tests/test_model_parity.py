"""End-to-end parity of the HIP path (DiffusionModel / DiffusionTrainer / FusedAdamWEMA)
against the golden fixtures the REFERENCE produced (tests/golden, see oracle/make_golden.py).

Tolerances (fp32 compute): forward v 5e-5 rel-L2, u 1e-5; sampler 1e-4 rel-L2 (north_star);
parameter gradients 1e-3 rel-L2 per tensor; two optimizer+EMA steps allclose(rtol 1e-4).
bf16 compute: forward error vs the fp32 reference must stay within 3x the error the reference's
own bf16-autocast run shows on the same inputs (recorded in the fixture).
Tiny configs run on the emulator build (CPU) and on the GPU; full-width configs on the GPU only.
"""
import os

import numpy as np
import pytest
import torch

from oracle import denoiser_oracle as O
from osu_dreamer_amd.model import BackboneArgs, DiffusionModel, DiffusionModelArgs
from osu_dreamer_amd.train import DiffusionTrainer
from osu_dreamer_amd.lr_schedule import LRScheduleArgs
from kernel_backend import dev, rel_l2  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def dims_of(fx):
    keys = list(O.Dims().to_dict().keys())
    return O.Dims(**{k: int(v) for k, v in zip(keys, fx["dims"].tolist())})


def margs(d):
    return DiffusionModelArgs(d.global_cond_dim, d.backbone_dim,
                              BackboneArgs(d.depth, d.expand, d.head_dim, d.n_heads, d.radius), d.u_head_dim)


def inputs(fx, d):
    if "in.h" in fx:
        return ({k[2:]: v for k, v in fx.items() if k.startswith("w.")},
                {k[3:]: v for k, v in fx.items() if k.startswith("in.")})
    seed = int(fx["seed"])
    return O.init_params(d, seed=seed), O.synthetic_batch(d, int(fx["B"]), int(fx["L"]), seed=seed + 1)


def make_trainer(d, P, device):
    tr = DiffusionTrainer(val_batches=2, opt_args=dict(lr=3e-4, weight_decay=0.01),
                          schedule_args=LRScheduleArgs(warmup_init=.3, warmup_steps=1000, decay_start=30000),
                          osl_weight=1., del_weight=30., emb_dim=d.emb_dim, a_dim=d.a_dim, style_dim=d.style_dim,
                          diffusion_args=margs(d))
    assert list(tr.diffusion.state_dict().keys()) == list(P.keys())
    tr.diffusion.load_state_dict(P)
    tr.diffusion_ema.module.load_state_dict(P)
    return tr.to(device)


CASES = [
    pytest.param("tiny_b3_l40", id="tiny_b3_l40"),
    pytest.param("tiny_b2_l77_bcast", id="tiny_b2_l77_bcast"),
]
GPU_ONLY = [
    pytest.param("full_d2_b2_l96", id="full_d2_b2_l96"),
    pytest.param("full_d8_b2_l64", id="full_d8_b2_l64"),
]


def run_case(name, dev):
    fx = load(name)
    d = dims_of(fx)
    P, data = inputs(fx, d)
    tr = make_trainer(d, P, dev)
    model = tr.diffusion
    dd = {k: v.to(dev) for k, v in data.items()}
    xt = torch.lerp(data["x0"], data["z"], data["t"][:, None, None]).to(dev)

    # ---- forward (no grad)
    with torch.no_grad():
        u, v = model(dd["h"], dd["s"], xt)
    assert rel_l2(u, fx["fwd_u"]) < 1e-5
    assert rel_l2(v, fx["fwd_v"]) < 5e-5

    # ---- bf16 compute, anchored to the reference's own bf16 error
    if "fwd_v_bf16" in fx:
        model.compute_dtype = torch.bfloat16
        with torch.no_grad():
            ub, vb = model(dd["h"], dd["s"], xt)
        model.compute_dtype = None
        ref_err = rel_l2(fx["fwd_v_bf16"], fx["fwd_v"])
        assert rel_l2(vb, fx["fwd_v"]) < 3 * ref_err + 1e-3, (rel_l2(vb, fx["fwd_v"]), ref_err)
        assert rel_l2(ub, fx["fwd_u"]) < 2e-2

    # ---- sampler
    xs = model.sample(dd["h"], dd["s"], int(fx["num_steps"]), x_init=dd["x_init"])
    assert rel_l2(xs, fx["sample_x"]) < 1e-4

    # ---- fp32 tensors, 3 x bf16 MFMA products (OD_F32X3): held to the same forward / sampler bounds
    model.f32_matmul = "bf16x3"
    with torch.no_grad():
        u3, v3 = model(dd["h"], dd["s"], xt)
    xs3 = model.sample(dd["h"], dd["s"], int(fx["num_steps"]), x_init=dd["x_init"])
    model.f32_matmul = "f32"
    assert not torch.equal(v3, v), "f32_matmul='bf16x3' must reach the f32x3 kernels"
    assert rel_l2(v3, fx["fwd_v"]) < 5e-5 and rel_l2(u3, fx["fwd_u"]) < 1e-5
    assert rel_l2(xs3, fx["sample_x"]) < 1e-4
    print(f"[{name}] f32x3: fwd_v {rel_l2(v3, fx['fwd_v']):.2e} (f32 {rel_l2(v, fx['fwd_v']):.2e}), "
          f"sampler {rel_l2(xs3, fx['sample_x']):.2e} (f32 {rel_l2(xs, fx['sample_x']):.2e})")

    # ---- training loss + gradients through the trainer's forward
    opt_cfg = tr.configure_optimizers()
    opt, sched = opt_cfg["optimizer"], opt_cfg["lr_scheduler"]["scheduler"]
    opt.max_grad_norm = 1.0
    opt.zero_grad()
    loss, logs = tr(model, dd["h"], dd["z"], dd["s"], None, t=fx["t_used"].to(dev), x0=dd["x0"])
    assert float(loss) == pytest.approx(float(fx["loss"]), rel=5e-5)
    for k in ("osl", "del", "u_mape"):
        assert float(logs[k]) == pytest.approx(float(fx["log_" + k]), rel=1e-4)
    loss.backward()
    gn = float(model.arena.grad.double().norm())
    assert gn == pytest.approx(float(fx["grad_norm"]), rel=2e-4)
    worst = 0.0
    for k, p in model.named_parameters():
        g = p.grad.cpu()
        if "grad." + k in fx:
            e = rel_l2(g, fx["grad." + k])
            assert e < 1e-3, (k, e)
            worst = max(worst, e)
        else:
            assert float(g.norm()) == pytest.approx(float(fx["gradnorm." + k]), rel=3e-3, abs=1e-6), k
            n = g.numel()
            assert torch.allclose(g.flatten()[::max(1, n // 64)][:64], fx["gradsub." + k], rtol=2e-2, atol=2e-5 * float(fx["gradnorm." + k]) + 1e-7), k

    # ---- two optimizer + EMA steps on those gradients (clip 1.0, lr schedule steps 0 and 1)
    for _ in range(2):
        opt.step()
        sched.step()
        tr.on_train_batch_end()
    assert int(tr.diffusion_ema.n_averaged) == int(fx["n_averaged"])
    assert opt.param_groups[0]["lr"] == pytest.approx(float(fx["lr_after"][0]), rel=1e-9)
    ema = tr.diffusion_ema.module
    for (k, p), (_, e) in zip(model.named_parameters(), ema.named_parameters()):
        if "p2." + k in fx:
            assert torch.allclose(p.detach().cpu(), fx["p2." + k], rtol=1e-4, atol=3e-6), k
            assert torch.allclose(e.detach().cpu(), fx["ema2." + k], rtol=1e-4, atol=3e-6), k
        else:
            n = p.numel()
            assert torch.allclose(p.detach().cpu().flatten()[::max(1, n // 64)][:64], fx["p2sub." + k], rtol=1e-4, atol=3e-6), k
            assert torch.allclose(e.detach().cpu().flatten()[::max(1, n // 64)][:64], fx["ema2sub." + k], rtol=1e-4, atol=3e-6), k
    return worst


@pytest.mark.parametrize("name", CASES)
def test_tiny_vs_reference(dev, name):
    run_case(name, dev)


def test_tiny_vs_reference_with_the_fused_film_conv_kernel(dev, monkeypatch):
    """OD_FUSE_FILM_DWCONV=1 (round 6): gate + residual, norm + FiLM and the SwiGLU branch's depthwise conv run as ONE kernel, in the
    no-grad forward, the sampler and the training step — the whole reference fixture (forward, sampler, loss, every gradient, two
    optimizer steps) holds to the same bounds."""
    monkeypatch.setenv("OD_FUSE_FILM_DWCONV", "1")
    run_case("tiny_b3_l40", dev)


@pytest.mark.gpu
@pytest.mark.parametrize("name", GPU_ONLY)
def test_full_width_vs_reference(name):
    from osu_dreamer_amd import _lib
    _lib._lib = None
    _lib.lib()
    run_case(name, torch.device("cuda:0"))


@pytest.mark.parametrize("B,L,Ba,steps", [(1, 5, 1, 1), (2, 64, 2, 3), (3, 129, 1, 2), (1, 200, 1, 5)])
def test_edge_shapes_vs_oracle(dev, B, L, Ba, steps):
    """Shapes the fixtures do not hold — a sequence shorter than one attention tile, exact tile multiples, one past a
    multiple, batch 1, a single sampler step (no captured loop) — against the oracle (itself pinned by the fixtures):
    forward, sampler, loss and every gradient."""
    d = O.TINY
    P = O.init_params(d, seed=900 + L)
    data = O.synthetic_batch(d, B, L, seed=901 + L)
    g = torch.Generator().manual_seed(902 + L)
    h = data["h"][:Ba]
    xt = torch.randn(B, d.emb_dim, L, generator=g)
    x_init = torch.randn(B, d.emb_dim, L, generator=g)
    m = DiffusionModel(d.emb_dim, d.a_dim, d.style_dim, margs(d))
    m.load_state_dict(P)
    m = m.to(dev)
    with torch.no_grad():
        u, v = m(h.to(dev), data["s"].to(dev), xt.to(dev))
    ref_u, ref_v = O.forward(h, data["s"], xt, P, d)
    assert rel_l2(u, ref_u) < 1e-5 and rel_l2(v, ref_v) < 5e-5
    xs = m.sample(h.to(dev), data["s"].to(dev), steps, x_init=x_init.to(dev))
    ref_x = O.sample(h, data["s"], steps, x_init, P, d)[0]
    assert rel_l2(xs, ref_x) < 1e-4
    if Ba == B:       # training never broadcasts audio
        tr = DiffusionTrainer(val_batches=2, opt_args=dict(lr=3e-4, weight_decay=0.01),
                              schedule_args=LRScheduleArgs(warmup_init=.3, warmup_steps=1000, decay_start=30000),
                              osl_weight=1., del_weight=30., emb_dim=d.emb_dim, a_dim=d.a_dim, style_dim=d.style_dim,
                              diffusion_args=margs(d))
        tr.diffusion.load_state_dict(P)
        tr = tr.to(dev)
        loss, _ = tr(tr.diffusion, data["h"].to(dev), data["z"].to(dev), data["s"].to(dev), None, t=data["t"].to(dev),
                     x0=data["x0"].to(dev))
        loss.backward()
        ref_loss, _, ref_grads = O.loss_and_grads(P, d, data["h"], data["z"], data["s"], data["t"], data["x0"])
        assert float(loss.detach()) == pytest.approx(float(ref_loss), rel=5e-5)
        for k, p in tr.diffusion.named_parameters():
            assert rel_l2(p.grad, ref_grads[k]) < 1e-3, k


def test_bf16x3_forward_when_the_qkv_epilogue_does_not_apply(dev):
    """f32_matmul = 'bf16x3' with an odd head count (2 * H * hd not a multiple of 128): the no-grad forward takes the qkv projection's
    two-kernel path (od_gemm_nt_qkrope_split) and has to unwrap the pre-split weight like every other GEMM wrapper."""
    d = O.Dims(global_cond_dim=64, backbone_dim=96, n_heads=3, head_dim=32, depth=2, expand=2, radius=1, u_head_dim=16)
    P = O.init_params(d, seed=41)
    data = O.synthetic_batch(d, 2, 50, seed=42)
    g = torch.Generator().manual_seed(43)
    xt = torch.randn(2, d.emb_dim, 50, generator=g)
    m = DiffusionModel(d.emb_dim, d.a_dim, d.style_dim, margs(d))
    m.load_state_dict(P)
    m = m.to(dev)
    m.f32_matmul = "bf16x3"
    with torch.no_grad():
        u, v = m(data["h"].to(dev), data["s"].to(dev), xt.to(dev))
    ref_u, ref_v = O.forward(data["h"], data["s"], xt, P, d)
    assert rel_l2(u, ref_u) < 5e-5 and rel_l2(v, ref_v) < 2e-4


# ------------------------------------------------------------------------------------------------------------------
# bf16 training step (model.yml:12 `precision: bf16-mixed` — what the headline bench runs) against the reference's own
# bf16-autocast loss and gradients.  Bound per tensor: error vs the reference's fp32 gradient <= BF16_K x the error of the
# reference's own bf16 run on that tensor (+ a floor for tensors where the reference's bf16 error is tiny).
BF16_K, BF16_FLOOR = 3.0, 4e-3


def run_bf16_training_case(name, dev, attn_dtype=None, target="f32"):
    """`attn_dtype` torch.float16 + target "f16": the step with the attention core on half operands, held to the reference's OWN
    fp16-autocast run (tests/golden/train_f16_*.npz, loss-scaled as Lightning's 16-mixed does) within the reference's bf16 noise."""
    fx = load(name)
    d = dims_of(fx)
    seed = int(fx["seed"])
    P = O.init_params(d, seed=seed)
    data = O.synthetic_batch(d, int(fx["B"]), int(fx["L"]), seed=seed + 1)
    tr = make_trainer(d, P, dev)
    model = tr.diffusion
    model.compute_dtype, model.attn_dtype = torch.bfloat16, attn_dtype
    dd = {k: v.to(dev) for k, v in data.items()}
    opt = tr.configure_optimizers()["optimizer"]
    opt.zero_grad()
    loss, logs = tr(model, dd["h"], dd["z"], dd["s"], None, t=fx["t_used"].to(dev), x0=dd["x0"])
    loss.backward()
    if attn_dtype is not None:
        assert model.engine.attn_f16 and model.engine.fused_attn_bwd() and model.engine.attn_bwd_passes() == 5
        model.engine.check_attn_status()
    T = target                                   # what the step is compared with; the bf16-vs-f32 distance of the reference is the yardstick
    ref32, ref16 = float(fx["f32.loss"]), float(fx["bf16.loss"])
    refT = float(fx[T + ".loss"])
    assert abs(float(loss.detach()) - refT) <= BF16_K * abs(ref16 - ref32) + 2e-3 * ref32, (float(loss.detach()), refT, ref32, ref16)
    gn = float(model.arena.grad.double().norm())
    gn32, gn16 = float(fx["f32.grad_norm"]), float(fx["bf16.grad_norm"])
    assert abs(gn - float(fx[T + ".grad_norm"])) <= BF16_K * abs(gn16 - gn32) + 5e-3 * gn32, (gn, gn32, gn16)
    ratios = []
    for k, p in model.named_parameters():
        g = p.grad.detach().cpu()
        if "f32.grad." + k in fx:
            r32, r16, rT = fx["f32.grad." + k], fx["bf16.grad." + k], fx[T + ".grad." + k]
            mine = g
        else:                                    # full-width fixture: strided sub-sample + norm
            r32, r16, rT = fx["f32.gradsub." + k], fx["bf16.gradsub." + k], fx[T + ".gradsub." + k]
            n = g.numel()
            mine = g.flatten()[::max(1, n // 64)][:64]
            n32 = float(fx["f32.gradnorm." + k])
            assert abs(float(g.norm()) - float(fx[T + ".gradnorm." + k])) <= BF16_K * abs(float(fx["bf16.gradnorm." + k]) - n32) + 2e-2 * n32 + 1e-7, k
        if float(r32.norm()) == 0:
            continue
        e_ref, e_mine = rel_l2(r16, r32), rel_l2(mine, rT)
        assert e_mine <= BF16_K * e_ref + BF16_FLOOR, (k, e_mine, e_ref)
        ratios.append(e_mine / max(e_ref, 1e-9))
    print(f"[{name}] bf16 step{' + f16 attention, against the reference fp16-autocast run' if attn_dtype is not None else ''}: "
          f"loss {float(loss.detach()):.5f} (ref fp32 {ref32:.5f}, ref bf16 {ref16:.5f}, ref {T} {refT:.5f}); "
          f"median grad distance = {float(np.median(ratios)):.2f} x the reference's own bf16 error, worst {max(ratios):.2f} x")


def test_f16_attention_step_small(dev):
    run_bf16_training_case("train_f16_small_hd64_b2_l230", dev, attn_dtype=torch.float16, target="f16")


@pytest.mark.gpu
def test_f16_attention_step_full_width():
    from osu_dreamer_amd import _lib
    _lib._lib = None
    _lib.lib()
    run_bf16_training_case("train_f16_full_d2_b2_l96", torch.device("cuda:0"), attn_dtype=torch.float16, target="f16")


@pytest.mark.parametrize("name", ["train_bf16_tiny_b3_l40"])
def test_bf16_training_step_tiny(dev, name):
    run_bf16_training_case(name, dev)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", ["0", "1"])
def test_bf16_training_step_full_width(monkeypatch, fused):
    """... with the attention backward as the two-kernel pair (7 MFMA passes) and as od_flash_attn_bwd_fused (5; the default from L = 4096)."""
    from osu_dreamer_amd import _lib
    _lib._lib = None
    _lib.lib()
    monkeypatch.setenv("OD_ATTN_BWD_FUSED", fused)
    run_bf16_training_case("train_bf16_full_d2_b2_l96", torch.device("cuda:0"))


def test_fused_attention_backward_in_the_step(dev, monkeypatch):
    """The training step's backward with od_flash_attn_bwd_fused (forced: L is short) against the same step with the two-kernel attention
    backward — bf16, head_dim 64, two key blocks, ragged tiles: every gradient agrees to the bf16 noise of dq, and the loss is identical."""
    d = O.Dims(global_cond_dim=64, backbone_dim=128, n_heads=2, head_dim=64, depth=2, expand=2, radius=1, u_head_dim=16)
    P = O.init_params(d, seed=11)
    data = O.synthetic_batch(d, 2, 230, seed=12)
    grads = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("OD_ATTN_BWD_FUSED", fused)
        tr = make_trainer(d, P, dev)
        model = tr.diffusion
        model.compute_dtype = torch.bfloat16
        dd = {k: v.to(dev) for k, v in data.items()}
        opt = tr.configure_optimizers()["optimizer"]
        opt.zero_grad()
        loss, _ = tr(model, dd["h"], dd["z"], dd["s"], None, t=dd["t"], x0=dd["x0"])
        loss.backward()
        assert model.engine.fused_attn_bwd() == (fused == "1") and model.engine.attn_bwd_passes() == (5 if fused == "1" else 7)
        grads[fused] = (float(loss.detach()), model.arena.grad.detach().cpu().clone())
    # the forwards are the same kernels on the same inputs — every layer buffer is bit-identical between the two runs (tools/dbg_fwd16x.py) — but the
    # loss scalars are summed with fp32 atomics, whose order is not fixed: the last bit of the loss may differ (it did once the forward's values
    # changed with the 16x16x32 kernel of round 6; OD_DETERMINISTIC=1 is the mode that makes it exact)
    assert grads["0"][0] == pytest.approx(grads["1"][0], rel=1e-6)
    assert rel_l2(grads["1"][1], grads["0"][1]) < 2e-3


def test_f16_attention_in_the_step(dev):
    """model.attn_dtype = float16 ("attention in fp16", BASELINE configs[4]): the bf16 training step with the attention core on half operands
    against the same step on bf16 operands and against the fp32 oracle — the half form must be at least as close to fp32 as the bf16 form
    (its P / dS / q / k / v carry 3 more mantissa bits), loss and every arena segment of the gradient."""
    d = O.Dims(global_cond_dim=64, backbone_dim=128, n_heads=2, head_dim=64, depth=2, expand=2, radius=1, u_head_dim=16)
    P = O.init_params(d, seed=11)
    data = O.synthetic_batch(d, 2, 230, seed=12)
    ref_loss, _, ref_grads = O.loss_and_grads(P, d, data["h"], data["z"], data["s"], data["t"], data["x0"])
    out = {}
    for name, adt in (("bf16", None), ("f16", torch.float16)):
        tr = make_trainer(d, P, dev)
        model = tr.diffusion
        model.compute_dtype, model.attn_dtype = torch.bfloat16, adt
        dd = {k: v.to(dev) for k, v in data.items()}
        opt = tr.configure_optimizers()["optimizer"]
        opt.zero_grad()
        loss, _ = tr(model, dd["h"], dd["z"], dd["s"], None, t=dd["t"], x0=dd["x0"])
        loss.backward()
        assert model.engine.attn_f16 == (adt is not None)
        if adt is not None:
            assert model.engine.fused_attn_bwd() and model.engine.attn_bwd_passes() == 5
            model.engine.check_attn_status()
        errs = {k: rel_l2(p.grad, ref_grads[k]) for k, p in model.named_parameters() if float(ref_grads[k].norm()) > 0}
        out[name] = (abs(float(loss.detach()) - float(ref_loss)) / abs(float(ref_loss)), errs)
    worst = max(out["f16"][1][k] / max(out["bf16"][1][k], 1e-4) for k in out["f16"][1])
    med = float(np.median([out["f16"][1][k] / max(out["bf16"][1][k], 1e-4) for k in out["f16"][1]]))
    print(f"loss error vs fp32: bf16 {out['bf16'][0]:.2e}, f16-attention {out['f16'][0]:.2e}; gradient error ratio f16/bf16: median {med:.2f}, worst {worst:.2f}")
    assert out["f16"][0] <= 1.5 * out["bf16"][0] + 1e-3
    assert med <= 1.1 and worst <= 2.0


def test_deterministic_mode_matches_the_plain_step(dev):
    """OD_DETERMINISTIC (osu_dreamer_amd/det.py): the same kernels accumulate 2^40-scaled integers into 64-bit shadows instead of fp32 atomics and
    the engine folds them in before each destination's first reader.  The step must be the plain step to fp32 rounding — loss, every gradient,
    two optimizer steps — and leave every shadow empty; a second run of the deterministic step must be bit-identical to the first.  (On the
    GPU, where the atomics' order really varies: tests/test_full_size.py::test_deterministic_full_batch_step_is_bit_identical.)"""
    from osu_dreamer_amd import det
    d = O.Dims(global_cond_dim=64, backbone_dim=128, n_heads=2, head_dim=64, depth=2, expand=2, radius=1, u_head_dim=16)
    P = O.init_params(d, seed=21)
    data = O.synthetic_batch(d, 3, 150, seed=22)
    out = {}
    try:
        for mode in ("plain", "det", "det2"):
            det.force(mode != "plain")
            tr = make_trainer(d, P, dev)
            model = tr.diffusion
            model.compute_dtype = torch.bfloat16
            dd = {k: v.to(dev) for k, v in data.items()}
            opt = tr.configure_optimizers()["optimizer"]
            opt.max_grad_norm = 1.0
            losses = []
            for _ in range(2):
                opt.zero_grad()
                loss, _ = tr(model, dd["h"], dd["z"], dd["s"], None, t=dd["t"], x0=dd["x0"])
                loss.backward()
                g1 = model.arena.grad.detach().cpu().clone()
                opt.step()
                losses.append(float(loss.detach()))
            out[mode] = (losses, g1, model.arena.data.detach().cpu().clone())
            if mode != "plain":
                ctx = det.context(dev)
                assert len(ctx.live_ranges()) >= 6
                for t, sh in ctx.live_ranges().values():
                    assert int(sh.abs().max()) == 0                   # every shadow was folded in
    finally:
        det.force(None)
    assert torch.equal(out["det"][1], out["det2"][1]) and torch.equal(out["det"][2], out["det2"][2]) and out["det"][0] == out["det2"][0]
    assert out["det"][0] == pytest.approx(out["plain"][0], rel=1e-6)
    assert rel_l2(out["det"][1], out["plain"][1]) < 1e-5       # = the plain step's own atomics-order noise (2e-6 .. 4e-6 measured at this size; 5.9e-5 at 32 x 8192)
    assert rel_l2(out["det"][2], out["plain"][2]) < 1e-6


def test_deterministic_mode_survives_replanning(dev):
    """under OD_DETERMINISTIC the validation loader (batch 1, whole maps of varying length) makes the engine re-plan per batch,
    and every plan registers the workspace buffers its kernels accumulate into.  The registrations follow the buffers' lifetime (det.py:
    weak references, dead plans pruned), so many plan changes neither exhaust the C side's fixed table nor leak shadows — and a training step
    after them is still bit-identical to the same step before them."""
    from osu_dreamer_amd import det
    d = O.TINY
    P = O.init_params(d, seed=31)
    try:
        det.force(True)
        tr = make_trainer(d, P, dev)
        model = tr.diffusion
        opt = tr.configure_optimizers()["optimizer"]
        opt.max_grad_norm = 1.0

        def step_grad(L):
            data = {k: v.to(dev) for k, v in O.synthetic_batch(d, 2, L, seed=32 + L).items()}
            opt.zero_grad()
            loss, _ = tr(model, data["h"], data["z"], data["s"], None, t=data["t"], x0=data["x0"])
            loss.backward()
            return float(loss.detach()), model.arena.grad.detach().cpu().clone()

        first = step_grad(24)
        ctx = det.context(dev)
        for L in (17, 40, 33, 56, 24, 29, 48, 21, 64, 37):          # ten more plans: 3-6 registrations each
            step_grad(L)
            assert len(ctx.live_ranges()) <= 14, len(ctx.live_ranges())
        again = step_grad(24)
        assert again[0] == first[0] and torch.equal(again[1], first[1])
        n_cap = int(_table_capacity())
        # a registration that does not fit raises AND leaves the previous table active (the mode keeps working)
        keep = [torch.zeros(8, device=dev) for _ in range(n_cap + 2)]
        with pytest.raises(Exception):
            for t in keep:
                ctx.register(t)
        del keep
        after = step_grad(24)
        assert after[0] == first[0] and torch.equal(after[1], first[1])
    finally:
        det.force(None)


def _table_capacity():
    from osu_dreamer_amd import _lib
    return (_lib.lib().cdll.od_det_table_bytes() - 8) // 24


def test_failed_attention_backward_is_caught_in_every_step(dev, monkeypatch):
    """The fused attention backward's sticky error word is folded into EVERY optimizer step on the device (no host round trip): after a good
    first step, a launch on a workspace whose write numbers were damaged ends with status 3 and NaN dq — the gradient norm turns NaN, the
    AdamW / EMA update is skipped (weights bit-identical to before) and the trainer's host-side check raises."""
    monkeypatch.setenv("OD_ATTN_BWD_FUSED", "1")
    monkeypatch.setenv("OD_FB_CHAIN_TIMEOUT_MS", "50")
    d = O.Dims(global_cond_dim=64, backbone_dim=128, n_heads=2, head_dim=64, depth=2, expand=2, radius=1, u_head_dim=16)
    P = O.init_params(d, seed=11)
    data = O.synthetic_batch(d, 2, 230, seed=12)
    tr = make_trainer(d, P, dev)
    model = tr.diffusion
    model.compute_dtype = torch.bfloat16
    dd = {k: v.to(dev) for k, v in data.items()}
    opt = tr.configure_optimizers()["optimizer"]
    opt.max_grad_norm = 1.0

    def step():
        opt.zero_grad()
        loss, _ = tr(model, dd["h"], dd["z"], dd["s"], None, t=dd["t"], x0=dd["x0"])
        loss.backward()
        opt.step()
    step()
    opt.check_device_status()                              # healthy: nothing raised
    assert torch.isfinite(opt.gnorm_sq).all()
    before = model.arena.data.detach().clone()
    m_before = opt.exp_avg.clone()
    ws = model.engine._attn_ws
    ws.buf[256:ws.zero_bytes].view(torch.int32).fill_(5)   # damaged write numbers in every running tile
    step()
    assert torch.isnan(opt.gnorm_sq).all()
    assert torch.equal(model.arena.data, before) and torch.equal(opt.exp_avg, m_before)
    with pytest.raises(RuntimeError, match="status 3"):
        opt.check_device_status()


# ------------------------------------------------------------------------------------------------------------------
# validation_step (train.py:128-139) against the reference's logs: segmenting of the full map, EMA weights, no_grad
def run_validation_case(name, dev):
    fx = load(name)
    d = dims_of(fx)
    seed = int(fx["seed"])
    tr = DiffusionTrainer(val_batches=int(fx["val_batches"]), opt_args=dict(lr=3e-4, weight_decay=0.01),
                          schedule_args=LRScheduleArgs(warmup_init=.3, warmup_steps=1000, decay_start=30000),
                          osl_weight=1., del_weight=30., emb_dim=d.emb_dim, a_dim=d.a_dim, style_dim=d.style_dim,
                          diffusion_args=margs(d))
    tr.diffusion.load_state_dict(O.init_params(d, seed=seed))
    tr.diffusion_ema.module.load_state_dict(O.init_params(d, seed=seed + 7))     # the EMA copy has its own weights
    tr = tr.to(dev)
    batch = tuple(fx[k].to(dev) for k in ("h", "z", "s", "labels"))
    logs = tr.validation_step(batch, 0, t=fx["t_used"].to(dev), x0=fx["x0"].to(dev))
    for k in ("loss", "osl", "del", "u_mape"):
        assert float(logs[k]) == pytest.approx(float(fx["log.val_" + k]), rel=1e-4), k
        assert float(tr._logged["val/" + k]) == float(logs[k])
    assert not any(p.grad is not None and float(p.grad.abs().sum()) > 0 for p in tr.diffusion_ema.module.parameters())


def test_validation_step_tiny(dev):
    run_validation_case("val_tiny", dev)


@pytest.mark.gpu
def test_validation_step_full_width():
    from osu_dreamer_amd import _lib
    _lib._lib = None
    _lib.lib()
    run_validation_case("val_full_d2", torch.device("cuda:0"))


# ------------------------------------------------------------------------------------------------------------------
# a cached sampler hipGraph must not outlive the buffers it captured.  sample(A) -> a train step /
# forward with another shape (re-plans: new workspace, re-packed weights) -> sample(A) again has an EQUAL plan key but
# different allocations; the second call must re-capture (engine.generation) and still equal the eager sampler.
def test_sampler_graph_survives_replanning(dev):
    d = O.TINY
    P = O.init_params(d, seed=77)
    data = O.synthetic_batch(d, 2, 40, seed=78)
    tr = make_trainer(d, P, dev)
    m = tr.diffusion
    dd = {k: v.to(dev) for k, v in data.items()}
    x1 = m.sample(dd["h"], dd["s"], 4, x_init=dd["x_init"])
    gen1 = m.engine.generation
    other = O.synthetic_batch(d, 3, 24, seed=79)
    oo = {k: v.to(dev) for k, v in other.items()}
    loss, _ = tr(m, oo["h"], oo["z"], oo["s"], None, t=oo["t"], x0=oo["x0"])       # train=True plan, another shape
    loss.backward()
    with torch.no_grad():
        m(oo["h"], oo["s"], oo["x_init"])                                         # no-grad plan, another shape
    x2 = m.sample(dd["h"], dd["s"], 4, x_init=dd["x_init"])
    assert m.engine.generation > gen1
    m.use_graph = False
    x3 = m.sample(dd["h"], dd["s"], 4, x_init=dd["x_init"])
    m.use_graph = True
    assert torch.equal(x2, x3) and torch.equal(x1, x3)
    ref = O.sample(data["h"], data["s"], 4, data["x_init"], P, d)[0]
    assert rel_l2(x2, ref) < 1e-4
    # requires_grad survives .to(): the EMA copy stays frozen
    assert not any(p.requires_grad for p in tr.diffusion_ema.module.parameters())


# ------------------------------------------------------------------------------------------------------------------
# SwiGLU generality (common/swiglu.py:19-23): radius 0 (nn.Identity), wider depthwise kernels, and nn.Dropout1d p > 0
@pytest.mark.parametrize("radius", [0, 1, 3])
def test_swiglu_radius_variants_vs_oracle(dev, radius):
    import dataclasses
    d = dataclasses.replace(O.TINY, radius=radius)
    P = O.init_params(d, seed=300 + radius)
    assert ("net.layers.0.ffn.proj_vg.0.weight" in P) == (radius > 0)
    data = O.synthetic_batch(d, 2, 37, seed=310 + radius)
    tr = make_trainer(d, P, dev)
    dd = {k: v.to(dev) for k, v in data.items()}
    xt = torch.lerp(data["x0"], data["z"], data["t"][:, None, None])
    with torch.no_grad():
        u, v = tr.diffusion(dd["h"], dd["s"], xt.to(dev))
    ref_u, ref_v = O.forward(data["h"], data["s"], xt, P, d)
    assert rel_l2(u, ref_u) < 1e-5 and rel_l2(v, ref_v) < 5e-5
    loss, _ = tr(tr.diffusion, dd["h"], dd["z"], dd["s"], None, t=dd["t"], x0=dd["x0"])
    loss.backward()
    ref_loss, _, ref_grads = O.loss_and_grads(P, d, data["h"], data["z"], data["s"], data["t"], data["x0"])
    assert float(loss.detach()) == pytest.approx(float(ref_loss), rel=5e-5)
    for k, p in tr.diffusion.named_parameters():
        assert rel_l2(p.grad, ref_grads[k]) < 1e-3, k


def test_dropout1d_training_mode(dev):
    """nn.Dropout1d(p) on the SwiGLU hidden state: whole channels of a sample dropped, survivors scaled by 1/(1-p), training mode
    only.  The kernel path (od_scale_channels forward and backward) is compared with the oracle given the SAME channel factors;
    the factors themselves are checked to be a Dropout1d pattern (per (sample, channel), two values, the right keep rate)."""
    import dataclasses
    from osu_dreamer_amd import ops
    d = O.TINY
    P = O.init_params(d, seed=320)
    data = O.synthetic_batch(d, 3, 29, seed=321)
    tr = DiffusionTrainer(val_batches=2, opt_args=dict(lr=3e-4, weight_decay=0.01),
                          schedule_args=LRScheduleArgs(warmup_init=.3, warmup_steps=1000, decay_start=30000),
                          osl_weight=1., del_weight=30., emb_dim=d.emb_dim, a_dim=d.a_dim, style_dim=d.style_dim,
                          diffusion_args=DiffusionModelArgs(d.global_cond_dim, d.backbone_dim,
                                                            BackboneArgs(d.depth, d.expand, d.head_dim, d.n_heads, d.radius, dropout=0.25),
                                                            d.u_head_dim))
    tr.diffusion.load_state_dict(P)
    tr = tr.to(dev)
    m = tr.diffusion
    dd = {k: v.to(dev) for k, v in data.items()}
    xt = torch.lerp(data["x0"], data["z"], data["t"][:, None, None]).to(dev)
    # eval mode: identity
    m.eval()
    with torch.no_grad():
        u0, v0 = m(dd["h"], dd["s"], xt)
    ref_u, ref_v = O.forward(data["h"], data["s"], xt.cpu(), P, d)
    assert rel_l2(v0, ref_v) < 5e-5
    # training mode: masked; replay the drawn factors through the oracle
    m.train()
    torch.manual_seed(5)
    loss, _ = tr(m, dd["h"], dd["z"], dd["s"], None, t=dd["t"], x0=dd["x0"])
    loss.backward()
    eng = m.engine
    masks = [eng.ws.t[f"drop.{i}"].cpu()[:, : eng.Hf].clone() for i in range(d.depth)]
    for mk in masks:
        vals = sorted(mk.unique().tolist())
        assert len(vals) == 2 and vals[0] == 0.0 and vals[1] == pytest.approx(1 / 0.75)
    keep = float(torch.cat([mk.flatten() for mk in masks]).ne(0).float().mean())
    assert 0.6 < keep < 0.9

    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref_loss, _ = O.train_loss(Pg, d, data["h"], data["z"], data["s"], data["t"], data["x0"], drop_scales=masks)
    ref_loss.backward()
    assert float(loss.detach()) == pytest.approx(float(ref_loss), rel=5e-5)
    for k, p in m.named_parameters():
        assert rel_l2(p.grad, Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k])) < 1e-3, k
    # and a different draw the next time
    loss2, _ = tr(m, dd["h"], dd["z"], dd["s"], None, t=dd["t"], x0=dd["x0"])
    assert not torch.equal(eng.ws.t["drop.0"].cpu()[:, : eng.Hf], masks[0])
