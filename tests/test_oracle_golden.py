"""Pins oracle/denoiser_oracle.py to the reference: replays the fixtures that
oracle/make_golden.py produced by running the reference itself.
CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import denoiser_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def dims_of(fx):
    keys = list(O.Dims().to_dict().keys())
    return O.Dims(**{k: int(v) for k, v in zip(keys, fx["dims"].tolist())})


def test_lr_schedule():
    fx = load("lr_schedule")
    for s, m in zip(fx["steps"].tolist(), fx["mult"].tolist()):
        assert O.lr_multiplier(int(s), 1000, .3, 30000) == pytest.approx(m, rel=1e-12)


def test_ops():
    fx = load("ops")
    assert rel_l2(O.rms_norm_channels(fx["x"]), fx["rms"]) < 1e-6
    assert rel_l2(O.rope_half_split(fx["xr"]), fx["rope"]) < 1e-6
    d = O.Dims(backbone_dim=48, n_heads=2, head_dim=16, expand=4, radius=2)
    P = {k[4:]: v for k, v in fx.items() if k.startswith("att.")}
    assert rel_l2(O.sdpsa(fx["xa"], P, "", d), fx["sdpsa"]) < 2e-6
    P = {k[4:]: v for k, v in fx.items() if k.startswith("ffn.")}
    assert rel_l2(O.swiglu(fx["xa"], P, "", d), fx["swiglu"]) < 2e-6


def _inputs(fx, d):
    if "in.h" in fx:
        data = {k[3:]: v for k, v in fx.items() if k.startswith("in.")}
        P = {k[2:]: v for k, v in fx.items() if k.startswith("w.")}
    else:
        B, L, seed = int(fx["B"]), int(fx["L"]), int(fx["seed"])
        P = O.init_params(d, seed=seed)
        data = O.synthetic_batch(d, B, L, seed=seed + 1)
    return P, data


@pytest.mark.parametrize("name", ["tiny_b3_l40", "tiny_b2_l77_bcast", "full_d2_b2_l96", "full_d8_b2_l64"])
def test_forward_sample_train(name):
    torch.set_num_threads(8)
    fx = load(name)
    d = dims_of(fx)
    P, data = _inputs(fx, d)
    xt = torch.lerp(data["x0"], data["z"], data["t"][:, None, None])
    with torch.no_grad():
        a, cg = O.precompute_conditioning(data["h"], data["s"], P)
        assert rel_l2(a, fx["cond_a"]) < 1e-6 and rel_l2(cg, fx["cond_cg"]) < 1e-6
        h0 = O.pointwise_conv(xt, P["proj_in.weight"], P["proj_in.bias"])
        assert rel_l2(O.backbone_layer(h0, a, cg, P, 0, d), fx["layer0"]) < 5e-6
        u, v = O.forward(data["h"], data["s"], xt, P, d)
        assert rel_l2(u, fx["fwd_u"]) < 1e-6
        assert rel_l2(v, fx["fwd_v"]) < 2e-5
        xs, u0, eta = O.sample(data["h"], data["s"], int(fx["num_steps"]), data["x_init"], P, d)
        assert rel_l2(xs, fx["sample_x"]) < 1e-4   # north_star sampler tolerance

    loss, logs, grads = O.loss_and_grads(P, d, data["h"], data["z"], data["s"], fx["t_used"], data["x0"])
    assert float(loss) == pytest.approx(float(fx["loss"]), rel=2e-5)
    for k in ("osl", "del", "u_mape"):
        assert float(logs[k]) == pytest.approx(float(fx["log_" + k]), rel=5e-5)
    total, coef = O.clip_coef(grads, 1.0)
    assert total == pytest.approx(float(fx["grad_norm"]), rel=1e-4)
    for k, g in grads.items():
        if "grad." + k in fx:
            assert rel_l2(g, fx["grad." + k]) < 2e-4, k
        else:
            assert float(g.norm()) == pytest.approx(float(fx["gradnorm." + k]), rel=2e-3, abs=1e-7), k

    # two optimizer + EMA steps on those grads (lr schedule steps 0 and 1)
    m = {k: torch.zeros_like(p) for k, p in P.items()}
    vv = {k: torch.zeros_like(p) for k, p in P.items()}
    ema = {k: p.clone() for k, p in P.items()}
    Pc = {k: p.clone() for k, p in P.items()}
    for step in (1, 2):
        lr = 3e-4 * O.lr_multiplier(step - 1, 1000, .3, 30000)
        O.adamw_ema_step(Pc, grads, m, vv, ema, step, lr, clip=coef, first_ema=(step == 1))
    assert float(fx["lr_after"][0]) == pytest.approx(3e-4 * O.lr_multiplier(2, 1000, .3, 30000), rel=1e-9)
    assert int(fx["n_averaged"]) == 2
    for k in Pc:
        if "p2." + k in fx:
            assert torch.allclose(Pc[k], fx["p2." + k], rtol=1e-4, atol=2e-6), k
            assert torch.allclose(ema[k], fx["ema2." + k], rtol=1e-4, atol=2e-6), k
        else:
            n = Pc[k].numel()
            sub = Pc[k].flatten()[::max(1, n // 64)][:64]
            assert torch.allclose(sub, fx["p2sub." + k], rtol=1e-4, atol=2e-6), k
            sub = ema[k].flatten()[::max(1, n // 64)][:64]
            assert torch.allclose(sub, fx["ema2sub." + k], rtol=1e-4, atol=2e-6), k


@pytest.mark.parametrize("name", ["val_tiny", "val_full_d2"])
def test_validation_step_vs_reference(name):
    """train.py:128-139: segmenting of the full map, EMA weights, no_grad loss."""
    torch.set_num_threads(8)
    fx = load(name)
    d = dims_of(fx)
    seed = int(fx["seed"])
    P_ema = O.init_params(d, seed=seed + 7)          # the generator gave the EMA copy its own weights
    logs = O.validation_step(P_ema, d, fx["h"], fx["z"], fx["s"], fx["labels"], int(fx["val_batches"]), fx["t_used"], fx["x0"])
    for k in ("loss", "osl", "del", "u_mape"):
        assert float(logs[k]) == pytest.approx(float(fx["log.val_" + k]), rel=5e-5), k
    # the live model's weights give a different loss: the fixture can tell the two apart
    wrong = O.validation_step(O.init_params(d, seed=seed), d, fx["h"], fx["z"], fx["s"], fx["labels"], int(fx["val_batches"]),
                              fx["t_used"], fx["x0"])
    assert abs(float(wrong["loss"]) - float(fx["log.val_loss"])) > 1e-2 * float(fx["log.val_loss"])


def test_rope_long_positions():
    """rope() at configs[4]'s length (positions up to 32767), sub-sampled rows."""
    fx = load("rope_long")
    N, D = int(fx["N"]), int(fx["D"])
    pos = fx["pos"].long()
    x = torch.zeros(1, 2, N, D)
    x[:, :, pos] = fx["x"]
    y = O.rope_half_split(x)[:, :, pos]
    assert torch.equal(y, fx["y"]) or rel_l2(y, fx["y"]) < 1e-7
    # the angle table itself (what od_rope_table must reproduce): l * 10000^(-2j/D) in fp32
    inv = 10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / -D)
    assert torch.equal(inv, fx["inv_freq"])
    assert torch.equal(torch.outer(pos.float(), inv), fx["angle"])


def test_bf16_training_fixture_is_anchored():
    """The bf16-autocast loss / gradients of the reference sit next to its fp32 ones on the same inputs; the fp32 half
    must be the very numbers of the round-1 fixture, and the bf16 half must differ from it by a bf16-sized amount."""
    for name, base in (("train_bf16_tiny_b3_l40", "tiny_b3_l40"), ("train_bf16_full_d2_b2_l96", "full_d2_b2_l96")):
        fx, fb = load(name), load(base)
        assert float(fx["f32.loss"]) == pytest.approx(float(fb["loss"]), rel=1e-6)
        assert torch.allclose(fx["t_used"], fb["t_used"])
        assert float(fx["f32.grad_norm"]) == pytest.approx(float(fb["grad_norm"]), rel=1e-5)
        e = abs(float(fx["bf16.loss"]) - float(fx["f32.loss"])) / float(fx["f32.loss"])
        assert 1e-6 < e < 2e-2
        errs = []
        for k in fx:
            if k.startswith("bf16.grad."):
                errs.append(rel_l2(fx[k], fx["f32.grad." + k[len("bf16.grad."):]]))
            elif k.startswith("bf16.gradsub."):
                ref = fx["f32.gradsub." + k[len("bf16.gradsub."):]]
                if float(ref.norm()) > 0:
                    errs.append(rel_l2(fx[k], ref))
        assert errs and 1e-4 < float(np.median(errs)) < 5e-2


def test_reference_bf16_error_is_recorded():
    """The fixture carries the reference's own bf16-autocast output so the GPU bf16
    tolerance is anchored to what the reference itself loses in bf16."""
    fx = load("tiny_b3_l40")
    e = rel_l2(fx["fwd_v_bf16"], fx["fwd_v"])
    assert 1e-4 < e < 5e-2


# ---------------------------------------------------------------- style model sampler
def _style_case(name):
    from oracle import style_oracle as SO
    fx = load(name)
    v = [int(x) for x in fx["dims"].tolist()]
    d = SO.StyleDims(style_dim=v[0], label_features=v[1], h_dim=v[2], depth=v[3], expand=v[4])
    P = ({k[2:]: t for k, t in fx.items() if k.startswith("w.")} if any(k.startswith("w.") for k in fx)
         else SO.init_style_params(d, int(fx["seed"])))
    return SO, fx, d, P


@pytest.mark.parametrize("name", ["style_tiny", "style_full"])
def test_style_oracle_vs_reference(name):
    SO, fx, d, P = _style_case(name)
    with torch.no_grad():
        assert rel_l2(SO.conditioning(fx["labels"], P, d), fx["cond"]) < 2e-6
        u, v = SO.style_forward(fx["st"], fx["labels"], P, d)
        assert rel_l2(u, fx["fwd_u"]) < 2e-6 and rel_l2(v, fx["fwd_v"]) < 5e-6
        s = SO.style_sample(fx["labels"], 16, fx["s_init"], P, d)
        assert rel_l2(s, fx["sample_s"]) < 1e-4
