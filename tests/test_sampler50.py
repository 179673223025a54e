"""BASELINE configs[3] where north_star claims it: the 50-step sampler (num_steps + 1 = 51 network evaluations) of the full
46.9 M-parameter model against the REFERENCE's own `DiffusionModel.sample` run (tests/golden/sample50_*.npz, written by
oracle/make_golden.py::gen_sampler50 from /root/reference in the build container; weights / inputs regenerated from the
seed on both sides).

  * fp32 and fp32-bf16x3 compute: < 1e-4 relative L2 (north_star's bound), at L = 64 and at B = 4, L = 1115 with batch-1 audio.
  * bf16 compute is OUTSIDE that bound by construction (the reference's own bf16-autocast sample differs from its fp32
    sample by ~6e-3): it is held to 3x the reference's own bf16 error on the same inputs.
  * CPU (`-m "not gpu"`): the oracle's restatement of the sampler against the same fixture at L = 64.
The GPU test also writes gpurun_out/sampler_parity.json (mode -> rel-L2, meets_1e-4, kernel source hash), which bench.py
quotes beside each sampler mode when the hash still matches the library it runs.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import denoiser_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOUND = 1e-4          # north_star: "sample-loop output within 1e-4 relative L2 of reference"


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def problem(fx):
    keys = list(O.Dims().to_dict().keys())
    d = O.Dims(**{k: int(v) for k, v in zip(keys, fx["dims"].tolist())})
    seed = int(fx["seed"])
    P = O.init_params(d, seed=seed)
    data = O.synthetic_batch(d, int(fx["B"]), int(fx["L"]), seed=seed + 1, audio_batch=1)
    return d, P, data


def test_oracle_sampler_50_steps_vs_reference():
    fx = load("sample50_full_d8_b2_l64")
    d, P, data = problem(fx)
    with torch.no_grad():
        xs, u0, eta_o = O.sample(data["h"], data["s"], int(fx["num_steps"]), data["x_init"], P, d)
    assert rel(xs, fx["sample_x"]) < 2e-5
    assert abs(eta_o - float(fx["eta"])) < 1e-6 and abs(u0 - float(fx["u0"].mean())) < 1e-5
    # the fixture's own consistency: eta follows model.py:131-132 from the stored first evaluation
    c0 = O.distance_constants(d.emb_dim)[0]
    eta = 1 - (c0 ** .5 / max(float(fx["u0"].mean()), c0 ** .5 + 1e-6)) ** (1 / int(fx["num_steps"]))
    assert abs(eta - float(fx["eta"])) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sample50_full_d8_b2_l64", "sample50_full_d8_b4_l1115"])
def test_hip_sampler_50_steps_vs_reference(name):
    from osu_dreamer_amd import _lib
    from osu_dreamer_amd.model import BackboneArgs, DiffusionModel, DiffusionModelArgs
    _lib.lib()
    dev = torch.device("cuda:0")
    fx = load(name)
    d, P, data = problem(fx)
    m = DiffusionModel(d.emb_dim, d.a_dim, d.style_dim,
                       DiffusionModelArgs(d.global_cond_dim, d.backbone_dim,
                                          BackboneArgs(d.depth, d.expand, d.head_dim, d.n_heads, d.radius), d.u_head_dim))
    m.load_state_dict(P)
    m = m.to(dev).eval()
    h, s, x_init = data["h"].to(dev), data["s"].to(dev), data["x_init"].to(dev)
    steps = int(fx["num_steps"])
    assert steps == 50 and h.shape[0] == 1
    ref, ref_bf16 = fx["sample_x"], fx["sample_x_bf16"]
    out = {}
    for mode, dt, mm in (("fp32", None, "f32"), ("fp32_bf16x3", None, "bf16x3"), ("bf16", torch.bfloat16, "f32")):
        m.compute_dtype, m.f32_matmul = dt, mm
        with torch.no_grad():
            u0, v0 = m(h, s, x_init)
        xs = m.sample(h, s, steps, x_init=x_init)
        torch.cuda.synchronize()
        assert torch.isfinite(xs).all()
        out[mode] = {"rel_l2_vs_reference_fp32": rel(xs, ref), "first_eval_v_rel_l2": rel(v0, fx["v0"]),
                     "first_eval_u_rel_l2": rel(u0, fx["u0"]), "eta": float(m.last_sample_stats[0])}
    m.compute_dtype, m.f32_matmul = None, "f32"
    ref_bf16_err = rel(ref_bf16, ref)
    print(f"[{name}] 50-step sampler rel-L2 vs reference: " + ", ".join(f"{k} {v['rel_l2_vs_reference_fp32']:.3e}" for k, v in out.items())
          + f"; reference's own bf16-autocast run {ref_bf16_err:.3e}")
    for mode in ("fp32", "fp32_bf16x3"):
        assert out[mode]["rel_l2_vs_reference_fp32"] < BOUND, (mode, out[mode])
        assert abs(out[mode]["eta"] - float(fx["eta"])) < 1e-5 * abs(float(fx["eta"])) + 1e-7
        out[mode]["meets_1e-4"] = True
    e16 = out["bf16"]["rel_l2_vs_reference_fp32"]
    assert e16 < 3 * ref_bf16_err + 1e-3, (e16, ref_bf16_err)
    out["bf16"]["meets_1e-4"] = bool(e16 < BOUND)
    out["bf16"]["reference_bf16_autocast_rel_l2"] = ref_bf16_err
    if name.endswith("b4_l1115"):
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, "gpurun_out", "sampler_parity.json"), "w") as f:
            json.dump({"fixture": f"tests/golden/{name}.npz", "workload": "50-step sampler, B=4, L=1115, audio batch 1, 46.9 M params",
                       "bound": BOUND, "kernel_src_sha": _lib.source_sha(), "modes": out}, f, indent=1)
