"""A pinned TRAINING TRAJECTORY: 40 consecutive optimizer steps of the reference's DiffusionTrainer (models/diffusion/train.py:69-126 —
loss, backward, clip 1.0, AdamW with bias correction and weight decay, the LR schedule's warm-up / plateau / decay branches
(common/lr_schedule.py:10-21) and the EMA copy-then-lerp), recorded by oracle/make_golden.py::gen_trajectory in fp32 and with the forward
under bf16 autocast.  Held to it:

  * the oracle (CPU restatement; `-m "not gpu"`): every step's loss, clip norm and learning rate, the final weights and EMA weights;
  * the HIP path (DiffusionTrainer + FusedAdamWEMA through the C ABI): a 10-step prefix on the emulator build, all 40 steps on the GPU —
      fp32:  loss within 1e-4 (relative) at EVERY step, learning rate exactly, final weights / EMA: per tensor the RMS distance to the
             reference's tensor <= 1e-5 of the tensor's RMS + 0.1 % of how far 40 steps of training moved it (an element whose gradient is
             ~0 takes AdamW's sign-like first steps, +-lr whatever the magnitude, so the bound scales with the move, not with the weight; measured on the MI355X: 7-9 % of it);
      bf16:  per step |loss - ref_bf16| <= 3 x |ref_bf16 - ref_f32| + 2e-3 |ref_f32| (the reference's own bf16-vs-fp32 drift is the
             yardstick, as in test_model_parity), final weights within 3 x the reference's own bf16-vs-fp32 weight distance per tensor (+ 2 % of the tensor's move).
"""
import os

import numpy as np
import pytest
import torch

from oracle import denoiser_oracle as O
from osu_dreamer_amd.model import BackboneArgs, DiffusionModelArgs
from osu_dreamer_amd.train import DiffusionTrainer
from osu_dreamer_amd.lr_schedule import LRScheduleArgs
from kernel_backend import dev  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["traj40_tiny_b3_l40", "traj40_small_hd64_b2_l130"]


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def dims_of(fx):
    keys = list(O.Dims().to_dict().keys())
    return O.Dims(**{k: int(v) for k, v in zip(keys, fx["dims"].tolist())})


def batch_of(fx, d, i):
    return O.synthetic_batch(d, int(fx["B"]), int(fx["L"]), seed=int(fx["seed"]) + 1 + 10 * i)


def sub(w):
    w = w.detach().float().cpu().flatten()
    return w[::max(1, w.numel() // 64)][:64]


def check_weights(fx, tag, P0, weights, ema, k_drift=0.0, label=""):
    """per tensor: |sub-sample - reference's| <= 1e-5 |w| + 0.1 % |w_final - w_0| (+ k_drift x the reference's own bf16-vs-fp32 distance), as RMS over
    the sub-sample"""
    worst = 0.0
    for k in weights:
        n_el = sub(P0[k]).numel()
        moved = float(fx[f"{tag}.dnorm." + k]) / max(1.0, float(P0[k].numel())) ** 0.5           # RMS move per element
        for pre, w in (("p", weights[k]), ("ema", ema[k])):
            ref = fx[f"{tag}.{pre}sub." + k]
            rms_w = float(fx[f"{tag}.{pre}norm." + k]) / max(1.0, float(P0[k].numel())) ** 0.5
            tol = 1e-5 * rms_w + 1e-3 * moved + 1e-7
            if k_drift:                          # bf16: the reference's own bf16-vs-fp32 distance on this tensor is the yardstick, 2 % of the move the floor
                drift = float((fx[f"bf16.{pre}sub." + k] - fx[f"f32.{pre}sub." + k]).norm()) / n_el ** 0.5
                tol += k_drift * drift + 0.02 * moved
            err = float((sub(w) - ref).norm()) / n_el ** 0.5
            assert err <= tol, (label, tag, pre, k, err, tol)
            worst = max(worst, err / tol)
    return worst


@pytest.mark.parametrize("name", NAMES)
def test_oracle_follows_the_reference_trajectory(name):
    """fp32: the oracle's own 40 steps (loss_and_grads + clip_coef + adamw_ema_step + lr_multiplier) against the reference's."""
    fx = load(name)
    d = dims_of(fx)
    steps = int(fx["steps"])
    P0 = O.init_params(d, seed=int(fx["seed"]))
    P = {k: v.clone() for k, v in P0.items()}
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v = {k: torch.zeros_like(v_) for k, v_ in P.items()}
    ema = {k: v_.clone() for k, v_ in P.items()}
    for i in range(steps):
        data = batch_of(fx, d, i)
        loss, _, grads = O.loss_and_grads(P, d, data["h"], data["z"], data["s"], fx["t_used"][i], data["x0"])
        assert float(loss) == pytest.approx(float(fx["f32.loss"][i]), rel=1e-4), i
        total, coef = O.clip_coef(grads, 1.0)
        assert total == pytest.approx(float(fx["f32.grad_norm"][i]), rel=2e-3), i
        lr = 3e-4 * O.lr_multiplier(i, warmup_steps=int(fx["warmup_steps"]), warmup_init=0.3, decay_start=int(fx["decay_start"]))
        assert lr == pytest.approx(float(fx["f32.lr"][i]), rel=1e-12), i
        O.adamw_ema_step(P, grads, m, v, ema, i + 1, lr, clip=coef, first_ema=(i == 0))
    worst = check_weights(fx, "f32", P0, P, ema, label="oracle")
    print(f"[{name}] oracle: final weights at {worst:.2e} of the tolerance")


def make_trainer(fx, d, P, device):
    tr = DiffusionTrainer(val_batches=2, opt_args=dict(lr=3e-4, weight_decay=0.01),
                          schedule_args=LRScheduleArgs(warmup_init=.3, warmup_steps=int(fx["warmup_steps"]), decay_start=int(fx["decay_start"])),
                          osl_weight=1., del_weight=30., emb_dim=d.emb_dim, a_dim=d.a_dim, style_dim=d.style_dim,
                          diffusion_args=DiffusionModelArgs(d.global_cond_dim, d.backbone_dim,
                                                            BackboneArgs(d.depth, d.expand, d.head_dim, d.n_heads, d.radius), d.u_head_dim))
    tr.diffusion.load_state_dict(P)
    tr.diffusion_ema.module.load_state_dict(P)
    return tr.to(device)


def run_hip_trajectory(name, device, tag, steps):
    fx = load(name)
    d = dims_of(fx)
    total = int(fx["steps"])
    steps = min(steps, total)
    P0 = O.init_params(d, seed=int(fx["seed"]))
    tr = make_trainer(fx, d, P0, device)
    model = tr.diffusion
    if tag == "bf16":
        model.compute_dtype = torch.bfloat16
    cfg = tr.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    opt.max_grad_norm = 1.0
    worst_loss = 0.0
    for i in range(steps):
        data = {k: v.to(device) for k, v in batch_of(fx, d, i).items()}
        assert opt.param_groups[0]["lr"] == pytest.approx(float(fx[f"{tag}.lr"][i]), rel=1e-9), i
        opt.zero_grad()
        loss, _ = tr(model, data["h"], data["z"], data["s"], None, t=fx["t_used"][i].to(device), x0=data["x0"])
        loss.backward()
        opt.step()
        sched.step()
        tr.on_train_batch_end()
        mine, ref = float(loss.detach()), float(fx[f"{tag}.loss"][i])
        if tag == "f32":
            assert mine == pytest.approx(ref, rel=1e-4), (i, mine, ref)
            worst_loss = max(worst_loss, abs(mine - ref) / abs(ref))
        else:
            ref32 = float(fx["f32.loss"][i])
            bound = 3.0 * abs(ref - ref32) + 2e-3 * abs(ref32)
            assert abs(mine - ref) <= bound, (i, mine, ref, ref32)
            worst_loss = max(worst_loss, abs(mine - ref) / bound)
    if hasattr(model.engine, "check_attn_status"):
        model.engine.check_attn_status()
    assert int(tr.diffusion_ema.n_averaged) == steps
    if steps == total:
        assert int(tr.diffusion_ema.n_averaged) == int(fx[f"{tag}.n_averaged"])
        weights = {k: p.detach().cpu() for k, p in model.named_parameters()}
        ema = {k: p.detach().cpu() for k, p in tr.diffusion_ema.module.named_parameters()}
        worst_w = check_weights(fx, tag, P0, weights, ema, k_drift=3.0 if tag == "bf16" else 0.0, label="hip")
        print(f"[{name}/{tag}] {steps} steps: worst loss error {worst_loss:.3e} ({'relative' if tag == 'f32' else 'of its bound'}), "
              f"final weights at {worst_w:.2e} of the tolerance")
    return worst_loss


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_hip_trajectory_prefix_tiny(dev, tag):
    """the first 10 steps (warm-up branch, EMA copy then lerp, bias correction) on the emulator build and on the GPU"""
    run_hip_trajectory("traj40_tiny_b3_l40", dev, tag, 10)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["f32", "bf16"])
@pytest.mark.parametrize("name", NAMES)
def test_hip_trajectory_40_steps(name, tag):
    from osu_dreamer_amd import _lib
    _lib._lib = None
    _lib.lib()
    run_hip_trajectory(name, torch.device("cuda:0"), tag, 40)
