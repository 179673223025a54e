"""N>1 path on CPU: two processes (gloo, world_size 2) each run the real training step — the
kernel sources on the SIMT emulator — on their own batch; GradBucketReducer averages the
gradient arena bucket by bucket; FusedAdamWEMA steps.  Checks: ranks end bit-identical, and
equal to a single process stepping on the oracle's average of the two per-rank gradients."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import denoiser_oracle as O
    from osu_dreamer_amd import _lib
    from osu_dreamer_amd.ddp import GradBucketReducer
    from kernel_backend import EMU_SO
    from test_model_parity import make_trainer
    _lib.use_library(EMU_SO)
    d = O.TINY
    P = O.init_params(d, seed=50 + rank)          # deliberately different: broadcast must fix it
    tr = make_trainer(d, P, torch.device("cpu"))
    red = GradBucketReducer(tr.diffusion)
    red.broadcast_parameters(0)
    tr.diffusion_ema.module.load_state_dict(tr.diffusion.state_dict())
    data = O.synthetic_batch(d, 2, 24, seed=60 + rank)
    cfg = tr.configure_optimizers()
    opt = cfg["optimizer"]
    opt.max_grad_norm = 1.0
    opt.zero_grad()
    loss, _ = tr(tr.diffusion, data["h"], data["z"], data["s"], None, t=data["t"], x0=data["x0"])
    loss.backward()
    opt.step()
    tr.on_train_batch_end()
    torch.save({"p": tr.diffusion.arena.data.clone(), "ema": tr.diffusion_ema.module.arena.data.clone(),
                "g": tr.diffusion.arena.grad.clone()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_step_matches_averaged_oracle(tmp_path):
    from kernel_backend import build_emu
    build_emu()
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["p"], r1["p"]) and torch.equal(r0["ema"], r1["ema"]) and torch.equal(r0["g"], r1["g"])

    from oracle import denoiser_oracle as O
    from osu_dreamer_amd.model import DiffusionModel
    from test_model_parity import margs
    d = O.TINY
    P = O.init_params(d, seed=50)
    grads = []
    for rank in range(world):
        data = O.synthetic_batch(d, 2, 24, seed=60 + rank)
        grads.append(O.loss_and_grads(P, d, data["h"], data["z"], data["s"], data["t"], data["x0"])[2])
    avg = {k: (grads[0][k] + grads[1][k]) / 2 for k in P}
    _, coef = O.clip_coef(avg, 1.0)
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    vv = {k: torch.zeros_like(v) for k, v in P.items()}
    ema = {k: v.clone() for k, v in P.items()}
    Pc = {k: v.clone() for k, v in P.items()}
    O.adamw_ema_step(Pc, avg, m, vv, ema, 1, 3e-4 * O.lr_multiplier(0, 1000, .3, 30000), clip=coef, first_ema=True)
    model = DiffusionModel(d.emb_dim, d.a_dim, d.style_dim, margs(d))
    for k in P:
        got = model.arena.view(k, r0["p"])
        assert torch.allclose(got, Pc[k], rtol=1e-4, atol=2e-6), k
        gg = model.arena.view(k, r0["g"])
        assert float((gg - avg[k]).norm() / (avg[k].norm() + 1e-12)) < 1e-3, k
