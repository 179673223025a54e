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
    torch.set_num_threads(2 if world <= 2 else 1)
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
                "g": tr.diffusion.arena.grad.clone(), "buckets": list(red.bucket_log)}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,det_mode", [(2, "0"), (8, "0"), (2, "1")])
def test_n_rank_step_matches_averaged_oracle(tmp_path, world, det_mode, monkeypatch):
    """World sizes 2 and 8 (the size the driver's scaling run uses): every rank ends the step with identical weights / EMA / gradients,
    equal to the oracle's step on the mean of the ranks' gradients, and every rank reduced its buckets in the same order.
    det_mode 1: the same under OD_DETERMINISTIC=1 — every arena segment's integer shadow is folded in BEFORE its bucket is handed to the exchange."""
    from kernel_backend import build_emu
    build_emu()
    monkeypatch.setenv("OD_DETERMINISTIC", det_mode)          # the spawned ranks inherit it
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    for r in range(1, world):
        rr = torch.load(tmp_path / f"rank{r}.pt")
        assert torch.equal(r0["p"], rr["p"]) and torch.equal(r0["ema"], rr["ema"]) and torch.equal(r0["g"], rr["g"])
        assert rr["buckets"] == r0["buckets"], "ranks must enqueue the same buckets in the same order (a collective per bucket)"
    assert len(r0["buckets"]) >= 3 and sum(n for _, n in r0["buckets"]) == r0["g"].numel()

    from oracle import denoiser_oracle as O
    from osu_dreamer_amd.model import DiffusionModel
    from test_model_parity import margs
    d = O.TINY
    P = O.init_params(d, seed=50)
    grads = []
    for rank in range(world):
        data = O.synthetic_batch(d, 2, 24, seed=60 + rank)
        grads.append(O.loss_and_grads(P, d, data["h"], data["z"], data["s"], data["t"], data["x0"])[2])
    avg = {k: sum(g[k] for g in grads) / world for k in P}
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


# ---------------------------------------------------------------------------------------------------------
# uneven shards: rank 0 gets two maps, rank 1 one map -> different numbers of windows per epoch.  The epoch must
# end for BOTH ranks after the shorter shard's steps (no rank left waiting in an all-reduce), with identical
# weights / EMA / AdamW moments, even though the ranks were built from different seeds (state broadcast).
def _fit_worker(rank, world, port, data_dir, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from osu_dreamer_amd import _lib
    from osu_dreamer_amd.data import LatentDataModule
    from osu_dreamer_amd.fit import build_from_config
    from kernel_backend import EMU_SO
    from test_fit import _cfg
    _lib.use_library(EMU_SO)
    cfg = _cfg(True)
    cfg["data"].update(data_path=data_dir, seq_len=24, batch_size=1, num_workers=0, shuffle_buffer_size=1,
                       max_val_count=1, max_val_frac=.3, max_per_map=-1)
    cfg["trainer"].update(max_epochs=1, max_steps=-1, log_every_n_steps=1, limit_val_batches=1, devices=world,
                          default_root_dir=os.path.join(out_dir, f"run{rank}"), precision="32", enable_checkpointing=False)
    torch.manual_seed(100 + rank)                 # different initial weights per rank: the broadcast must fix it
    module, trainer = build_from_config(cfg)
    with torch.no_grad():
        for n, p in module.diffusion.named_parameters():
            if any(z in n for z in ("ssg1.", "ssg2.", "proj_out.", "u_mod.")):
                p.normal_(0, 0.02)
    dm = LatentDataModule(**cfg["data"], rank=rank, world_size=world)
    n_local = sum(1 for _ in dm.train_set)        # windows this rank's shard would yield alone
    trainer.fit(module, dm)
    torch.save({"p": module.diffusion.arena.data.clone(), "ema": module.diffusion_ema.module.arena.data.clone(),
                "steps": trainer.global_step, "n_local": n_local, "n_avg": int(module.diffusion_ema.n_averaged)},
               os.path.join(out_dir, f"fit{rank}.pt"))
    dist.destroy_process_group()


def test_uneven_shards_end_epoch_together(tmp_path):
    from kernel_backend import build_emu
    from osu_dreamer_amd.data import write_synthetic_dataset
    build_emu()
    data_dir = tmp_path / "data"
    # 4 maps of 72 frames: 1 held out for validation, 3 for training -> rank 0 reads maps {0, 2}, rank 1 map {1}
    write_synthetic_dataset(str(data_dir), n_maps=4, frames=72, a_dim=16, emb_dim=6, style_dim=8, seed=3)
    world = 2
    mp.spawn(_fit_worker, args=(world, _free_port(), str(data_dir), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "fit0.pt"), torch.load(tmp_path / "fit1.pt")
    assert r0["n_local"] != r1["n_local"], "the fixture is supposed to shard unevenly"
    assert r0["steps"] == r1["steps"] == min(r0["n_local"], r1["n_local"]) > 0
    assert r0["n_avg"] == r1["n_avg"] == r0["steps"]
    assert torch.equal(r0["p"], r1["p"]) and torch.equal(r0["ema"], r1["ema"])


# ---------------------------------------------------------------------------------------------------------
# `--gpus N` / `trainer.devices: N` start N ranks themselves (children of torch.distributed.run) with the right env
def test_launcher_spawns_n_ranks(tmp_path):
    from osu_dreamer_amd import launch
    script = tmp_path / "rank_probe.py"
    script.write_text(
        "import os, sys\n"
        "open(os.path.join(sys.argv[1], 'rank' + os.environ['RANK']), 'w').write("
        "' '.join(os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR')) + ' ' + sys.argv[2])\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = launch.torchrun_command(3, [str(script), str(tmp_path), "--flag"], port=_free_port())
    assert cmd[1:5] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3"] and "127.0.0.1" in cmd
    rc = launch.spawn_ranks_if_needed(3, [str(script), str(tmp_path), "--flag"], env=env)
    assert rc == 0
    got = sorted((tmp_path / f"rank{r}").read_text() for r in range(3))
    assert got == [f"{r} {r} 3 127.0.0.1 --flag" for r in range(3)]
    # inside a rank (WORLD_SIZE set) nothing is spawned, and a contradicting --gpus is refused
    os.environ["WORLD_SIZE"] = "3"
    try:
        assert launch.spawn_ranks_if_needed(3, ["x"]) is None
        assert launch.check_world(3) == 3
        with pytest.raises(SystemExit):
            launch.check_world(8)
    finally:
        del os.environ["WORLD_SIZE"]
    assert launch.spawn_ranks_if_needed(1, ["x"]) is None


def test_launcher_runs_eight_ranks_that_rendezvous(tmp_path):
    """`devices: 8` — the launcher's eight torchrun children find each other on 127.0.0.1 and complete a collective (gloo here; the GPU job
    hands the same rendezvous the RCCL unique id)."""
    from osu_dreamer_amd import launch
    script = tmp_path / "rank_sum.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        "torch.set_num_threads(1)\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.tensor([float(dist.get_rank())])\n"
        "dist.all_reduce(t)\n"
        "open(os.path.join(sys.argv[1], 'sum' + os.environ['RANK']), 'w').write(f'{int(t.item())} {dist.get_world_size()}')\n"
        "dist.destroy_process_group()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    rc = launch.spawn_ranks_if_needed(8, [str(script), str(tmp_path)], env=env)
    assert rc == 0
    assert [(tmp_path / f"sum{r}").read_text() for r in range(8)] == ["28 8"] * 8


def test_bench_gpus_flag_starts_ranks(monkeypatch):
    """`python bench.py --gpus 4` must hand the job to the launcher before touching the GPU (round-1 bug: the flag was
    parsed and ignored, so an 8-GPU run would have reported n_gpus 1)."""
    import importlib
    from osu_dreamer_amd import launch
    monkeypatch.syspath_prepend(REPO)
    bench = importlib.import_module("bench")
    calls = []
    monkeypatch.setattr(launch, "spawn_ranks_if_needed", lambda n, args, **kw: calls.append((n, list(args))) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    assert calls and calls[0][0] == 4 and calls[0][1][0].endswith("bench.py") and calls[0][1][1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]


# ---------------------------------------------------------------------------------------------------------
# a rank that dies mid-epoch must take the job down with a non-zero exit, not leave its peers waiting in a collective
_DEAD_RANK_SCRIPT = r'''
import os, sys, time
repo, data_dir, out_dir = sys.argv[1:4]
sys.path.insert(0, repo); sys.path.insert(0, os.path.join(repo, "tests"))
import torch
torch.set_num_threads(2)
from osu_dreamer_amd import _lib
from osu_dreamer_amd.data import LatentDataModule
from osu_dreamer_amd.fit import build_from_config
from kernel_backend import EMU_SO
from test_fit import _cfg
_lib.use_library(EMU_SO)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
cfg = _cfg(True)
cfg["data"].update(data_path=data_dir, seq_len=24, batch_size=1, num_workers=0, shuffle_buffer_size=1, max_val_count=1, max_val_frac=.3, max_per_map=-1)
cfg["trainer"].update(max_epochs=50, max_steps=-1, log_every_n_steps=1, limit_val_batches=1, devices=world,
                      default_root_dir=os.path.join(out_dir, f"run{rank}"), precision="32", enable_checkpointing=False)
module, trainer = build_from_config(cfg)
hook = module.on_train_batch_end
def dying_hook(*a, **k):
    hook(*a, **k)
    open(os.path.join(out_dir, f"alive{rank}"), "a").write("x")
    if rank == 1 and trainer.global_step >= 2:
        os._exit(17)                      # no teardown, no goodbye: the process is simply gone
module.on_train_batch_end = dying_hook
dm = LatentDataModule(**cfg["data"], rank=rank, world_size=world)
trainer.fit(module, dm)                   # initialises (and would destroy) the gloo group itself
open(os.path.join(out_dir, f"finished{rank}"), "w").write("done")
'''


@pytest.mark.timeout(600)
def test_dead_rank_ends_the_job_nonzero(tmp_path):
    """Rank 1 vanishes (os._exit) after its 3rd step of a 50-epoch run.  The launcher's job must END — rank 0 must not sit in the
    step agreement / all-reduce forever — and must report failure (non-zero).  The watchdog is the launcher's agent plus the
    failure path of Trainer.fit (abort the communicator, re-raise): never a re-exec of a process that has touched the GPU."""
    import time
    from kernel_backend import build_emu
    from osu_dreamer_amd import launch
    from osu_dreamer_amd.data import write_synthetic_dataset
    build_emu()
    data_dir = tmp_path / "data"
    write_synthetic_dataset(str(data_dir), n_maps=5, frames=96, a_dim=16, emb_dim=6, style_dim=8, seed=4)
    script = tmp_path / "dying_fit.py"
    script.write_text(_DEAD_RANK_SCRIPT)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    rc = launch.spawn_ranks_if_needed(2, [str(script), REPO, str(data_dir), str(tmp_path)], env=env)
    dt = time.time() - t0
    assert rc not in (0, None), "a job that lost a rank must not report success"
    assert dt < 300, f"the surviving rank took {dt:.0f} s to give up"
    assert (tmp_path / "alive0").exists() and (tmp_path / "alive1").exists(), "both ranks were training before the failure"
    assert not (tmp_path / "finished0").exists() and not (tmp_path / "finished1").exists()


# ---------------------------------------------------------------------------------------------------------
# torch's own DistributedDataParallel around DiffusionTrainer (what Lightning's `devices: N` / strategy "ddp" builds around the
# reference's module, model.yml:11).  The denoiser's autograd node returns None for every parameter and writes
# the gradient arena directly (train.py, model.py: `p.grad` aliases the arena), which is the shape that usually breaks DDP's reducer:
# its post-accumulate hooks must still fire and must read `param.grad`.  Checked: after backward through the wrapper every rank holds
# the MEAN of the ranks' local gradients (bit for bit against a gloo all-reduce of the local arenas), for two iterations (the second
# runs on the reducer's rebuilt buckets), and with gradient_as_bucket_view=True — where DDP re-points `.grad` into its buckets — the
# arena still holds the averaged gradient that FusedAdamWEMA reads.
def _torch_ddp_worker(rank, world, port, out_dir, bucket_view):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch.nn.parallel import DistributedDataParallel as DDP
    from oracle import denoiser_oracle as O
    from osu_dreamer_amd import _lib
    from kernel_backend import EMU_SO
    from test_model_parity import make_trainer
    _lib.use_library(EMU_SO)
    d = O.TINY
    P = O.init_params(d, seed=70)

    def local_grads(it):
        """the gradient this rank's batch gives WITHOUT the wrapper (same weights): what DDP has to average"""
        tr0 = make_trainer(d, P, torch.device("cpu"))
        data = O.synthetic_batch(d, 2, 24, seed=80 + 10 * it + rank)
        loss, _ = tr0(tr0.diffusion, data["h"], data["z"], data["s"], None, t=data["t"], x0=data["x0"])
        loss.backward()
        return tr0.diffusion.arena.grad.clone(), data

    tr = make_trainer(d, P, torch.device("cpu"))
    for p in tr.diffusion_ema.parameters():
        p.requires_grad_(False)
    ddp = DDP(tr, gradient_as_bucket_view=bucket_view)
    out = {"max_diff": [], "arena_is_grad": []}
    for it in range(2):
        g_local, data = local_grads(it)
        want = g_local.clone()
        dist.all_reduce(want)
        want /= world
        if tr.diffusion.arena.grad is not None:
            tr.diffusion.arena.grad.zero_()
        loss, _ = ddp(tr.diffusion, data["h"], data["z"], data["s"], None, t=data["t"], x0=data["x0"])
        loss.backward()
        model = tr.diffusion
        # what the optimizer reads: the arena's gradient buffer; and what every parameter's .grad says
        got_arena = model.arena.grad
        per_param = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for _, p in model.named_parameters()])
        want_params = torch.cat([model.arena.view(k, want).reshape(-1) for k, _ in model.named_parameters()])
        out["max_diff"].append((float((got_arena - want).abs().max()), float((per_param - want_params).abs().max())))
        out["arena_is_grad"].append(all(p.grad is not None and p.grad.data_ptr() == model.arena.view(k, model.arena.grad).data_ptr()
                                        for k, p in model.named_parameters()))
        # what FusedAdamWEMA.step does first: adopt whatever .grad the wrapper left (a no-op when .grad still aliases the arena)
        out.setdefault("adopted_diff", []).append(float((model.adopt_grads() - want).abs().max()))
    torch.save(out, os.path.join(out_dir, f"ddp{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_view", [False, True])
def test_torch_ddp_wrapper_averages_the_arena_gradients(tmp_path, bucket_view):
    from kernel_backend import build_emu
    build_emu()
    mp.spawn(_torch_ddp_worker, args=(2, _free_port(), str(tmp_path), bucket_view), nprocs=2, join=True)
    for r in range(2):
        o = torch.load(tmp_path / f"ddp{r}.pt")
        for it, (d_arena, d_params) in enumerate(o["max_diff"]):
            assert d_params == 0.0, (r, it, d_params)       # every p.grad is the mean of the ranks' local gradients, bit for bit
            if not bucket_view:
                assert d_arena == 0.0, (r, it, d_arena)     # ... and it still lives in the arena FusedAdamWEMA reads
        if not bucket_view:
            assert all(o["arena_is_grad"]), "DDP without bucket views must leave p.grad aliasing the arena"
        else:
            # gradient_as_bucket_view=True re-points p.grad into DDP's buckets (the arena keeps the LOCAL gradient): FusedAdamWEMA.step adopts
            # the averaged .grad into the arena before it reads it (DiffusionModel.adopt_grads)
            assert not any(o["arena_is_grad"])
        assert all(a == 0.0 for a in o["adopted_diff"]), o["adopted_diff"]


# ---------------------------------------------------------------------------------------------------------
# A failed attention backward on ONE rank.  Its incomplete / NaN gradients are already in everybody's all-reduced arena, so the
# status word every rank's optimizer acts on is the OR over the ranks (GradBucketReducer.global_status): all ranks skip the update (weights
# bit-identical to before, and still identical across ranks) and all ranks raise at their next check — the failing one with its own code,
# the healthy one with "another rank".
def _failed_rank_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OD_ATTN_BWD_FUSED="1", OD_FB_CHAIN_TIMEOUT_MS="50")
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import denoiser_oracle as O
    from osu_dreamer_amd import _lib
    from osu_dreamer_amd.ddp import GradBucketReducer
    from kernel_backend import EMU_SO
    from test_model_parity import make_trainer
    _lib.use_library(EMU_SO)
    d = O.Dims(global_cond_dim=64, backbone_dim=128, n_heads=2, head_dim=64, depth=2, expand=2, radius=1, u_head_dim=16)
    P = O.init_params(d, seed=11)
    tr = make_trainer(d, P, torch.device("cpu"))
    model = tr.diffusion
    model.compute_dtype = torch.bfloat16
    red = GradBucketReducer(model)
    red.broadcast_parameters(0)
    data = O.synthetic_batch(d, 2, 230, seed=12 + rank)
    opt = tr.configure_optimizers()["optimizer"]
    opt.max_grad_norm = 1.0

    def step():
        opt.zero_grad()
        loss, _ = tr(model, data["h"], data["z"], data["s"], None, t=data["t"], x0=data["x0"])
        loss.backward()
        opt.step()
    step()
    opt.check_device_status()                                  # healthy step on both ranks
    good = model.arena.data.detach().clone()
    if rank == 1:                                              # damaged write numbers in every running tile of rank 1's workspace
        ws = model.engine._attn_ws
        ws.buf[256:ws.zero_bytes].view(torch.int32).fill_(5)
    step()
    raised = ""
    try:
        opt.check_device_status()
    except RuntimeError as e:
        raised = str(e)
    torch.save({"skipped": bool(torch.equal(model.arena.data, good)), "p": model.arena.data.clone(), "raised": raised,
                "gnorm_nan": bool(torch.isnan(opt.gnorm_sq).all())}, os.path.join(out_dir, f"fail{rank}.pt"))
    dist.destroy_process_group()


def test_failed_attention_backward_on_one_rank_stops_every_rank(tmp_path):
    from kernel_backend import build_emu
    build_emu()
    mp.spawn(_failed_rank_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "fail0.pt"), torch.load(tmp_path / "fail1.pt")
    assert r0["skipped"] and r1["skipped"] and r0["gnorm_nan"] and r1["gnorm_nan"]
    assert torch.equal(r0["p"], r1["p"])                        # the replicas did not diverge
    assert "another rank" in r0["raised"] and "status 3" in r1["raised"]
