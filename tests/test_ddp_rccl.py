"""The RCCL exchange through the C ABI (od_comm_* / od_allreduce_grads / od_broadcast_f32) on a real MI355X.  The pool
gives one GPU, so the communicator has one rank: every call still goes through librccl (ncclCommInitRank, ncclAllReduce
with ncclAvg, ncclBroadcast on a side stream), and a data-parallel training step with world size 1 must equal the plain
step bit for bit.  The 2-rank semantics (mean of the per-rank gradients, lock-step epochs) are covered on CPU by
tests/test_ddp_gloo.py."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def pg():
    import torch.distributed as dist
    from osu_dreamer_amd import _lib
    _lib._lib = None
    _lib.lib()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    yield dist
    dist.destroy_process_group()
    for k in ("WORLD_SIZE", "RANK", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)


def test_comm_allreduce_broadcast(pg):
    from osu_dreamer_amd.ddp import RcclComm
    dev = torch.device("cuda:0")
    c = RcclComm(dev)
    assert c.version >= 20000 and c.world == 1
    x = torch.randn(1 << 20, device=dev)
    y = x.clone()
    c.allreduce_mean_(y)
    c.broadcast_(y, 0)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        c.allreduce_mean_(y[: 1000])
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    c.close()


def test_ddp_step_world1_equals_plain_step(pg):
    """Two plain runs of the same two steps differ in the last bits (the weight-gradient GEMMs accumulate with fp32 atomics, and
    AdamW's first steps turn a sign flip of a near-zero gradient into a 2*lr difference); the run that exchanges its gradients
    through od_allreduce_grads (world size 1: the mean of one rank) must sit inside that same noise."""
    import bench
    from osu_dreamer_amd.ddp import GradBucketReducer
    dev = torch.device("cuda:0")
    outs = []
    for ddp in (False, False, True):
        tr = bench.make_trainer(dev, seed=31)
        tr.diffusion.compute_dtype = torch.bfloat16
        cfg = tr.configure_optimizers()
        opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
        red = None
        if ddp:
            red = GradBucketReducer(tr.diffusion)
            red.broadcast_state(opt, tr.diffusion_ema, src=0)
        batch = bench.synthetic_batch(2, 512, dev, seed=32)
        torch.manual_seed(33)
        for i in range(2):
            opt.zero_grad()
            loss = tr.training_step(batch, i)
            loss.backward()
            opt.step()
            sched.step()
            tr.on_train_batch_end()
        torch.cuda.synchronize()
        outs.append((float(loss.detach()), tr.diffusion.arena.data.clone(), tr.diffusion_ema.module.arena.data.clone()))
        if red is not None:
            red.close()
            tr.diffusion._reducer = None
    rel = lambda a, b: float((a - b).norm() / b.norm())
    noise_loss = abs(outs[0][0] - outs[1][0]) / abs(outs[0][0])
    noise_p, noise_e = rel(outs[0][1], outs[1][1]), rel(outs[0][2], outs[1][2])
    print(f"plain-vs-plain noise: loss {noise_loss:.2e}, params {noise_p:.2e}, ema {noise_e:.2e}; "
          f"ddp-vs-plain: loss {abs(outs[2][0] - outs[0][0]) / abs(outs[0][0]):.2e}, params {rel(outs[2][1], outs[0][1]):.2e}")
    assert abs(outs[2][0] - outs[0][0]) <= 3 * noise_loss * abs(outs[0][0]) + 2e-4 * abs(outs[0][0])
    # one plain-vs-plain pair is a noisy estimate of the noise (a handful of AdamW sign flips decide it: 1e-5 .. 6e-5 seen over runs)
    assert rel(outs[2][1], outs[0][1]) <= 3 * noise_p + 1e-4 and rel(outs[2][2], outs[0][2]) <= 3 * noise_e + 1e-4


def test_global_status_through_rccl(pg):
    """GradBucketReducer.global_status on the RCCL path (world size 1: the OR over one rank): 0 stays 0, any local status becomes 4, the word is
    a device tensor the optimizer kernels can read, and the communicator keeps working for the gradient buckets afterwards.  (The two-rank
    behaviour — a failed rank stops every rank — is tests/test_ddp_gloo.py::test_failed_attention_backward_on_one_rank_stops_every_rank.)"""
    import bench
    from osu_dreamer_amd.ddp import GradBucketReducer
    dev = torch.device("cuda:0")
    tr = bench.make_trainer(dev, seed=31)
    red = GradBucketReducer(tr.diffusion)
    local = torch.zeros(1, dtype=torch.int32, device=dev)
    assert int(red.global_status(local).item()) == 0
    local.fill_(3)
    g = red.global_status(local)
    assert g.is_cuda and g.dtype == torch.int32 and int(g.item()) == 4
    side = torch.cuda.Stream(dev)                               # the next step's first bucket goes to the side stream, behind the compute stream
    side.wait_stream(torch.cuda.current_stream(dev))
    x = torch.randn(1 << 16, device=dev)
    y = x.clone()
    with torch.cuda.stream(side):
        red.comm.allreduce_mean_(y)
    torch.cuda.current_stream(dev).wait_stream(side)
    local.zero_()
    assert int(red.global_status(local).item()) == 0
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    red.close()
    tr.diffusion._reducer = None


def test_bench_line_through_the_rccl_path():
    """`bench.py` as the driver's torchrun starts it at N > 1, on the one GPU the pool has: a 1-rank torchrun job (WORLD_SIZE=1 set by
    the launcher, so bench does not re-spawn) with OD_FORCE_DDP=1 — NCCL process group, RCCL communicator through the C ABI, state
    broadcast, bucketed all-reduce overlapped with the backward, barrier + MAX over ranks — must print ONE JSON line carrying the
    collective block and a finite loss."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OD_FORCE_DDP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--frames", "1024", "--no-extras"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["collective"]["backend"].startswith("RCCL") and d["collective"]["version"] > 0
    assert d["final_loss"] == d["final_loss"] and d["value"] > 0
    # what the exchange moves and what it costs alone (the figures an N-rank line is sanity-checked against)
    c = d["collective"]
    assert abs(c["bytes_per_step"] - 187.5e6) < 1e6           # the arena (46,877,504 floats: every tensor's offset rounded to 64)
    assert c["buckets_per_step"] == 10 and c["allreduce_alone_ms"] >= 0.0


@pytest.mark.parametrize("priority", [0, -1])
def test_fused_attention_backward_beside_the_exchange(priority):
    """od_flash_attn_bwd_fused with the gradient exchange of a data-parallel step running beside it on the side stream — od_allreduce_grads on
    the arena's ten segments plus od_comm_ring_standin (resident channel workgroups streaming each bucket: what RCCL's ring kernels occupy;
    a one-rank all-reduce in place launches nothing) — at the bench shape: every launch's dq / dk / dv bit-identical to the quiet launch,
    status 0, with the side stream at default and at high priority.  (tools/soak_fused.py runs the same for 200 steps:
    profiles/r05_soak_fused_exchange.txt.)"""
    from osu_dreamer_amd import _lib
    _lib._lib = None
    _lib.lib()
    from tools.soak_fused import soak
    rec = soak(4, 32, 8192, mode="exchange", priority=priority, channels=32, threads=512, rounds=40)
    assert rec["differ"] == 0 and rec["status"] == 0 and rec["grads_intact"], rec
