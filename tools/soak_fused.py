"""Soak of od_flash_attn_bwd_fused: N launches on one workspace while a side stream keeps the chip unevenly busy; every launch's
dq / dk / dv must be bit-identical to the first (a quiet launch) and the workspace's status word 0.

    python tools/soak_fused.py [N] [B L] [--mode gemm|exchange] [--priority P] [--channels C] [--threads T] [--rounds R]

mode gemm      GEMMs of varying size on the side stream (round 4's soak).
mode exchange  the gradient exchange of a data-parallel step as ddp.GradBucketReducer issues it: the side stream waits for the compute
               stream, then all-reduces the arena's ten segments (tail, layer 7..0, head: 187.5 MB fp32) through od_allreduce_grads on a
               communicator the C ABI created — and, because a single-rank ncclAllReduce in place launches nothing, runs
               od_comm_ring_standin over each segment: `channels` resident workgroups streaming the bucket, the footprint of RCCL's ring
               kernels on the CUs and on HBM.  One "step" = eight fused launches (one per layer) with the exchange of the previous layers
               running beside them.  `--priority` is the side stream's priority (0 = default, -1 = high: the exchange's workgroups are
               dispatched ahead of the backward's; positive values are clamped to 0 by the runtime and mean nothing)."""
import argparse
import ctypes
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import _lib, ops  # noqa: E402


def world1_comm(dev):
    """An RCCL communicator of one rank through the C ABI (no torch.distributed needed for a 1-rank rendezvous)."""
    L = _lib.lib()
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    L.od_comm_load((os.environ.get("OD_RCCL_LIB") or (p if os.path.exists(p) else "")).encode())
    ident = (ctypes.c_ubyte * 128)()
    L.od_comm_unique_id(ident, 128)
    comm = ctypes.c_void_p()
    with torch.cuda.device(dev):
        L.od_comm_init(ctypes.byref(comm), 1, 0, ident, 128)
    return comm


def arena_segments(n_params=46_877_103, depth=8):
    """Sizes (floats) of the gradient arena's segments in the order the backward completes them."""
    layer = 5_850_000
    tail = 40_000
    head = n_params - depth * layer - tail
    return [tail] + [layer] * depth + [head]


def soak(n_steps, B, L, mode="exchange", priority=0, channels=32, threads=512, rounds=40, layers=8, dev=None, log=print):
    dev = dev or torch.device("cuda:0")
    H, hd = 16, 64
    M, dh = B * L, H * hd
    bf = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
    qk, qkv, do = r(M, 2 * dh), r(M, 3 * dh), r(M, dh)
    qk[:, :dh] = (qk[:, :dh].float() * math.log2(math.e) / 8).to(bf)
    o = torch.zeros(M, dh, dtype=bf, device=dev)
    lse = torch.zeros(B, H, L, device=dev)
    q, k, v = qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:]
    ops.flash_attn_fwd(q, k, v, o, lse, B, H, L, hd, 0.125, q_prescaled=True)
    ws = ops.FusedAttnBwdWorkspace(B, H, L, dev)
    out = [torch.zeros(M, 2 * dh, dtype=bf, device=dev), torch.zeros(M, 3 * dh, dtype=bf, device=dev)]

    def launch():
        out[0].fill_(float("nan")); out[1][:, 2 * dh:].fill_(float("nan"))
        ops.flash_attn_bwd_fused(q, k, v, o, do, lse, out[0][:, :dh], out[0][:, dh:], out[1][:, 2 * dh:], B, H, L, hd, 0.125, ws, q_prescaled=True)

    def timed(fn, reps=5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # the quiet reference
    launch()
    torch.cuda.synchronize()
    ref = (out[0].clone(), out[1][:, 2 * dh:].clone())
    assert not torch.isnan(ref[0].float()).any() and not torch.isnan(ref[1].float()).any()
    quiet_ms = timed(launch)

    side = torch.cuda.Stream(dev, priority=priority) if priority else torch.cuda.Stream(dev)
    cur = torch.cuda.current_stream(dev)
    comm, grads, segs = None, None, None
    if mode == "exchange":
        comm = world1_comm(dev)
        sizes = arena_segments()
        grads = torch.randn(sum(sizes), device=dev)
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        segs = [grads[offs[i]:offs[i + 1]] for i in range(len(sizes))]
        grads0 = grads.clone()
    noise_a = [torch.randn(s, s, device=dev, dtype=bf) for s in (512, 1024, 2048, 4096)]
    L_ = _lib.lib()

    def exchange(seg):
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            L_.od_allreduce_grads(comm, seg.data_ptr(), seg.numel(), 1, side.cuda_stream)
            L_.od_comm_ring_standin(seg.data_ptr(), seg.numel() // 4 * 4, channels, threads, rounds, side.cuda_stream)

    bad = launches = 0
    t0 = time.time()
    beside_ms = []
    for i in range(n_steps):
        if mode == "exchange":
            exchange(segs[0])                                  # tail: done before the first layer's backward
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for layer in range(layers):
                e0.record()
                launch()
                e1.record()
                cur_out = (out[0].clone(), out[1][:, 2 * dh:].clone())
                exchange(segs[1 + layer])                      # this layer's bucket runs beside the next layer's backward
                launches += 1
                if not (torch.equal(cur_out[0], ref[0]) and torch.equal(cur_out[1], ref[1])):
                    bad += 1
                    d = (cur_out[0].float() - ref[0].float()).abs()
                    log(f"step {i} layer {layer}: differs from the quiet launch in {int((d > 0).sum())} elements of dq/dk (max {float(d.max()):.3e})")
                if i == n_steps - 1:
                    torch.cuda.synchronize()
                    beside_ms.append(e0.elapsed_time(e1))
            exchange(segs[-1])
            cur.wait_stream(side)
        else:
            with torch.cuda.stream(side):                      # uneven load: a few GEMMs of a size that changes from launch to launch
                a = noise_a[i % 4]
                for _ in range(1 + i % 3):
                    a @ a
            launch()
            launches += 1
            cur_out = (out[0].clone(), out[1][:, 2 * dh:].clone())
            if not (torch.equal(cur_out[0], ref[0]) and torch.equal(cur_out[1], ref[1])):
                bad += 1
    torch.cuda.synchronize()
    status = ws.status()
    intact = True
    if mode == "exchange":
        intact = bool(torch.equal(grads, grads0))              # mean over one rank + the stand-in's write-back: unchanged
        L_.od_comm_destroy(comm)
    rec = dict(mode=mode, steps=n_steps, launches=launches, B=B, L=L, differ=bad, status=status, quiet_ms=round(quiet_ms, 3),
               beside_ms=[round(x, 3) for x in beside_ms], side_priority=priority, channels=channels, threads=threads, rounds=rounds,
               grads_intact=intact, wall_s=round(time.time() - t0, 1), kernel_src_sha=_lib.source_sha())
    log(str(rec))
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("n", nargs="?", type=int, default=200)
    ap.add_argument("B", nargs="?", type=int, default=32)
    ap.add_argument("L", nargs="?", type=int, default=8192)
    ap.add_argument("--mode", default="exchange")
    ap.add_argument("--priority", type=int, default=0)
    ap.add_argument("--channels", type=int, default=32)
    ap.add_argument("--threads", type=int, default=512)
    ap.add_argument("--rounds", type=int, default=40)
    a = ap.parse_args()
    rec = soak(a.n, a.B, a.L, a.mode, a.priority, a.channels, a.threads, a.rounds)
    sys.exit(1 if rec["differ"] or rec["status"] or not rec["grads_intact"] else 0)
