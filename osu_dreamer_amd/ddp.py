"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI through the C ABI
(`od_allreduce_grads` / `od_broadcast_f32`, csrc/comm.hip).

The reference is single-device (models/diffusion/model.yml:11); the only exchange a
data-parallel denoiser needs is the mean of the 46.9 M fp32 parameter gradients (187.5 MB).
Because the backward is an explicit launch sequence over a flat gradient arena, the buckets
are simply the arena's contiguous segments in completion order — tail (proj_out/u_head),
layer 7 … layer 0 (23.4 MB each), head — each all-reduced on a side stream the moment the
backward kernels that write it have been enqueued, overlapping the remaining backward.  The
EMA copy is never reduced.  Gradient clipping happens after the exchange, so the norm is
identical on every rank without a second collective.

`torch.distributed` is used for rendezvous only: it ships the 128-byte RCCL unique id from rank 0 to the
other ranks (and provides the host-side barrier / step-agreement of the trainer shell).  When the bound
kernel library is the CPU emulator build (the test-suite's world-size-2 gloo run) there is no RCCL and the
same buckets go through `torch.distributed.all_reduce` on the gloo group instead.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from . import _lib

UNIQUE_ID_BYTES = 128


def torch_rccl_path() -> Optional[str]:
    """PyTorch's own librccl.so, so that this process holds ONE RCCL instance."""
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else None


class RcclComm:
    """A communicator created through the C ABI (ncclCommInitRank on the current device)."""

    def __init__(self, device: torch.device, process_group=None):
        L = _lib.lib()
        L.od_comm_load((os.environ.get("OD_RCCL_LIB") or torch_rccl_path() or "").encode())
        self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        ident = (ctypes.c_ubyte * UNIQUE_ID_BYTES)()
        if self.rank == 0:
            L.od_comm_unique_id(ident, UNIQUE_ID_BYTES)
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0, group=process_group)      # rendezvous only: 128 bytes of host memory
        ident = (ctypes.c_ubyte * UNIQUE_ID_BYTES).from_buffer_copy(box[0])
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(device):
            L.od_comm_init(ctypes.byref(self._comm), self.world, self.rank, ident, UNIQUE_ID_BYTES)
        self.version = int(L.cdll.od_comm_version())
        self.ranks_seen = int(L.cdll.od_comm_count(self._comm))     # ncclCommCount: what RCCL itself believes the world is
        if self.ranks_seen != self.world:
            raise RuntimeError(f"RCCL communicator holds {self.ranks_seen} ranks, torch.distributed says {self.world}")

    def allreduce_mean_(self, t: torch.Tensor):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        _lib.lib().od_allreduce_grads(self._comm, t.data_ptr(), t.numel(), 1, torch.cuda.current_stream(t.device).cuda_stream)

    def broadcast_(self, t: torch.Tensor, root: int = 0):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        _lib.lib().od_broadcast_f32(self._comm, t.data_ptr(), t.numel(), root, torch.cuda.current_stream(t.device).cuda_stream)

    def close(self, abort: bool = False):
        """`abort`: the failure path (a peer died, collectives in flight can never complete) — ncclCommAbort, no device sync."""
        if self._comm:
            if abort:
                _lib.lib().od_comm_abort(self._comm)
            else:
                torch.cuda.synchronize()
                _lib.lib().od_comm_destroy(self._comm)
            self._comm = ctypes.c_void_p()


class GradBucketReducer:
    def __init__(self, model, process_group=None, overlap: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.segments = model.arena.segments(model.args.backbone_args.depth)
        self.overlap = overlap
        self._handles: List = []
        self._comm_stream: Optional[torch.cuda.Stream] = None
        self.bucket_log: list = []
        self._done = set()
        # exposed communication: HIP events on the compute stream either side of its wait for the side stream, i.e. how long the
        # optimizer had to wait for the exchange after the backward's own kernels were done (0 when the overlap is perfect)
        self.time_exposed = False
        self._exposed_events: List = []
        dev = model.arena.data.device
        self.comm: Optional[RcclComm] = RcclComm(dev, process_group) if dev.type == "cuda" else None
        model._reducer = self

    # ---- start-up / resume: every rank continues from rank `src`'s state ------------------------------
    def _bcast(self, t: torch.Tensor, src: int):
        if self.comm is not None:
            self.comm.broadcast_(t, src)
        else:
            dist.broadcast(t, src=src, group=self.pg)

    def broadcast_parameters(self, src: int = 0):
        """Every rank starts from rank `src`'s weights (what DDP does at construction)."""
        self._bcast(self.model.arena.data, src)

    def broadcast_state(self, optimizer=None, ema=None, src: int = 0):
        """Parameters, AdamW moments / step count and the EMA copy: everything a step reads besides the batch.
        Identical seeding is not relied on (a resumed rank 0 and fresh peers would otherwise diverge silently)."""
        self.broadcast_parameters(src)
        if optimizer is not None:
            self._bcast(optimizer.exp_avg, src)
            self._bcast(optimizer.exp_avg_sq, src)
            meta = [optimizer.step_count]
            dist.broadcast_object_list(meta, src=src, group=self.pg)
            optimizer.step_count = int(meta[0])
        if ema is not None:
            self._bcast(ema.module.arena.data, src)
            meta = [ema.count]
            dist.broadcast_object_list(meta, src=src, group=self.pg)
            ema.count = int(meta[0])
            ema.n_averaged.fill_(ema.count)

    # ---- the per-step exchange -----------------------------------------------------------------------
    def _reduce(self, g: torch.Tensor):
        if self.comm is not None:
            self.comm.allreduce_mean_(g)          # enqueued on the current stream (the side stream when overlapping)
            return None
        return dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def segment_done(self, name: str):
        """Called by DenoiserEngine.backward when all kernels writing `name`'s gradients are enqueued."""
        if name in self._done:
            return
        self._done.add(name)
        s, e = self.segments[name]
        g = self.model.arena.ensure_grad()[s:e]
        self.bucket_log.append((name, e - s))   # the order the collectives were enqueued in (tests: identical on every rank)
        if g.is_cuda and self.overlap:
            if self._comm_stream is None:
                # OD_COMM_STREAM_PRIORITY: 0 (default) = a plain side stream; -1 = HIGH priority — the exchange's workgroups are dispatched
                # ahead of the backward's (the knob for the first real multi-GPU A/B: the ring's latency against the CUs it takes).  The
                # runtime clamps positive values to 0 (0 already is the lowest priority a stream can have): they are refused here rather
                # than recorded in a bench line as if they had meant something.
                import os
                prio = int(os.environ.get("OD_COMM_STREAM_PRIORITY", "0"))
                if prio > 0:
                    raise ValueError("OD_COMM_STREAM_PRIORITY > 0 has no effect (HIP clamps it to the default); use 0 or -1")
                self._comm_stream = torch.cuda.Stream(g.device, priority=prio) if prio else torch.cuda.Stream(g.device)
            self._comm_stream.wait_stream(torch.cuda.current_stream(g.device))
            with torch.cuda.stream(self._comm_stream):
                self._handles.append((self._reduce(g), g))
        else:
            self._handles.append((self._reduce(g), g))

    def wait(self):
        """Block the compute stream until every bucket has been reduced (called by the optimizer)."""
        for name in self.segments:              # anything backward did not report (e.g. frozen parts)
            if name not in self._done:
                self.segment_done(name)
        for h, g in self._handles:
            if h is not None:                   # gloo (emulator build): host-side handle, sum -> mean
                h.wait()
                g.div_(self.world)
        if self._comm_stream is not None:
            cur = torch.cuda.current_stream(self._comm_stream.device)
            if self.time_exposed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self._comm_stream)
                e1.record(cur)
                self._exposed_events.append((e0, e1))
            else:
                cur.wait_stream(self._comm_stream)
        self._handles.clear()
        self._done.clear()
        self.bucket_log = self.bucket_log[-64:]

    def global_status(self, local: torch.Tensor) -> torch.Tensor:
        """The ranks' attention-backward status words OR-ed into one device word every rank agrees on: a rank whose fused backward
        failed has already sent its incomplete / NaN gradients into the all-reduce, so EVERY rank must poison its norm and skip the update —
        with a rank-local word the healthy ranks would step on them and the replicas diverge.  `local`: this rank's int32[1] word.  One
        4-byte all-reduce per step on the compute stream (which has just waited for the bucket exchange: the communicator's collectives stay
        ordered).  Returns an int32[1] tensor: 0, or 4 = "some rank's attention backward failed" (this rank's own code stays in its workspace)."""
        if getattr(self, "_st_f", None) is None or self._st_f.device != local.device:
            self._st_f = torch.zeros(1, dtype=torch.float32, device=local.device)
            self._st_i = torch.zeros(1, dtype=torch.int32, device=local.device)
        self._st_f.copy_((local != 0).to(torch.float32))
        if self.comm is not None:
            self.comm.allreduce_mean_(self._st_f)
        else:
            dist.all_reduce(self._st_f, op=dist.ReduceOp.SUM, group=self.pg)
        self._st_i.copy_((self._st_f > 0).to(torch.int32) * 4)
        return self._st_i

    def exposed_ms(self, reset: bool = True) -> List[float]:
        """Per step since the last reset: milliseconds the compute stream spent waiting for the gradient exchange (synchronises)."""
        torch.cuda.synchronize()
        out = [e0.elapsed_time(e1) for e0, e1 in self._exposed_events]
        if reset:
            self._exposed_events.clear()
        return out

    def close(self, abort: bool = False):
        if self.comm is not None:
            self.comm.close(abort=abort)


class StepAgreement:
    """Keeps the ranks of a data-parallel epoch in lock step when their shards differ in length: before every
    optimizer step the ranks agree (MIN over a host-side flag) whether ALL of them still have a batch; the epoch
    ends for everyone as soon as one rank runs dry, so no rank is left waiting in an all-reduce."""

    def __init__(self, world: int):
        self.world = world
        self._group = None
        if world > 1:
            # a host-side (gloo) group: the agreement must not synchronise the device stream
            self._group = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else dist.group.WORLD

    def all_have(self, have_batch: bool) -> bool:
        if self.world <= 1:
            return have_batch
        flag = torch.tensor([1 if have_batch else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self._group)
        return bool(flag.item())
