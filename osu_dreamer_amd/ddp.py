"""Data-parallel gradient exchange: one process per GPU, `torch.distributed` (backend "nccl"
is RCCL on ROCm) over xGMI.

The reference is single-device (models/diffusion/model.yml:11); the only exchange a
data-parallel denoiser needs is the mean of the 46.9 M fp32 parameter gradients (187.5 MB).
Because the backward is an explicit launch sequence over a flat gradient arena, the buckets
are simply the arena's contiguous segments in completion order — tail (proj_out/u_head),
layer 7 … layer 0 (23.4 MB each), head — each all-reduced on a side stream the moment the
backward kernels that write it have been enqueued, overlapping the remaining backward.  The
EMA copy is never reduced.  Gradient clipping happens after the exchange, so the norm is
identical on every rank without a second collective.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


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
        self._done = set()
        backend = dist.get_backend(process_group)
        self._avg = backend == "nccl"
        model._reducer = self

    def broadcast_parameters(self, src: int = 0):
        """Every rank starts from rank `src`'s weights (what DDP does at construction)."""
        dist.broadcast(self.model.arena.data, src=src, group=self.pg)

    def _reduce(self, t: torch.Tensor):
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        return dist.all_reduce(t, op=op, group=self.pg, async_op=True)

    def segment_done(self, name: str):
        """Called by DenoiserEngine.backward when all kernels writing `name`'s gradients are enqueued."""
        if name in self._done:
            return
        self._done.add(name)
        s, e = self.segments[name]
        g = self.model.arena.ensure_grad()[s:e]
        if g.is_cuda and self.overlap:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(g.device)
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
            h.wait()
            if not self._avg:
                g.div_(self.world)
        if self._comm_stream is not None:
            torch.cuda.current_stream(self._comm_stream.device).wait_stream(self._comm_stream)
        self._handles.clear()
        self._done.clear()
