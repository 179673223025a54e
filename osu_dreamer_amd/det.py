"""Deterministic gradients (OD_DETERMINISTIC=1, or `det.force(True)`): run-to-run bit-identical training steps.

The step forms its weight / bias / norm-weight / modulation gradients and its loss scalars with fp32 atomics (weight-gradient GEMM
split-K partials, one atomic per channel per block in the row kernels): the order blocks arrive in changes from run to run, and with it
the last bits of every gradient (5.9e-5 rel-L2 between two identical full-batch steps) — which makes "rank r diverged" unanswerable.
In this mode the SAME kernels add 2^40-scaled integers into 64-bit shadows of their destinations (od_det_*, csrc/det.hip: integer
addition is associative) and the engine folds each shadow into its fp32 destination before the destination's first reader
(DenoiserEngine._det_flush).  Nothing in the reference corresponds to this (torch's own autograd on this path is not
deterministic either); it is a debugging aid, off by default.  Cost: 8 bytes of shadow per gradient element (375 MB) and a flush pass
per arena segment.
"""
from __future__ import annotations

import os
import weakref
from typing import Dict, Optional, Tuple

import torch

from . import _lib

_forced: Optional[bool] = None
_ctx: Dict[str, "DetContext"] = {}

# workspace buffers (engine.Workspace names) that kernels accumulate into with atomics
WS_NAMES = ("d.ssg1", "d.ssg2", "d.cg", "fsum", "loss.dsq", "loss.sums")


def force(on: Optional[bool]):
    """Override the environment switch (tests).  None: follow OD_DETERMINISTIC again."""
    global _forced
    _forced = on
    if not on:
        for c in _ctx.values():
            c.close()
        _ctx.clear()


def enabled() -> bool:
    return _forced if _forced is not None else os.environ.get("OD_DETERMINISTIC", "0") == "1"


def context(device: torch.device) -> Optional["DetContext"]:
    """The process's table for `device` (one process drives one GPU), or None when the mode is off."""
    if not enabled():
        return None
    key = str(device)
    if key not in _ctx:
        for c in _ctx.values():          # the C side holds ONE table: a second device replaces it
            c.close()
        _ctx.clear()
        _ctx[key] = DetContext(device)
    return _ctx[key]


class DetContext:
    """The registered ranges follow the LIFETIME of their buffers: a range is held through a weak reference to its tensor, and ranges
    whose tensor has been freed (a Workspace the engine replaced on a new plan: the validation loader re-plans per map length) are dropped
    — with their int64 shadows — at the next registration, so the C side's fixed table (od_common.h OdDetTable) only ever holds the live
    plans' buffers.  A registration that does not fit leaves the previous table active and raises."""

    def __init__(self, device: torch.device):
        L = _lib.lib()
        self.device = device
        self.table = torch.zeros(int(L.cdll.od_det_table_bytes()), dtype=torch.uint8, device=device)
        self.ranges: Dict[Tuple[int, int], Tuple["weakref.ref[torch.Tensor]", torch.Tensor]] = {}
        L.od_det_clear()

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    def _upload(self, ranges):
        """Rebuild the C table from `ranges`; raises (HipKernelError) when they do not fit."""
        L = _lib.lib()
        L.od_det_clear()
        for (ptr, n), (_, sh) in ranges.items():
            L.od_det_register(ptr, n, sh.data_ptr())
        L.od_det_enable(self.table.data_ptr(), self._stream())

    def live_ranges(self):
        return {k: v for k, v in self.ranges.items() if v[0]() is not None}

    def register(self, t: torch.Tensor):
        """`t`: a contiguous fp32 buffer kernels accumulate into.  Idempotent; a buffer that was re-allocated replaces its old entry."""
        assert t.dtype == torch.float32 and t.is_contiguous()
        key = (t.data_ptr(), t.numel())
        cur = self.ranges.get(key)
        if cur is not None and cur[0]() is t:
            return
        # drop entries whose tensor is gone (freed workspaces) or overlaps the new one (a re-allocated arena, a recycled address)
        lo, hi = t.data_ptr(), t.data_ptr() + t.numel() * 4
        new = {k: v for k, v in self.live_ranges().items() if not (k[0] < hi and lo < k[0] + k[1] * 4)}
        new[key] = (weakref.ref(t), torch.zeros(t.numel(), dtype=torch.int64, device=t.device))
        try:
            self._upload(new)
        except Exception:
            self._upload(self.live_ranges())      # the previous table stays active
            raise
        self.ranges = new

    def flush(self, t: torch.Tensor):
        """Fold the shadow of `t` (a registered buffer or a contiguous slice of one) into it: call before its first reader."""
        _lib.lib().od_det_flush(t.data_ptr(), t.numel(), self._stream())

    def close(self):
        _lib.lib().od_det_clear()
        self.ranges.clear()
