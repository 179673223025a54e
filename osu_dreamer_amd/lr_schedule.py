"""Learning-rate multiplier for LambdaLR — osu_dreamer/common/lr_schedule.py:4-21."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(kw_only=True)
class LRScheduleArgs:
    warmup_steps: int = 0
    warmup_init: float = 1
    decay_start: float = float("inf")


def make_lr_schedule(lr: LRScheduleArgs):
    """Exponential warm-up from `warmup_init` to 1 over `warmup_steps`, flat, then
    1/sqrt(step/decay_start) after `decay_start`."""
    if isinstance(lr, dict):
        lr = LRScheduleArgs(**lr)
    if lr.warmup_steps > lr.decay_start:
        raise ValueError("warmup_steps must not exceed decay_start")
    w, w0, d = lr.warmup_steps, lr.warmup_init, lr.decay_start

    def multiplier(step: int) -> float:
        if step < w:
            return w0 ** (1 - step / w)
        if step > d:
            return (step / d) ** -0.5
        return 1.0

    return multiplier
