"""`predict` — the sampling entry of the hot path (reference: osu_dreamer/scripts/predict.py:28-77).

The reference command does five things: read audio tags, `load_inference(model_path)`, audio -> spectrogram
(`make_spec(load_wave(...))`, CPU: torchcodec + resonators), `model.sample(audio, labels, num_steps, show_progress)`,
and package the decoded beatmaps into an .osz.  The first, third and fifth are CPU pre/post-processing outside the
denoiser path; this entry is the part between them, with the same option names:

    python -m osu_dreamer_amd predict --model-path inference.pt --spec song.spec.npy \
        --diff 5.5 9 8 4 6 --diff 3.2 7 6 4 5 --sample-steps 8 --out pred.npz

`--spec` is what `make_spec` returns (float array (72, L)); `pred.npz` holds what `decode_beatmap` consumes:
`pred_signals` (B, 9, L) and `pred_labels` (B, 5), one row per `--diff`.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

PRECISIONS = {"fp32": (None, "f32"), "fp32_bf16x3": (None, "bf16x3"), "bf16": (torch.bfloat16, "f32")}


def predict(model_path: str, spec: np.ndarray, diff: Sequence[Sequence[float]], sample_steps: int = 8,
            precision: str = "fp32", show_progress: bool = False, device: str = "cuda",
            seed: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """scripts/predict.py:56-77 without the audio decode and the .osz packaging: artifact + spectrogram + difficulty
    rows -> (pred_signals (B, 9, L), pred_labels (B, 5)) as numpy arrays."""
    from . import _lib
    from .ldm import load_inference
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
    labels = np.asarray(diff, dtype=np.float32)
    if labels.ndim != 2 or labels.shape[1] != 5 or labels.shape[0] < 1:
        raise ValueError("--diff takes five numbers (sr, ar, od, cs, hp), at least once")
    _lib.lib()                                   # no CPU path: fail before loading anything else
    if seed is not None:
        torch.manual_seed(seed)
    model = load_inference(model_path, device=device)
    model.set_precision(*PRECISIONS[precision])
    dev = next(model.parameters()).device
    audio = torch.as_tensor(np.asarray(spec), device=dev).float()
    with torch.no_grad():
        signals, out_labels = model.sample(audio, torch.as_tensor(labels, device=dev), num_steps=sample_steps,
                                           show_progress=show_progress)
    return signals.cpu().numpy(), out_labels.cpu().numpy()


def add_parser(sub):
    p = sub.add_parser("predict", help="generate osu!std chart signals from a spectrogram (the sampler path of scripts/predict.py)")
    p.add_argument("--model-path", required=True, help="inference artifact (.pt)")
    p.add_argument("--spec", required=True, help=".npy spectrogram (72, L) as make_spec() returns it")
    p.add_argument("--diff", type=float, nargs=5, action="append", required=True, metavar=("SR", "AR", "OD", "CS", "HP"),
                   help="difficulty conditioning (sr, ar, od, cs, hp); repeat for several difficulties")
    p.add_argument("--sample-steps", type=int, default=8, help="number of diffusion steps to sample")
    p.add_argument("--precision", default="fp32", choices=sorted(PRECISIONS))
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--device", default="cuda")
    p.add_argument("--out", default="pred.npz", help="output .npz: pred_signals (B, 9, L), pred_labels (B, 5)")
    return p


def run(a):
    signals, labels = predict(a.model_path, np.load(a.spec), a.diff, a.sample_steps, a.precision, show_progress=True, device=a.device, seed=a.seed)
    np.savez(a.out, pred_signals=signals, pred_labels=labels)
    print(f"wrote {a.out}: pred_signals {signals.shape}, pred_labels {labels.shape}")
