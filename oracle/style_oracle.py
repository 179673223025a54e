"""CPU oracle for the style model's distance-field sampler.  TEST INFRASTRUCTURE ONLY.

Functional torch restatement of osu_dreamer/models/style/model.py:29-119 (`StyleModel.forward`,
`compute_conditioning`, `sample`) over a flat dict keyed like `StyleModel.state_dict()` (the two random
Fourier-feature buffers `rff.W`, `rff.b` included).  Pinned by tests/golden/style_*.npz, which
oracle/make_golden.py produced by running the reference's StyleModel itself.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict

import torch

NUM_LABELS = 5


@dataclass(frozen=True)
class StyleDims:
    style_dim: int = 32
    label_features: int = 128
    h_dim: int = 256
    depth: int = 8
    expand: int = 4


STYLE_TINY = StyleDims(style_dim=8, label_features=16, h_dim=32, depth=2, expand=4)
STYLE_FULL = StyleDims()


def style_constants(style_dim: int):
    d0_sq = 2.0 * style_dim
    t99 = torch.tensor(2.3263478740408408).sigmoid().item()
    return (1 - t99) ** 2 * d0_sq, math.sqrt(d0_sq)      # c0, u_scale  (style/model.py:33-40)


def style_param_shapes(d: StyleDims):
    H, F, S = d.h_dim, d.label_features, d.style_dim
    s = {"rff.W": (F, 1), "rff.b": (F,), "cond_proj_w": (NUM_LABELS, F, H), "cond_proj_b": (NUM_LABELS, H),
         "null_labels": (NUM_LABELS, H), "proj_in.weight": (H, S), "proj_in.bias": (H,),
         "proj_out.0.weight": (H,), "proj_out.1.weight": (S, H), "proj_out.1.bias": (S,),
         "u_out.weight": (1, H), "u_out.bias": (1,)}
    for i in range(d.depth):
        s[f"films.{i}.weight"] = (3 * H, H); s[f"films.{i}.bias"] = (3 * H,)
    for i in range(d.depth):
        s[f"blocks.{i}.0.weight"] = (d.expand * H, H); s[f"blocks.{i}.0.bias"] = (d.expand * H,)
        s[f"blocks.{i}.3.weight"] = (H, d.expand * H); s[f"blocks.{i}.3.bias"] = (H,)
    return s


def init_style_params(d: StyleDims, seed: int) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name, shape in style_param_shapes(d).items():
        if name == "rff.W":
            t = torch.randn(shape, generator=g) * 32.0
        elif name == "rff.b":
            t = (torch.rand(shape, generator=g) * 2 - 1) * math.pi
        elif name == "proj_out.0.weight":
            t = 1 + 0.1 * torch.randn(shape, generator=g)
        elif name == "u_out.bias":
            t = torch.full(shape, -0.4328)
        else:
            fan = shape[-1] if len(shape) > 1 else shape[0]
            t = torch.randn(shape, generator=g) * (0.5 / math.sqrt(fan))
        P[name] = t
    return P


def rms_rows(x, eps):
    return x * torch.rsqrt((x * x).mean(dim=1, keepdim=True) + eps)


def conditioning(labels, P, d: StyleDims):
    """style/model.py:73-80: random Fourier features of label/10, one linear map per label, missing
    labels (negative) replaced by a learned vector, summed over the labels."""
    F = d.label_features
    x = labels[:, :, None] / 10                                              # B N 1
    rff = math.sqrt(2 / F) * torch.cos(x @ P["rff.W"].t() + P["rff.b"])      # B N F
    h = torch.einsum("bnf,nfh->bnh", rff, P["cond_proj_w"]) + P["cond_proj_b"]
    h = torch.where(labels[:, :, None] < 0, P["null_labels"][None], h)
    return h.sum(1)


def style_forward(st, labels, P, d: StyleDims):
    """(u, v) — style/model.py:82-100."""
    _, u_scale = style_constants(d.style_dim)
    H = d.h_dim
    c = conditioning(labels, P, d)
    x = st @ P["proj_in.weight"].t() + P["proj_in.bias"]
    for i in range(d.depth):
        ssg = c @ P[f"films.{i}.weight"].t() + P[f"films.{i}.bias"]
        scale, shift, gate = ssg[:, :H], ssg[:, H:2 * H], ssg[:, 2 * H:]
        h = rms_rows(x, 1e-6) * (1 + scale) + shift
        h = h @ P[f"blocks.{i}.0.weight"].t() + P[f"blocks.{i}.0.bias"]
        h = h * torch.sigmoid(h)
        h = h @ P[f"blocks.{i}.3.weight"].t() + P[f"blocks.{i}.3.bias"]
        x = x + rms_rows(h, 1e-6) * gate
    eps32 = torch.finfo(torch.float32).eps                                   # nn.RMSNorm(eps=None)
    v = (rms_rows(x, eps32) * P["proj_out.0.weight"]) @ P["proj_out.1.weight"].t() + P["proj_out.1.bias"]
    u = u_scale * torch.nn.functional.softplus(rms_rows(x, 1e-6) @ P["u_out.weight"].t() + P["u_out.bias"]).squeeze(-1)
    return u, v


@torch.no_grad()
def style_sample(labels, num_steps, s_init, P, d: StyleDims):
    """style/model.py:102-119 with the initial noise passed in."""
    c0, _ = style_constants(d.style_dim)
    s = s_init.clone()
    u0 = style_forward(s, labels, P, d)[0].mean().item()
    eta = 1.0 - (math.sqrt(c0) / max(u0, math.sqrt(c0) + 1e-6)) ** (1.0 / num_steps)
    for _ in range(num_steps):
        u, v = style_forward(s, labels, P, d)
        s = s - eta * u[:, None] * v
    return s
