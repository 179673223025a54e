"""`StyleModel` inference on the HIP path — the step before `diffusion.sample` in `LDM.sample`
(osu_dreamer/models/inference/model.py:49; reference model: osu_dreamer/models/style/model.py:29-119).

Same constructor, state-dict keys (`rff.W`, `rff.b` buffers included), `forward(st, labels) -> (u, v)`,
`compute_conditioning(labels)` and `sample(labels, num_steps=16)`.  The conditioning and every FiLM
(`films[i](c)`) depend only on the labels, so `sample` computes them once instead of on each of its
`num_steps + 1` evaluations; the step body is captured into a hipGraph like the denoiser's.  Training of
the style model is out of scope (inference only; parameters do not receive gradients here).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import nn

from . import ops
from ._lib import OD_ACT_NONE, OD_ACT_SILU

NUM_LABELS = 5
FP32_EPS = float(torch.finfo(torch.float32).eps)


@dataclass
class StyleModelArgs:
    label_features: int
    h_dim: int
    depth: int
    expand: int
    dropout: float = 0.


class _Node(nn.Module):
    pass


class StyleModel(nn.Module):
    def __init__(self, style_dim: int, args: StyleModelArgs):
        super().__init__()
        if isinstance(args, dict):
            args = StyleModelArgs(**args)
        self.style_dim, self.args = style_dim, args
        d0_sq = 2.0 * style_dim
        t99 = torch.tensor(2.3263478740408408).sigmoid().item()
        self.c0 = (1 - t99) ** 2 * d0_sq
        self.u_scale = math.sqrt(d0_sq)
        self.use_graph = True
        H, F, S = args.h_dim, args.label_features, style_dim

        def P(*shape, std=None):
            t = torch.empty(*shape)
            fan = shape[-1] if len(shape) > 1 else shape[0]
            return nn.Parameter(t.normal_(0, std if std is not None else 1.0 / math.sqrt(fan)))

        self.rff = _Node()
        self.rff.register_buffer("W", torch.randn(F, 1) * 32.0)              # FourierFeatures(1, F, n_bins=32)
        self.rff.register_buffer("b", torch.empty(F).uniform_(-math.pi, math.pi))
        self.cond_proj_w = P(NUM_LABELS, F, H, std=math.sqrt(2.0 / (F + H)))
        self.cond_proj_b = nn.Parameter(torch.zeros(NUM_LABELS, H))
        self.null_labels = P(NUM_LABELS, H, std=H ** -0.5)
        self.proj_in = _Node(); self.proj_in.weight = P(H, S); self.proj_in.bias = nn.Parameter(torch.zeros(H))
        self.proj_out = _Node()
        n0, n1 = _Node(), _Node()
        n0.weight = nn.Parameter(torch.ones(H))
        n1.weight = nn.Parameter(torch.zeros(S, H)); n1.bias = nn.Parameter(torch.zeros(S))
        self.proj_out.add_module("0", n0); self.proj_out.add_module("1", n1)
        self.u_out = _Node(); self.u_out.weight = nn.Parameter(torch.zeros(1, H)); self.u_out.bias = nn.Parameter(torch.full((1,), -0.4328))
        self.films, self.blocks = _Node(), _Node()
        for i in range(args.depth):
            f = _Node(); f.weight = nn.Parameter(torch.zeros(3 * H, H)); f.bias = nn.Parameter(torch.zeros(3 * H))
            self.films.add_module(str(i), f)
            blk, l0, l3 = _Node(), _Node(), _Node()
            l0.weight = P(args.expand * H, H); l0.bias = nn.Parameter(torch.zeros(args.expand * H))
            l3.weight = P(H, args.expand * H); l3.bias = nn.Parameter(torch.zeros(H))
            blk.add_module("0", l0); blk.add_module("3", l3)
            self.blocks.add_module(str(i), blk)
        self.requires_grad_(False)
        self._buf: Dict[str, torch.Tensor] = {}
        self._buf_gen = 0          # bumped on every (re)allocation: a captured graph holding old addresses is stale

    # ------------------------------------------------------------------
    def _b(self, name, shape, like):
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != like.device:
            t = torch.zeros(shape, dtype=torch.float32, device=like.device)
            self._buf[name] = t
            self._buf_gen += 1
        return t

    def _w(self):
        return {k: v.detach() for k, v in self.state_dict().items()}

    def compute_conditioning(self, labels: torch.Tensor) -> torch.Tensor:
        labels = labels.detach().float().contiguous()
        W = self._w()
        c = self._b("c", (labels.shape[0], self.args.h_dim), labels)
        ops.style_conditioning(labels, W["rff.W"].contiguous(), W["rff.b"], W["cond_proj_w"], W["cond_proj_b"],
                               W["null_labels"], c)
        return c

    def _films(self, c, W):
        B, H = c.shape
        out = []
        for i in range(self.args.depth):
            ssg = self._b(f"ssg.{i}", (B, 3 * H), c)
            ops.linear_small(c, W[f"films.{i}.weight"], W[f"films.{i}.bias"], ssg)
            out.append(ssg)
        return out

    def _eval(self, st, ssgs, W, u, v):
        """One network evaluation given the (label-only) FiLM tensors."""
        B, H, a = st.shape[0], self.args.h_dim, self.args
        x, h = self._b("x", (B, H), st), self._b("h", (B, H), st)
        h1, h2 = self._b("h1", (B, a.expand * H), st), self._b("h2", (B, H), st)
        inv = self._b("inv", (B,), st)
        ops.linear_small(st, W["proj_in.weight"], W["proj_in.bias"], x)
        for i in range(a.depth):
            ops.rmsnorm_film(x, ssgs[i], None, False, h, inv, B, 1)
            ops.linear_small(h, W[f"blocks.{i}.0.weight"], W[f"blocks.{i}.0.bias"], h1, None, OD_ACT_SILU)
            ops.linear_small(h1, W[f"blocks.{i}.3.weight"], W[f"blocks.{i}.3.bias"], h2)
            ops.rmsnorm_gate_residual(x, h2, ssgs[i], x, inv, B, 1)
        xn = self._b("xn", (B, H), st)
        ops.rmsnorm_rows(x, W["proj_out.0.weight"], xn, FP32_EPS)
        ops.linear_small(xn, W["proj_out.1.weight"], W["proj_out.1.bias"], v)
        ops.rmsnorm_rows(x, None, xn, 1e-6)
        zero_mod = self._b("zero_mod", (B, 2 * H), st)
        ops.uhead_tail(xn, zero_mod, W["u_out.weight"], W["u_out.bias"], u, 1, self.u_scale)

    @torch.no_grad()
    def forward(self, st: torch.Tensor, labels: torch.Tensor):
        st = st.detach().float().contiguous()
        W = self._w()
        ssgs = self._films(self.compute_conditioning(labels), W)
        B = st.shape[0]
        u = torch.empty(B, dtype=torch.float32, device=st.device)
        v = torch.empty(B, self.style_dim, dtype=torch.float32, device=st.device)
        self._eval(st, ssgs, W, u, v)
        return u, v

    @torch.no_grad()
    def sample(self, labels: torch.Tensor, num_steps: int = 16, s_init: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, dev = labels.shape[0], labels.device
        s = torch.randn(B, self.style_dim, device=dev) if s_init is None else s_init.detach().float().clone()
        W = self._w()
        ssgs = self._films(self.compute_conditioning(labels), W)          # label-only: once per call
        u, eta = self._b("smp.u", (B,), s), self._b("smp.eta", (2,), s)
        v = self._b("smp.v", (B, self.style_dim), s)
        x = self._b("smp.s", (B, self.style_dim, 1), s)
        x.copy_(s.view(B, self.style_dim, 1))
        xs = x.view(B, self.style_dim)
        self._eval(xs, ssgs, W, u, v)
        ops.sampler_eta(u, eta, self.c0, num_steps)

        def step():
            self._eval(xs, ssgs, W, u, v)
            ops.sampler_step(x, u, v.view(B, self.style_dim, 1), eta)

        if self.use_graph and dev.type == "cuda" and num_steps > 1:
            from .graph import CapturedLoop
            key = (B, dev, self._buf_gen, tuple(t.data_ptr() for t in W.values()))
            if getattr(self, "_graph", None) is None or self._graph[0] != key:
                if getattr(self, "_graph", None) is not None:
                    self._graph[1].close()
                self._graph = (key, CapturedLoop(step, dev))
            loop = self._graph[1]
            loop.begin()
            for _ in range(num_steps):
                loop.replay()
            loop.end()
        else:
            for _ in range(num_steps):
                step()
        return xs.clone()
