"""`DiffusionModel` — drop-in for osu_dreamer/models/diffusion/model.py:24-138 whose arithmetic
runs on hand-written HIP kernels.

Same constructor (`DiffusionModel(emb_dim, a_dim, style_dim, args)`), same attributes
(`c0`, `u_scale`, `emb_dim`, `style_dim`), same `forward(audio, style, xt) -> (u, v)` and
`sample(audio, style, num_steps, show_progress)`, and the same `state_dict()` keys / shapes
(329 entries at default hparams) so reference checkpoints and `export-inference` re-keying
(osu_dreamer/models/inference/artifact.py:24-36) interchange.

All parameters live in one flat fp32 arena (`ParamArena`): every nn.Parameter is a view of it,
so the optimizer, EMA, gradient norm and the RCCL all-reduce are single passes over one
buffer, and gradients are written by the backward kernels straight into `arena.grad`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import ops
from .engine import DenoiserEngine


@dataclass
class BackboneArgs:
    """models/diffusion/backbone.py:18-25"""
    depth: int
    expand: int
    head_dim: int
    n_heads: int
    radius: int = 1
    dropout: float = 0.


@dataclass
class DiffusionModelArgs:
    """models/diffusion/model.py:16-21"""
    global_cond_dim: int
    backbone_dim: int
    backbone_args: BackboneArgs
    u_head_dim: int = 64


def _coerce_args(args) -> DiffusionModelArgs:
    if isinstance(args, dict):
        ba = args["backbone_args"]
        ba = BackboneArgs(**ba) if isinstance(ba, dict) else ba
        return DiffusionModelArgs(**{**args, "backbone_args": ba})
    if isinstance(args.backbone_args, dict):
        args.backbone_args = BackboneArgs(**args.backbone_args)
    return args


def parameter_inventory(emb_dim: int, a_dim: int, style_dim: int, args: DiffusionModelArgs) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(dotted name, shape, init rule) in the reference's registration order.
    init rules: 'fan_in' = U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (torch's default for Conv1d/Linear
    weight and bias), 'zero' (backbone.py:12-16 `zero`, model.py:51-53,66-68), 'one' (nn.RMSNorm),
    'u_out_bias' (model.py:71)."""
    ba = args.backbone_args
    D, A, Cg, U, E = args.backbone_dim, a_dim, args.global_cond_dim, args.u_head_dim, emb_dim
    dh = ba.n_heads * ba.head_dim
    hf = int(D * ba.expand * 2 / 3)
    k = 1 + 2 * ba.radius
    inv: List[Tuple[str, Tuple[int, ...], str]] = []

    def wb(name, wshape, rule="fan_in"):
        inv.append((name + ".weight", tuple(wshape), rule))
        inv.append((name + ".bias", (wshape[0],), rule))

    wb("proj_audio.0", (A, A, 1))
    wb("proj_style.0", (Cg, style_dim))
    wb("proj_in", (D, E, 1))
    for i in range(ba.depth):
        p = f"net.layers.{i}."
        wb(p + "ssg1", (3 * D, Cg), "zero")
        wb(p + "proj_cl", (D, A, 1))
        wb(p + "attn.qkv_proj", (3 * dh, D, 1))
        wb(p + "attn.out_proj", (D, dh, 1))
        inv.append((p + "attn.q_norm.weight", (ba.head_dim,), "one"))
        inv.append((p + "attn.k_norm.weight", (ba.head_dim,), "one"))
        wb(p + "ssg2", (3 * D, Cg), "zero")
        if ba.radius > 0:
            wb(p + "ffn.proj_vg.0", (D, 1, k))
        wb(p + "ffn.proj_vg.1", (2 * hf, D, 1))
        wb(p + "ffn.proj_o", (D, hf, 1))
    wb("proj_out", (E, D, 1), "zero")
    wb("u_head.0", (E, 1, 3))
    wb("u_head.1", (U, E, 1))
    wb("u_head.3", (U, 1, 3))
    wb("u_head.4", (U, U, 1))
    wb("u_mod", (2 * U, Cg), "zero")
    inv.append(("u_out.weight", (1, U), "zero"))
    inv.append(("u_out.bias", (1,), "u_out_bias"))
    return inv


class ParamArena:
    """One flat fp32 buffer for all parameters (+ a twin for gradients).  Offsets are rounded
    to 64 floats so every tensor is 256-byte aligned for the vector loads in the kernels."""

    ALIGN = 64

    def __init__(self, inventory, device="cpu"):
        self.entries: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for name, shape, _ in inventory:
            n = int(math.prod(shape))
            self.entries[name] = (off, shape)
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad: Optional[torch.Tensor] = None

    def view(self, name: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        off, shape = self.entries[name]
        b = self.data if buf is None else buf
        return b[off: off + int(math.prod(shape))].view(shape)

    def ensure_grad(self) -> torch.Tensor:
        if self.grad is None or self.grad.device != self.data.device:
            self.grad = torch.zeros_like(self.data)
        return self.grad

    def grad_view(self, name: str) -> torch.Tensor:
        return self.view(name, self.ensure_grad())

    def segments(self, depth: int) -> Dict[str, Tuple[int, int]]:
        """Contiguous [start, end) ranges of the arena in backward-completion order:
        'tail' (proj_out, u_head, u_mod, u_out), 'layer{i}', 'head' (proj_audio, proj_style, proj_in)."""
        names = list(self.entries)
        ends = {n: self.entries[n][0] for n in names}
        first_layer = self.entries["net.layers.0.ssg1.weight"][0]
        tail_start = self.entries["proj_out.weight"][0]
        seg = {"head": (0, first_layer), "tail": (tail_start, self.numel)}
        for i in range(depth):
            s = self.entries[f"net.layers.{i}.ssg1.weight"][0]
            e = self.entries[f"net.layers.{i + 1}.ssg1.weight"][0] if i + 1 < depth else tail_start
            seg[f"layer{i}"] = (s, e)
        return seg


class _Node(nn.Module):
    """Bare container so dotted parameter names reproduce the reference's module tree."""


def _init_rule_(t: torch.Tensor, rule: str, fan_in: int):
    if rule == "zero":
        t.zero_()
    elif rule == "one":
        t.fill_(1.0)
    elif rule == "u_out_bias":
        t.fill_(-0.4328)
    else:
        bound = 1.0 / math.sqrt(max(fan_in, 1))
        t.uniform_(-bound, bound)


class DiffusionModel(nn.Module):
    def __init__(self, emb_dim: int, a_dim: int, style_dim: int, args: DiffusionModelArgs):
        super().__init__()
        args = _coerce_args(args)
        self.emb_dim, self.a_dim, self.style_dim, self.args = emb_dim, a_dim, style_dim, args
        d0_sq = 2.0 * emb_dim                                            # model.py:35-43
        t99 = torch.tensor(2.3263478740408408).sigmoid().item()
        self.c0 = (1 - t99) ** 2 * d0_sq
        self.u_scale = math.sqrt(d0_sq)
        self.compute_dtype: Optional[torch.dtype] = None                 # None: follow autocast, else fp32
        # torch.float16: "attention in fp16" — q, k, v reach the attention core as IEEE half and its MFMAs are the f16 ones, with bf16
        # compute around it (BASELINE configs[4]; what Lightning's precision: 16-mixed, model.yml:12, asks of attn.py:82)
        self.attn_dtype: Optional[torch.dtype] = None
        # fp32 no-grad forward / sampler: "f32" = exact fp32 MFMA chain (parity mode), "bf16x3" = three
        # bf16 MFMAs per product (~4e-6 per GEMM, 1.7x faster).  Training always uses "f32".
        self.f32_matmul = "f32"
        self.use_graph = True                                            # hipGraph the sampler loop

        self._inventory = parameter_inventory(emb_dim, a_dim, style_dim, args)
        self.arena = ParamArena(self._inventory)
        weights = {n: s for n, s, _ in self._inventory if n.endswith(".weight")}
        with torch.no_grad():
            for name, shape, rule in self._inventory:
                wshape = weights.get(name[:-4] + "weight", shape) if name.endswith(".bias") else shape
                fan_in = int(math.prod(wshape[1:])) if len(wshape) > 1 else 1
                _init_rule_(self.arena.view(name), rule, fan_in)
        self._register_views()
        self._engine: Optional[DenoiserEngine] = None
        self._graph = None

    # ---- parameter plumbing -------------------------------------------------------
    def _register_views(self):
        for name, _, _ in self._inventory:
            parts = name.split(".")
            mod = self
            for part in parts[:-1]:
                if part not in mod._modules:
                    mod.add_module(part, _Node())
                mod = mod._modules[part]
            old = mod._parameters.get(parts[-1])
            # .to()/.cuda() re-create the views: keep requires_grad (EMA / inference copies are frozen)
            mod._parameters[parts[-1]] = nn.Parameter(self.arena.view(name), requires_grad=True if old is None else old.requires_grad)

    def _apply(self, fn, recurse=True):
        """`.to()/.cuda()/.float()` move the arena as a whole and re-create the views."""
        new = fn(self.arena.data)
        if new.dtype != torch.float32:
            raise TypeError("DiffusionModel keeps fp32 master parameters; use compute_dtype / autocast for bf16")
        if new is not self.arena.data:
            self.arena.data = new.contiguous()
            self.arena.grad = None
            self._register_views()
            self._engine, self._graph = None, None
        return self

    def attach_grads(self):
        """Point every parameter's .grad at its slice of the arena's grad buffer (zeroing it if
        any .grad was dropped, e.g. by zero_grad(set_to_none=True))."""
        g = self.arena.ensure_grad()
        fresh = False
        for name, p in self.named_parameters():
            gv = self.arena.grad_view(name)
            if p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                p.grad = gv
                fresh = True
        if fresh:
            g.zero_()
        return g

    def adopt_grads(self):
        """Before an optimizer step: make the arena's gradient buffer hold what every parameter's `.grad` says.  Normally they are the same
        memory (attach_grads) and nothing is copied.  A wrapper that RE-POINTS `.grad` after backward — torch DDP with
        `gradient_as_bucket_view=True` leaves each `.grad` a view of its all-reduced bucket — is honoured by copying those gradients into the
        arena (the fused optimizer reads the arena; re-attaching alone would have zeroed it and stepped on nothing)."""
        g = self.arena.ensure_grad()
        for name, p in self.named_parameters():
            gv = self.arena.grad_view(name)
            if p.grad is None:
                gv.zero_()
                p.grad = gv
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad.reshape(gv.shape))
                p.grad = gv
        return g

    @property
    def engine(self) -> DenoiserEngine:
        if self._engine is None:
            self._engine = DenoiserEngine(self)
        return self._engine

    def _dtype(self) -> torch.dtype:
        if self.compute_dtype is not None:
            return self.compute_dtype
        dev = self.arena.data.device.type
        if torch.is_autocast_enabled(dev) if hasattr(torch, "is_autocast_enabled") else False:
            dt = torch.get_autocast_dtype(dev)
            if dt == torch.bfloat16:
                return torch.bfloat16
        return torch.float32

    def _x3(self) -> bool:
        if self.f32_matmul not in ("f32", "bf16x3"):
            raise ValueError(f"f32_matmul must be 'f32' or 'bf16x3', got {self.f32_matmul!r}")
        return self.f32_matmul == "bf16x3"

    @staticmethod
    def _f32c(t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(torch.float32).contiguous()

    # ---- forward (model.py:105-114) --------------------------------------------------
    def forward(self, audio: torch.Tensor, style: torch.Tensor, xt: torch.Tensor):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _DenoiserFn.apply(self, audio, style, xt, *self.parameters())
        return self._forward_nograd(audio, style, xt)

    def _forward_nograd(self, audio, style, xt):
        audio, style, xt = self._f32c(audio), self._f32c(style), self._f32c(xt)
        B, _, L = xt.shape
        eng, dt = self.engine, self._dtype()
        eng.pack_weights(dt, train=False, x3=self._x3())
        eng.plan(B, L, audio.shape[0], dt, train=False, x3=self._x3())
        eng.conditioning(audio, style)
        u = torch.empty(B, dtype=torch.float32, device=xt.device)
        v = torch.empty_like(xt)
        eng.pred(xt, u, v)
        return u, v

    # ---- sampler (model.py:117-138) -----------------------------------------------------
    @torch.no_grad()
    def sample(self, audio: torch.Tensor, style: torch.Tensor, num_steps: int, show_progress: bool = False,
               x_init: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Sphere tracing with a self-calibrating step.  `x_init` (optional, not in the reference's
        signature) pins the starting noise for parity tests; default draws N(0, I) like the reference."""
        audio, style = self._f32c(audio), self._f32c(style)
        B, L, dev = style.shape[0], audio.shape[-1], audio.device
        x = torch.randn(B, self.emb_dim, L, device=dev) if x_init is None else self._f32c(x_init).clone()
        eng, dt = self.engine, self._dtype()
        eng.pack_weights(dt, train=False, x3=self._x3())
        eng.plan(B, L, audio.shape[0], dt, train=False, x3=self._x3())
        eng.conditioning(audio, style)                   # loop invariants, once
        u = eng.buf("smp.u", (B,), torch.float32)
        v = eng.buf("smp.v", (B, self.emb_dim, L), torch.float32)
        eta = eng.buf("smp.eta", (2,), torch.float32)
        xs = eng.buf("smp.x", (B, self.emb_dim, L), torch.float32)
        xs.copy_(x)
        eng.pred(xs, u, v)                               # u0 = mean u(x_init)
        ops.sampler_eta(u, eta, self.c0, num_steps)      # eta stays on the device: no host sync
        self.last_sample_stats = eta

        def step():
            eng.pred(xs, u, v)
            ops.sampler_step(xs, u, v, eta)

        steps = range(num_steps)
        if show_progress:
            try:
                import tqdm
                steps = tqdm.trange(num_steps)
            except ImportError:
                pass
        if self.use_graph and dev.type == "cuda" and num_steps > 1 and not eng._drop_active():     # (a training-mode Dropout1d draws a mask per step)
            # the captured step only references plan-owned buffers (x, u, v, eta, workspace, packed weights), so one
            # instantiated graph serves later calls — as long as those buffers are the SAME allocations: a forward or
            # train step with another shape in between re-plans (new workspace / packed weights), which bumps
            # `generation` and retires the graph even when the plan key comes back equal
            from .graph import CapturedLoop
            from . import det
            key = (eng._plan_key, eng._packed_key, eng.generation, det.enabled())     # (a captured step carries the deterministic mode's launches)
            if self._graph is None or self._graph[0] != key:
                if self._graph is not None:
                    self._graph[1].close()
                self._graph = (key, CapturedLoop(step, dev))
            loop = self._graph[1]
            loop.begin()
            for _ in steps:
                loop.replay()
            loop.end()
        else:
            for _ in steps:
                step()
        return xs.clone()


class _DenoiserFn(torch.autograd.Function):
    """Autograd glue for `DiffusionModel.forward`: one node for the whole network.  Parameter
    gradients are accumulated by the backward kernels directly into `arena.grad` (which every
    parameter's .grad aliases), so None is returned for them."""

    @staticmethod
    def forward(ctx, model: DiffusionModel, audio, style, xt, *params):
        audio, style, xt = model._f32c(audio), model._f32c(style), model._f32c(xt)
        B, _, L = xt.shape
        if audio.shape[0] == 1 and B > 1:
            audio = audio.expand(B, -1, -1).contiguous()   # training never broadcasts (data/modules/latent.py:55-62)
        eng, dt = model.engine, model._dtype()
        eng.pack_weights(dt, train=True)
        eng.plan(B, L, audio.shape[0], dt, train=True)
        eng.conditioning(audio, style)
        u = torch.empty(B, dtype=torch.float32, device=xt.device)
        v = torch.empty_like(xt)
        eng.pred(xt, u, v)
        ctx.model, ctx.xt, ctx.style = model, xt, style
        ctx.nparams = len(params)
        return u, v

    @staticmethod
    def backward(ctx, du, dv):
        model = ctx.model
        model.attach_grads()
        B = ctx.xt.shape[0]
        du = torch.zeros(B, device=ctx.xt.device) if du is None else du.to(torch.float32).contiguous()
        dv = torch.zeros_like(ctx.xt) if dv is None else dv.to(torch.float32).contiguous()
        model.engine.backward(ctx.xt, ctx.style, du, dv, reducer=getattr(model, "_reducer", None))
        return (None, None, None, None) + (None,) * ctx.nparams
