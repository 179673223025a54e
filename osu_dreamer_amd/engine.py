"""Launch plan for the denoiser: the explicit forward / backward kernel sequence over
pre-allocated frame-major workspaces.

This is the host-side replacement of the reference's module graph
(osu_dreamer/models/diffusion/model.py:86-103 `_pred`, backbone.py:69-88
`BackboneLayer.forward`, common/attn.py:74-84, common/swiglu.py:27-32) and of the autograd
tape torch would record for it.  Every arithmetic step is an `od_*` HIP kernel
(osu_dreamer_amd/ops.py); torch tensors are storage only.  Nothing is allocated inside
`pred` / `forward_train` / `backward`, so a sampler step is hipGraph-capturable.

Layout: activations are [M = B*L, C] frame-major (csrc/od_common.h).  The SwiGLU hidden
width Hf = int(D*expand*2/3) (1365 at default hparams, odd) is padded to Hp = ceil64(Hf):
packed vg weights put v at rows [0,Hf) and g at rows [Hp,Hp+Hf) with zero rows between,
so every kernel sees aligned, zero-padded columns.
"""
from __future__ import annotations


import math
import os
from typing import Dict, List, Optional

import torch

from . import det, ops
from ._lib import OD_ACT_NONE, OD_ACT_SILU

OD_FUSE_FILM_DWCONV_DEFAULT = "0"       # (see DenoiserEngine.pred)

FP32_EPS = float(torch.finfo(torch.float32).eps)   # nn.RMSNorm(eps=None), common/attn.py:71-72


def _ceil(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class Workspace:
    """Named device buffers, allocated once per plan."""

    def __init__(self, device):
        self.device = device
        self.t: Dict[str, torch.Tensor] = {}
        self.bytes = 0

    def get(self, name: str, shape, dtype) -> torch.Tensor:
        t = self.t.get(name)
        if t is None:
            t = torch.zeros(shape, dtype=dtype, device=self.device)
            self.t[name] = t
            self.bytes += t.numel() * t.element_size()
            if name in det.WS_NAMES:              # OD_DETERMINISTIC: a buffer kernels accumulate into with atomics gets its integer shadow
                ctx = det.context(self.device)
                if ctx is not None:
                    ctx.register(t)
        return t


class DenoiserEngine:
    def __init__(self, model):
        self.model = model
        a = model.args
        ba = a.backbone_args
        self.E, self.A, self.S = model.emb_dim, model.a_dim, model.style_dim
        self.Cg, self.D, self.U = a.global_cond_dim, a.backbone_dim, a.u_head_dim
        self.H, self.hd, self.depth = ba.n_heads, ba.head_dim, ba.depth
        self.dh = self.H * self.hd
        # q leaves the norm + RoPE step multiplied by scale * log2(e): q.k is then the softmax's base-2 exponent and the
        # attention loops drop one multiply per score (od_flash_attn_*'s q_prescaled mode)
        self.q_scale = math.log2(math.e) / math.sqrt(self.hd)
        self.radius = ba.radius
        self.ksize = 1 + 2 * ba.radius
        self.Hf = int(self.D * ba.expand * 2 / 3)
        self.Hp = _ceil(self.Hf, 64)
        if ba.radius not in (0, 1, 2, 3, 4):
            raise NotImplementedError("depthwise kernel sizes 3, 5, 7, 9 (radius 1-4) are compiled; radius 0 is the identity")
        if not (0.0 <= ba.dropout < 1.0):
            raise ValueError(f"dropout probability has to be in [0, 1), got {ba.dropout}")
        self.dropout = float(ba.dropout)
        self._packed: Dict[str, torch.Tensor] = {}
        self._packed_key = None
        self._plan_key = None
        self.ws: Optional[Workspace] = None
        self._rowmap = None
        # bumped whenever the workspace or the packed weights are re-allocated: anything that captured their addresses
        # (the sampler's hipGraph) is stale from then on
        self.generation = 0
        self._attn_aux = None
        # workspaces of the fused (5-pass) attention backward, one per (B, H, L, device) that has run (a ragged last batch or a second
        # sequence length alternates between two of them without re-allocating, re-zeroing or synchronising); `_attn_ws` = the one the
        # last backward used: its sticky error word is folded into every optimizer step on the device (attn_status_ptr)
        self._attn_ws_cache: Dict[tuple, "ops.FusedAttnBwdWorkspace"] = {}
        self._attn_ws = None
        # "attention in fp16" (BASELINE configs[4]; Lightning precision 16-mixed in the reference's trainer, model.yml:12): q, k, v reach the
        # attention core as IEEE half and its MFMAs are the f16 ones; everything around it stays bf16.  model.attn_dtype = torch.float16.
        self.attn_f16 = False

    # ------------------------------------------------------------------ OD_DETERMINISTIC (det.py)
    def _det_flush(self, *tensors):
        """Fold the integer shadows of these accumulation targets into them (no-op unless the mode is on): before their first reader."""
        ctx = det.context(self.model.arena.data.device)
        if ctx is not None:
            for t in tensors:
                ctx.flush(t)

    def _det_flush_segment(self, name: str):
        ctx = det.context(self.model.arena.data.device)
        if ctx is not None:
            s, e = self.model.arena.segments(self.depth)[name]
            ctx.flush(self.model.arena.ensure_grad()[s:e])

    # ------------------------------------------------------------------ parameters
    def P(self, name: str) -> torch.Tensor:
        return self.model.arena.view(name)

    def G(self, name: str) -> torch.Tensor:
        return self.model.arena.grad_view(name)

    def _gemm_weights(self) -> List[str]:
        names = ["proj_audio.0"]
        for i in range(self.depth):
            p = f"net.layers.{i}."
            names += [p + "proj_cl", p + "attn.qkv_proj", p + "attn.out_proj", p + "ffn.proj_vg.1", p + "ffn.proj_o"]
        return names

    def pack_weights(self, dtype: torch.dtype, train: bool, x3: bool = False):
        """fp32 masters -> compute-dtype GEMM operands (and their transposes for backward-data).  `x3` (fp32 tensors, three bf16
        MFMAs per product; no-grad only): the weights are split into their bf16 (hi, lo) halves HERE, once per call, instead of in every
        fragment load of every GEMM of the 51 evaluations (ops.SplitWeight, OD_F32X3W) — wherever K is a multiple of 32."""
        dev = self.model.arena.data.device
        import os
        x3 = bool(x3 and dtype == torch.float32 and not train and os.environ.get("OD_X3_SPLIT", "1") != "0")      # (env: A/B only)
        key = (dtype, train, dev, x3)
        if self._packed_key != key:
            self._packed = {}
            self._packed_key = key
            self.generation += 1
            Hf, Hp = self.Hf, self.Hp
            rm = [-1] * (2 * Hp)
            for j in range(Hf):
                rm[j] = j
                rm[Hp + j] = Hf + j
            self._rowmap = torch.tensor(rm, dtype=torch.int32, device=dev)
        pk = self._packed
        for n in self._gemm_weights():
            w = self.P(n + ".weight")
            N, K = w.shape[0], w.shape[1]
            Np, Kp, rmap = N, K, None
            if n.endswith("ffn.proj_vg.1"):
                Np, rmap = 2 * self.Hp, self._rowmap
            if n.endswith("ffn.proj_o"):
                Kp = self.Hp
            if n not in pk:
                pk[n] = torch.empty(Np, Kp, dtype=dtype, device=dev)
                if x3 and Kp % 32 == 0:
                    pk[n] = ops.SplitWeight(pk[n])
                if train:
                    pk[n + ".T"] = torch.empty(Kp, Np, dtype=dtype, device=dev)
                if rmap is not None:
                    pk[n + ".b"] = torch.empty(Np, dtype=torch.float32, device=dev)
            ops.pack_weight(w, pk[n], row_map=rmap)
            if train:
                ops.pack_weight(w, pk[n + ".T"], transpose=True, row_map=rmap)
            if rmap is not None:
                ops.pack_weight(self.P(n + ".bias"), pk[n + ".b"].view(Np, 1), row_map=rmap)

    def W(self, name: str, T: bool = False) -> torch.Tensor:
        return self._packed[name + (".T" if T else "")]

    def plain_gemm(self, A: torch.Tensor, W: torch.Tensor, bias, C: torch.Tensor):
        """C = A W^T (+ bias), no epilogue: od_gemm_nt (the 4-wave persistent kernel at training size)."""
        ops.gemm_nt(A, W, bias, C, x3=self.x3)

    # ------------------------------------------------------------------ plan / workspace
    def plan(self, B: int, L: int, Ba: int, dtype: torch.dtype, train: bool, x3: bool = False):
        """`x3`: run the fp32 forward's MFMA products as 3 bf16 MFMAs (OD_F32X3; no-grad paths only)."""
        dev = self.model.arena.data.device
        assert not (x3 and train), "the f32x3 product exists for the forward kernels only"
        x3 = bool(x3 and dtype == torch.float32)
        key = (B, L, Ba, dtype, train, dev, x3, getattr(self.model, "attn_dtype", None))
        if key != self._plan_key:
            self.ws = Workspace(dev)
            self._plan_key = key
            self.generation += 1
            tab = self.ws.get("rope", (L, self.hd // 2, 2), torch.float32)
            ops.rope_table(tab, L, self.hd)
        self.B, self.L, self.Ba, self.dtype, self.train, self.x3 = B, L, Ba, dtype, train, x3
        self.M, self.Ma = B * L, Ba * L
        self.attn_f16 = getattr(self.model, "attn_dtype", None) == torch.float16
        if self.attn_f16 and (dtype != torch.bfloat16 or self.hd != 64):
            # the half-operand core exists for bf16 compute and head_dim 64 only; anything else — an fp32 no-grad forward or sample() after a
            # `precision: 16-mixed` fit, a model with another head_dim — runs the attention of the compute dtype, said once
            if not getattr(self, "_warned_f16_fallback", False):
                import warnings
                warnings.warn(f"attn_dtype = float16 applies to compute_dtype bfloat16 with head_dim 64; this plan (dtype {dtype}, head_dim "
                              f"{self.hd}) runs its attention in {dtype}")
                self._warned_f16_fallback = True
            self.attn_f16 = False
        return self.ws

    def buf(self, name, shape, dtype=None):
        return self.ws.get(name, shape, self.dtype if dtype is None else dtype)

    def lbuf(self, name, i, shape, dtype=None):
        """Per-layer buffer when training (saved for backward), shared scratch otherwise."""
        return self.buf(f"{name}.{i}" if self.train else name, shape, dtype)

    # ------------------------------------------------------------------ conditioning (model.py:73-84)
    def conditioning(self, audio: torch.Tensor, style: torch.Tensor):
        """proj_audio, proj_style and everything that depends only on (audio, style):
        ssg1/ssg2 of every layer, u_mod, and (inference) every layer's proj_cl(a) — the
        loop invariants the reference recomputes on each of the sampler's evaluations."""
        B, Ba, L = self.B, self.Ba, self.L
        D, A, Cg, U = self.D, self.A, self.Cg, self.U
        f32 = torch.float32
        a_t = self.buf("a_t", (self.Ma, A))
        a_pre = self.buf("a_pre", (self.Ma, A))
        a = self.buf("a", (self.Ma, A))
        ops.cl_to_frames(audio, a_t)
        ops.gemm_nt(a_t, self.W("proj_audio.0"), self.P("proj_audio.0.bias"), a_pre, x3=self.x3)
        ops.silu(a_pre, a)
        cg, cg_pre = self.buf("cg", (B, Cg), f32), self.buf("cg_pre", (B, Cg), f32)
        ops.linear_small(style, self.P("proj_style.0.weight"), self.P("proj_style.0.bias"), cg, cg_pre, OD_ACT_SILU)
        for i in range(self.depth):
            p = f"net.layers.{i}."
            for s in ("ssg1", "ssg2"):
                ops.linear_small(cg, self.P(p + s + ".weight"), self.P(p + s + ".bias"),
                                 self.buf(f"{s}.{i}", (B, 3 * D), f32))
            if not self.train:
                ops.gemm_nt(a, self.W(p + "proj_cl"), self.P(p + "proj_cl.bias"), self.buf(f"cl.{i}", (self.Ma, D)), x3=self.x3)
        ops.linear_small(cg, self.P("u_mod.weight"), self.P("u_mod.bias"), self.buf("umod", (B, 2 * U), f32))

    # ------------------------------------------------------------------ forward (model.py:86-103)
    def pred(self, xt: torch.Tensor, u: torch.Tensor, v: torch.Tensor):
        B, L, M, D, dh, Hp = self.B, self.L, self.M, self.D, self.dh, self.Hp
        f32 = torch.float32
        tab = self.ws.t["rope"]
        bcast = self.Ba == 1 and B > 1
        x = self.lbuf("x_in", 0, (M, D))
        ops.proj_in(xt, self.P("proj_in.weight"), self.P("proj_in.bias"), x)

        def cl_of(i):
            if not self.train:
                return self.ws.t[f"cl.{i}"]
            cl = self.buf("cl", (self.Ma, D))                      # consumed by the FiLM right after: one buffer
            ops.gemm_nt(self.ws.t["a"], self.W(f"net.layers.{i}.proj_cl"), self.P(f"net.layers.{i}.proj_cl.bias"), cl)
            return cl

        h1 = self.lbuf("h1", 0, (M, D))
        ops.rmsnorm_film(x, self.ws.t["ssg1.0"], cl_of(0), bcast, h1, self.lbuf("inv1", 0, (M,), f32), B, L)
        for i in range(self.depth):
            p = f"net.layers.{i}."
            ssg1, ssg2 = self.ws.t[f"ssg1.{i}"], self.ws.t[f"ssg2.{i}"]
            # --- attention branch (backbone.py:76-80, attn.py:74-84); h1 = norm + FiLM + proj_cl(a) came from the
            #     kernel that closed the previous layer
            qkv = self.lbuf("qkv", i, (M, 3 * dh))
            if self.train or self.attn_f16 or self.hd not in (32, 64) or (2 * dh) % 128:
                # backward needs the pre-norm q, k as well.  attn_f16: qk and the v columns of qkv are written as IEEE half
                qk = self.lbuf("qk16" if self.attn_f16 else "qk", i, (M, 2 * dh), torch.float16 if self.attn_f16 else None)
                ops.gemm_nt_qkrope_split(h1, self.W(p + "attn.qkv_proj"), self.P(p + "attn.qkv_proj.bias"), qkv, qk,
                                         self.P(p + "attn.q_norm.weight"), self.P(p + "attn.k_norm.weight"), tab, L, self.H,
                                         self.hd, FP32_EPS, x3=self.x3, q_scale=self.q_scale)
            else:                                             # no-grad: norm + RoPE in the GEMM's epilogue, in place
                ops.gemm_nt_qkrope(h1, self.W(p + "attn.qkv_proj"), self.P(p + "attn.qkv_proj.bias"), qkv,
                                   self.P(p + "attn.q_norm.weight"), self.P(p + "attn.k_norm.weight"), tab, L,
                                   self.H, self.hd, FP32_EPS, x3=self.x3, q_scale=self.q_scale)
                qk = qkv
            y = self.lbuf("y", i, (M, dh))
            lse = self.lbuf("lse", i, (B, self.H, L), f32)
            ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], self._v_of(qkv), y, lse, B, self.H, L, self.hd,
                               1.0 / math.sqrt(self.hd), x3=self.x3, q_prescaled=True)
            ao = self.lbuf("ao", i, (M, D))
            self.plain_gemm(y, self.W(p + "attn.out_proj"), self.P(p + "attn.out_proj.bias"), ao)
            # --- gate + residual of the attention branch and norm + FiLM of the feed-forward branch, one pass
            x_mid = self.lbuf("x_mid", i, (M, D))
            h2 = self.lbuf("h2", i, (M, D))
            # (OD_FUSE_FILM_DWCONV=1: those two AND the SwiGLU branch's depthwise conv as one kernel — bit-identical, 0.3 ms per step at the
            #  bench shape: profiles/r06q_ab_film_dwconv.txt; large training shapes only: a wave walks 32 frames)
            fuse_dw = (self.radius > 0 and D <= 512 and x.data_ptr() != x_mid.data_ptr() and x.dtype in (torch.bfloat16, torch.float32)
                       and os.environ.get("OD_FUSE_FILM_DWCONV", OD_FUSE_FILM_DWCONV_DEFAULT) == "1")
            if fuse_dw:
                hdw = self.lbuf("hdw", i, (M, D))
                ops.rmsnorm_gate_residual_film_dwconv(x, ao, ssg1, x_mid, self.lbuf("inv2", i, (M,), f32), ssg2, h2 if self.train else None,
                                                      self.lbuf("inv3", i, (M,), f32),
                                                      self.P(p + "ffn.proj_vg.0.weight"), self.P(p + "ffn.proj_vg.0.bias"), hdw, B, L, self.ksize)
            else:
                ops.rmsnorm_gate_residual_film(x, ao, ssg1, x_mid, self.lbuf("inv2", i, (M,), f32), ssg2, None, False, h2,
                                               self.lbuf("inv3", i, (M,), f32), B, L)
            # --- feed-forward branch (backbone.py:82-86, swiglu.py:27-32)
            if fuse_dw:
                pass
            elif self.radius > 0:
                hdw = self.lbuf("hdw", i, (M, D))
                ops.dwconv(h2, self.P(p + "ffn.proj_vg.0.weight"), self.P(p + "ffn.proj_vg.0.bias"), hdw, B, L, self.ksize)
            else:
                hdw = h2                                      # radius 0: nn.Identity (swiglu.py:20)
            vg = self.lbuf("vg", i, (M, 2 * Hp))
            ops.gemm_nt(hdw, self.W(p + "ffn.proj_vg.1"), self._packed[p + "ffn.proj_vg.1.b"], vg, x3=self.x3)
            hh = self.lbuf("hh", i, (M, Hp))
            ops.swiglu_rmsnorm(vg, hh, self.lbuf("inv4", i, (M,), f32), self.Hf, Hp)
            if self._drop_active():                           # nn.Dropout1d, training mode only (swiglu.py:23,30)
                ops.scale_channels(hh, self.draw_dropout_mask(i), B, L)
            fo = self.lbuf("fo", i, (M, D))
            self.plain_gemm(hh, self.W(p + "ffn.proj_o"), self.P(p + "ffn.proj_o.bias"), fo)
            # next layer input (at inference x_in.0 / x_in.1 ping-pong), together with the next layer's h1
            x = self.buf(f"x_in.{i + 1}", (M, D)) if self.train else self.buf(f"x_in.{(i + 1) & 1}", (M, D))
            if i + 1 < self.depth:
                h1 = self.lbuf("h1", i + 1, (M, D))
                ops.rmsnorm_gate_residual_film(x_mid, fo, ssg2, x, self.lbuf("inv5", i, (M,), f32),
                                               self.ws.t[f"ssg1.{i + 1}"], cl_of(i + 1), bcast, h1,
                                               self.lbuf("inv1", i + 1, (M,), f32), B, L)
            else:
                ops.rmsnorm_gate_residual(x_mid, fo, ssg2, x, self.lbuf("inv5", i, (M,), f32), B, L)
        self._x_last = x
        ops.final_norm_proj_out(x, self.P("proj_out.weight"), self.P("proj_out.bias"), v,
                                self.buf("invf", (M,), f32), B, L)
        # distance head on the raw latent (model.py:99-102)
        fsum = self.buf("fsum", (B, self.U), f32)
        fsum.zero_()
        ops.uhead_fwd(xt, self._uhead_w(), fsum, self.U)
        self._det_flush(fsum)
        ops.uhead_tail(fsum, self.ws.t["umod"], self.P("u_out.weight"), self.P("u_out.bias"), u, L, self.model.u_scale)

    def _v_of(self, qkv: torch.Tensor) -> torch.Tensor:
        """The v columns of a layer's qkv buffer as the attention core reads them (attn_f16: the same bytes hold IEEE half)."""
        v = qkv[:, 2 * self.dh:]
        return v.view(torch.float16) if self.attn_f16 else v

    def attn_bwd_passes(self) -> int:
        """L x L x hd MFMA passes the attention backward of this plan executes (5 = the algorithmic count: one fused kernel; 7 = the
        dK/dV + dQ kernel pair, which recomputes S and dP)."""
        from . import _lib
        return int(_lib.lib().cdll.od_flash_attn_bwd_fused_passes() if self.fused_attn_bwd() else _lib.lib().cdll.od_flash_attn_bwd_passes())

    def fused_attn_bwd(self) -> bool:
        """Whether the attention backward runs as od_flash_attn_bwd_fused: one kernel, the 5 algorithmic MFMA passes, dQ summed over key blocks by
        the chain through the XCD's L2 (bf16, head_dim 64).  Default for L >= 2048, measured per call at batch 32 (profiles/r04ak_crossover.txt):
        L = 2048: 1.64-1.70 against 1.71-1.88 ms; 4096: 5.7 against 6.3-6.4; 8192: 21.0-21.4 against 23.3-24.1; batch 8 x 32768: 81.0 against 94.0.
        (Shorter sequences have not been measured: they run the chain in lock step — a key block trails its predecessor by ~1.5 query tiles and
        consecutive key blocks start (L / 64) / (CUs per XCD) tiles apart.)
        OD_ATTN_BWD_FUSED=0 / 1 forces the two-kernel / the fused path."""
        import os
        if self.dtype != torch.bfloat16 or self.hd != 64:
            return False
        if self.attn_f16:
            return True                     # the half-operand backward exists as the fused kernel only
        env = os.environ.get("OD_ATTN_BWD_FUSED", "")
        if env in ("0", "1"):
            return env == "1"
        return self.L >= 2048

    def attn_bwd_launch(self, i: int, dy: torch.Tensor, delta: torch.Tensor, dqkv: torch.Tensor, core_only: bool = False):
        """The backward of layer i's attention core + q / k norm + RoPE exactly as `backward` launches it: dy = gradient of the attention
        output, dqkv <- gradient of the qkv projection; norm weight gradients accumulate.  `core_only` (bench.py's roofline leg): just
        the attention backward as the step launches it, without the separate norm + RoPE backward pass that follows it."""
        t, dh, p = self.ws.t, self.dh, f"net.layers.{i}."
        qk, qkv, y, lse = t[f"qk16.{i}" if self.attn_f16 else f"qk.{i}"], t[f"qkv.{i}"], t[f"y.{i}"], t[f"lse.{i}"]
        wq, wk = self.P(p + "attn.q_norm.weight"), self.P(p + "attn.k_norm.weight")
        gq, gk = self.G(p + "attn.q_norm.weight"), self.G(p + "attn.k_norm.weight")
        scale = 1.0 / math.sqrt(self.hd)
        dqk = self.buf("d.qk", (self.M, 2 * dh))
        if self.fused_attn_bwd():
            key = (self.B, self.H, self.L, qk.device, qk.dtype)
            ws = self._attn_ws_cache.get(key)
            if ws is None:
                if len(self._attn_ws_cache) >= 4:              # bounded: the oldest shape goes
                    self._attn_ws_cache.pop(next(iter(self._attn_ws_cache)))
                ws = self._attn_ws_cache[key] = ops.FusedAttnBwdWorkspace(self.B, self.H, self.L, qk.device, qk.dtype)
                ws.checked = False
            self._attn_ws = ws
            ops.flash_attn_bwd_fused(qk[:, :dh], qk[:, dh:], self._v_of(qkv), y, dy, lse, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:],
                                     self.B, self.H, self.L, self.hd, scale, self._attn_ws, q_prescaled=True)
        else:
            if (self._attn_aux is None or self._attn_aux.device != qk.device) and qk.is_cuda:
                if self._attn_aux is not None:
                    self._attn_aux.close()
                with torch.cuda.device(qk.device):          # the side stream and its events belong to the tensors' device
                    self._attn_aux = ops.AttnAux()          # side stream + events: the dQ kernel runs beside the dK/dV kernel
                self._attn_aux.device = qk.device
            ops.flash_attn_bwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], y, dy, lse, delta, dqk[:, :dh], dqk[:, dh:],
                               dqkv[:, 2 * dh:], self.B, self.H, self.L, self.hd, scale, q_prescaled=True, aux=self._attn_aux)
        if not core_only:
            ops.qk_norm_rope_bwd(qkv, wq, wk, t["rope"], dqk, dqkv, gq, gk, self.B, self.L, self.H, self.hd, FP32_EPS, q_scale=self.q_scale)

    # ---- nn.Dropout1d(p) of the SwiGLU hidden state: a (B, Hp) factor per layer, 0 or 1 / (1 - p), drawn by torch's generator
    #      (as the reference's mask is) when the module is in training mode; `hh` is stored masked, so proj_o's weight gradient
    #      sees the dropped activations and the data gradient is masked again on its way back
    def _drop_active(self) -> bool:
        return self.dropout > 0.0 and self.model.training

    def _drop_mask(self, i: int) -> torch.Tensor:
        return self.ws.t[f"drop.{i}"]

    def draw_dropout_mask(self, i: int) -> torch.Tensor:
        """Layer i's mask, drawn where the reference draws it (inside the layer's SwiGLU, swiglu.py:30) and the way nn.Dropout1d does:
        one Bernoulli(keep) per (batch, channel) over the Hf real channels, scaled by 1/keep; zero-extended to the padded width.
        (The draw consumes the device generator in the reference's order; bit-equal masks across devices are not a goal.)"""
        keep = 1.0 - self.dropout
        m = self.buf(f"drop.{i}", (self.B, self.Hp), torch.float32)
        m.zero_()
        m[:, :self.Hf] = torch.empty(self.B, self.Hf, device=m.device, dtype=torch.float32).bernoulli_(keep) / keep
        return m

    def _uhead_w(self, grads=False):
        f = self.G if grads else self.P
        return [f(f"u_head.{j}.{s}") for j in (0, 1, 3, 4) for s in ("weight", "bias")]

    # ------------------------------------------------------------------ backward
    def backward(self, xt: torch.Tensor, style: torch.Tensor, du: torch.Tensor, dv: torch.Tensor, reducer=None):
        """Accumulate dLoss/dparam into the arena's grad buffer given dLoss/du [B], dLoss/dv (B,E,L).
        Mirrors `pred` in reverse.  `reducer.layer_done(i)` (if given) is told when a layer's
        gradients are final so their all-reduce can overlap the rest of backward."""
        assert self.train
        B, L, M, D, dh, Hf, Hp, A = self.B, self.L, self.M, self.D, self.dh, self.Hf, self.Hp, self.A
        f32 = torch.float32
        dctx = det.context(self.model.arena.data.device)
        if dctx is not None:
            dctx.register(self.model.arena.ensure_grad())
        tab = self.ws.t["rope"]
        cg, a = self.ws.t["cg"], self.ws.t["a"]
        # residual-stream gradient and scratch
        dx = self.buf("d.x", (M, D))
        dbr = self.buf("d.branch", (M, D))
        dtmp = self.buf("d.tmp", (M, D))
        dhh = self.buf("d.hh", (M, Hp))
        dvg = self.buf("d.vg", (M, 2 * Hp))
        dy = self.buf("d.y", (M, dh))
        dqkv = self.buf("d.qkv", (M, 3 * dh))
        da = self.buf("d.a", (self.Ma, A))
        delta = self.buf("d.delta", (B, self.H, L), f32)
        dcg = self.buf("d.cg", (B, self.Cg), f32)
        dssg1, dssg2 = self.buf("d.ssg1", (B, 3 * D), f32), self.buf("d.ssg2", (B, 3 * D), f32)

        ops.final_norm_proj_out_bwd(self._x_last, self.ws.t["invf"], self.P("proj_out.weight"), dv, dx,
                                    self.G("proj_out.weight"), self.G("proj_out.bias"), B, L)
        # distance head
        dfm, dmod = self.buf("d.fm", (B, self.U), f32), self.buf("d.mod", (B, 2 * self.U), f32)
        ops.uhead_tail_bwd(self.ws.t["fsum"], self.ws.t["umod"], self.P("u_out.weight"), self.P("u_out.bias"), du,
                           dfm, dmod, self.G("u_out.weight"), self.G("u_out.bias"), L, self.model.u_scale)
        ops.uhead_bwd(xt, self._uhead_w(), dfm, self._uhead_w(grads=True), self.U)
        ops.linear_small_bwd(cg, self.P("u_mod.weight"), None, dmod, self.buf("d.lin_pre_umod", (B, 2 * self.U), f32),
                             self.G("u_mod.weight"), self.G("u_mod.bias"), dcg, False, OD_ACT_NONE)
        self._det_flush_segment("tail")
        if reducer is not None:
            reducer.segment_done("tail")

        for i in reversed(range(self.depth)):
            p = f"net.layers.{i}."
            t = self.ws.t
            ssg1, ssg2 = t[f"ssg1.{i}"], t[f"ssg2.{i}"]
            dssg1.zero_()
            dssg2.zero_()
            # ---- feed-forward branch
            ops.rmsnorm_gate_residual_bwd(t[f"fo.{i}"], t[f"inv5.{i}"], ssg2, dx, dbr, dssg2, B, L)
            ops.gemm_tn(dbr, t[f"hh.{i}"], self.G(p + "ffn.proj_o.weight"), n_cols=D, k_cols=Hf,
                        dbias=self.G(p + "ffn.proj_o.bias"))
            ops.gemm_nt(dbr, self.W(p + "ffn.proj_o", T=True), None, dhh)
            if self._drop_active():
                ops.scale_channels(dhh, self._drop_mask(i), B, L)
            ops.swiglu_rmsnorm_bwd(t[f"vg.{i}"], t[f"inv4.{i}"], dhh, dvg, Hf, Hp)
            gw, gb = self.G(p + "ffn.proj_vg.1.weight"), self.G(p + "ffn.proj_vg.1.bias")
            hdw = t[f"hdw.{i}"] if self.radius > 0 else t[f"h2.{i}"]
            # one launch over the padded width (v columns [0, Hf), g columns [Hp, Hp + Hf)): 11 column tiles instead of 2 x 6
            ops.gemm_tn(dvg, hdw, gw, n_cols=2 * Hp, k_cols=D, dbias=gb, n_block=Hp, n_valid=Hf)
            if self.radius > 0:
                self.plain_gemm(dvg, self.W(p + "ffn.proj_vg.1", T=True), None, dtmp)
                ops.dwconv_bwd(t[f"h2.{i}"], self.P(p + "ffn.proj_vg.0.weight"), dtmp, dbr,
                               self.G(p + "ffn.proj_vg.0.weight"), self.G(p + "ffn.proj_vg.0.bias"), B, L, self.ksize)
            else:
                self.plain_gemm(dvg, self.W(p + "ffn.proj_vg.1", T=True), None, dbr)
            ops.rmsnorm_film_bwd(t[f"x_mid.{i}"], t[f"inv3.{i}"], ssg2, dbr, dx, dssg2, B, L)
            # ---- attention branch
            ops.rmsnorm_gate_residual_bwd(t[f"ao.{i}"], t[f"inv2.{i}"], ssg1, dx, dbr, dssg1, B, L)
            ops.gemm_tn(dbr, t[f"y.{i}"], self.G(p + "attn.out_proj.weight"), dbias=self.G(p + "attn.out_proj.bias"))
            ops.gemm_nt(dbr, self.W(p + "attn.out_proj", T=True), None, dy)
            self.attn_bwd_launch(i, dy, delta, dqkv)
            ops.gemm_tn(dqkv, t[f"h1.{i}"], self.G(p + "attn.qkv_proj.weight"), dbias=self.G(p + "attn.qkv_proj.bias"))
            self.plain_gemm(dqkv, self.W(p + "attn.qkv_proj", T=True), None, dtmp)      # d h1 (== d cl)
            ops.gemm_tn(dtmp, a, self.G(p + "proj_cl.weight"), dbias=self.G(p + "proj_cl.bias"))
            ops.gemm_nt(dtmp, self.W(p + "proj_cl", T=True), None, da, accumulate=(i != self.depth - 1))
            ops.rmsnorm_film_bwd(t[f"x_in.{i}"], t[f"inv1.{i}"], ssg1, dtmp, dx, dssg1, B, L)
            # ---- the two modulation linears of this layer
            lp = self.buf("d.lin_pre_ssg", (B, 3 * D), f32)
            self._det_flush(dssg1, dssg2)
            ops.linear_small_bwd(cg, self.P(p + "ssg2.weight"), None, dssg2, lp, self.G(p + "ssg2.weight"),
                                 self.G(p + "ssg2.bias"), dcg, True, OD_ACT_NONE)
            ops.linear_small_bwd(cg, self.P(p + "ssg1.weight"), None, dssg1, lp, self.G(p + "ssg1.weight"),
                                 self.G(p + "ssg1.bias"), dcg, True, OD_ACT_NONE)
            self._det_flush_segment(f"layer{i}")
            if reducer is not None:
                reducer.segment_done(f"layer{i}")

        ops.proj_in_bwd(xt, dx, self.G("proj_in.weight"), self.G("proj_in.bias"))
        da_pre = self.buf("d.a_pre", (self.Ma, A))
        ops.silu_bwd(self.ws.t["a_pre"], da, da_pre)
        ops.gemm_tn(da_pre, self.ws.t["a_t"], self.G("proj_audio.0.weight"), dbias=self.G("proj_audio.0.bias"))
        self._det_flush(dcg)
        ops.linear_small_bwd(style, self.P("proj_style.0.weight"), self.ws.t["cg_pre"], dcg,
                             self.buf("d.lin_pre_style", (B, self.Cg), f32), self.G("proj_style.0.weight"),
                             self.G("proj_style.0.bias"), None, False, OD_ACT_SILU)
        self._det_flush_segment("head")
        if reducer is not None:
            reducer.segment_done("head")
        # The fused attention backward draws its jobs from one queue per XCD and needs every XCD to run at least one of its workgroups (observed
        # on every launch so far: block b runs on XCD b % 8), and its chain waits are bounded.  Its workspace keeps a sticky error word for the
        # launches where either went wrong.  EVERY step checks it on the device: FusedAdamWEMA hands attn_status_ptr() to od_sqnorm /
        # od_adamw_ema (norm -> NaN, update skipped) and the trainer raises at its next logging point (check_attn_status).  The first backward
        # on a workspace is also checked here, on the host (one stream synchronisation per shape): fail before the first optimizer step.
        if self.fused_attn_bwd() and self._attn_ws is not None and not self._attn_ws.checked:
            self._attn_ws.checked = True
            self.check_attn_status()

    def attn_status_ptr(self) -> int:
        """Device address of the sticky error word of the fused attention backward the last `backward` used (0: none ran)."""
        return self._attn_ws.err_ptr() if self._attn_ws is not None else 0

    def attn_status_view(self) -> torch.Tensor:
        """The same word as an int32[1] tensor (data-parallel steps reduce it over the ranks)."""
        return self._attn_ws.err_view()

    ATTN_STATUS = {1: "a launch left jobs unprocessed (an XCD ran none of its workgroups): gradients incomplete",
                   2: "a launch with another (B, H, L) than the workspace's earlier launches was refused",
                   3: "a chain wait ran out of time (the predecessor's write never came; OD_FB_CHAIN_TIMEOUT_MS): dq is NaN"}

    def check_attn_status(self):
        """Host-side read of that word (waits for the current stream).  Raises on anything but 0; the optimizer steps since the failing
        launch were skipped on the device."""
        if self._attn_ws is None:
            return
        # every cached shape's workspace (a sticky error on the shape a ragged last batch uses must not wait for that shape's turn: ADVICE r5)
        code = 0
        for ws in {id(w): w for w in [self._attn_ws, *self._attn_ws_cache.values()]}.values():
            code = code or ws.status()
        if code != 0:
            raise RuntimeError(f"od_flash_attn_bwd_fused status {code}: {self.ATTN_STATUS.get(code, '?')}; the optimizer steps since then "
                               "were skipped.  OD_ATTN_BWD_FUSED=0 selects the two-kernel backward.")
