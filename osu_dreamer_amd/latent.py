"""`LatentModel` inference on the HIP path — the steps either side of `diffusion.sample` in `LDM.sample`
(osu_dreamer/models/inference/model.py:48,51; reference model: osu_dreamer/models/latent/model.py:38-134,
unet.py:21-126, spec_features.py:10-32).

Same constructor, same `state_dict()` keys and shapes as the reference's `LatentModel` (all of them, so a
reference checkpoint / inference artifact loads with `strict=True`), and the calls `LDM.sample` makes:

    skips, h = model.audio_encoder(audio)           # (Ba,72,L) -> [ (Ba,h_dim,L/stride^i) ], (Ba,h_dim,L/stride^n)
    chart, labels = model.decode(z, s, skips=skips) # or audio=...;   also decode_logits(z, s, ...)
    z, s = model.encode_chart(chart)                # (B,9,L) -> (B,emb_dim,L/chunk), (B,style_dim)

Activations are frame-major [B*L][h_dim] on the device; the tensors handed back are (B, C, L)-shaped *views* of
those buffers (no transposing copy), and `decode` takes them back without one.  `encode_chart(chart) -> (z, s)`
(chart encoder, style head, temporal head: the dataset-encoding direction of scripts/encode_latents.py) runs on
the same kernels.  Training of the latent model is out of scope: parameters do not receive gradients here.
"""
from __future__ import annotations

import math
import weakref
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import ops
from ._lib import OD_ACT_NONE, OD_ACT_SILU
from .engine import Workspace

A_DIM = 72           # data/load_audio.py:11-15
X_DIM = 9            # data/beatmap/encode.py:14-29
N_HIT = 7            # HitSignals are channels 0..6 (encode.py:31-39)
NUM_LABELS = 5       # encode.py:50


@dataclass
class LayerArgs:                      # unet.py:9-13
    n_layers: int
    expand: int
    radius: int


@dataclass
class LatentModelArgs:                # latent/model.py:15-21
    h_dim: int
    ae_args: LayerArgs
    style_head_dim: int
    style_heads: int


def _ceil(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class _Node(nn.Module):
    pass


class _AudioEncoder(_Node):
    """`model.audio_encoder(audio)` — holds the reference's `audio_encoder.{0,1}.*` parameters."""

    def forward(self, audio: torch.Tensor):
        return self._owner()._audio_encoder(audio)


def _shapes(emb_dim: int, style_dim: int, n_downs: int, stride: int, a: LatentModelArgs):
    """Every key of the reference's `LatentModel.state_dict()` -> (shape, init) in its order; init is
    'w' (default conv/linear init), 'b:<fan_in>' (bias), 'g:<gain>' (RMSNorm gamma) or 'z' (zero-initialised)."""
    D, L_ = a.h_dim, a.ae_args
    hf, k, ks = int(D * L_.expand * 2 / 3), 1 + 2 * L_.radius, 1 + 2 * (stride // 2)
    s: Dict[str, Tuple[tuple, str]] = {}

    def conv(name, shape, zero=False):
        fan = 1
        for n in shape[1:]:
            fan *= n
        s[name + ".weight"] = (shape, "z" if zero else "w")
        s[name + ".bias"] = ((shape[0],), "z" if zero else f"b:{fan}")

    def layer(p, cond_dim):
        for i in range(L_.n_layers):
            s[f"{p}norms.{i}.gamma"] = ((D,), "g:1.0")
        for i in range(L_.n_layers):
            b = f"{p}blocks.{i}."
            conv(b + "0.proj_vg.0", (D, 1, k))
            conv(b + "0.proj_vg.1", (2 * hf, D, 1))
            conv(b + "0.proj_o", (D, hf, 1))
            s[b + "1.gamma"] = ((D,), "g:0.001")
        s[p + "out_norm.gamma"] = ((D,), "g:1.0")
        if cond_dim > 0:
            for i in range(L_.n_layers):
                conv(f"{p}films.{i}", (3 * D, cond_dim), zero=True)

    def encoder(p):
        for i in range(n_downs):
            conv(f"{p}downs.{i}.0", (D, 1, ks))
        for i in range(n_downs):
            layer(f"{p}layers.{i}.", 0)

    conv("chart_encoder.0", (D, X_DIM, 1))
    encoder("chart_encoder.1.")
    conv("audio_encoder.0.net.1", (8, 1, 8, 3)); s["audio_encoder.0.net.2.gamma"] = ((8,), "g:1.0")
    conv("audio_encoder.0.net.4", (32, 8, 6, 3)); s["audio_encoder.0.net.5.gamma"] = ((32,), "g:1.0")
    conv("audio_encoder.0.net.8", (D, 32 * (A_DIM // 24), 1)); s["audio_encoder.0.net.9.gamma"] = ((D,), "g:1.0")
    encoder("audio_encoder.1.")
    layer("style_head.0.", 0)
    hd = a.style_head_dim * a.style_heads
    conv("style_head.1.scores", (a.style_heads, D, 1))
    conv("style_head.1.values", (hd, D, 1))
    conv("style_head.1.proj_out", (style_dim, hd))
    layer("temporal_layer.", style_dim)
    conv("temporal_head.0", (emb_dim, D, 1))
    conv("proj_emb", (D, emb_dim, 1))
    for i in range(n_downs):
        conv(f"decoder.ups.{i}.1", (D, 1, ks))
    for i in range(n_downs):
        layer(f"decoder.layers.{i}.", style_dim)
    for i in range(n_downs):
        m = f"decoder.mixers.{i}."
        conv(m + "proj.0", (D, D, 1)); s[m + "proj.1.gamma"] = ((D,), "g:1.0")
        conv(m + "gate", (D, D, 1), zero=True)
    conv("proj_out", (X_DIM, D, 1))
    conv("label_predictor.0", (D, style_dim))
    conv("label_predictor.2", (NUM_LABELS, D))
    return s


class LatentModel(nn.Module):
    def __init__(self, emb_dim: int, style_dim: int, n_downs: int, stride: int, args: LatentModelArgs):
        super().__init__()
        if isinstance(args, dict):
            args = LatentModelArgs(**args)
        if isinstance(args.ae_args, dict):
            args.ae_args = LayerArgs(**args.ae_args)
        self.emb_dim, self.style_dim, self.n_downs, self.stride, self.args = emb_dim, style_dim, n_downs, stride, args
        self.a_dim = args.h_dim
        self.chunk_size = stride ** n_downs
        D = args.h_dim
        if D < 8 or D > 512 or D & (D - 1):
            raise NotImplementedError("h_dim must be a power of two in 8..512 on the compiled path")
        if args.ae_args.radius not in (1, 2):
            raise NotImplementedError("depthwise kernel sizes 3 and 5 (radius 1, 2) are compiled")
        self.hf = int(D * args.ae_args.expand * 2 / 3)
        self.hp = _ceil(self.hf, 64)
        self.compute_dtype: Optional[torch.dtype] = None     # None: fp32 (bf16 when set)
        self.f32_matmul = "f32"                              # or "bf16x3", see DiffusionModel.f32_matmul
        for name, (shape, init) in _shapes(emb_dim, style_dim, n_downs, stride, args).items():
            t = torch.empty(*shape)
            if init == "w":
                fan = 1
                for n in shape[1:]:
                    fan *= n
                t.uniform_(-1 / math.sqrt(fan), 1 / math.sqrt(fan))
            elif init.startswith("b:"):
                bound = 1 / math.sqrt(int(init[2:]))
                t.uniform_(-bound, bound)
            elif init.startswith("g:"):
                t.fill_(float(init[2:]))
            else:
                t.zero_()
            self._register(name, nn.Parameter(t, requires_grad=False))
        object.__setattr__(self.audio_encoder, "_owner", weakref.ref(self))
        self._ws: Dict[tuple, Workspace] = {}
        self._packed: Dict[str, torch.Tensor] = {}
        self._packed_key = None
        self._rowmap = None
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate())

    def _register(self, dotted: str, p: nn.Parameter):
        node, parts = self, dotted.split(".")
        for i, part in enumerate(parts[:-1]):
            nxt = node._modules.get(part)
            if nxt is None:
                nxt = _AudioEncoder() if (i == 0 and part == "audio_encoder") else _Node()
                node.add_module(part, nxt)
            node = nxt
        node.register_parameter(parts[-1], p)

    # ------------------------------------------------------------------ plumbing
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.invalidate()
        return out

    def invalidate(self):
        """Forget packed GEMM operands and workspaces (call after editing parameters in place)."""
        self._packed, self._packed_key, self._ws = {}, None, {}

    def P(self, name: str) -> torch.Tensor:
        return self.get_parameter(name)

    def _dtype(self) -> torch.dtype:
        return self.compute_dtype or torch.float32

    def _x3(self) -> bool:
        if self.f32_matmul not in ("f32", "bf16x3"):
            raise ValueError(f"f32_matmul must be 'f32' or 'bf16x3', got {self.f32_matmul!r}")
        return self.f32_matmul == "bf16x3"

    def _layer_prefixes(self) -> List[Tuple[str, bool]]:
        out = [(f"audio_encoder.1.layers.{i}.", False) for i in range(self.n_downs)]
        out += [(f"decoder.layers.{i}.", True) for i in range(self.n_downs)]
        out += [(f"chart_encoder.1.layers.{i}.", False) for i in range(self.n_downs)]
        return out + [("style_head.0.", False), ("temporal_layer.", True)]

    def _pack(self, dt: torch.dtype):
        """fp32 parameters -> GEMM operands of the compute dtype; SwiGLU width padded hf -> hp (zero rows/cols)."""
        dev = self.P("proj_out.weight").device
        key = (dt, dev)
        if self._packed_key == key:
            return
        pk: Dict[str, torch.Tensor] = {}
        hf, hp = self.hf, self.hp
        rm = [-1] * (2 * hp)
        for j in range(hf):
            rm[j], rm[hp + j] = j, hf + j
        rowmap = torch.tensor(rm, dtype=torch.int32, device=dev)

        def pack(name, Np=None, Kp=None, rmap=None):
            w = self.P(name + ".weight")
            N, K = w.shape[0], w.numel() // w.shape[0]
            pk[name] = torch.empty(Np or N, Kp or K, dtype=dt, device=dev)
            ops.pack_weight(w, pk[name], row_map=rmap)
            if rmap is not None:
                pk[name + ".b"] = torch.empty(Np, dtype=torch.float32, device=dev)
                ops.pack_weight(self.P(name + ".bias"), pk[name + ".b"].view(Np, 1), row_map=rmap)

        pack("audio_encoder.0.net.8")
        pack("chart_encoder.0", Kp=16)                      # 9 chart channels, zero-padded to one 16-wide k chunk
        pack("style_head.1.scores")
        pack("style_head.1.values")
        for p, _ in self._layer_prefixes():
            for i in range(self.args.ae_args.n_layers):
                pack(f"{p}blocks.{i}.0.proj_vg.1", Np=2 * hp, rmap=rowmap)
                pack(f"{p}blocks.{i}.0.proj_o", Kp=hp)
        for i in range(self.n_downs):
            pack(f"decoder.mixers.{i}.proj.0")
            pack(f"decoder.mixers.{i}.gate")
        self._packed, self._packed_key = pk, key

    def _workspace(self, tag: str, B: int, L: int, dt) -> Workspace:
        dev = self.P("proj_out.weight").device
        key = (tag, B, L, dt, dev)
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) >= 8:
                self._ws.clear()
            ws = self._ws[key] = Workspace(dev)
        return ws

    # ------------------------------------------------------------------ unet.py:21-53 (layer)
    def _layer(self, ws: Workspace, tag: str, p: str, x: torch.Tensor, cond: Optional[torch.Tensor], B: int, L: int,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [B*L][D] is updated in place by the blocks; returns out_norm(x) in `out` (or a workspace buffer)."""
        M, D, dt, x3 = B * L, self.a_dim, x.dtype, self._x3()
        h = ws.get(tag + ".h", (M, D), dt)
        hd = ws.get(tag + ".hd", (M, D), dt)
        vg = ws.get(tag + ".vg", (M, 2 * self.hp), dt)
        hh = ws.get(tag + ".hh", (M, self.hp), dt)
        fo = ws.get(tag + ".fo", (M, D), dt)
        inv = ws.get(tag + ".inv", (M,), torch.float32)
        k = 1 + 2 * self.args.ae_args.radius
        for i in range(self.args.ae_args.n_layers):
            ssg = None
            if cond is not None:
                ssg = ws.get(f"{tag}.ssg{i}", (B, 3 * D), torch.float32)
                ops.linear_small(cond, self.P(f"{p}films.{i}.weight"), self.P(f"{p}films.{i}.bias"), ssg)
            b = f"{p}blocks.{i}.0."
            ops.rmsnorm_affine_film(x, self.P(f"{p}norms.{i}.gamma"), ssg, h, B, L)
            ops.dwconv(h, self.P(b + "proj_vg.0.weight"), self.P(b + "proj_vg.0.bias"), hd, B, L, k)
            ops.gemm_nt(hd, self._packed[b + "proj_vg.1"], self._packed[b + "proj_vg.1.b"], vg, x3=x3)
            ops.swiglu_rmsnorm(vg, hh, inv, self.hf, self.hp)
            ops.gemm_nt(hh, self._packed[b + "proj_o"], self.P(b + "proj_o.bias"), fo, x3=x3)
            ops.rmsnorm_affine_gate_residual(x, fo, self.P(f"{p}blocks.{i}.1.gamma"), ssg, x, B, L)
        if out is None:
            out = ws.get(tag + ".out", (M, D), dt)
        ops.rmsnorm_affine_film(x, self.P(p + "out_norm.gamma"), None, out, B, L)
        return out

    # ------------------------------------------------------------------ latent/model.py:54 (audio_encoder)
    @torch.no_grad()
    def _audio_encoder(self, audio: torch.Tensor):
        audio = audio.detach().to(torch.float32).contiguous()
        Ba, F, L = audio.shape
        if F != A_DIM:
            raise ValueError(f"audio must have {A_DIM} spectrogram bins, got {F}")
        if L % self.chunk_size:
            raise ValueError(f"audio length {L} is not a multiple of chunk_size {self.chunk_size} (pad_to_multiple first)")
        dt, D, x3 = self._dtype(), self.a_dim, self._x3()
        self._pack(dt)
        ws = self._workspace("enc", Ba, L, dt)
        p = "audio_encoder.0.net."
        f96 = ws.get("f96", (Ba * L, 32 * (A_DIM // 24)), dt)
        ops.spec_features_conv(audio, self.P(p + "1.weight"), self.P(p + "1.bias"), self.P(p + "2.gamma"),
                               self.P(p + "4.weight"), self.P(p + "4.bias"), self.P(p + "5.gamma"), f96)
        pre = ws.get("pre", (Ba * L, D), dt)
        ops.gemm_nt(f96, self._packed["audio_encoder.0.net.8"], self.P(p + "8.bias"), pre, x3=x3)
        x = ws.get("x0", (Ba * L, D), dt)
        ops.rmsnorm_affine_film(pre, self.P(p + "9.gamma"), None, x, Ba, L, act=OD_ACT_SILU)
        skips, Li = [], L
        for i in range(self.n_downs):
            # the skip outlives this call: it gets its own storage, not a workspace slot
            skip = torch.empty(Ba * Li, D, dtype=dt, device=audio.device)
            self._layer(ws, f"l{i}", f"audio_encoder.1.layers.{i}.", x, None, Ba, Li, out=skip)
            skips.append(skip.view(Ba, Li, D).permute(0, 2, 1))
            Lo = Li // self.stride
            x = (torch.empty(Ba * Lo, D, dtype=dt, device=audio.device) if i == self.n_downs - 1
                 else ws.get(f"x{i + 1}", (Ba * Lo, D), dt))
            ops.unet_down(skip, self.P(f"audio_encoder.1.downs.{i}.0.weight"), self.P(f"audio_encoder.1.downs.{i}.0.bias"),
                          x, Ba, Lo, self.stride)
            Li = Lo
        return skips, x.view(Ba, Li, D).permute(0, 2, 1)

    def _frames(self, t: torch.Tensor, dt: torch.dtype) -> torch.Tensor:
        """(B, C, L) tensor -> frame-major [B*L][C] of the compute dtype; free for the views audio_encoder returns."""
        B, C, L = t.shape
        fm = t.permute(0, 2, 1)
        if fm.is_contiguous() and t.dtype == dt:
            return fm.reshape(B * L, C)
        out = torch.empty(B * L, C, dtype=dt, device=t.device)
        ops.cl_to_frames(t.detach().to(torch.float32).contiguous(), out)
        return out

    # ------------------------------------------------------------------ latent/model.py:103-114
    @torch.no_grad()
    def _decode(self, z, s, audio, skips, n_sigmoid: int):
        if skips is None:
            if audio is None:
                raise ValueError("decode needs `audio` or `skips`")
            skips, _ = self._audio_encoder(audio)
        z = z.detach().to(torch.float32).contiguous()
        s = s.detach().to(torch.float32).contiguous()
        B, E, l = z.shape
        dt, D, x3 = self._dtype(), self.a_dim, self._x3()
        self._pack(dt)
        ws = self._workspace("dec", B, l, dt)
        skips = list(skips)
        if len(skips) != self.n_downs:
            raise ValueError(f"expected {self.n_downs} skips, got {len(skips)}")
        x = ws.get("x0", (B * l, D), dt)
        ops.proj_in(z, self.P("proj_emb.weight").view(D, E), self.P("proj_emb.bias"), x)
        Li = l
        for i in range(self.n_downs):
            Lu = Li * self.stride
            xu = ws.get(f"xu{i}", (B * Lu, D), dt)
            ops.unet_up(x, self.P(f"decoder.ups.{i}.1.weight"), self.P(f"decoder.ups.{i}.1.bias"), xu, B, Li, self.stride)
            sk = skips.pop()
            Bs = sk.shape[0]
            if sk.shape[1] != D or sk.shape[2] != Lu or Bs not in (1, B):
                raise ValueError(f"skip {tuple(sk.shape)} does not match ({B}|1, {D}, {Lu})")
            skf = self._frames(sk, dt)
            m = f"decoder.mixers.{i}."
            pr = ws.get(f"pr{i}", (Bs * Lu, D), dt)
            ops.gemm_nt(skf, self._packed[m + "proj.0"], self.P(m + "proj.0.bias"), pr, x3=x3)
            gx = ws.get(f"gx{i}", (B * Lu, D), dt)
            ops.gemm_nt(xu, self._packed[m + "gate"], self.P(m + "gate.bias"), gx, x3=x3)
            ops.unet_mixer(xu, pr, Bs == 1 and B > 1, gx, self.P(m + "proj.1.gamma"), xu, B, Lu)
            x = self._layer(ws, f"l{i}", f"decoder.layers.{i}.", xu, s, B, Lu)
            Li = Lu
        out = torch.empty(B, X_DIM, Li, dtype=torch.float32, device=z.device)
        ops.chart_head(x, self.P("proj_out.weight").view(X_DIM, D), self.P("proj_out.bias"), out, B, Li, n_sigmoid)
        return out, s

    def decode_logits(self, z, s, *, audio=None, skips=None) -> torch.Tensor:
        return self._decode(z, s, audio, skips, n_sigmoid=0)[0]

    def _label_predictor(self, s: torch.Tensor) -> torch.Tensor:            # latent/model.py:72-76
        B = s.shape[0]
        hid = torch.empty(B, self.a_dim, dtype=torch.float32, device=s.device)
        out = torch.empty(B, NUM_LABELS, dtype=torch.float32, device=s.device)
        ops.linear_small(s, self.P("label_predictor.0.weight"), self.P("label_predictor.0.bias"), hid, None, OD_ACT_SILU)
        ops.linear_small(hid, self.P("label_predictor.2.weight"), self.P("label_predictor.2.bias"), out)
        return out

    @torch.no_grad()
    def decode(self, z, s, *, audio=None, skips=None):
        """(chart, labels): sigmoid on the hit signals, cursor signals raw, labels clamped to [0, 10]
        (latent/model.py:116-134)."""
        chart, s32 = self._decode(z, s, audio, skips, n_sigmoid=N_HIT)
        return chart, self._label_predictor(s32).clamp_(0, 10)

    @torch.no_grad()
    def forward(self, audio, z, s):                                          # latent/model.py:78-91
        return self.decode_logits(z, s, audio=audio), self._label_predictor(s.detach().to(torch.float32).contiguous())

    # ------------------------------------------------------------------ latent/model.py:93-101 (encode_chart)
    @torch.no_grad()
    def encode_chart(self, chart: torch.Tensor):
        """chart (B, 9, L) -> z (B, emb_dim, L / chunk_size), s (B, style_dim): the dataset-encoding direction
        (scripts/encode_latents.py): chart encoder, style head (layer + AttnPool + rms_norm), temporal layer / head."""
        chart = chart.detach().to(torch.float32).contiguous()
        B, X, L = chart.shape
        if X != X_DIM:
            raise ValueError(f"chart must have {X_DIM} signals, got {X}")
        if L % self.chunk_size:
            raise ValueError(f"chart length {L} is not a multiple of chunk_size {self.chunk_size} (pad_to_multiple first)")
        dt, D, x3 = self._dtype(), self.a_dim, self._x3()
        self._pack(dt)
        ws = self._workspace("chart", B, L, dt)
        cf = ws.get("cf", (B * L, 16), dt)                   # columns 9..15 stay zero
        ops.cl_to_frames(chart, cf)
        x = ws.get("x0", (B * L, D), dt)
        ops.gemm_nt(cf, self._packed["chart_encoder.0"], self.P("chart_encoder.0.bias"), x, x3=x3)
        Li = L
        for i in range(self.n_downs):
            y = self._layer(ws, f"l{i}", f"chart_encoder.1.layers.{i}.", x, None, B, Li)
            Lo = Li // self.stride
            x = ws.get(f"x{i + 1}", (B * Lo, D), dt)
            ops.unet_down(y, self.P(f"chart_encoder.1.downs.{i}.0.weight"), self.P(f"chart_encoder.1.downs.{i}.0.bias"),
                          x, B, Lo, self.stride)
            Li = Lo
        h = x                                                 # [B*l][D]; both heads below read it
        # style head: layer -> AttnPool -> rms_norm (no gain)
        hs = ws.get("hs", (B * Li, D), dt)
        hs.copy_(h)                                           # _layer updates its input in place
        y = self._layer(ws, "sh", "style_head.0.", hs, None, B, Li)
        heads, hd = self.args.style_heads, self.args.style_head_dim
        sc = ws.get("sc", (B * Li, heads), dt)
        va = ws.get("va", (B * Li, heads * hd), dt)
        ops.gemm_nt(y, self._packed["style_head.1.scores"], self.P("style_head.1.scores.bias"), sc, x3=x3)
        ops.gemm_nt(y, self._packed["style_head.1.values"], self.P("style_head.1.values.bias"), va, x3=x3)
        pooled = ws.get("pooled", (B, heads * hd), torch.float32)
        ops.attn_pool(sc, va, pooled, B, Li, heads, hd)
        s_pre = ws.get("s_pre", (B, self.style_dim), torch.float32)
        ops.linear_small(pooled, self.P("style_head.1.proj_out.weight"), self.P("style_head.1.proj_out.bias"), s_pre)
        s = torch.empty(B, self.style_dim, dtype=torch.float32, device=chart.device)
        ops.rmsnorm_rows(s_pre, None, s, 1e-6)
        # temporal layer (FiLM from s) -> Conv1d(D -> emb_dim) -> rms_norm over the emb_dim channels
        y = self._layer(ws, "tl", "temporal_layer.", h, s, B, Li)
        z = torch.empty(B, self.emb_dim, Li, dtype=torch.float32, device=chart.device)
        ops.chart_head(y, self.P("temporal_head.0.weight").view(self.emb_dim, D), self.P("temporal_head.0.bias"), z, B, Li, 0,
                       rms=True)
        return z, s
