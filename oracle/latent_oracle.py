"""CPU oracle for the latent model's inference path.  TEST INFRASTRUCTURE ONLY.

Functional torch restatement of the two steps either side of `diffusion.sample` in `LDM.sample`
(osu_dreamer/models/inference/model.py:44-51):

  * `LatentModel.audio_encoder`  = SpecFeatures (models/latent/spec_features.py:10-32) + UNetEncoder
    (models/latent/unet.py:55-75)                                         -> (skips, h)
  * `LatentModel.encode_chart`   = chart_encoder -> style_head (layer, AttnPool, rms_norm) -> temporal layer / head
    (models/latent/model.py:23-36,93-101) — the dataset-encoding direction (scripts/encode_latents.py)
  * `LatentModel.decode`         = proj_emb -> UNetDecoder (unet.py:77-101, mixer :117-126, layer :21-53)
    -> proj_out -> sigmoid on the hit signals, label_predictor clamp       (models/latent/model.py:103-134)

over a flat dict keyed like `LatentModel.state_dict()` restricted to the parameters those two calls read.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.  Pinned by
tests/golden/latent_*.npz, which oracle/make_golden.py produced by running the reference's LatentModel itself.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

A_DIM = 72          # data/load_audio.py:11-15  (9 bins x 8 octaves)
X_DIM = 9           # data/beatmap/encode.py:14-29
N_HIT = 7           # HitSignals = channels 0..6, CursorSignals = 7, 8  (encode.py:31-46)
NUM_LABELS = 5      # encode.py:50


@dataclass(frozen=True)
class LatentDims:
    emb_dim: int = 6
    style_dim: int = 32
    n_downs: int = 3
    stride: int = 3
    h_dim: int = 128
    n_layers: int = 8
    expand: int = 4
    radius: int = 2

    @property
    def hf(self) -> int:
        return int(self.h_dim * self.expand * 2 / 3)     # common/swiglu.py:17

    @property
    def chunk_size(self) -> int:
        return self.stride ** self.n_downs               # latent/model.py:51


LATENT_TINY = LatentDims(n_downs=2, h_dim=32, n_layers=2)
LATENT_FULL = LatentDims()


def _layer_shapes(p: str, d: LatentDims, cond: bool):
    D, hf, k = d.h_dim, d.hf, 1 + 2 * d.radius
    s = {}
    for i in range(d.n_layers):
        s[f"{p}norms.{i}.gamma"] = (D,)
    for i in range(d.n_layers):
        b = f"{p}blocks.{i}."
        s[b + "0.proj_vg.0.weight"] = (D, 1, k); s[b + "0.proj_vg.0.bias"] = (D,)
        s[b + "0.proj_vg.1.weight"] = (2 * hf, D, 1); s[b + "0.proj_vg.1.bias"] = (2 * hf,)
        s[b + "0.proj_o.weight"] = (D, hf, 1); s[b + "0.proj_o.bias"] = (D,)
        s[b + "1.gamma"] = (D,)
    s[p + "out_norm.gamma"] = (D,)
    if cond:
        for i in range(d.n_layers):
            s[f"{p}films.{i}.weight"] = (3 * D, d.style_dim); s[f"{p}films.{i}.bias"] = (3 * D,)
    return s


def latent_param_shapes(d: LatentDims) -> Dict[str, tuple]:
    """Keys/shapes of `LatentModel.state_dict()` read by audio_encoder and decode, in state-dict order."""
    D, ks = d.h_dim, 1 + 2 * (d.stride // 2)
    s = {"audio_encoder.0.net.1.weight": (8, 1, 8, 3), "audio_encoder.0.net.1.bias": (8,),
         "audio_encoder.0.net.2.gamma": (8,),
         "audio_encoder.0.net.4.weight": (32, 8, 6, 3), "audio_encoder.0.net.4.bias": (32,),
         "audio_encoder.0.net.5.gamma": (32,),
         "audio_encoder.0.net.8.weight": (D, 32 * (A_DIM // 24), 1), "audio_encoder.0.net.8.bias": (D,),
         "audio_encoder.0.net.9.gamma": (D,)}
    for i in range(d.n_downs):
        s[f"audio_encoder.1.downs.{i}.0.weight"] = (D, 1, ks); s[f"audio_encoder.1.downs.{i}.0.bias"] = (D,)
    for i in range(d.n_downs):
        s.update(_layer_shapes(f"audio_encoder.1.layers.{i}.", d, cond=False))
    s["proj_emb.weight"] = (D, d.emb_dim, 1); s["proj_emb.bias"] = (D,)
    for i in range(d.n_downs):
        s[f"decoder.ups.{i}.1.weight"] = (D, 1, ks); s[f"decoder.ups.{i}.1.bias"] = (D,)
    for i in range(d.n_downs):
        s.update(_layer_shapes(f"decoder.layers.{i}.", d, cond=True))
    for i in range(d.n_downs):
        m = f"decoder.mixers.{i}."
        s[m + "proj.0.weight"] = (D, D, 1); s[m + "proj.0.bias"] = (D,); s[m + "proj.1.gamma"] = (D,)
        s[m + "gate.weight"] = (D, D, 1); s[m + "gate.bias"] = (D,)
    s["proj_out.weight"] = (X_DIM, D, 1); s["proj_out.bias"] = (X_DIM,)
    s["label_predictor.0.weight"] = (D, d.style_dim); s["label_predictor.0.bias"] = (D,)
    s["label_predictor.2.weight"] = (NUM_LABELS, D); s["label_predictor.2.bias"] = (NUM_LABELS,)
    return s


def init_latent_params(d: LatentDims, seed: int) -> Dict[str, torch.Tensor]:
    """Seeded, non-degenerate values for every parameter (the reference zero-initialises the FiLMs and the
    mixer gates, unet.py:15-19,34,120, and gives the block norms a 1e-3 gain, :28 — a parity test on those
    would see nothing): fan-in-scaled normals for weights, O(0.1) biases, gains around their init value."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name, shape in latent_param_shapes(d).items():
        if name.endswith("gamma"):
            base = 0.3 if ".blocks." in name else 1.0
            t = base * (1.0 + 0.2 * torch.randn(shape, generator=g))
        elif name.endswith("bias"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif ".films." in name:
            t = 0.3 * torch.randn(shape, generator=g) / math.sqrt(shape[1])
        else:
            fan_in = 1
            for n in shape[1:]:
                fan_in *= n
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        P[name] = t
    return P


def latent_encoder_param_shapes(d: LatentDims, style_head_dim: int, style_heads: int) -> Dict[str, tuple]:
    """The rest of `LatentModel.state_dict()`: what `encode_chart` reads (chart_encoder, style_head, temporal_*)."""
    D, ks = d.h_dim, 1 + 2 * (d.stride // 2)
    s = {"chart_encoder.0.weight": (D, X_DIM, 1), "chart_encoder.0.bias": (D,)}
    for i in range(d.n_downs):
        s[f"chart_encoder.1.downs.{i}.0.weight"] = (D, 1, ks); s[f"chart_encoder.1.downs.{i}.0.bias"] = (D,)
    for i in range(d.n_downs):
        s.update(_layer_shapes(f"chart_encoder.1.layers.{i}.", d, cond=False))
    s.update(_layer_shapes("style_head.0.", d, cond=False))
    hd = style_head_dim * style_heads
    s["style_head.1.scores.weight"] = (style_heads, D, 1); s["style_head.1.scores.bias"] = (style_heads,)
    s["style_head.1.values.weight"] = (hd, D, 1); s["style_head.1.values.bias"] = (hd,)
    s["style_head.1.proj_out.weight"] = (d.style_dim, hd); s["style_head.1.proj_out.bias"] = (d.style_dim,)
    s.update(_layer_shapes("temporal_layer.", d, cond=True))
    s["temporal_head.0.weight"] = (d.emb_dim, D, 1); s["temporal_head.0.bias"] = (d.emb_dim,)
    return s


def init_latent_encoder_params(d: LatentDims, seed: int, style_head_dim: int, style_heads: int) -> Dict[str, torch.Tensor]:
    """Seeded values for the encode_chart parameters (own generator: init_latent_params' stream is untouched)."""
    g = torch.Generator().manual_seed(seed + 50)
    P = {}
    for name, shape in latent_encoder_param_shapes(d, style_head_dim, style_heads).items():
        if name.endswith("gamma"):
            base = 0.3 if ".blocks." in name else 1.0
            t = base * (1.0 + 0.2 * torch.randn(shape, generator=g))
        elif name.endswith("bias"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif ".films." in name:
            t = 0.3 * torch.randn(shape, generator=g) / math.sqrt(shape[1])
        else:
            fan_in = 1
            for n in shape[1:]:
                fan_in *= n
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        P[name] = t
    return P


# ---------------------------------------------------------------------------------------------------
def rms_norm(x, gamma=None):                      # common/rms_norm.py:6-16 (dim 1, eps 1e-6)
    y = x * x.pow(2).mean(dim=1, keepdim=True).add(1e-6).rsqrt()
    if gamma is not None:
        y = y * gamma.view((-1,) + (1,) * (x.dim() - 2))
    return y


def spec_features(P, audio, p="audio_encoder.0.net."):
    """spec_features.py:17-32: (B, 72, L) -> (B, h_dim, L)."""
    x = audio[:, None]                                                          # Unflatten(1, (1, -1))
    x = F.conv2d(x, P[p + "1.weight"], P[p + "1.bias"], stride=(6, 1), padding=(1, 1))
    x = F.silu(rms_norm(x, P[p + "2.gamma"]))
    x = F.conv2d(x, P[p + "4.weight"], P[p + "4.bias"], stride=(4, 1), padding=(1, 1))
    x = F.silu(rms_norm(x, P[p + "5.gamma"]))
    x = x.flatten(1, 2)                                                         # 'b c a l -> b (c a) l'
    x = F.conv1d(x, P[p + "8.weight"], P[p + "8.bias"])
    return F.silu(rms_norm(x, P[p + "9.gamma"]))


def swiglu(P, p, x, d: LatentDims):               # common/swiglu.py:27-32 (dropout p = 0)
    h = F.conv1d(x, P[p + "proj_vg.0.weight"], P[p + "proj_vg.0.bias"], padding=d.radius, groups=x.shape[1])
    v, g = F.conv1d(h, P[p + "proj_vg.1.weight"], P[p + "proj_vg.1.bias"]).chunk(2, dim=1)
    return F.conv1d(rms_norm(v * F.silu(g)), P[p + "proj_o.weight"], P[p + "proj_o.bias"])


def layer(P, p, x, cond, d: LatentDims):          # unet.py:36-53
    for i in range(d.n_layers):
        if cond is not None:
            scale, shift, gate = F.linear(cond, P[f"{p}films.{i}.weight"], P[f"{p}films.{i}.bias"])[:, :, None].chunk(3, dim=1)
        else:
            scale = shift = gate = 0.0
        h = rms_norm(x, P[f"{p}norms.{i}.gamma"]) * (1 + scale) + shift
        h = rms_norm(swiglu(P, f"{p}blocks.{i}.0.", h, d), P[f"{p}blocks.{i}.1.gamma"])
        x = x + h * (1 + gate)
    return rms_norm(x, P[p + "out_norm.gamma"])


def unet_encoder(P, p, x, d: LatentDims) -> Tuple[List[torch.Tensor], torch.Tensor]:     # unet.py:68-75
    skips = []
    for i in range(d.n_downs):
        x = layer(P, f"{p}layers.{i}.", x, None, d)
        skips.append(x)                                                         # unmixer = identity pair (:103-115)
        x = F.conv1d(x, P[f"{p}downs.{i}.0.weight"], P[f"{p}downs.{i}.0.bias"], padding=d.stride // 2, groups=x.shape[1])
        x = F.avg_pool1d(x, d.stride)
    return skips, x


def audio_encoder(P, audio, d: LatentDims):       # latent/model.py:54
    return unet_encoder(P, "audio_encoder.1.", spec_features(P, audio), d)


def unet_decoder(P, p, skips, x, cond, d: LatentDims):                                  # unet.py:90-101
    skips = list(skips)
    for i in range(d.n_downs):
        x = F.interpolate(x, scale_factor=d.stride, mode="nearest")
        x = F.conv1d(x, P[f"{p}ups.{i}.1.weight"], P[f"{p}ups.{i}.1.bias"], padding=d.stride // 2, groups=x.shape[1])
        skip = skips.pop().expand(x.shape[0], -1, -1)
        m = f"{p}mixers.{i}."
        proj = rms_norm(F.conv1d(skip, P[m + "proj.0.weight"], P[m + "proj.0.bias"]), P[m + "proj.1.gamma"])
        x = x + proj * F.conv1d(x, P[m + "gate.weight"], P[m + "gate.bias"])      # mixer, :117-126
        x = layer(P, f"{p}layers.{i}.", x, cond, d)
    return x


def decode_logits(P, z, s, skips, d: LatentDims):                                       # latent/model.py:103-114
    x = F.conv1d(z, P["proj_emb.weight"], P["proj_emb.bias"])
    x = unet_decoder(P, "decoder.", skips, x, s, d)
    return F.conv1d(x, P["proj_out.weight"], P["proj_out.bias"])


def label_predictor(P, s):                                                              # latent/model.py:72-76
    return F.linear(F.silu(F.linear(s, P["label_predictor.0.weight"], P["label_predictor.0.bias"])),
                    P["label_predictor.2.weight"], P["label_predictor.2.bias"])


def decode(P, z, s, skips, d: LatentDims):                                              # latent/model.py:116-134
    logits = decode_logits(P, z, s, skips, d)
    chart = torch.cat([logits[:, :N_HIT].sigmoid(), logits[:, N_HIT:]], dim=1)
    return chart, label_predictor(P, s).clamp(0, 10)


def attn_pool(P, p, x, heads: int):                                                     # latent/model.py:23-36
    a = F.conv1d(x, P[p + "scores.weight"], P[p + "scores.bias"]).softmax(dim=-1)       # B H L
    v = F.conv1d(x, P[p + "values.weight"], P[p + "values.bias"]).unflatten(1, (heads, -1))   # B H D L
    return F.linear(torch.einsum("bhl,bhdl->bhd", a, v).flatten(1), P[p + "proj_out.weight"], P[p + "proj_out.bias"])


def encode_chart(P, chart, d: LatentDims, style_heads: int):                            # latent/model.py:93-101
    x = F.conv1d(chart, P["chart_encoder.0.weight"], P["chart_encoder.0.bias"])
    _, h = unet_encoder(P, "chart_encoder.1.", x, d)
    s = rms_norm(attn_pool(P, "style_head.1.", layer(P, "style_head.0.", h, None, d), style_heads))
    z = rms_norm(F.conv1d(layer(P, "temporal_layer.", h, s, d), P["temporal_head.0.weight"], P["temporal_head.0.bias"]))
    return z, s


def pad_to_multiple(x, chunk):                                                          # data/modules/beatmap.py:26-30
    pad = (chunk - x.shape[-1] % chunk) % chunk
    return F.pad(x, (0, pad), mode="replicate") if pad > 0 else x
