"""Context number, not a test (pytest does not collect it): the reference's algorithm as plain PyTorch-ROCm ops on the
same MI355X — the oracle's train step moved to cuda under bf16 autocast, with its three heavy helpers swapped for the torch
calls the reference itself makes (F.scaled_dot_product_attention at attn.py:82, nn.Conv1d at swiglu.py:20-21 / attn.py:68).
Forward + loss + backward + clip + AdamW + EMA, B = 32 x L = 8192 (BASELINE configs[1]).
usage (GPU box):  python tests/perf_torch_reference_gpu.py [B] [L]"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import denoiser_oracle as O  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = torch.device("cuda:0")
d = O.FULL
# (.contiguous(): as written at attn.py:75-82 `v` is a strided view, which sends torch to the math path that
# materialises the L x L scores — 34 GiB per 8 samples at L = 8192; the flash path is the favourable case for torch)
O.attention_core = lambda q, k, v, q_chunk=0: F.scaled_dot_product_attention(q.contiguous(), k.contiguous(), v.contiguous())
O.pointwise_conv = lambda x, w, b: F.conv1d(x, w, b)
O.depthwise_conv = lambda x, w, b: F.conv1d(x, w, b, padding=w.shape[-1] // 2, groups=x.shape[1])
_rope = {}


def rope(x):        # attn.py:12-29: cos/sin cached per (N, hd, device), cast to x.dtype
    n, hd = x.shape[-2], x.shape[-1]
    if (n, hd) not in _rope:
        inv_freq = 10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32, device=dev) / -hd)
        ang = torch.outer(torch.arange(n, dtype=torch.float32, device=dev), inv_freq)
        _rope[(n, hd)] = (ang.cos(), ang.sin())
    c, s = (t.to(x.dtype) for t in _rope[(n, hd)])
    a, b = x[..., : hd // 2], x[..., hd // 2:]
    return torch.cat([a * c - b * s, a * s + b * c], dim=-1)


O.rope_half_split = rope
P = {k: v.to(dev).requires_grad_(True) for k, v in O.init_params(d, seed=1234).items()}
ema = {k: v.detach().clone() for k, v in P.items()}
data = {k: v.to(dev) for k, v in O.synthetic_batch(d, B, L, seed=1234).items()}
opt = torch.optim.AdamW(list(P.values()), lr=3e-4, weight_decay=0.01)


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _ = O.train_loss(P, d, data["h"], data["z"], data["s"], data["t"], data["x0"])
    loss.backward()
    torch.nn.utils.clip_grad_norm_(list(P.values()), 1.0)
    opt.step()
    with torch.no_grad():
        torch._foreach_lerp_(list(ema.values()), [p.detach() for p in P.values()], 0.01)
    return loss


for _ in range(2):
    loss = step()
torch.cuda.synchronize()
t0 = time.time()
n = 3
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
ms = (time.time() - t0) / n * 1e3
print(f"torch-eager reference algorithm on cuda:0, bf16 autocast, B={B} L={L}: {ms:.1f} ms/step "
      f"({1e3 / ms:.3f} steps/s), peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, loss {float(loss):.4f}")

# ---- sampler (BASELINE configs[3]): 50 steps, 4 difficulties, L = 1115, audio batch 1
if len(sys.argv) <= 1:
    del P, ema, opt, data
    torch.cuda.empty_cache()
    Ps = {k: v.to(dev) for k, v in O.init_params(d, seed=5).items()}
    g = torch.Generator().manual_seed(5)
    Bs, Ls = 4, 1115
    h = torch.randn(1, d.a_dim, Ls, generator=g).to(dev)
    s_ = torch.randn(Bs, d.style_dim, generator=g).to(dev)
    x_init = torch.randn(Bs, d.emb_dim, Ls, generator=g).to(dev)
    for name, ctx in (("fp32", torch.autocast("cuda", enabled=False)), ("bf16 autocast", torch.autocast("cuda", dtype=torch.bfloat16))):
        with torch.no_grad(), ctx:
            O.sample(h, s_, 50, x_init, Ps, d)
            torch.cuda.synchronize()
            t0 = time.time()
            O.sample(h, s_, 50, x_init, Ps, d)
            torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"torch-eager reference sampler on cuda:0, {name}, 50 steps B={Bs} L={Ls}: {dt * 1e3:.1f} ms ({Bs * Ls / dt:.0f} latents/s)")
