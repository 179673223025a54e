"""Attention fwd / bwd at the bench shape with and without the pre-multiplied q (q_prescaled); bf16.
usage: python tools/mb_attn_pre.py [0|1]"""
import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
from tools.microbench import timeit
pre = len(sys.argv) > 1 and sys.argv[1] == "1"
dev = torch.device("cuda:0")
B, L, H, hd = 32, 8192, 16, 64
M, dh = B * L, H * hd
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
qk, qkv = r(M, 2 * dh), r(M, 3 * dh)
if pre:
    qk[:, :dh] *= math.log2(math.e) / math.sqrt(hd)
o, do = torch.zeros(M, dh, dtype=bf, device=dev), r(M, dh)
lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
dqk, dqkv = torch.zeros_like(qk), torch.zeros_like(qkv)
sc = 1 / math.sqrt(hd)
kw = dict(q_prescaled=True) if pre else {}
unit = 2.0 * B * H * L * L * hd
t = timeit(lambda: ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, lse, B, H, L, hd, sc, **kw), 5)
print(f"pre={int(pre)} fwd {t:.3f} ms {2 * unit / t / 1e9:.1f} TF/s")
t = timeit(lambda: ops.flash_attn_bwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, do, lse, delta, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:], B, H, L, hd, sc, **kw), 5)
print(f"pre={int(pre)} bwd {t:.3f} ms {7 * unit / t / 1e9:.1f} TF/s")
