"""Attention backward alone (od_flash_attn_bwd = delta + dK/dV + dQ kernels) at the bench shape, bf16; used for
A/B runs of kernel variants (tools/build_variant.sh + OSU_DREAMER_HIP_LIB).  usage: python tools/mb_bwd.py"""
import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
from tools.microbench import timeit
dev = torch.device("cuda:0")
B, L, H, hd = 32, 8192, 16, 64
M, dh = B * L, H * hd
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
qk, qkv = r(M, 2 * dh), r(M, 3 * dh)
o, do = torch.zeros(M, dh, dtype=bf, device=dev), r(M, dh)
lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
dqk, dqkv = torch.zeros_like(qk), torch.zeros_like(qkv)
sc = 1 / math.sqrt(hd)
ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, lse, B, H, L, hd, sc)
t = timeit(lambda: ops.flash_attn_bwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, do, lse, delta, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:], B, H, L, hd, sc), 5)
print("bwd total %.3f ms" % t)
