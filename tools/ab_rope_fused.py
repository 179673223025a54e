"""A/B at the bench shape (B=32, L=8192, 16 heads x 64, bf16): od_flash_attn_bwd + od_qk_norm_rope_bwd against od_flash_attn_bwd_qkrope
(the norm + RoPE backward in the attention backward's epilogues).  One process, interleaved repetitions."""
import math, os, sys
import torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
from tools.microbench import timeit
dev = torch.device("cuda:0")
B, L, H, hd = 32, 8192, 16, 64
M, dh, bf = B * L, H * hd, torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
sc = 1 / math.sqrt(hd); qs = sc * math.log2(math.e); eps = float(torch.finfo(torch.float32).eps)
qkv, do = r(M, 3 * dh), r(M, dh)
wq, wk = torch.ones(hd, device=dev), torch.ones(hd, device=dev)
tab = torch.zeros(L, hd // 2, 2, device=dev); ops.rope_table(tab, L, hd)
qk = torch.zeros(M, 2 * dh, dtype=bf, device=dev)
ops.qk_norm_rope(qkv, wq, wk, tab, qk, B, L, H, hd, eps, q_scale=qs)
o = torch.zeros(M, dh, dtype=bf, device=dev)
lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, lse, B, H, L, hd, sc, q_prescaled=True)
dqk, dqkv, dqkv2 = torch.zeros_like(qk), torch.zeros_like(qkv), torch.zeros_like(qkv)
dwq, dwk = torch.zeros(hd, device=dev), torch.zeros(hd, device=dev)
def attn(): ops.flash_attn_bwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, do, lse, delta, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:], B, H, L, hd, sc, q_prescaled=True)
def rope(): ops.qk_norm_rope_bwd(qkv, wq, wk, tab, dqk, dqkv, dwq, dwk, B, L, H, hd, eps, q_scale=qs)
def fused(): ops.flash_attn_bwd_qkrope(qk[:, :dh], qk[:, dh:], qkv, o, do, lse, delta, dqkv2, wq, wk, tab, dwq, dwk, B, H, L, hd, sc, eps, q_scale=qs, q_prescaled=True)
attn(); rope(); fused(); torch.cuda.synchronize()
print("fused vs separate rel diff", float((dqkv2.float() - dqkv.float()).norm() / dqkv.float().norm()))
for rep in range(3):
    ta, tr, tf = timeit(attn, 5), timeit(rope, 5), timeit(fused, 5)
    print(f"[{rep}] attention bwd {ta:.3f} ms + norm/RoPE bwd {tr:.3f} ms = {ta + tr:.3f} ms | fused {tf:.3f} ms", flush=True)
