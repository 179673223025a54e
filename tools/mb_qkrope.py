"""qkv projection + q/k norm + RoPE at the bench shape (M = 262144, K = 512, 16 x 64): two kernels vs the fused large-M epilogue.
usage: python tools/mb_qkrope.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import timeit
from osu_dreamer_amd import ops
dev, bf = torch.device("cuda:0"), torch.bfloat16
B, L, H, hd, K = 32, 8192, 16, 64, 512
M, dh = B * L, H * hd
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).to(bf); W = (torch.randn(3 * dh, K, device=dev, generator=g) * 0.05).to(bf)
b = torch.zeros(3 * dh, device=dev); wq = torch.ones(hd, device=dev); wk = torch.ones(hd, device=dev)
tab = torch.zeros(L, hd // 2, 2, device=dev); ops.rope_table(tab, L, hd)
qkv = torch.zeros(M, 3 * dh, dtype=bf, device=dev); qk = torch.zeros(M, 2 * dh, dtype=bf, device=dev)
t1 = timeit(lambda: ops.gemm_nt(A, W, b, qkv), 20)
t2 = timeit(lambda: ops.qk_norm_rope(qkv, wq, wk, tab, qk, B, L, H, hd, 1.2e-7, q_scale=0.18), 20)
t3 = timeit(lambda: ops.gemm_nt_qkrope_split(A, W, b, qkv, qk, wq, wk, tab, L, H, hd, 1.2e-7, q_scale=0.18), 20)
t4 = timeit(lambda: ops.gemm_nt_qkrope(A, W, b, qkv, wq, wk, tab, L, H, hd, 1.2e-7, q_scale=0.18), 20)
print(f"gemm_nt {t1:.3f} ms + qk_norm_rope {t2:.3f} ms = {t1 + t2:.3f} ms | fused split {t3:.3f} ms | fused in place {t4:.3f} ms")
