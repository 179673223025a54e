"""od_dwconv_bwd at the bench shape (B=32, L=8192, C=512, k=5, bf16): ms and HBM GB/s (reads x, dy; writes dx)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
from tools.microbench import timeit
dev = torch.device("cuda:0")
B, L, C, ks = 32, 8192, 512, 5
M = B * L
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, C, device=dev, generator=g).bfloat16()
dy = torch.randn(M, C, device=dev, generator=g).bfloat16()
dx = torch.zeros_like(x)
w = torch.randn(C, ks, device=dev, generator=g)
dw, db = torch.zeros(C, ks, device=dev), torch.zeros(C, device=dev)
t = timeit(lambda: ops.dwconv_bwd(x, w, dy, dx, dw, db, B, L, ks), 20)
print(f"dwconv_bwd {t * 1e3:7.1f} us  {3 * M * C * 2 / t / 1e6:7.1f} GB/s")
