"""od_gemm_tn at the proj_cl weight-gradient shape (M = 262144, N = 512, K = 128, bf16): python tools/mb_tn_small.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
from tools.microbench import timeit
dev, bf, M = torch.device("cuda:0"), torch.bfloat16, 32 * 8192
g = torch.Generator(device=dev).manual_seed(0)
G, A = torch.randn(M, 512, device=dev, generator=g).to(bf), torch.randn(M, 128, device=dev, generator=g).to(bf)
dW, db = torch.zeros(512, 128, device=dev), torch.zeros(512, device=dev)
ops.gemm_tn(G, A, dW, dbias=db)
ref = G.float().t() @ A.float()
err = float((dW - ref).norm() / ref.norm())
t = timeit(lambda: ops.gemm_tn(G, A, dW, dbias=db), 20)
print(f"w_proj_cl {t * 1e3:.1f} us, {(G.numel() + A.numel()) * 2 / t / 1e6:.0f} GB/s, err {err:.1e}")
