"""Vendor-library reference point: torch.matmul (hipBLASLt / rocBLAS) at the GEMM shapes of the bench, bf16.
usage: python tools/mb_torch_gemm.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import timeit
from osu_dreamer_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
for M in (262144, 4460):
    for name, N, K in (("qkv", 3072, 512), ("out", 512, 1024), ("vg", 2816, 512), ("proj_o", 512, 1408), ("d_qkv", 512, 3072), ("d_vg", 512, 2816),
                       ("d_out", 1024, 512), ("d_proj_o", 1408, 512)):
        A = torch.randn(M, K, device=dev).to(bf); W = torch.randn(N, K, device=dev).to(bf)
        C = torch.empty(M, N, dtype=bf, device=dev); bias = torch.zeros(N, device=dev)
        it = 5 if M > 100000 else 50
        t_lib = timeit(lambda: torch.matmul(A, W.t(), out=C), it)
        t_own = timeit(lambda: ops.gemm_nt(A, W, bias, C), it)
        fl = 2.0 * M * N * K / 1e9
        print(f"M={M:7d} {name:7s} N={N:5d} K={K:5d}  torch.matmul {t_lib:8.3f} ms {fl / t_lib:7.1f} TF/s | od_gemm_nt {t_own:8.3f} ms {fl / t_own:7.1f} TF/s")

print("# weight-gradient shapes: dW[N,K] = G[M,N]^T A[M,K], M = 262144")
M = 262144
for name, N, K in (("qkv", 3072, 512), ("out", 512, 1024), ("vg", 2816, 512), ("proj_o", 512, 1408)):
    G = torch.randn(M, N, device=dev).to(bf); A = torch.randn(M, K, device=dev).to(bf)
    dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    out = torch.empty(N, K, dtype=bf, device=dev)
    t_lib = timeit(lambda: torch.matmul(G.t(), A, out=out), 5)
    t_own = timeit(lambda: ops.gemm_tn(G, A, dW, dbias=db), 5)
    fl = 2.0 * M * N * K / 1e9
    print(f"M={M:7d} {name:7s} N={N:5d} K={K:5d}  torch.matmul(G^T A) {t_lib:8.3f} ms {fl / t_lib:7.1f} TF/s | od_gemm_tn (+bias grad, fp32 out) {t_own:8.3f} ms {fl / t_own:7.1f} TF/s")
