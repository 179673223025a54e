"""od_gemm_nt at the sampler's shapes (M = 4 x 1115 = 4460 rows) for bf16 / fp32 / fp32-as-3xbf16, against torch.matmul (bf16).
usage: python tools/mb_gemm_small.py      (OSU_DREAMER_HIP_LIB selects the library)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import timeit
from osu_dreamer_amd import ops
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4460
out = []
for name, N, K in (("qkv", 3072, 512), ("out", 512, 1024), ("vg", 2816, 512), ("proj_o", 512, 1408)):
    row = f"{name:7s} N={N:5d} K={K:5d} "
    for dt, label in ((torch.bfloat16, "bf16"), (torch.float32, "fp32")):
        A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        C = torch.empty(M, N, dtype=dt, device=dev); bias = torch.zeros(N, device=dev)
        t = timeit(lambda: ops.gemm_nt(A, W, bias, C), 200)
        row += f"| {label} {t * 1e3:7.1f} us {2.0 * M * N * K / t / 1e9:6.0f} TF/s "
        if dt == torch.bfloat16:
            t = timeit(lambda: torch.matmul(A, W.t(), out=C), 200)
            row += f"(torch {t * 1e3:6.1f} us) "
    print(row, flush=True)
