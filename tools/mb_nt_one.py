"""One NT GEMM shape, a few launches, for counter runs: python3 tools/mb_nt_one.py N K [vendor]
(the 4-wave kernel is the default; OD_NT_W4=0 in the environment selects the 8-wave kernel)."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops

N, K = int(sys.argv[1]), int(sys.argv[2])
vendor = len(sys.argv) > 3 and sys.argv[3] == "vendor"
dev, bf, M = torch.device("cuda:0"), torch.bfloat16, 32 * 8192
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).to(bf)
W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(bf)
C = torch.zeros(M, N, dtype=bf, device=dev)
if vendor:
    for _ in range(6):
        torch.matmul(A, W.t(), out=C)
else:
    for _ in range(6):
        ops.gemm_nt(A, W, None, C)
torch.cuda.synchronize()
