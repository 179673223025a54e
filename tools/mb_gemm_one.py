"""One process, a few launches of od_gemm_nt at one bench shape (M = 262144, bf16): the target of
`rocprofv3 --pmc ... -- python3 tools/mb_gemm_one.py N K [iters]`."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
N, K = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev, bf, M = torch.device("cuda:0"), torch.bfloat16, 32 * 8192
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).to(bf)
W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(bf)
bias = torch.zeros(N, device=dev)
C = torch.zeros(M, N, dtype=bf, device=dev)
for _ in range(iters):
    ops.gemm_nt(A, W, bias, C)
torch.cuda.synchronize()
print("done")
