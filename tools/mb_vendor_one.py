"""torch.matmul (hipBLASLt / rocBLAS) at the N = 512 shapes of the step, for `rocprofv3 --kernel-trace --stats` (which kernels does the vendor pick?)."""
import torch
dev, bf, M = torch.device("cuda:0"), torch.bfloat16, 262144
for N, K in ((512, 3072), (512, 1024), (512, 1408), (3072, 512)):
    A = torch.randn(M, K, device=dev).to(bf); W = torch.randn(N, K, device=dev).to(bf); C = torch.empty(M, N, dtype=bf, device=dev)
    for _ in range(5):
        torch.matmul(A, W.t(), out=C)
torch.cuda.synchronize()
