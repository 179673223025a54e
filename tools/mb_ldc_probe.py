"""Does the pitch of the output rows change the NT GEMM's write rate?  The K = 512 products write 512-byte row segments at a 2 N-byte stride and reach
~2 TB/s where a contiguous fill reaches 6.8 (profiles/r06g_nt_store_policy.txt).  Times one product per output pitch (ldc), sustained ~1 s each:
    python3 tools/mb_ldc_probe.py [N K]"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
dev, bf, M = torch.device("cuda:0"), torch.bfloat16, 32 * 8192
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3072, 512)
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).to(bf)
W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(bf)
for ldc in [N, N + 64, N + 128, N + 512, N + 1024, 4096, 4096 + 64, 8192]:
    if ldc < N:
        continue
    C = torch.zeros(M, ldc, dtype=bf, device=dev)
    run = lambda: ops.gemm_nt(A, W, None, C[:, :N])
    for _ in range(3):
        run()
    n = max(50, int(1.0e-3 * 1.0e15 / (2.0 * M * N * K) * 1000))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"N={N} K={K} ldc={ldc:5d} ({ldc * 2} B rows): {ms * 1e3:7.1f} us  {2.0 * M * N * K / ms / 1e9:6.0f} TF/s  writes {M * N * 2 / ms / 1e9:5.2f} TB/s", flush=True)
    del C
