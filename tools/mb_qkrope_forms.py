import os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
dev, bf = torch.device("cuda:0"), torch.bfloat16
B, L, H, hd, K = 32, 8192, 16, 64, 512
M, dh = B * L, H * hd
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).to(bf); W = (torch.randn(3 * dh, K, device=dev, generator=g) * 0.05).to(bf)
b = torch.zeros(3 * dh, device=dev); wq = torch.ones(hd, device=dev); wk = torch.ones(hd, device=dev)
tab = torch.zeros(L, hd // 2, 2, device=dev); ops.rope_table(tab, L, hd)
qkv = torch.zeros(M, 3 * dh, dtype=bf, device=dev); qk = torch.zeros(M, 2 * dh, dtype=bf, device=dev)
fns = {"plain": lambda: ops.gemm_nt(A, W, b, qkv),
       "split (training form: pre-norm q,k + rotated copy)": lambda: ops.gemm_nt_qkrope_split(A, W, b, qkv, qk, wq, wk, tab, L, H, hd, 1.2e-7, q_scale=0.18),
       "in place (no-grad form: rotated q,k only)": lambda: ops.gemm_nt_qkrope(A, W, b, qkv, wq, wk, tab, L, H, hd, 1.2e-7, q_scale=0.18)}
for rd in range(2):
    for name, fn in fns.items():
        for _ in range(3): fn()
        n = 300
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"[{rd}] {name}: {e0.elapsed_time(e1) / n:.3f} ms per launch (sustained, {n} launches)", flush=True)
