"""The qkv projection with the q/k norm + RoPE epilogue (training form: two outputs) at the bench shape, a few launches, for counter runs and
variant A/Bs:  python3 tools/mb_qkrope_one.py [plain] [time]
`plain` = od_gemm_nt of the same shape instead (gemm_nt_w4_kernel<0>); `time` prints ms per launch (events, 30 launches)."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops, _lib
dev, bf = torch.device("cuda:0"), torch.bfloat16
B, L, H, hd, K = 32, 8192, 16, 64, 512
M, dh = B * L, H * hd
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).to(bf); W = (torch.randn(3 * dh, K, device=dev, generator=g) * 0.05).to(bf)
b = torch.zeros(3 * dh, device=dev); wq = torch.ones(hd, device=dev); wk = torch.ones(hd, device=dev)
tab = torch.zeros(L, hd // 2, 2, device=dev); ops.rope_table(tab, L, hd)
qkv = torch.zeros(M, 3 * dh, dtype=bf, device=dev); qk = torch.zeros(M, 2 * dh, dtype=bf, device=dev)
plain = "plain" in sys.argv
fn = (lambda: ops.gemm_nt(A, W, b, qkv)) if plain else (lambda: ops.gemm_nt_qkrope_split(A, W, b, qkv, qk, wq, wk, tab, L, H, hd, 1.2e-7, q_scale=0.18))
n = 30 if "time" in sys.argv else 4
for _ in range(2):
    fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(n):
    fn()
e1.record()
torch.cuda.synchronize()
if "time" in sys.argv:
    ms = e0.elapsed_time(e1) / n
    print(f"{'plain' if plain else 'qkrope'} lib {os.path.basename(_lib.loaded_path())}: {ms:.3f} ms per launch, {2.0 * M * 3 * dh * K / ms / 1e9:.0f} TF/s", flush=True)
