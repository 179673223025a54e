#!/usr/bin/env python3
"""Per-kernel micro-benchmarks at the bench workload's shapes (B=32, L=8192, default hparams).
usage: python tools/microbench.py [attn|gemm|all] [--iters N]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_dreamer_amd import ops  # noqa: E402


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--L", type=int, default=8192)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, L, H, hd = a.B, a.L, 16, 64
    M, dh = B * L, H * hd
    bf = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
    if a.what in ("attn", "all"):
        qk, qkv = r(M, 2 * dh), r(M, 3 * dh)
        o, do = torch.zeros(M, dh, dtype=bf, device=dev), r(M, dh)
        lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
        dqk, dqkv = torch.zeros_like(qk), torch.zeros_like(qkv)
        sc = 1 / math.sqrt(hd)
        unit = 2.0 * B * H * L * L * hd
        t = timeit(lambda: ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, lse, B, H, L, hd, sc), a.iters)
        print(f"flash_fwd  {t:8.3f} ms  {2 * unit / t / 1e9:7.1f} TF/s")
        t = timeit(lambda: ops.flash_attn_bwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, do, lse, delta, dqk[:, :dh],
                                              dqk[:, dh:], dqkv[:, 2 * dh:], B, H, L, hd, sc), a.iters)
        print(f"flash_bwd  {t:8.3f} ms  {7 * unit / t / 1e9:7.1f} TF/s (7 passes)")
    if a.what in ("rope", "all"):
        qkv, qk = r(M, 3 * dh), torch.zeros(M, 2 * dh, dtype=bf, device=dev)
        wq, wk = torch.ones(hd, device=dev), torch.ones(hd, device=dev)
        tab = torch.zeros(L, hd // 2, 2, device=dev); ops.rope_table(tab, L, hd)
        eps = 1.2e-7
        t = timeit(lambda: ops.qk_norm_rope(qkv, wq, wk, tab, qk, B, L, H, hd, eps), a.iters)
        print(f"qk_norm_rope fwd {t:8.3f} ms  {M * 2 * dh * 2 * 2 / t / 1e6:7.1f} GB/s")
        dqk, dqkv2 = r(M, 2 * dh), torch.zeros(M, 3 * dh, dtype=bf, device=dev)
        dwq, dwk = torch.zeros(hd, device=dev), torch.zeros(hd, device=dev)
        t = timeit(lambda: ops.qk_norm_rope_bwd(qkv, wq, wk, tab, dqk, dqkv2, dwq, dwk, B, L, H, hd, eps), a.iters)
        print(f"qk_norm_rope bwd {t:8.3f} ms  {M * 2 * dh * 2 * 3 / t / 1e6:7.1f} GB/s")
    if a.what in ("gemm", "all"):
        for name, N, K in (("qkv", 3072, 512), ("out", 512, 1024), ("vg", 2816, 512), ("proj_o", 512, 1408), ("cl", 512, 128),
                           ("d_qkv", 512, 3072), ("d_vg", 512, 2816)):
            A, W, bias = r(M, K), r(N, K), torch.zeros(N, device=dev)
            C = torch.zeros(M, N, dtype=bf, device=dev)
            t = timeit(lambda: ops.gemm_nt(A, W, bias, C), a.iters)
            print(f"gemm_nt {name:7s} N={N:5d} K={K:5d} {t:8.3f} ms  {2.0 * M * N * K / t / 1e9:7.1f} TF/s")
        for name, N, K in (("qkv", 3072, 512), ("out", 512, 1024), ("vg_half", 1365, 512), ("proj_o", 512, 1365)):
            Gm, A = r(M, (N + 7) // 8 * 8), r(M, (K + 7) // 8 * 8)
            dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
            t = timeit(lambda: ops.gemm_tn(Gm, A, dW, n_cols=N, k_cols=K, dbias=db), a.iters)
            print(f"gemm_tn {name:7s} N={N:5d} K={K:5d} {t:8.3f} ms  {2.0 * M * N * K / t / 1e9:7.1f} TF/s")


if __name__ == "__main__":
    main()
