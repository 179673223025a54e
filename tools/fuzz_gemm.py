"""Random-shape check of the large-M GEMM paths on the GPU against torch (bf16): od_gemm_nt (plain / SiLU / bias / no bias), od_gemm_tn (+ bias
gradient, + the block row map).  python tools/fuzz_gemm.py [n_shapes] [seed]"""
import os, random, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
dev, bf = torch.device("cuda:0"), torch.bfloat16
n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 40), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rnd = random.Random(seed)
g = torch.Generator(device=dev).manual_seed(seed)
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
bad = 0
for it in range(n):
    M = rnd.choice([32768, 32769, 40000, 65536, 70001, 131072]) if it % 3 else rnd.randrange(256, 3000)
    N = 8 * rnd.randrange(32, 400)
    K = rnd.choice([128, 256, 384, 512, 640, 1024, 1408, 8 * rnd.randrange(16, 200)])
    A = torch.randn(M, K, device=dev, generator=g).to(bf); W = (torch.randn(N, K, device=dev, generator=g) * 0.1).to(bf)
    bias = torch.randn(N, device=dev, generator=g) if rnd.random() < 0.6 else None
    C = torch.full((M, N), float("nan"), dtype=bf, device=dev)
    silu = rnd.random() < 0.3
    ops.gemm_nt(A, W, bias, C, epilogue=ops.OD_EPI_SILU if silu else ops.OD_EPI_NONE)
    ref = A.float() @ W.float().t() + (bias if bias is not None else 0)
    if silu: ref = torch.nn.functional.silu(ref)
    e1 = rel(C.float(), ref)
    # weight gradient: G = C-like [M, N], A [M, K]
    G = torch.randn(M, N, device=dev, generator=g).to(bf)
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    ops.gemm_tn(G, A, dW, dbias=db)
    e2, e3 = rel(dW, G.float().t() @ A.float()), rel(db, G.float().sum(0))
    ok = e1 < 2e-2 and e2 < 2e-2 and e3 < 2e-2 and not torch.isnan(C.float()).any()
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} M={M} N={N} K={K} bias={bias is not None} silu={silu}: nt {e1:.1e} tn {e2:.1e} dbias {e3:.1e}", flush=True)
    del A, W, C, G, dW, ref
print(f"{bad} bad of {n}")
sys.exit(1 if bad else 0)
