import os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
dev, bf, M = torch.device("cuda:0"), torch.bfloat16, 32 * 8192
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
Hf, Hp = 1365, 1408
h1, dqkv, y, dbr, dvg, hdw, hh = r(M, 512), r(M, 3072), r(M, 1024), r(M, 512), r(M, 2 * Hp), r(M, 512), r(M, Hp)
shapes = [("w_qkv", dqkv, h1, 3072, 512), ("w_out", dbr, y, 512, 1024), ("w_vg_v", dvg[:, :Hp], hdw, Hf, 512), ("w_proj_o", dbr, hh, 512, Hf),
          ("w_vg_merged", dvg, hdw, 2 * Hp, 512)]
out = []
for name, G, A, N, K in shapes:
    merged = name == "w_vg_merged"
    rows = 2 * Hf if merged else N
    kw = dict(n_block=Hp, n_valid=Hf) if merged else {}
    dW, db = torch.zeros(rows, K, device=dev), torch.zeros(rows, device=dev)
    for _ in range(3): ops.gemm_tn(G, A, dW, n_cols=N, k_cols=K, dbias=db, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n): ops.gemm_tn(G, A, dW, n_cols=N, k_cols=K, dbias=db, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    out.append(f"{name} {ms*1e3:.0f} us {2.0*M*rows*K/ms/1e9:.0f} TF/s")
print(os.path.basename(os.environ.get("OSU_DREAMER_HIP_LIB", "libosudreamer_hip.so")) + ": " + " | ".join(out), flush=True)
