"""One process, ONE launch of od_gemm_tn per weight-gradient shape of the bench step (M = 262144, bf16, the model's own layouts: the two
halves of d_vg are column ranges of one [M, 2816] tensor, hh is [M, 1408] with 1365 live columns): the target of
`rocprofv3 --pmc FETCH_SIZE ... -- python3 tools/mb_tn_all.py`.  Launch order = the order printed by tools/rocpd_pmc_dispatch.py."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
dev, bf, M = torch.device("cuda:0"), torch.bfloat16, 32 * 8192
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
Hf, Hp = 1365, 1408
h1, dqkv, y, dbr, dvg, hdw, hh = r(M, 512), r(M, 3072), r(M, 1024), r(M, 512), r(M, 2 * Hp), r(M, 512), r(M, Hp)
shapes = [("w_qkv", dqkv, h1, 3072, 512), ("w_out", dbr, y, 512, 1024), ("w_vg_v", dvg[:, :Hp], hdw, Hf, 512), ("w_vg_g", dvg[:, Hp:], hdw, Hf, 512),
          ("w_proj_o", dbr, hh, 512, Hf)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for name, G, A, N, K in shapes:
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    for _ in range(reps):
        ops.gemm_tn(G, A, dW, n_cols=N, k_cols=K, dbias=db)
    torch.cuda.synchronize()
    alg = (M * N + M * K) * 2 / 1e6
    print(f"{name}: N={N} K={K} algorithmic read {alg:.0f} MB", flush=True)
