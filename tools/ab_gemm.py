"""A/B of GEMM kernel variants at the bench shapes (M = 262144, bf16): python tools/ab_gemm.py lib_a.so lib_b.so [--rounds=N]
Each library runs in its own process; every run also checks the NT result against torch.matmul on a row sub-sample."""
import os, subprocess, sys

def child():
    import torch
    sys.path.insert(0, os.getcwd())
    from osu_dreamer_amd import ops
    from tools.microbench import timeit
    dev = torch.device("cuda:0")
    M, bf = 32 * 8192, torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
    out = []
    for name, N, K in (("qkv", 3072, 512), ("out", 512, 1024), ("vg", 2816, 512), ("proj_o", 512, 1408), ("d_qkv", 512, 3072), ("d_vg", 512, 2816),
                       ("d_out", 1024, 512), ("d_proj_o", 1408, 512)):
        A, W, bias = r(M, K), (r(N, K).float() * 0.05).to(bf), torch.randn(N, device=dev, generator=g)
        C = torch.zeros(M, N, dtype=bf, device=dev)
        ops.gemm_nt(A, W, bias, C)
        rows = torch.arange(0, M, 4099, device=dev)
        ref = A[rows].float() @ W.float().t() + bias
        err = float((C[rows].float() - ref).norm() / ref.norm())
        t = timeit(lambda: ops.gemm_nt(A, W, bias, C), 10)
        out.append(f"{name} {2.0 * M * N * K / t / 1e9:6.0f}{'' if err < 5e-3 else ' WRONG(%.1e)' % err}")
    # weight gradients in the model's own layouts: the two halves of d_vg are column ranges of one [M, 2816] tensor, hh is [M, 1408]
    Hf, Hp = 1365, 1408
    dvg = r(M, 2 * Hp)
    for name, Gm, A, N, K in (("w_qkv", r(M, 3072), r(M, 512), 3072, 512), ("w_out", r(M, 512), r(M, 1024), 512, 1024),
                              ("w_vg", dvg[:, Hp:], r(M, 512), Hf, 512), ("w_proj_o", r(M, 512), r(M, Hp), 512, Hf)):
        dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        ops.gemm_tn(Gm, A, dW, n_cols=N, k_cols=K, dbias=db)
        ref = Gm[:, :N].float().t() @ A[:, :K].float()
        err = float((dW - ref).norm() / ref.norm())
        bref = Gm[:, :N].float().sum(0)
        errb = float((db - bref).norm() / bref.norm())
        del ref
        t = timeit(lambda: ops.gemm_tn(Gm, A, dW, n_cols=N, k_cols=K, dbias=db), 10)
        out.append(f"{name} {2.0 * M * N * K / t / 1e9:6.0f}{'' if max(err, errb) < 3e-3 else ' WRONG(%.1e,%.1e)' % (err, errb)}")
    print("TF/s: " + " | ".join(out), flush=True)

if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("AB_CHILD"):
        child()
    else:
        libs = [a for a in sys.argv[1:] if not a.startswith("--")] or [os.path.join("osu_dreamer_amd", "libosudreamer_hip.so")]
        rounds = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--rounds=")), 2))
        for rd in range(rounds):
            for lib in libs:
                from ab_common import parse
                label, libpath, extra = parse(lib)
                env = dict(os.environ, AB_CHILD="1", OSU_DREAMER_HIP_LIB=libpath, **extra)
                o = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=900)
                line = [l for l in o.stdout.splitlines() if l.startswith("TF/s")]
                print(f"[round {rd}] {label:44s} " + (line[0] if line else "FAILED: " + o.stderr[-400:]), flush=True)
