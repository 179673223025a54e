"""Sustained-rate A/B of NT GEMM variants: every (variant, shape) runs back to back for ~1.2 s (the board sits at its power limit in these
GEMMs, and a 10-launch timing sees a different clock than a training step does).
  python tools/ab_sustained.py spec1 spec2 ... [--shapes=d_qkv,out,...]     spec = lib.so@ENV=VAL,... | @ENV=VAL | vendor"""
import os, subprocess, sys

SHAPES = {"qkv": (3072, 512), "out": (512, 1024), "vg": (2816, 512), "proj_o": (512, 1408), "d_qkv": (512, 3072), "d_vg": (512, 2816),
          "d_out": (1024, 512), "d_proj_o": (1408, 512)}


def child(names, vendor):
    import torch
    sys.path.insert(0, os.getcwd())
    from osu_dreamer_amd import ops
    dev, bf, M = torch.device("cuda:0"), torch.bfloat16, 32 * 8192
    g = torch.Generator(device=dev).manual_seed(0)
    out = []
    for name in names:
        N, K = SHAPES[name]
        A = torch.randn(M, K, device=dev, generator=g).to(bf)
        W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(bf)
        bias = torch.randn(N, device=dev, generator=g)
        C = torch.zeros(M, N, dtype=bf, device=dev)
        run = (lambda: torch.matmul(A, W.t(), out=C)) if vendor else (lambda: ops.gemm_nt(A, W, bias, C))
        for _ in range(3):
            run()
        rows = torch.arange(0, M, 4099, device=dev)
        ref = A[rows].float() @ W.float().t() + (0 if vendor else bias)
        err = float((C[rows].float() - ref).norm() / ref.norm())
        n = max(50, int(1.2e-3 * 1.0e15 / (2.0 * M * N * K) * 1000))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        out.append(f"{name} {2.0 * M * N * K / ms / 1e9:5.0f}{'' if err < 5e-3 else ' WRONG(%.1e)' % err}")
        del A, W, C
    print("TF/s: " + " | ".join(out), flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("AB_CHILD"):
        child(os.environ["AB_SHAPES"].split(","), os.environ.get("AB_VENDOR") == "1")
    else:
        from ab_common import parse
        specs = [a for a in sys.argv[1:] if not a.startswith("--")]
        shapes = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--shapes=")), ",".join(SHAPES))
        rounds = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--rounds=")), 1))
        for rd in range(rounds):
            for spec in specs:
                if spec == "vendor":
                    label, env = "vendor (torch.matmul)", dict(os.environ, AB_CHILD="1", AB_VENDOR="1", AB_SHAPES=shapes)
                else:
                    label, libpath, extra = parse(spec)
                    env = dict(os.environ, AB_CHILD="1", OSU_DREAMER_HIP_LIB=libpath, AB_SHAPES=shapes, **extra)
                o = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=900)
                line = [l for l in o.stdout.splitlines() if l.startswith("TF/s")]
                print(f"[{rd}] {label:40s} " + (line[0] if line else "FAILED: " + o.stderr[-400:]), flush=True)
