"""Time od_flash_attn_bwd_fused of several library variants at the bench shape (one process per library): python tools/ab_bwd_fused.py lib.so ..."""
import math, os, subprocess, sys

def child():
    import torch
    sys.path.insert(0, os.getcwd())
    from osu_dreamer_amd import ops
    from tools.microbench import timeit
    dev = torch.device("cuda:0")
    B, L, H, hd = 32, 8192, 16, 64
    M, dh = B * L, H * hd
    bf = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
    qk, qkv, do = r(M, 2 * dh), r(M, 3 * dh), r(M, dh)
    qk[:, :dh] = (qk[:, :dh].float() * math.log2(math.e) / 8).to(bf)
    o = torch.zeros(M, dh, dtype=bf, device=dev)
    lse = torch.zeros(B, H, L, device=dev)
    q, k, v = qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:]
    ops.flash_attn_fwd(q, k, v, o, lse, B, H, L, hd, 0.125, q_prescaled=True)
    dqk, dqkv = torch.zeros_like(qk), torch.zeros_like(qkv)
    ws = ops.FusedAttnBwdWorkspace(B, H, L, dev)
    f5 = lambda: ops.flash_attn_bwd_fused(q, k, v, o, do, lse, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:], B, H, L, hd, 0.125, ws, q_prescaled=True)
    t = timeit(f5, 5)
    import ctypes
    from osu_dreamer_amd import _lib
    arr = (ctypes.c_long * 16)()
    _lib.lib().od_flash_attn_bwd_fused_prof(ops._p(ws.buf), arr)
    f5(); torch.cuda.synchronize()
    _lib.lib().od_flash_attn_bwd_fused_prof(ops._p(ws.buf), arr)
    pr = list(arr)
    if any(pr):
        n = 256 * 1.0
        names = ["key:barrier", "key:loop", "q:issue", "q:phase_b", "q:wait", "q:chain", "q:barrier", "q:retries"]
        print("prof (Mcycles per CU-wave; retries = count): " + "  ".join(f"{nm}={v / n / 1e6:.3f}" if nm != "q:retries" else f"{nm}={v}" for nm, v in zip(names, pr)), flush=True)
    print(f"fused {t:7.3f} ms {5 * 2.0 * B * H * L * L * hd / t / 1e9:7.1f} TF/s (5-pass algorithmic) status {ws.status()}", flush=True)

if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("AB_CHILD"):
        child()
    else:
        from ab_common import parse
        args = [a for a in sys.argv[1:] if not a.startswith("--")]
        rounds = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--rounds=")), 2))
        for rd in range(rounds):
            for lib in args or [""]:
                label, libpath, extra = parse(lib)
                env = dict(os.environ, AB_CHILD="1", OSU_DREAMER_HIP_LIB=libpath, **extra)
                out = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=int(os.environ.get("AB_TIMEOUT", "120")))
                line = [l for l in out.stdout.splitlines() if l.startswith("fused")]
                for l in out.stdout.splitlines():
                    if l.startswith("prof"): print("      " + l)
                print(f"[round {rd}] {label:40s} " + (line[0] if line else "FAILED: " + out.stderr[-400:]), flush=True)
