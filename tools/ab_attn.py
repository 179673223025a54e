"""A/B of attention kernel variants at the bench shape (B=32, L=8192, 16 heads x 64, bf16, pre-multiplied q):
    python tools/ab_attn.py lib_a.so lib_b.so ...   (default: the in-tree library)
Each library is measured in its own process (one library per process), `--rounds` times round-robin; each run also checks
forward / backward against a dense fp32 torch reference at L = 1000 (ragged) so a fast-but-wrong variant is flagged."""
import math, os, subprocess, sys

def child():
    import torch
    sys.path.insert(0, os.getcwd())
    from osu_dreamer_amd import ops
    from tools.microbench import timeit
    dev = torch.device("cuda:0")
    H, hd = 16, 64
    dh = H * hd
    bf = torch.bfloat16
    qs = math.log2(math.e) / math.sqrt(hd)
    sc = 1 / math.sqrt(hd)
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
    # ---- correctness at a ragged length
    B, L = 2, 1000
    M = B * L
    qk, qkv, do = r(M, 2 * dh), r(M, 3 * dh), r(M, dh)
    qk[:, :dh] = (qk[:, :dh].float() * qs).to(bf)
    o = torch.zeros(M, dh, dtype=bf, device=dev)
    lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
    dqk, dqkv = torch.zeros_like(qk), torch.zeros_like(qkv)
    ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, lse, B, H, L, hd, sc, q_prescaled=True)
    ops.flash_attn_bwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, do, lse, delta, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:], B, H, L, hd, sc, q_prescaled=True)
    hd_ = lambda t: t.float().reshape(B, L, H, hd).permute(0, 2, 1, 3)
    qr, kr, vr = (hd_(t).clone().requires_grad_() for t in (qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:]))
    s = qr @ kr.transpose(-1, -2) * math.log(2.0)           # q carries scale * log2(e)
    ref = torch.softmax(s, -1) @ vr
    ref.backward(hd_(do))
    rel = lambda a, b: float((a.float() - b).norm() / b.norm())
    errs = dict(o=rel(hd_(o), ref), lse=rel(lse, torch.logsumexp(s, -1)), dq=rel(hd_(dqk[:, :dh]), qr.grad), dk=rel(hd_(dqk[:, dh:]), kr.grad),
                dv=rel(hd_(dqkv[:, 2 * dh:]), vr.grad))
    # ---- timing at the bench shape
    B, L = 32, 8192
    M = B * L
    qk, qkv, do = r(M, 2 * dh), r(M, 3 * dh), r(M, dh)
    qk[:, :dh] = (qk[:, :dh].float() * qs).to(bf)
    o = torch.zeros(M, dh, dtype=bf, device=dev)
    lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
    dqk, dqkv = torch.zeros_like(qk), torch.zeros_like(qkv)
    unit = 2.0 * B * H * L * L * hd
    tf = timeit(lambda: ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, lse, B, H, L, hd, sc, q_prescaled=True), 5)
    tb = timeit(lambda: ops.flash_attn_bwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, do, lse, delta, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:], B, H, L, hd, sc, q_prescaled=True), 5)
    print(f"fwd {tf:7.3f} ms {2 * unit / tf / 1e9:7.1f} TF/s | bwd {tb:7.3f} ms {5 * unit / tb / 1e9:7.1f} TF/s (5-pass algorithmic) | err " +
          " ".join(f"{k}={v:.1e}" for k, v in errs.items()), flush=True)

if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("AB_CHILD"):
        child()
    else:
        args = [a for a in sys.argv[1:] if not a.startswith("--")]
        rounds = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--rounds=")), 2))
        libs = args or [os.path.join("osu_dreamer_amd", "libosudreamer_hip.so")]
        for rd in range(rounds):
            for lib in libs:
                from ab_common import parse
                label, libpath, extra = parse(lib)
                env = dict(os.environ, AB_CHILD="1", OSU_DREAMER_HIP_LIB=libpath, **extra)
                out = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=600)
                line = [l for l in out.stdout.splitlines() if l.startswith("fwd")]
                print(f"[round {rd}] {label:44s} " + (line[0] if line else "FAILED: " + out.stderr[-400:]), flush=True)
