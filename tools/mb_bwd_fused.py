"""od_flash_attn_bwd_fused against od_flash_attn_bwd at the bench shape (B=32, L=8192, 16 heads x 64, bf16, pre-multiplied q): time per call,
agreement of the two results, run-to-run determinism of the fused kernel, and the workspace's status word.
    python tools/mb_bwd_fused.py [B L]"""
import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
from tools.microbench import timeit
dev = torch.device("cuda:0")
B, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 8192)
H, hd = 16, 64
M, dh = B * L, H * hd
bf = torch.bfloat16
qs = math.log2(math.e) / math.sqrt(hd)
sc = 1 / math.sqrt(hd)
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
qk, qkv, do = r(M, 2 * dh), r(M, 3 * dh), r(M, dh)
qk[:, :dh] = (qk[:, :dh].float() * qs).to(bf)
o = torch.zeros(M, dh, dtype=bf, device=dev)
lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
q, k, v = qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:]
ops.flash_attn_fwd(q, k, v, o, lse, B, H, L, hd, sc, q_prescaled=True)
dqk, dqkv = torch.zeros_like(qk), torch.zeros_like(qkv)
dqk2, dqkv2 = torch.zeros_like(qk), torch.zeros_like(qkv)
aux = ops.AttnAux()
ws = ops.FusedAttnBwdWorkspace(B, H, L, dev)
f7 = lambda: ops.flash_attn_bwd(q, k, v, o, do, lse, delta, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:], B, H, L, hd, sc, q_prescaled=True, aux=aux)
f5 = lambda: ops.flash_attn_bwd_fused(q, k, v, o, do, lse, dqk2[:, :dh], dqk2[:, dh:], dqkv2[:, 2 * dh:], B, H, L, hd, sc, ws, q_prescaled=True)
f7(); f5(); torch.cuda.synchronize()
rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
print(f"status {ws.status()}  fused vs 7-pass: dq {rel(dqk2[:, :dh], dqk[:, :dh]):.2e} dk {rel(dqk2[:, dh:], dqk[:, dh:]):.2e} dv {rel(dqkv2[:, 2 * dh:], dqkv[:, 2 * dh:]):.2e}", flush=True)
first = dqk2.clone()
same = True
for _ in range(3):
    dqk2.zero_(); f5(); torch.cuda.synchronize()
    same &= bool(torch.equal(dqk2, first))
print(f"fused run-to-run bit-identical: {same}  status {ws.status()}", flush=True)
unit = 2.0 * B * H * L * L * hd
for rd in range(2):
    t7 = timeit(f7, 5)
    t5 = timeit(f5, 5)
    print(f"[{rd}] 7-pass {t7:7.3f} ms {5 * unit / t7 / 1e9:7.1f} TF/s | fused {t5:7.3f} ms {5 * unit / t5 / 1e9:7.1f} TF/s (5-pass algorithmic)", flush=True)
