"""Soak of od_flash_attn_bwd_fused: N launches on one workspace at the bench shape while a side stream keeps the chip unevenly busy (GEMMs of
varying size); every launch's dq / dk / dv must be bit-identical to the first and the workspace's status word 0.
    python tools/soak_fused.py [N] [B L]"""
import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, L = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 8192)
H, hd = 16, 64
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
ws = ops.FusedAttnBwdWorkspace(B, H, L, dev)
out = [torch.zeros(M, 2 * dh, dtype=bf, device=dev), torch.zeros(M, 3 * dh, dtype=bf, device=dev)]
ref = None
side = torch.cuda.Stream()
noise_a = [torch.randn(s, s, device=dev, dtype=bf) for s in (512, 1024, 2048, 4096)]
bad = 0
for i in range(n):
    out[0].fill_(float("nan")); out[1][:, 2 * dh:].fill_(float("nan"))
    with torch.cuda.stream(side):                      # uneven load: a few GEMMs of a size that changes from launch to launch
        a = noise_a[i % 4]
        for _ in range(1 + i % 3):
            a @ a
    ops.flash_attn_bwd_fused(q, k, v, o, do, lse, out[0][:, :dh], out[0][:, dh:], out[1][:, 2 * dh:], B, H, L, hd, 0.125, ws, q_prescaled=True)
    cur = (out[0].clone(), out[1][:, 2 * dh:].clone())
    if ref is None:
        ref = cur
        assert not torch.isnan(ref[0].float()).any() and not torch.isnan(ref[1].float()).any()
    elif not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])):
        bad += 1
        d = (cur[0].float() - ref[0].float()).abs()
        print(f"launch {i}: differs from the first in {int((d > 0).sum())} elements of dq/dk (max {float(d.max()):.3e})")
torch.cuda.synchronize()
print(f"{n} launches at B={B} L={L}: {bad} differ from the first; status {ws.status()}")
sys.exit(1 if bad or ws.status() else 0)
