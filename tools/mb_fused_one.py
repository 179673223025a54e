"""One process, N calls of od_flash_attn_bwd_fused at the bench shape (for rocprofv3 --pmc / --kernel-trace runs): python3 tools/mb_fused_one.py [N]"""
import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
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
for _ in range(n):
    ops.flash_attn_bwd_fused(q, k, v, o, do, lse, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:], B, H, L, hd, 0.125, ws, q_prescaled=True)
torch.cuda.synchronize()
print("status", ws.status())
