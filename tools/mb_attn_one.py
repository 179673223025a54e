"""One process, a few launches of the attention forward + backward at the bench shape (B=32, L=8192, 16 x 64, bf16,
pre-multiplied q): the target of `rocprofv3 --pmc ... -- python3 tools/mb_attn_one.py [iters] [B] [L]`."""
import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
L = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
dev = torch.device("cuda:0")
H, hd = 16, 64
M, dh = B * L, H * hd
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
qk, qkv, do = r(M, 2 * dh), r(M, 3 * dh), r(M, dh)
qk[:, :dh] = (qk[:, :dh].float() * (math.log2(math.e) / math.sqrt(hd))).to(bf)
o = torch.zeros(M, dh, dtype=bf, device=dev)
lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
dqk, dqkv = torch.zeros_like(qk), torch.zeros_like(qkv)
sc = 1 / math.sqrt(hd)
for _ in range(iters):
    ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, lse, B, H, L, hd, sc, q_prescaled=True)
    ops.flash_attn_bwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, do, lse, delta, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:], B, H, L, hd, sc, q_prescaled=True)
torch.cuda.synchronize()
print("done")
