"""Phase timeline of one wave of the attention forward (debug build with -DOD_FWD32_TRACE=<block id>): shader-clock stamps around
DMA issue | score MFMAs | exp + sums (+ cvt) | PV MFMAs | barrier for tiles 32..39.  usage: OSU_DREAMER_HIP_LIB=... python tools/trace_fwd.py"""
import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
dev = torch.device("cuda:0")
B, L, H, hd = 32, 8192, 16, 64
M, dh = B * L, H * hd
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
qk, qkv = r(M, 2 * dh), r(M, 3 * dh)
qk[:, :dh] = (qk[:, :dh].float() * (math.log2(math.e) / math.sqrt(hd))).to(bf)
o = torch.zeros(M, dh, dtype=bf, device=dev)
lse_full = torch.zeros(B * H * L + 1024, device=dev)
lse = lse_full[: B * H * L].view(B, H, L)
for _ in range(2):
    ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, lse, B, H, L, hd, 1 / math.sqrt(hd), q_prescaled=True)
torch.cuda.synchronize()
t = lse_full[B * H * L:].view(torch.int64)[: 64].view(8, 8).cpu()
names = ["dma issue", "scores (8 K reads + 8 mfma)", "exp+sum (+guard)", "cvt + PV (16 tr reads + 8 mfma)", "barrier"]
print("tile   " + "  ".join(f"{n:>30s}" for n in names) + "   total")
for i in range(8):
    d = [int(t[i, j + 1] - t[i, j]) for j in range(5)]
    nxt = int(t[i + 1, 0] - t[i, 0]) if i < 7 else 0
    print(f"{32 + i:4d}   " + "  ".join(f"{x:30d}" for x in d) + f"   {nxt}")
