# experiment: do TN GEMMs overlap usefully with attention backward when issued on a second stream?
import math, os, sys, time, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
dev = torch.device("cuda:0")
B, L, H, hd = 32, 8192, 16, 64
M, dh = B * L, H * hd
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
qk, qkv = r(M, 2 * dh), r(M, 3 * dh)
o, do = torch.zeros(M, dh, dtype=bf, device=dev), r(M, dh)
lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
dqk, dqkv = torch.zeros_like(qk), torch.zeros_like(qkv)
sc = 1 / math.sqrt(hd)
ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, lse, B, H, L, hd, sc)
G1, A1 = r(M, 3072), r(M, 512)
G2, A2 = r(M, 2816), r(M, 512)
dW1, dW2 = torch.zeros(3072, 512, device=dev), torch.zeros(2816, 512, device=dev)
x, h = r(M, 512), r(M, 512)
def attn_bwd():
    ops.flash_attn_bwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, do, lse, delta, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:], B, H, L, hd, sc)
def gemms():
    ops.gemm_tn(G1, A1, dW1); ops.gemm_tn(G2, A2, dW2)
side = torch.cuda.Stream()
def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
def serial():
    attn_bwd(); gemms()
def overlapped():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gemms()
    attn_bwd()
    torch.cuda.current_stream().wait_stream(side)
print("attn_bwd alone %.2f ms, gemms alone %.2f ms" % (timed(attn_bwd), timed(gemms)))
print("serial %.2f ms, overlapped %.2f ms" % (timed(serial), timed(overlapped)))
