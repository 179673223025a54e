"""Reference point: torch's own F.scaled_dot_product_attention (what the reference's attn.py:82 would run on this GPU) at
the bench shape, bf16, forward and forward+backward, next to od_flash_attn_fwd / _bwd.  usage: python tools/mb_torch_sdpa.py"""
import math, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import timeit
from osu_dreamer_amd import ops
dev = torch.device("cuda:0")
B, H, L, hd = 32, 16, 8192, 64
bf = torch.bfloat16
q, k, v = (torch.randn(B, H, L, hd, device=dev, dtype=bf, requires_grad=True) for _ in range(3))
do = torch.randn(B, H, L, hd, device=dev, dtype=bf)
unit = 2.0 * B * H * L * L * hd
def fwd():
    with torch.no_grad():
        return F.scaled_dot_product_attention(q, k, v)
t_f = timeit(fwd, 3)
def fb():
    o = F.scaled_dot_product_attention(q, k, v)
    o.backward(do)
    q.grad = k.grad = v.grad = None
t_fb = timeit(fb, 3)
print(f"torch SDPA    fwd {t_f:8.3f} ms {2 * unit / t_f / 1e9:7.1f} TF/s | fwd+bwd {t_fb:8.3f} ms | bwd ~{t_fb - t_f:8.3f} ms")
M, dh = B * L, H * hd
qk, qkv = torch.randn(M, 2 * dh, device=dev).to(bf), torch.randn(M, 3 * dh, device=dev).to(bf)
o, dO = torch.zeros(M, dh, dtype=bf, device=dev), torch.randn(M, dh, device=dev).to(bf)
lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
dqk, dqkv = torch.zeros_like(qk), torch.zeros_like(qkv)
sc = 1 / math.sqrt(hd)
t1 = timeit(lambda: ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, lse, B, H, L, hd, sc, q_prescaled=True), 3)
t2 = timeit(lambda: ops.flash_attn_bwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, dO, lse, delta, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:], B, H, L, hd, sc, q_prescaled=True), 3)
print(f"od_flash_attn fwd {t1:8.3f} ms {2 * unit / t1 / 1e9:7.1f} TF/s | fwd+bwd {t1 + t2:8.3f} ms | bwd  {t2:8.3f} ms")
