"""The attention forward or backward at the bench shape, a few launches, for counter runs: python3 tools/mb_attn_one.py fwd|bwd [n]"""
import math, os, sys
import torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops


def setup(dev, B=32, L=8192, H=16, hd=64):
    dh, bf = H * hd, torch.bfloat16
    qs, sc = math.log2(math.e) / math.sqrt(hd), 1 / math.sqrt(hd)
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
    M = B * L
    qk, qkv, do = r(M, 2 * dh), r(M, 3 * dh), r(M, dh)
    qk[:, :dh] = (qk[:, :dh].float() * qs).to(bf)
    o = torch.zeros(M, dh, dtype=bf, device=dev)
    lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
    dqk, dqkv = torch.zeros_like(qk), torch.zeros_like(qkv)
    aux = ops.AttnAux() if os.environ.get("OD_BWD_2STREAM", "1") != "0" and hasattr(ops, "AttnAux") else None
    fwd = lambda: ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, lse, B, H, L, hd, sc, q_prescaled=True)
    bwd = lambda: ops.flash_attn_bwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, do, lse, delta, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:],
                                     B, H, L, hd, sc, q_prescaled=True, aux=aux)
    fwd()
    return fwd, bwd, 2.0 * B * H * L * L * hd


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    fwd, bwd, unit = setup(dev)
    run = fwd if sys.argv[1] == "fwd" else bwd
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
        run()
    torch.cuda.synchronize()
