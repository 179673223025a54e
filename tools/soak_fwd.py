"""Run-to-run identity of od_flash_attn_fwd: N launches per shape on fresh-but-equal inputs, every output (o, lse) compared bit for bit with the first.
    python3 tools/soak_fwd.py [N]     (library from OSU_DREAMER_HIP_LIB)"""
import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops, _lib
dev, bf = torch.device("cuda:0"), torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for (B, H, L) in [(2, 2, 230), (3, 2, 150), (4, 16, 1115), (32, 16, 8192), (8, 16, 8191)]:
    hd = 64
    M, dh = B * L, H * hd
    g = torch.Generator(device=dev).manual_seed(1)
    qk = torch.randn(M, 2 * dh, device=dev, generator=g).to(bf)
    qk[:, :dh] = (qk[:, :dh].float() * math.log2(math.e) / 8).to(bf)
    v = torch.randn(M, dh, device=dev, generator=g).to(bf)
    outs = []
    differ = 0
    n = N if L < 4096 else max(10, N // 5)
    # a side stream keeps the chip unevenly busy (GEMMs of changing size), as tools/soak_fused.py does
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev, dtype=bf)
    for i in range(n):
        o = torch.full((M, dh), float("nan"), dtype=bf, device=dev)
        lse = torch.full((B, H, L), float("nan"), device=dev)
        if i % 3:
            with torch.cuda.stream(side):
                k = 512 * (1 + i % 5)
                torch.matmul(a[:k], a)
        ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], v, o, lse, B, H, L, hd, 0.125, q_prescaled=True)
        torch.cuda.synchronize()
        if i == 0:
            o0, l0 = o.clone(), lse.clone()
            assert not bool(torch.isnan(o0.float()).any()) and not bool(torch.isnan(l0).any())
        else:
            differ += int(not (torch.equal(o.view(torch.int16), o0.view(torch.int16)) and torch.equal(lse, l0)))
    print(f"{os.path.basename(_lib.loaded_path())}: B={B} H={H} L={L}: {n} launches, {differ} differ from the first", flush=True)
