"""Random-shape check of the attention core on the GPU against dense fp32 autograd: od_flash_attn_fwd + od_flash_attn_bwd_fused with bf16 and with
IEEE-half operands (q pre-scaled or not, ragged lengths, (batch, head) counts around the XCD queues), two launches per workspace.
python tools/fuzz_attn.py [n] [seed]"""
import math, os, random, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
dev, bf, f16 = torch.device("cuda:0"), torch.bfloat16, torch.float16
n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 30), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rnd = random.Random(seed)
g = torch.Generator(device=dev).manual_seed(seed)
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
hd, scale = 64, 0.125
bad = 0
for it in range(n):
    B, H = rnd.randrange(1, 5), rnd.randrange(1, 20)
    L = rnd.choice([64, 65, 191, 192, 193, 384, 777, 1000, 2048, 2049, 3000, 4096])
    half, pre = rnd.random() < 0.5, rnd.random() < 0.5
    dt = f16 if half else bf
    M, dh = B * L, H * hd
    c = scale * math.log2(math.e)
    q = (torch.randn(M, dh, device=dev, generator=g) * (c if pre else 1.0)).to(dt)
    k = torch.randn(M, dh, device=dev, generator=g).to(dt)
    v = (torch.randn(M, dh, device=dev, generator=g) * 2).to(dt)
    do = (torch.randn(M, dh, device=dev, generator=g) * rnd.choice([1.0, 1e-3, 1e-6])).to(bf)
    o = torch.zeros(M, dh, dtype=bf, device=dev); lse = torch.zeros(B, H, L, device=dev)
    ops.flash_attn_fwd(q, k, v, o, lse, B, H, L, hd, scale, q_prescaled=pre)
    heads = lambda t: t.reshape(B, L, H, hd).permute(0, 2, 1, 3)
    qr, kr, vr = (t.float().clone().requires_grad_() for t in (q, k, v))
    s = heads(qr) @ heads(kr).transpose(-1, -2) * (math.log(2.0) if pre else scale)
    ref = (torch.softmax(s, -1) @ heads(vr)).permute(0, 2, 1, 3).reshape(M, dh)
    ref.backward(do.float())
    ws = ops.FusedAttnBwdWorkspace(B, H, L, dev, dt)
    errs = []
    for rep in range(2):
        dq, dk, dv = (torch.full((M, dh), float("nan"), dtype=bf, device=dev) for _ in range(3))
        ops.flash_attn_bwd_fused(q, k, v, o, do, lse, dq, dk, dv, B, H, L, hd, scale, ws, q_prescaled=pre)
        errs.append((rel(dq.float(), qr.grad), rel(dk.float(), kr.grad), rel(dv.float(), vr.grad)))
    eo = rel(o.float(), ref)
    tol = 1.2e-2 if half else 3e-2
    ok = ws.status() == 0 and eo < 6e-3 and all(e < tol for t in errs for e in t) and errs[0] == errs[1]
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} B={B} H={H} L={L} {'f16' if half else 'bf16'} pre={pre}: o {eo:.1e} dq/dk/dv {errs[0][0]:.1e} {errs[0][1]:.1e} {errs[0][2]:.1e} repeat identical {errs[0] == errs[1]}", flush=True)
    del s, ref, qr, kr, vr
print(f"{bad} bad of {n}")
sys.exit(1 if bad else 0)
