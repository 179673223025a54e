import math, os, sys, torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops
dev = torch.device("cuda:0")
B, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 8192)
H, hd = 16, 64
M, dh = B * L, H * hd
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
qk, qkv, do = r(M, 2 * dh), r(M, 3 * dh), r(M, dh)
qk[:, :dh] = (qk[:, :dh].float() * math.log2(math.e) / 8).to(bf)
o = torch.zeros(M, dh, dtype=bf, device=dev)
lse, delta = torch.zeros(B, H, L, device=dev), torch.zeros(B, H, L, device=dev)
q, k, v = qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:]
ops.flash_attn_fwd(q, k, v, o, lse, B, H, L, hd, 0.125, q_prescaled=True)
dqk, dqkv = torch.zeros_like(qk), torch.zeros_like(qkv)
ops.flash_attn_bwd(q, k, v, o, do, lse, delta, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:], B, H, L, hd, 0.125, q_prescaled=True)
ref = dqk[:, :dh].float().reshape(B, L // 64, 64, H, hd)
ws = ops.FusedAttnBwdWorkspace(B, H, L, dev)
for rep in range(3):
    d2, dv2 = torch.zeros_like(qk), torch.zeros_like(qkv)
    ops.flash_attn_bwd_fused(q, k, v, o, do, lse, d2[:, :dh], d2[:, dh:], dv2[:, 2 * dh:], B, H, L, hd, 0.125, ws, q_prescaled=True)
    torch.cuda.synchronize()
    x = d2[:, :dh].float().reshape(B, L // 64, 64, H, hd)
    err = (x - ref).pow(2).sum((2, 4)).sqrt() / ref.pow(2).sum((2, 4)).sqrt()      # [B, tiles, H]
    bad = (err > 1e-2).nonzero()
    print(f"rep {rep}: tiles with rel err > 1e-2: {bad.shape[0]} of {err.numel()}; max {float(err.max()):.3e}; median {float(err.median()):.2e}")
    for i in bad[:12].tolist():
        b_, t_, h_ = i
        e16 = (x[b_, t_, :, h_] - ref[b_, t_, :, h_]).pow(2).sum(0).reshape(4, 16).sum(1).sqrt()   # per 16-feature slice (helper wave)
        print("   b %d tile %d h %d (bh %d, xcd %d) err %.3e per-slice %s" % (b_, t_, h_, b_ * H + h_, (b_ * H + h_) % 8, float(err[b_, t_, h_]), [f"{float(v):.2e}" for v in e16]))
    # which key block's share explains the error of the first bad tile?
    if bad.shape[0]:
        b_, t_, h_ = bad[0].tolist()
        hs = slice(h_ * hd, (h_ + 1) * hd)
        rows = slice(b_ * L, (b_ + 1) * L)
        qf, kf, vf, dof, of = (t[rows, hs].float() for t in (q, k, v, do, o))
        S = qf @ kf.T * math.log(2.0)
        P = torch.softmax(S, -1)
        dP = dof @ vf.T
        dl = (dof * of).sum(-1, keepdim=True)
        dS = P * (dP - dl)                                        # [L, L]
        e = (x[b_, t_, :, h_] - ref[b_, t_, :, h_])              # [64, 64]
        nkb_ = (L + 191) // 192
        sl = int(torch.argmax((e.pow(2).reshape(64, 4, 16).sum((0, 2)))))            # worst 16-feature slice
        cols = slice(sl * 16, sl * 16 + 16)
        shares = torch.stack([dS[t_ * 64:(t_ + 1) * 64, s * 192:min(L, (s + 1) * 192)] @ kf[s * 192:min(L, (s + 1) * 192)] * math.log(2.0) for s in range(nkb_)])   # [nkb, 64, 64]
        A = shares[:, :, cols].reshape(nkb_, -1).T                                    # [1024, nkb]
        y = e[:, cols].reshape(-1, 1)
        sol = torch.linalg.lstsq(A, y).solution[:, 0]
        res = float((A @ sol[:, None] - y).norm() / y.norm())
        big = [(i, round(float(v), 2)) for i, v in enumerate(sol.tolist()) if abs(v) > 0.2]
        print(f"   slice {sl}: lstsq over the {nkb_} shares of this tile: residual {res:.2f}, |coef| > 0.2: {big}")
        # the same against the shares of the neighbouring tiles
        for tt in (t_ - 2, t_ - 1, t_ + 1):
            if 0 <= tt < L // 64:
                sh2 = torch.stack([dS[tt * 64:(tt + 1) * 64, s * 192:min(L, (s + 1) * 192)] @ kf[s * 192:min(L, (s + 1) * 192)] * math.log(2.0) for s in range(nkb_)])
                A2 = torch.cat([A, sh2[:, :, cols].reshape(nkb_, -1).T], 1)
                sol2 = torch.linalg.lstsq(A2, y).solution[:, 0]
                res2 = float((A2 @ sol2[:, None] - y).norm() / y.norm())
                big2 = [(("t" if i < nkb_ else f"t{tt - t_:+d}") + ":" + str(i % nkb_), round(float(v), 2)) for i, v in enumerate(sol2.tolist()) if abs(v) > 0.2]
                print(f"   + shares of tile {tt}: residual {res2:.2f}, |coef| > 0.2: {big2}")
