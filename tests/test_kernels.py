"""Per-kernel parity: every od_* entry point against the oracle's op on the same seeded
inputs, in fp32 (tolerance 2e-5 rel-L2) and bf16 (2e-2).  Runs on the emulator build
(CPU) and, marked gpu, on the MI355X through the real C ABI."""
import math
import os

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import denoiser_oracle as O
from osu_dreamer_amd import ops
from kernel_backend import dev, frames, unframes, rel_l2, TOL  # noqa: F401

DTYPES = [torch.float32, torch.bfloat16]


def leaf(t):
    return t.detach().float().cpu().clone().requires_grad_()


def mk(shape, g, dev, dtype=torch.float32, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


@pytest.mark.parametrize("dtype", DTYPES)
# (2100, 520, 128) and (1500, 264, 256): on the emulator build (large-M threshold 256 rows, 16 "CUs") the persistent 4-wave kernel walks 2-3 output
# tiles per workgroup with ragged row / column tiles, its operand pipeline crossing the tile boundary (K = 256: also inside a tile)
@pytest.mark.parametrize("M,N,K", [(200, 136, 96), (128, 128, 64), (77, 24, 16), (300, 384, 192), (2100, 520, 128), (1500, 264, 256)])
def test_gemm_nt(dev, dtype, M, N, K):
    g = torch.Generator().manual_seed(1)
    A, W, b = mk((M, K), g, dev, dtype), mk((N, K), g, dev, dtype, 0.2), mk((N,), g, dev)
    C = torch.zeros(M, N, dtype=dtype, device=dev)
    ops.gemm_nt(A, W, b, C)
    ref = A.float().cpu() @ W.float().cpu().t() + b.cpu()
    assert rel_l2(C.float(), ref) < TOL[dtype]
    ops.gemm_nt(A, W, b, C, epilogue=ops.OD_EPI_SILU)
    assert rel_l2(C.float(), O.silu(ref)) < TOL[dtype]
    C0 = mk((M, N), g, dev, dtype)
    C = C0.clone()
    ops.gemm_nt(A, W, None, C, accumulate=True)
    assert rel_l2(C.float(), ref - b.cpu() + C0.float().cpu()) < TOL[dtype]


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(40000, 512, 1024), (33001, 1408, 512), (32768, 3072, 128), (70000, 264, 64)])
def test_gemm_nt_large_m(M, N, K):
    """The large-M bf16 path (256x256 tiles; outputs >= 1024 wide stored non-temporally): ragged row and column tiles, one to
    many k stages per tile, bias / SiLU / accumulate epilogues, against torch on the device."""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    dev, bf = torch.device("cuda:0"), torch.bfloat16
    g = torch.Generator().manual_seed(5)
    A, W, b = mk((M, K), g, dev, bf), mk((N, K), g, dev, bf, 0.2), mk((N,), g, dev)
    C = torch.zeros(M, N, dtype=bf, device=dev)
    ops.gemm_nt(A, W, b, C)
    ref = A.float() @ W.float().t() + b
    assert float((C.float() - ref).norm() / ref.norm()) < TOL[bf]
    assert float((C.float() - ref).abs().max()) < 0.05 * float(ref.abs().max())          # no stale / missing tile anywhere
    ops.gemm_nt(A, W, b, C, epilogue=ops.OD_EPI_SILU)
    assert float((C.float() - torch.nn.functional.silu(ref)).norm() / ref.norm()) < TOL[bf]
    C0 = mk((M, N), g, dev, bf)
    C = C0.clone()
    ops.gemm_nt(A, W, None, C, accumulate=True)
    ref2 = ref - b + C0.float()
    assert float((C.float() - ref2).norm() / ref2.norm()) < TOL[bf]
    # no bias (the 4-wave kernel's accumulators then start at zero), into a column range of a wider tensor (ldc > N)
    wide = torch.full((M, N + 64), 7.0, dtype=bf, device=dev)
    ops.gemm_nt(A, W, None, wide[:, :N])
    assert float((wide[:, :N].float() - (ref - b)).norm() / (ref - b).norm()) < TOL[bf]
    assert bool((wide[:, N:] == 7.0).all())                                               # nothing written past column N


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,hd,L,B,K", [(2, 64, 70, 2, 64), (4, 32, 33, 3, 96), (1, 64, 300, 1, 40)])
def test_gemm_nt_qkrope_equals_gemm_then_rope(dev, dtype, H, hd, L, B, K):
    """The fused epilogue against od_gemm_nt followed by od_qk_norm_rope (same rounding point; only the order of
    the 64-term sum of squares differs)."""
    g = torch.Generator().manual_seed(31)
    M, dh = B * L, H * hd
    A, W, b = mk((M, K), g, dev, dtype), mk((3 * dh, K), g, dev, dtype, 0.3), mk((3 * dh,), g, dev)
    wq, wk = 1 + 0.2 * mk((hd,), g, dev), 1 + 0.2 * mk((hd,), g, dev)
    tab = torch.zeros(L, hd // 2, 2, device=dev)
    ops.rope_table(tab, L, hd)
    eps = 1.2e-7
    qkv = torch.zeros(M, 3 * dh, dtype=dtype, device=dev)
    ops.gemm_nt(A, W, b, qkv)
    qk = torch.zeros(M, 2 * dh, dtype=dtype, device=dev)
    ops.qk_norm_rope(qkv, wq, wk, tab, qk, B, L, H, hd, eps, q_scale=0.18)
    fused = torch.zeros(M, 3 * dh, dtype=dtype, device=dev)
    ops.gemm_nt_qkrope(A, W, b, fused, wq, wk, tab, L, H, hd, eps, q_scale=0.18)
    assert torch.equal(fused[:, 2 * dh:], qkv[:, 2 * dh:])                      # v columns untouched
    assert rel_l2(fused[:, :2 * dh].float(), qk.float()) < (2e-6 if dtype == torch.float32 else 3e-3)
    if dtype == torch.float32:
        ops.gemm_nt_qkrope(A, W, b, fused, wq, wk, tab, L, H, hd, eps, x3=True, q_scale=0.18)
        assert rel_l2(fused[:, :2 * dh], qk) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("N,K", [(512, 1024), (512, 1408), (3072, 512)])
def test_gemm_nt_sampler_rows(N, K):
    """od_gemm_nt at the sampler's M = 4 x 1115 = 4460 rows, bf16 and fp32-as-3-x-bf16 with a pre-split weight, against torch."""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    M = 4460
    A32, W32, b = mk((M, K), g, dev), mk((N, K), g, dev, scale=0.1), mk((N,), g, dev)
    ref = A32.double() @ W32.double().t() + b.double()
    C = torch.zeros(M, N, device=dev)
    Ws = ops.SplitWeight(torch.zeros(N, K, device=dev)); ops.pack_weight(W32, Ws)
    ops.gemm_nt(A32, Ws, b, C, x3=True)
    assert float((C.double() - ref).norm() / ref.norm()) < 2e-5
    A16, W16 = A32.bfloat16(), W32.bfloat16()
    C16 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm_nt(A16, W16, b, C16)
    ref16 = A16.double() @ W16.double().t() + b.double()
    assert float((C16.double() - ref16).norm() / ref16.norm()) < 5e-3


@pytest.mark.parametrize("M,N,K", [(200, 136, 96), (300, 384, 192), (512, 512, 256)])
def test_gemm_nt_f32x3(dev, M, N, K):
    """OD_F32X3: fp32 tensors, three bf16 MFMAs per product.  Tolerance 2e-5 rel-L2 against fp64
    (measured ~4e-6: two dropped terms of 2^-17 relative size each)."""
    g = torch.Generator().manual_seed(11)
    A, W, b = mk((M, K), g, dev), mk((N, K), g, dev, scale=0.2), mk((N,), g, dev)
    C = torch.zeros(M, N, device=dev)
    ops.gemm_nt(A, W, b, C, x3=True)
    ref = (A.double().cpu() @ W.double().cpu().t() + b.double().cpu()).float()
    assert rel_l2(C, ref) < 2e-5
    C32 = torch.zeros(M, N, device=dev)
    ops.gemm_nt(A, W, b, C32)
    assert not torch.equal(C, C32), "the x3 request must reach a different kernel"
    ops.gemm_nt(A, W, b, C, epilogue=ops.OD_EPI_SILU, x3=True)
    assert rel_l2(C, O.silu(ref)) < 2e-5
    # OD_F32X3W: the weight split into its (hi, lo) bf16 halves once at pack time — the same three products, bit for bit
    Ws = ops.SplitWeight(torch.zeros(N, K, device=dev))
    ops.pack_weight(W, Ws)
    Cs = torch.zeros(M, N, device=dev)
    ops.gemm_nt(A, Ws, b, Cs, x3=True)
    ops.gemm_nt(A, W, b, C, x3=True)
    assert torch.equal(Cs, C)
    with pytest.raises(ValueError):
        ops.gemm_nt(A, Ws, b, Cs)                       # a split weight has no meaning for the plain fp32 product


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("L,hd", [(300, 64), (257, 32)])
def test_flash_attention_reference_moves(dev, dtype, L, hd):
    """Scores whose row maximum keeps growing along the key axis (and spans ~100 nats): the forward's lazy
    softmax reference has to move on later key tiles, not only on the first one."""
    g = torch.Generator().manual_seed(13)
    B, H = 1, 2
    M, dh = B * L, H * hd
    q = (torch.randn(M, dh, generator=g) * 3).to(dtype)
    k = (torch.randn(M, dh, generator=g) * (0.2 + 4.0 * torch.arange(L)[:, None] / L)).to(dtype)
    v = torch.randn(M, dh, generator=g).to(dtype)
    scale = 1 / math.sqrt(hd)

    def heads(t):
        return t.double().reshape(B, L, H, hd).permute(0, 2, 1, 3)
    s = heads(q) @ heads(k).transpose(-1, -2) * scale
    first, later = s[..., :64].amax(-1), s[..., 64:].amax(-1)
    assert float((later - first).max()) > 30, "the fixture must push the maximum up after the first tile"
    ref = (torch.softmax(s, -1) @ heads(v)).permute(0, 2, 1, 3).reshape(M, dh).float()
    o = torch.zeros(M, dh, dtype=dtype, device=dev)
    lse = torch.zeros(B, H, L, device=dev)
    ops.flash_attn_fwd(q.to(dev), k.to(dev), v.to(dev), o, lse, B, H, L, hd, scale)
    assert rel_l2(o.float(), ref) < TOL[dtype]
    assert rel_l2(lse, torch.logsumexp(s, -1).float()) < (1e-2 if dtype == torch.bfloat16 else 1e-5)
    if dtype == torch.float32:
        ops.flash_attn_fwd(q.to(dev), k.to(dev), v.to(dev), o, lse, B, H, L, hd, scale, x3=True)
        assert rel_l2(o, ref) < 5e-5


@pytest.mark.parametrize("B,H,L,hd", [(1, 2, 150, 32), (1, 1, 257, 64), (2, 1, 64, 64)])
def test_flash_attention_fwd_f32x3(dev, B, H, L, hd):
    g = torch.Generator().manual_seed(12)
    M, dh = B * L, H * hd
    q, k, v = (mk((M, dh), g, dev) for _ in range(3))
    o = torch.zeros(M, dh, device=dev)
    lse = torch.zeros(B, H, L, device=dev)
    scale = 1 / math.sqrt(hd)
    ops.flash_attn_fwd(q, k, v, o, lse, B, H, L, hd, scale, x3=True)

    def heads(t):
        return t.double().cpu().reshape(B, L, H, hd).permute(0, 2, 1, 3)
    s = heads(q) @ heads(k).transpose(-1, -2) * scale
    ref = (torch.softmax(s, -1) @ heads(v)).permute(0, 2, 1, 3).reshape(M, dh).float()
    assert rel_l2(o, ref) < 2e-5
    assert rel_l2(lse, torch.logsumexp(s, -1).float()) < 1e-5


def same_up_to_an_ulp(a, b, frac=2e-3):
    """bf16 tensors equal except for a small fraction of elements that sit on neighbouring bf16 values (or, where the sum cancels to
    nearly nothing, differ by the fp32 rounding of its terms)."""
    a, b = a.float().cpu(), b.float().cpu()
    diff = (a - b).abs()
    assert float((diff > 0).float().mean()) < frac
    assert bool((diff <= torch.maximum(a.abs(), b.abs()) * 2.0 ** -7 + 1e-5 * float(a.abs().max())).all())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,hd,L,B,K", [(4, 64, 150, 2, 64), (2, 64, 70, 2, 64), (4, 32, 33, 3, 96), (6, 64, 130, 3, 128)])
def test_gemm_nt_qkrope_split_equals_gemm_then_rope(dev, dtype, H, hd, L, B, K):
    """The training-time form (pre-norm projection kept in C, normed + rotated q/k to a second buffer): identical C, q/k equal to
    od_gemm_nt + od_qk_norm_rope up to the order of the 64-term sum of squares.  (4, 64, 150, 2) and (6, 64, 130, 3) run the
    large-M kernel's fused epilogue on the emulator build (its large-M threshold is 256 rows); the rest take the two-kernel path."""
    g = torch.Generator().manual_seed(33)
    M, dh = B * L, H * hd
    A, W, b = mk((M, K), g, dev, dtype), mk((3 * dh, K), g, dev, dtype, 0.3), mk((3 * dh,), g, dev)
    wq, wk = 1 + 0.2 * mk((hd,), g, dev), 1 + 0.2 * mk((hd,), g, dev)
    tab = torch.zeros(L, hd // 2, 2, device=dev)
    ops.rope_table(tab, L, hd)
    eps = 1.2e-7
    qkv = torch.zeros(M, 3 * dh, dtype=dtype, device=dev)
    ops.gemm_nt(A, W, b, qkv)
    qk = torch.zeros(M, 2 * dh, dtype=dtype, device=dev)
    ops.qk_norm_rope(qkv, wq, wk, tab, qk, B, L, H, hd, eps, q_scale=0.18)
    raw = torch.zeros(M, 3 * dh, dtype=dtype, device=dev)
    qk2 = torch.zeros(M, 2 * dh, dtype=dtype, device=dev)
    ops.gemm_nt_qkrope_split(A, W, b, raw, qk2, wq, wk, tab, L, H, hd, eps, q_scale=0.18)
    if dtype == torch.bfloat16 and M >= 256 and K % 128 == 0:
        same_up_to_an_ulp(raw, qkv)          # od_gemm_nt takes the 4-wave kernel there (bias first, not last: see the large-M test)
    else:
        assert torch.equal(raw, qkv)
    assert rel_l2(qk2.float(), qk.float()) < (2e-6 if dtype == torch.float32 else 3e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,K", [(5, 8192, 512), (3, 11001, 128)])
def test_gemm_nt_qkrope_large_m(B, L, K):
    """The fused epilogue of the 256x256 kernel at training-size M (ragged last row tile), both forms: split (training) and in place
    (no-grad), against od_gemm_nt + od_qk_norm_rope."""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    dev, bf = torch.device("cuda:0"), torch.bfloat16
    H, hd = 16, 64
    g = torch.Generator().manual_seed(35)
    M, dh = B * L, H * hd
    A, W, b = mk((M, K), g, dev, bf), mk((3 * dh, K), g, dev, bf, 0.3), mk((3 * dh,), g, dev)
    wq, wk = 1 + 0.2 * mk((hd,), g, dev), 1 + 0.2 * mk((hd,), g, dev)
    tab = torch.zeros(L, hd // 2, 2, device=dev)
    ops.rope_table(tab, L, hd)
    eps = 1.2e-7
    qkv = torch.zeros(M, 3 * dh, dtype=bf, device=dev)
    ops.gemm_nt(A, W, b, qkv)
    qk = torch.zeros(M, 2 * dh, dtype=bf, device=dev)
    ops.qk_norm_rope(qkv, wq, wk, tab, qk, B, L, H, hd, eps, q_scale=0.18)
    raw, qk2 = torch.zeros_like(qkv), torch.zeros_like(qk)
    ops.gemm_nt_qkrope_split(A, W, b, raw, qk2, wq, wk, tab, L, H, hd, eps, q_scale=0.18)
    # od_gemm_nt runs the 4-wave kernel here, whose accumulators START at the bias (the 8-wave kernel under the fused epilogue adds it last):
    # the fp32 sums differ in their last bit now and then, i.e. a bf16 output may land on the neighbouring value
    same_up_to_an_ulp(raw, qkv)
    assert float((qk2.float() - qk.float()).norm() / qk.float().norm()) < 3e-3
    assert float((qk2.float() - qk.float()).abs().max()) < 0.1
    fused = torch.zeros_like(qkv)
    ops.gemm_nt_qkrope(A, W, b, fused, wq, wk, tab, L, H, hd, eps, q_scale=0.18)
    assert torch.equal(fused[:, 2 * dh:], raw[:, 2 * dh:])
    assert torch.equal(fused[:, :2 * dh], qk2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 136, 72), (64, 16, 24), (1000, 130, 170), (600, 264, 300)])
def test_gemm_tn_colsum(dev, dtype, M, N, K):
    g = torch.Generator().manual_seed(2)
    ldg, lda = (N + 7) // 8 * 8 + 8, (K + 7) // 8 * 8
    Gf = mk((M, ldg), g, dev, dtype)
    Af = mk((M, lda), g, dev, dtype)
    dW0 = mk((N, K), g, dev)
    dW = dW0.clone()
    db = torch.ones(N, device=dev)
    ops.gemm_tn(Gf, Af, dW, n_cols=N, k_cols=K, dbias=db)
    ref = Gf.float().cpu()[:, :N].t() @ Af.float().cpu()[:, :K] + dW0.cpu()
    assert rel_l2(dW, ref) < TOL[dtype]
    assert rel_l2(db - 1, Gf.float().cpu()[:, :N].sum(0)) < TOL[dtype]
    out = torch.zeros(N, device=dev)
    ops.colsum(Gf, out, n_cols=N)
    assert rel_l2(out, Gf.float().cpu()[:, :N].sum(0)) < TOL[dtype]


@pytest.mark.parametrize("M,N,K", [(1000, 512, 1365), (257, 1365, 512), (777, 261, 1365)])
def test_gemm_tn_odd_width_counts_every_row(dev, M, N, K):
    """Hf = 1365 is odd: the last valid column shares a dword with the first padding column, and the buffer descriptor of each
    M-split ends at its last row.  Hardware range-checks per dword, so a descriptor cut at the exact element count drops that
    column of the split's last row.  Integer-valued operands make the loss exact: every row must be counted."""
    bf = torch.bfloat16
    lda, ldg = (K + 7) // 8 * 8, (N + 7) // 8 * 8
    A = torch.zeros(M, lda, dtype=bf, device=dev)
    G = torch.zeros(M, ldg, dtype=bf, device=dev)
    A[:, K - 1] = 1.0
    A[:, 0] = 2.0
    G[:, N - 1] = 1.0
    G[:, 0] = 1.0
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    ops.gemm_tn(G, A, dW, n_cols=N, k_cols=K, dbias=db)
    dW = dW.cpu()
    assert float(dW[N - 1, K - 1]) == M and float(dW[0, K - 1]) == M and float(dW[N - 1, 0]) == 2 * M and float(dW[0, 0]) == 2 * M
    assert float(db.cpu()[N - 1]) == M and float(dW.sum()) == 6 * M


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(40000, 3072, 512), (33001, 1365, 512), (65536, 512, 1365), (32768, 512, 1024), (50001, 136, 512)])
def test_gemm_tn_large_m(M, N, K):
    """The weight-gradient GEMMs at training-size M (256x256 kernel from 8 output tiles on, 128x128 below; both stage their slabs
    by buffer-addressed asm LDS-DMA whose descriptor ends at the workgroup's last row): ragged M / N / K, operands that are
    column sub-views of wider buffers holding non-zero neighbours, accumulation into a non-zero dW, the fused bias gradient."""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    dev, bf = torch.device("cuda:0"), torch.bfloat16
    g = torch.Generator().manual_seed(9)
    ldg, lda = (N + 7) // 8 * 8 + 64, (K + 7) // 8 * 8 + 8
    Gf, Af = mk((M, ldg), g, dev, bf), mk((M, lda), g, dev, bf)
    Gv, Av = Gf[:, 8:], Af[:, 8:]                         # 16-byte-aligned column sub-views: live data on both sides
    dW0 = mk((N, K), g, dev)
    dW, db = dW0.clone(), torch.ones(N, device=dev)
    ops.gemm_tn(Gv, Av, dW, n_cols=N, k_cols=K, dbias=db)
    ref = Gv.float()[:, :N].t() @ Av.float()[:, :K] + dW0
    assert float((dW - ref).norm() / ref.norm()) < 1e-3
    assert float((dW - ref).abs().max()) < 0.02 * float(ref.abs().max())       # no missing / doubled slab anywhere
    cs = Gv.float()[:, :N].sum(0)
    assert float((db - 1 - cs).norm() / cs.norm()) < 5e-3      # sums of +-1-sized terms: |sum| ~ sqrt(M), fp32 order-of-summation noise


@pytest.mark.parametrize("dtype", DTYPES)
def test_pack_weight(dev, dtype):
    g = torch.Generator().manual_seed(3)
    src = mk((10, 6, 1), g, dev)
    dst = torch.full((16, 8), 7.0, dtype=dtype, device=dev)
    ops.pack_weight(src, dst)
    ref = torch.zeros(16, 8); ref[:10, :6] = src.cpu()[:, :, 0]
    assert torch.equal(dst.float().cpu(), ref.to(dtype).float())
    dstT = torch.full((8, 16), 7.0, dtype=dtype, device=dev)
    ops.pack_weight(src, dstT, transpose=True)
    assert torch.equal(dstT.float().cpu(), ref.t().to(dtype).float())
    rm = torch.tensor([0, 1, 2, -1, 5, 6, 7, -1], dtype=torch.int32, device=dev)
    d2 = torch.full((8, 8), 7.0, dtype=dtype, device=dev)
    ops.pack_weight(src, d2, row_map=rm)
    ref2 = torch.zeros(8, 8)
    for i, s in enumerate(rm.tolist()):
        if s >= 0:
            ref2[i, :6] = src.cpu()[s, :, 0]
    assert torch.equal(d2.float().cpu(), ref2.to(dtype).float())


@pytest.mark.parametrize("B,N,K", [(3, 50, 40), (32, 96, 512), (35, 70, 300), (4, 33, 513)])     # one tile; the ssg shape (two K chunks); ragged B > 32 / N / K
def test_linear_small(dev, B, N, K):
    g = torch.Generator().manual_seed(4)
    x, W, b, dout = mk((B, K), g, dev), mk((N, K), g, dev, scale=.3), mk((N,), g, dev), mk((B, N), g, dev)
    for act in (ops.OD_ACT_NONE, ops.OD_ACT_SILU):
        out, pre = torch.zeros(B, N, device=dev), torch.zeros(B, N, device=dev)
        ops.linear_small(x, W, b, out, pre, act)
        xr, Wr, br = leaf(x), leaf(W), leaf(b)
        ref = xr @ Wr.t() + br
        if act:
            ref = O.silu(ref)
        assert rel_l2(out, ref) < 1e-5
        ref.backward(dout.cpu())
        dW, db, dx, dpre = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev), torch.ones(B, K, device=dev), torch.zeros(B, N, device=dev)
        ops.linear_small_bwd(x, W, pre, dout, dpre, dW, db, dx, True, act)
        assert rel_l2(dW, Wr.grad) < 1e-5 and rel_l2(db, br.grad) < 1e-5 and rel_l2(dx - 1, xr.grad) < 1e-5
        dx2 = torch.full((B, K), 7.0, device=dev)                 # not accumulating: whatever dx held is overwritten
        ops.linear_small_bwd(x, W, pre, dout, dpre, None, None, dx2, False, act)
        assert rel_l2(dx2, xr.grad) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_projin_silu(dev, dtype):
    g = torch.Generator().manual_seed(5)
    B, C, L, E, D = 2, 24, 70, 6, 64
    src = mk((B, C, L), g, dev)
    dst = torch.zeros(B * L, C, dtype=dtype, device=dev)
    ops.cl_to_frames(src, dst)
    assert torch.equal(dst.float().cpu(), frames(src.cpu()).to(dtype).float())
    xt, W, b = mk((B, E, L), g, dev), mk((D, E, 1), g, dev), mk((D,), g, dev)
    x = torch.zeros(B * L, D, dtype=dtype, device=dev)
    ops.proj_in(xt, W, b, x)
    ref = O.pointwise_conv(xt.cpu(), W.cpu(), b.cpu())
    assert rel_l2(x.float(), frames(ref)) < TOL[dtype]
    dx = mk((B * L, D), g, dev, dtype)
    dW, db = torch.zeros(D, E, 1, device=dev), torch.zeros(D, device=dev)
    ops.proj_in_bwd(xt, dx, dW, db)
    dxr = unframes(dx.float().cpu(), B, L)
    assert rel_l2(dW[:, :, 0], torch.einsum("bdl,bel->de", dxr, xt.cpu())) < TOL[dtype]
    assert rel_l2(db, dxr.sum((0, 2))) < TOL[dtype]
    y = torch.zeros_like(x)
    ops.silu(x, y)
    assert rel_l2(y.float(), O.silu(x.float().cpu())) < TOL[dtype]
    xr = leaf(x)
    O.silu(xr).backward(dx.float().cpu())
    d2 = torch.zeros_like(x)
    ops.silu_bwd(x, dx, d2)
    assert rel_l2(d2.float(), xr.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,bcast", [(64, False), (512, True), (136, False)])
def test_rmsnorm_film_and_gate(dev, dtype, C, bcast):
    g = torch.Generator().manual_seed(6)
    B, L = 2, 37
    M = B * L
    x, h = mk((M, C), g, dev, dtype), mk((M, C), g, dev, dtype)
    ssg = mk((B, 3 * C), g, dev, scale=.5)
    cl = mk((L if bcast else M, C), g, dev, dtype)
    out = torch.zeros_like(x)
    inv = torch.zeros(M, device=dev)
    ops.rmsnorm_film(x, ssg, cl, bcast, out, inv, B, L)

    def film(xr, ssgr, clr):
        xb = unframes(xr, B, L)
        sc, sh = ssgr[:, :C, None], ssgr[:, C:2 * C, None]
        clb = unframes(clr, 1 if bcast else B, L)
        return frames(O.rms_norm_channels(xb) * (1 + sc) + sh + clb)
    xr, sr = leaf(x), leaf(ssg)
    ref = film(xr, sr, cl.float().cpu())
    assert rel_l2(out.float(), ref) < TOL[dtype]
    dh = mk((M, C), g, dev, dtype)
    ref.backward(dh.float().cpu())
    dres0 = mk((M, C), g, dev, dtype)
    dres = dres0.clone()
    dssg = torch.zeros(B, 3 * C, device=dev)
    ops.rmsnorm_film_bwd(x, inv, ssg, dh, dres, dssg, B, L)
    assert rel_l2(dres.float() - dres0.float(), xr.grad) < 2.5 * TOL[dtype]
    assert rel_l2(dssg[:, :2 * C], sr.grad[:, :2 * C]) < TOL[dtype]

    xo = torch.zeros_like(x)
    ops.rmsnorm_gate_residual(x, h, ssg, xo, inv, B, L)
    hr, sr = leaf(h), leaf(ssg)
    ref = x.float().cpu() + frames(O.rms_norm_channels(unframes(hr, B, L)) * sr[:, 2 * C:, None])
    assert rel_l2(xo.float(), ref) < TOL[dtype]
    ref.backward(dh.float().cpu())
    dhh = torch.zeros_like(x)
    dssg.zero_()
    ops.rmsnorm_gate_residual_bwd(h, inv, ssg, dh, dhh, dssg, B, L)
    assert rel_l2(dhh.float(), hr.grad) < TOL[dtype]
    assert rel_l2(dssg[:, 2 * C:], sr.grad[:, 2 * C:]) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,hd,B,L", [(2, 32, 2, 19), (16, 64, 2, 19), (3, 16, 2, 19),
                                      (16, 64, 11, 5), (16, 32, 3, 9), (32, 64, 1, 21)])   # last three: position-major kernels, ragged batch / position groups
@pytest.mark.parametrize("q_scale", [1.0, 0.18])        # 1: the reference's tensor; scale*log2(e): what the engine feeds attention
def test_qk_norm_rope(dev, dtype, H, hd, B, L, q_scale):
    g = torch.Generator().manual_seed(7)
    M, dh = B * L, H * hd
    qkv = mk((M, 3 * dh), g, dev, dtype)
    wq, wk = 1 + .2 * mk((hd,), g, dev), 1 + .2 * mk((hd,), g, dev)
    table = torch.zeros(L, hd // 2, 2, device=dev)
    ops.rope_table(table, L, hd)
    inv_freq = 10000.0 ** (torch.arange(0, hd, 2).float() / -hd)
    ang = torch.outer(torch.arange(L).float(), inv_freq)
    assert rel_l2(table[..., 0], ang.cos()) < 1e-5 and rel_l2(table[..., 1], ang.sin()) < 1e-5
    out = torch.zeros(M, 2 * dh, dtype=dtype, device=dev)
    eps = torch.finfo(torch.float32).eps
    ops.qk_norm_rope(qkv, wq, wk, table, out, B, L, H, hd, eps, q_scale=q_scale)

    def ref_fn(qkvr, wqr, wkr):
        t = qkvr.reshape(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)   # (3,B,H,L,hd)
        q = O.rope_half_split(O.head_rms_norm(t[0], wqr)) * q_scale
        k = O.rope_half_split(O.head_rms_norm(t[1], wkr))
        return torch.stack([q, k], 0).permute(1, 3, 0, 2, 4).reshape(M, 2 * dh)
    qr, wqr, wkr = leaf(qkv), leaf(wq), leaf(wk)
    ref = ref_fn(qr, wqr, wkr)
    assert rel_l2(out.float(), ref) < TOL[dtype]
    dqk = mk((M, 2 * dh), g, dev, dtype)
    ref.backward(dqk.float().cpu())
    dqkv = torch.zeros(M, 3 * dh, dtype=dtype, device=dev)
    dwq, dwk = torch.zeros(hd, device=dev), torch.zeros(hd, device=dev)
    ops.qk_norm_rope_bwd(qkv, wq, wk, table, dqk, dqkv, dwq, dwk, B, L, H, hd, eps, q_scale=q_scale)
    assert rel_l2(dqkv.float()[:, :2 * dh], qr.grad[:, :2 * dh]) < TOL[dtype]
    assert rel_l2(dwq, wqr.grad) < TOL[dtype] and rel_l2(dwk, wkr.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,with_cl", [(64, True), (512, False), (1024, True)])
def test_rmsnorm_gate_residual_film_equals_pair(dev, dtype, C, with_cl):
    """The fused forward kernel against the two kernels it replaces: identical bits (xo is rounded before the
    second norm), broadcast and per-sample cl."""
    g = torch.Generator().manual_seed(21)
    B, L = 2, 19
    M = B * L
    x, h = mk((M, C), g, dev, dtype), mk((M, C), g, dev, dtype)
    ssg_a, ssg_b = mk((B, 3 * C), g, dev, scale=0.3), mk((B, 3 * C), g, dev, scale=0.3)
    for bcast in (False, True):
        cl = mk((L if bcast else M, C), g, dev, dtype) if with_cl else None
        xo_ref, h2_ref = torch.zeros_like(x), torch.zeros_like(x)
        inv_a_ref, inv_b_ref = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
        ops.rmsnorm_gate_residual(x, h, ssg_a, xo_ref, inv_a_ref, B, L)
        ops.rmsnorm_film(xo_ref, ssg_b, cl, bcast, h2_ref, inv_b_ref, B, L)
        xo, h2 = torch.zeros_like(x), torch.zeros_like(x)
        inv_a, inv_b = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
        ops.rmsnorm_gate_residual_film(x, h, ssg_a, xo, inv_a, ssg_b, cl, bcast, h2, inv_b, B, L)
        assert torch.equal(xo, xo_ref) and torch.equal(h2, h2_ref)
        assert torch.equal(inv_a, inv_a_ref) and torch.equal(inv_b, inv_b_ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,L,hd", [(1, 2, 150, 32), (2, 1, 64, 64), (1, 1, 257, 64), (1, 1, 5, 32), (2, 2, 40, 64), (1, 2, 321, 64)])
def test_flash_attention(dev, dtype, B, H, L, hd):
    g = torch.Generator().manual_seed(8)
    M, dh = B * L, H * hd
    qk = mk((M, 2 * dh), g, dev, dtype)
    qkv = mk((M, 3 * dh), g, dev, dtype)
    q, k, v = qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:]
    o = torch.zeros(M, dh, dtype=dtype, device=dev)
    lse = torch.zeros(B, H, L, device=dev)
    scale = 1 / math.sqrt(hd)
    ops.flash_attn_fwd(q, k, v, o, lse, B, H, L, hd, scale)

    def heads(t):   # [M, dh] -> (B,H,L,hd)
        return t.reshape(B, L, H, hd).permute(0, 2, 1, 3)
    qr, kr, vr = (leaf(t) for t in (q, k, v))
    s = heads(qr) @ heads(kr).transpose(-1, -2) * scale
    ref = (torch.softmax(s, -1) @ heads(vr)).permute(0, 2, 1, 3).reshape(M, dh)
    assert rel_l2(o.float(), ref) < TOL[dtype]
    assert rel_l2(lse, torch.logsumexp(s, -1)) < (1e-2 if dtype == torch.bfloat16 else 1e-5)
    do = mk((M, dh), g, dev, dtype)
    ref.backward(do.float().cpu())
    dq, dk, dv = (torch.zeros(M, dh, dtype=dtype, device=dev) for _ in range(3))
    delta = torch.zeros(B, H, L, device=dev)
    ops.flash_attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, B, H, L, hd, scale)
    tol = 1.5 * TOL[dtype]
    assert rel_l2(dv.float(), vr.grad) < tol
    assert rel_l2(dk.float(), kr.grad) < tol
    assert rel_l2(dq.float(), qr.grad) < tol
    # q_prescaled: q' = q * scale * log2(e) handed over instead of q; same softmax, dq returned for q'
    c = scale * math.log2(math.e)
    qp = (q.float() * c).to(dtype).contiguous()
    qs = leaf(qp.float() / c)                                   # the q that q' stands for
    kr2, vr2 = leaf(k), leaf(v)
    s2 = heads(qs) @ heads(kr2).transpose(-1, -2) * scale
    ref2 = (torch.softmax(s2, -1) @ heads(vr2)).permute(0, 2, 1, 3).reshape(M, dh)
    o2, lse2 = torch.zeros_like(o), torch.zeros_like(lse)
    ops.flash_attn_fwd(qp, k, v, o2, lse2, B, H, L, hd, scale, q_prescaled=True)
    assert rel_l2(o2.float(), ref2) < TOL[dtype]
    assert rel_l2(lse2, torch.logsumexp(s2, -1)) < (1e-2 if dtype == torch.bfloat16 else 1e-5)
    ref2.backward(do.float().cpu())
    dq2, dk2, dv2 = (torch.zeros(M, dh, dtype=dtype, device=dev) for _ in range(3))
    ops.flash_attn_bwd(qp, k, v, o2, do, lse2, delta, dq2, dk2, dv2, B, H, L, hd, scale, q_prescaled=True)
    assert rel_l2(dv2.float(), vr2.grad) < tol and rel_l2(dk2.float(), kr2.grad) < tol
    assert rel_l2(dq2.float(), qs.grad / c) < tol


@pytest.mark.parametrize("pre", [False, True])
@pytest.mark.parametrize("B,H,L", [(2, 2, 75), (1, 3, 200), (2, 1, 64), (1, 9, 450), (3, 1, 192), (1, 2, 600), (1, 1, 65), (1, 11, 193), (2, 1, 385), (5, 7, 200), (1, 41, 250)])
def test_flash_attn_bwd_fused(dev, B, H, L, pre):
    """od_flash_attn_bwd_fused (5 MFMA passes, dQ through the key-block chain; bf16, head_dim 64) against dense fp32 autograd and against
    od_flash_attn_bwd.  Lengths cover one and several key blocks (192 keys each), ragged key blocks and query tiles, more (batch, head)
    pairs than XCD queues, more than 8 x FB_SLOTS of them (the running-tile slots are reused within a launch, with ragged queues), and a second
    call on the same workspace (the control block, flags and write numbers must be left re-armed)."""
    hd, dtype = 64, torch.bfloat16
    g = torch.Generator().manual_seed(31)
    M, dh = B * L, H * hd
    scale = 1 / math.sqrt(hd)
    c = scale * math.log2(math.e)
    qk = mk((M, 2 * dh), g, dev, dtype)
    qkv = mk((M, 3 * dh), g, dev, dtype)
    if pre:
        qk[:, :dh] = (qk[:, :dh].float() * c).to(dtype)
    q, k, v = qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:]
    o = torch.zeros(M, dh, dtype=dtype, device=dev)
    lse = torch.zeros(B, H, L, device=dev)
    ops.flash_attn_fwd(q, k, v, o, lse, B, H, L, hd, scale, q_prescaled=pre)
    do = mk((M, dh), g, dev, dtype)

    def heads(t):
        return t.reshape(B, L, H, hd).permute(0, 2, 1, 3)
    qr, kr, vr = (leaf(t) for t in (q, k, v))
    s = heads(qr) @ heads(kr).transpose(-1, -2) * (math.log(2.0) if pre else scale)
    ref = (torch.softmax(s, -1) @ heads(vr)).permute(0, 2, 1, 3).reshape(M, dh)
    ref.backward(do.float().cpu())
    ws = ops.FusedAttnBwdWorkspace(B, H, L, dev)
    tol = 1.5 * TOL[dtype]
    for rep in range(2):
        dqk = torch.full((M, 2 * dh), float("nan"), dtype=dtype, device=dev)
        dqkv = torch.full((M, 3 * dh), float("nan"), dtype=dtype, device=dev)
        dq, dk, dv = dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:]
        ops.flash_attn_bwd_fused(q, k, v, o, do, lse, dq, dk, dv, B, H, L, hd, scale, ws, q_prescaled=pre)
        assert ws.status() == 0
        assert rel_l2(dv.float(), vr.grad) < tol
        assert rel_l2(dk.float(), kr.grad) < tol
        assert rel_l2(dq.float(), qr.grad) < tol
    dq7, dk7, dv7 = (torch.zeros(M, dh, dtype=dtype, device=dev) for _ in range(3))
    delta = torch.zeros(B, H, L, device=dev)
    ops.flash_attn_bwd(q, k, v, o, do, lse, delta, dq7, dk7, dv7, B, H, L, hd, scale, q_prescaled=pre)
    assert rel_l2(dk.float(), dk7.float().cpu()) < 2e-3 and rel_l2(dv.float(), dv7.float().cpu()) < 2e-3      # same algorithm, same tiles
    assert rel_l2(dq.float(), dq7.float().cpu()) < 6e-3                                                     # fp32 chain vs in-register sum


@pytest.mark.parametrize("pre", [False, True])
@pytest.mark.parametrize("B,H,L", [(2, 2, 75), (1, 9, 450)])
def test_flash_attn_f16_operands(dev, B, H, L, pre):
    """"Attention in fp16" (BASELINE configs[4]): q, k, v as IEEE half, P / dS / the staged dO half inside the kernels (v_mfma_f32_*_f16),
    o, dO, dq, dk, dv bf16 — od_flash_attn_fwd and od_flash_attn_bwd_fused with OD_F16 against dense fp32 autograd on the SAME (half-rounded)
    inputs.  Half carries 3 more mantissa bits than bf16: P and dS are 8x finer, so the error is what the bf16 OUTPUT rounding leaves.
    dO is tiny on purpose (1e-6: a mean-reduced loss) — it must survive through the power-of-two re-scaling, not vanish in half."""
    hd = 64
    g = torch.Generator().manual_seed(41)
    M, dh = B * L, H * hd
    scale = 1 / math.sqrt(hd)
    c = scale * math.log2(math.e)
    f16, bf = torch.float16, torch.bfloat16
    q = (torch.randn(M, dh, generator=g) * (c if pre else 1.0)).to(f16).to(dev)
    k = torch.randn(M, dh, generator=g).to(f16).to(dev)
    v = (torch.randn(M, dh, generator=g) * 3.0).to(f16).to(dev)
    o = torch.zeros(M, dh, dtype=bf, device=dev)
    lse = torch.zeros(B, H, L, device=dev)
    ops.flash_attn_fwd(q, k, v, o, lse, B, H, L, hd, scale, q_prescaled=pre)
    do = (torch.randn(M, dh, generator=g) * 1e-6).to(bf).to(dev)

    def heads(t):
        return t.reshape(B, L, H, hd).permute(0, 2, 1, 3)
    qr, kr, vr = (leaf(t) for t in (q, k, v))
    s = heads(qr) @ heads(kr).transpose(-1, -2) * (math.log(2.0) if pre else scale)
    ref = (torch.softmax(s, -1) @ heads(vr)).permute(0, 2, 1, 3).reshape(M, dh)
    assert rel_l2(o.float(), ref) < 4e-3                       # bf16 output rounding (2^-9 per element)
    ref.backward(do.float().cpu())
    ws = ops.FusedAttnBwdWorkspace(B, H, L, dev, torch.float16)
    for rep in range(2):
        dq, dk, dv = (torch.full((M, dh), float("nan"), dtype=bf, device=dev) for _ in range(3))
        ops.flash_attn_bwd_fused(q, k, v, o, do, lse, dq, dk, dv, B, H, L, hd, scale, ws, q_prescaled=pre)
        assert ws.status() == 0
        # o is bf16 (delta = sum dO o carries its rounding) and the outputs are bf16: 5e-3; the bf16-operand kernel is held to 3e-2
        assert rel_l2(dv.float(), vr.grad) < 5e-3
        assert rel_l2(dk.float(), kr.grad) < 8e-3
        assert rel_l2(dq.float(), qr.grad) < 8e-3


def test_flash_attn_bwd_fused_refuses_another_shape(dev):
    """A workspace's running tiles carry write numbers that continue from launch to launch: a launch with another (B, H, L) must be refused
    (status 2, nothing computed) instead of waiting for numbers that never come."""
    from osu_dreamer_amd import _lib
    hd, dtype, B, H, L = 64, torch.bfloat16, 1, 2, 200
    g = torch.Generator().manual_seed(5)
    M, dh = B * L, H * hd
    q, k, v, do = (mk((M, dh), g, dev, dtype) for _ in range(4))
    o = torch.zeros(M, dh, dtype=dtype, device=dev)
    lse = torch.zeros(B, H, L, device=dev)
    scale = 1 / math.sqrt(hd)
    ops.flash_attn_fwd(q, k, v, o, lse, B, H, L, hd, scale)
    ws = ops.FusedAttnBwdWorkspace(B, H, L, dev)
    dq, dk, dv = (torch.zeros(M, dh, dtype=dtype, device=dev) for _ in range(3))
    ops.flash_attn_bwd_fused(q, k, v, o, do, lse, dq, dk, dv, B, H, L, hd, scale, ws)
    assert ws.status() == 0
    with pytest.raises(AssertionError):
        ops.flash_attn_bwd_fused(q, k, v, o, do, lse, dq, dk, dv, B, H, L - 70, hd, scale, ws)      # the Python wrapper checks first
    p, ld = ops._p, ops._ld
    _lib.lib().od_flash_attn_bwd_fused(ops.dt_code(dtype), p(q), ld(q), p(k), ld(k), p(v), ld(v), p(o), ld(o), p(do), ld(do), p(lse),
                                       p(dq), ld(dq), p(dk), ld(dk), p(dv), ld(dv), B, H, L - 70, hd, scale, 0, p(ws.buf), ws.bytes, ops._stream(q))
    assert ws.status() == 2


def test_flash_attn_bwd_fused_corrupt_workspace_terminates(dev):
    """A workspace whose running tiles do not carry the write numbers a launch expects (not zeroed, or left mid-launch by an aborted process):
    the chain wait is BOUNDED — the launch must terminate (no hung GPU), report status 3 and poison dq with NaN; the device-side fold turns
    the gradient norm into NaN and the optimizer update into a no-op; after a reset the workspace serves again."""
    hd, dtype, B, H, L = 64, torch.bfloat16, 1, 2, 400
    g = torch.Generator().manual_seed(6)
    M, dh = B * L, H * hd
    q, k, v, do = (mk((M, dh), g, dev, dtype) for _ in range(4))
    o = torch.zeros(M, dh, dtype=dtype, device=dev)
    lse = torch.zeros(B, H, L, device=dev)
    scale = 1 / math.sqrt(hd)
    ops.flash_attn_fwd(q, k, v, o, lse, B, H, L, hd, scale)
    os.environ["OD_FB_CHAIN_TIMEOUT_MS"] = "50"          # read once per process, before the first fused launch (harmless if it already was)
    ws = ops.FusedAttnBwdWorkspace(B, H, L, dev)
    dq, dk, dv = (torch.zeros(M, dh, dtype=dtype, device=dev) for _ in range(3))
    ops.flash_attn_bwd_fused(q, k, v, o, do, lse, dq, dk, dv, B, H, L, hd, scale, ws)
    assert ws.status() == 0
    good = dq.float().cpu().clone()
    # garbage write numbers in every running tile (the control block, 256 bytes, stays intact)
    ws.buf[256:ws.zero_bytes].view(torch.int32).fill_(5)
    ops.flash_attn_bwd_fused(q, k, v, o, do, lse, dq, dk, dv, B, H, L, hd, scale, ws)
    assert ws.status() == 3                                # ... and we got here: the launch ended
    assert torch.isnan(dq.float()).any()
    # the same word folded into the step on the device: norm -> NaN, update skipped
    n = 1000
    gr, p0 = mk((n,), g, dev), mk((n,), g, dev)
    p, m, vv, ema = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    gn = torch.zeros(1, device=dev)
    ops.sqnorm(gr, gn, ws.err_ptr())
    assert torch.isnan(gn).all()
    ops.adamw_ema(p, gr, m, vv, ema, 3e-4, .9, .999, 1e-8, .01, 1, .99, 1, gn, 1.0, ws.err_ptr())
    assert torch.equal(p, p0) and float(m.abs().sum()) == 0 and float(ema.abs().sum()) == 0
    # a healthy word changes nothing
    ws.reset()
    assert ws.status() == 0
    gn.zero_()
    ops.sqnorm(gr, gn, ws.err_ptr())
    assert abs(float(gn) - float((gr.double() ** 2).sum())) < 1e-3 * float(gn)
    ops.adamw_ema(p, gr, m, vv, ema, 3e-4, .9, .999, 1e-8, .01, 1, .99, 1, gn, 1.0, ws.err_ptr())
    assert not torch.equal(p, p0)
    ops.flash_attn_bwd_fused(q, k, v, o, do, lse, dq, dk, dv, B, H, L, hd, scale, ws)
    assert ws.status() == 0 and torch.equal(dq.float().cpu(), good)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ks", [3, 5])
@pytest.mark.parametrize("L", [75, 21])          # long-run and short-run launch shapes (od_dwconv picks by size)
def test_dwconv(dev, dtype, ks, L):
    g = torch.Generator().manual_seed(9)
    B, C = 2, 24
    x = mk((B * L, C), g, dev, dtype)
    w, b = mk((C, 1, ks), g, dev, scale=.4), mk((C,), g, dev)
    y = torch.zeros_like(x)
    ops.dwconv(x, w, b, y, B, L, ks)
    xr, wr, br = leaf(x), leaf(w), leaf(b)
    ref = frames(O.depthwise_conv(unframes(xr, B, L), wr, br))
    assert rel_l2(y.float(), ref) < TOL[dtype]
    dy = mk((B * L, C), g, dev, dtype)
    ref.backward(dy.float().cpu())
    dx, dw, db = torch.zeros_like(x), torch.zeros(C, 1, ks, device=dev), torch.zeros(C, device=dev)
    ops.dwconv_bwd(x, w, dy, dx, dw, db, B, L, ks)
    assert rel_l2(dx.float(), xr.grad) < TOL[dtype]
    assert rel_l2(dw, wr.grad) < TOL[dtype] and rel_l2(db, br.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Hf,Hp", [(170, 192), (1365, 1408)])
def test_swiglu_rmsnorm(dev, dtype, Hf, Hp):
    g = torch.Generator().manual_seed(10)
    M = 9
    vg = torch.zeros(M, 2 * Hp, dtype=dtype, device=dev)
    vg[:, :Hf] = mk((M, Hf), g, dev, dtype)
    vg[:, Hp:Hp + Hf] = mk((M, Hf), g, dev, dtype)
    hh = torch.zeros(M, Hp, dtype=dtype, device=dev)
    inv = torch.zeros(M, device=dev)
    ops.swiglu_rmsnorm(vg, hh, inv, Hf, Hp)
    vr = leaf(vg)
    ref = O.rms_norm_channels(vr[:, :Hf] * O.silu(vr[:, Hp:Hp + Hf]))
    assert rel_l2(hh.float()[:, :Hf], ref) < TOL[dtype]
    assert float(hh.float()[:, Hf:].abs().max()) == 0.0
    dhh = torch.zeros(M, Hp, dtype=dtype, device=dev)
    dhh[:, :Hf] = mk((M, Hf), g, dev, dtype)
    ref.backward(dhh.float().cpu()[:, :Hf])
    dvg = torch.zeros_like(vg)
    ops.swiglu_rmsnorm_bwd(vg, inv, dhh, dvg, Hf, Hp)
    assert rel_l2(dvg.float(), vr.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_final_norm_proj_out(dev, dtype):
    g = torch.Generator().manual_seed(11)
    B, L, C, E = 2, 45, 64, 6
    x = mk((B * L, C), g, dev, dtype)
    W, b = mk((E, C, 1), g, dev, scale=.3), mk((E,), g, dev)
    v, inv = torch.zeros(B, E, L, device=dev), torch.zeros(B * L, device=dev)
    ops.final_norm_proj_out(x, W, b, v, inv, B, L)
    xr, Wr, br = leaf(x), leaf(W), leaf(b)
    ref = O.pointwise_conv(O.rms_norm_channels(unframes(xr, B, L)), Wr, br)
    assert rel_l2(v, ref) < 1e-5
    dv = mk((B, E, L), g, dev)
    ref.backward(dv.cpu())
    dx, dW, db = torch.zeros_like(x), torch.zeros(E, C, 1, device=dev), torch.zeros(E, device=dev)
    ops.final_norm_proj_out_bwd(x, inv, W, dv, dx, dW, db, B, L)
    assert rel_l2(dx.float(), xr.grad) < TOL[dtype]
    assert rel_l2(dW, Wr.grad) < 1e-5 and rel_l2(db, br.grad) < 1e-5


@pytest.mark.parametrize("U,L", [(16, 77), (64, 40), (64, 301)])
def test_uhead(dev, U, L):
    g = torch.Generator().manual_seed(12)
    B, E = 2, 6
    names = ["u_head.0", "u_head.1", "u_head.3", "u_head.4"]
    shapes = {"u_head.0": (E, 1, 3), "u_head.1": (U, E, 1), "u_head.3": (U, 1, 3), "u_head.4": (U, U, 1)}
    P = {}
    for n in names:
        P[n + ".weight"] = mk(shapes[n], g, dev, scale=.5)
        P[n + ".bias"] = mk((shapes[n][0],), g, dev, scale=.3)
    xt = mk((B, E, L), g, dev)
    mod, w_out, b_out = mk((B, 2 * U), g, dev, scale=.3), mk((1, U), g, dev, scale=.3), mk((1,), g, dev)
    wlist = [P[n + s] for n in names for s in (".weight", ".bias")]
    fsum, u = torch.zeros(B, U, device=dev), torch.zeros(B, device=dev)
    ops.uhead_fwd(xt, wlist, fsum, U)
    ops.uhead_tail(fsum, mod, w_out, b_out, u, L, 3.4641016)
    Pr = {k: leaf(v) for k, v in P.items()}
    modr, wor, bor = leaf(mod), leaf(w_out), leaf(b_out)
    f = O.u_head(xt.cpu(), Pr).mean(-1)
    assert rel_l2(fsum / L, f) < 1e-5
    fm = f * (1 + modr[:, :U]) + modr[:, U:]
    uref = 3.4641016 * torch.nn.functional.softplus(fm @ wor.t() + bor).squeeze(-1)
    assert rel_l2(u, uref) < 1e-5
    du = mk((B,), g, dev)
    uref.backward(du.cpu())
    dfm, dmod = torch.zeros(B, U, device=dev), torch.zeros(B, 2 * U, device=dev)
    dwo, dbo = torch.zeros(1, U, device=dev), torch.zeros(1, device=dev)
    ops.uhead_tail_bwd(fsum, mod, w_out, b_out, du, dfm, dmod, dwo, dbo, L, 3.4641016)
    assert rel_l2(dmod, modr.grad) < 1e-5 and rel_l2(dwo, wor.grad) < 1e-5 and rel_l2(dbo, bor.grad) < 1e-5
    grads = [torch.zeros_like(t) for t in wlist]
    ops.uhead_bwd(xt, wlist, dfm, grads, U)
    for t, gr, n in zip(wlist, grads, [n + s for n in names for s in (".weight", ".bias")]):
        assert rel_l2(gr, Pr[n].grad) < 2e-5, n


def test_loss_and_sampler(dev):
    g = torch.Generator().manual_seed(13)
    B, E, L = 3, 6, 50
    x0, x1, v = mk((B, E, L), g, dev), mk((B, E, L), g, dev), mk((B, E, L), g, dev)
    t = torch.tensor([0.1, 0.5, 0.93], device=dev)
    u = torch.tensor([1.5, 0.7, 2.2], device=dev)
    c0, _ = O.distance_constants(E)
    xt, dsq = torch.zeros_like(x0), torch.zeros(B, device=dev)
    ops.make_xt(x0, x1, t, xt, dsq)
    xtr = torch.lerp(x0.cpu(), x1.cpu(), t.cpu()[:, None, None])
    assert rel_l2(xt, xtr) < 1e-6 and rel_l2(dsq, O.frame_dist_sq(xtr, x1.cpu())) < 1e-5
    ur, vr = leaf(u), leaf(v)
    d_sq = O.frame_dist_sq(xtr, x1.cpu())
    ut = (d_sq + c0).sqrt()
    osl = (O.frame_dist_sq(xtr - ur[:, None, None] * vr, x1.cpu()) / (d_sq + c0)).mean()
    dl = O.frame_dist_sq(vr, (xtr - x1.cpu()) / ut[:, None, None]).mean()
    loss = osl + 30 * dl
    loss.backward()
    dv, sums, out, du = torch.zeros_like(v), torch.zeros(B, 3, device=dev), torch.zeros(4, device=dev), torch.zeros(B, device=dev)
    ops.loss_grad(xt, x1, u, v, dsq, dv, sums, c0, 1.0, 30.0)
    ops.loss_finalize(sums, dsq, u, out, du, c0, 1.0, 30.0)
    assert rel_l2(out[0], loss) < 1e-5 and rel_l2(out[1], osl) < 1e-5 and rel_l2(out[2], dl) < 1e-5
    assert rel_l2(out[3], ((ur - ut) / ut).abs().mean()) < 1e-5
    assert rel_l2(dv, vr.grad) < 1e-5 and rel_l2(du, ur.grad) < 1e-5
    eta = torch.zeros(2, device=dev)
    ops.sampler_eta(u, eta, c0, 8)
    u0 = float(u.mean())
    assert float(eta[0]) == pytest.approx(1 - (math.sqrt(c0) / max(u0, math.sqrt(c0) + 1e-6)) ** (1 / 8), rel=1e-5)
    x = x0.clone()
    ops.sampler_step(x, u, v, eta)
    assert rel_l2(x, x0.cpu() - float(eta[0]) * u.cpu()[:, None, None] * v.cpu()) < 1e-6


def test_optimizer(dev):
    g = torch.Generator().manual_seed(14)
    n = 1003
    p, gr = mk((n,), g, dev), mk((n,), g, dev, scale=3.0)
    m, v, ema = torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    P, G = {"w": p.cpu().clone()}, {"w": gr.cpu().clone()}
    Mo, Vo, Eo = {"w": torch.zeros(n)}, {"w": torch.zeros(n)}, {"w": torch.zeros(n)}
    gn = torch.zeros(1, device=dev)
    ops.sqnorm(gr, gn)
    total, coef = O.clip_coef(G, 1.0)
    assert float(gn.sqrt()) == pytest.approx(total, rel=1e-5)
    for step in (1, 2, 3):
        ops.adamw_ema(p, gr, m, v, ema, 3e-4, .9, .999, 1e-8, .01, step, .99, 1 if step == 1 else 2, gn, 1.0)
        O.adamw_ema_step(P, G, Mo, Vo, Eo, step, 3e-4, clip=coef, first_ema=(step == 1))
    assert torch.allclose(p.cpu(), P["w"], rtol=1e-5, atol=1e-7)
    assert torch.allclose(ema.cpu(), Eo["w"], rtol=1e-5, atol=1e-7)
    assert torch.allclose(v.cpu(), Vo["w"], rtol=1e-5, atol=1e-9)


def test_rope_table_long_positions(dev):
    """od_rope_table at BASELINE configs[4]'s length against the REFERENCE's own angles (tests/golden/rope_long.npz, rows
    sub-sampled): at position 32767 one ulp of a high inverse frequency is already 2e-3 rad, so the table must not depend
    on a fast device powf.  Then q/k norm + RoPE through od_qk_norm_rope on those rows against the reference's rope()."""
    import numpy as np
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "rope_long.npz"))
    N, D = int(z["N"]), int(z["D"])
    pos = torch.from_numpy(z["pos"]).long()
    ang = torch.from_numpy(z["angle"])
    table = torch.zeros(N, D // 2, 2, device=dev)
    ops.rope_table(table, N, D)
    t = table.cpu()[pos]
    # cos/sin of the reference's fp32 angle, evaluated in double: the table may differ by the device's cosf/sinf rounding and
    # by <= 1 ulp of inv_freq at the two low frequencies where torch's own fp32 pow is not correctly rounded (<= 8e-6 rad)
    assert float((t[..., 0].double() - ang.double().cos()).abs().max()) < 1e-5
    assert float((t[..., 1].double() - ang.double().sin()).abs().max()) < 1e-5
    if dev.type != "cuda":
        return                       # the norm + RoPE kernel over 32768 frames is a GPU-sized launch
    H = 2
    x = torch.from_numpy(z["x"])                       # (1, H, P, D) at rows `pos`
    qkv = torch.zeros(N, 3 * H * D)
    qkv[pos, : H * D] = x[0].permute(1, 0, 2).reshape(len(pos), H * D)
    qkv[pos, H * D: 2 * H * D] = qkv[pos, : H * D]
    qkv = qkv.to(dev)
    ones = torch.ones(D, device=dev)
    out = torch.zeros(N, 2 * H * D, device=dev)
    eps = torch.finfo(torch.float32).eps
    ops.qk_norm_rope(qkv, ones, ones, table, out, 1, N, H, D, eps)
    # reference: rope(q_norm(x)) with unit gain; the fixture's y = rope(x), and rope is linear per position, so
    # rope(x * r) = y * r with r the per-(row, head) inverse RMS
    y = torch.from_numpy(z["y"])[0]                    # (H, P, D)
    r = torch.rsqrt(x[0].pow(2).mean(-1, keepdim=True) + eps)
    want = (y * r).permute(1, 0, 2).reshape(len(pos), H * D)
    got = out.cpu()[pos]
    assert rel_l2(got[:, : H * D], want) < 2e-6 and rel_l2(got[:, H * D:], want) < 2e-6
    assert float((got[:, : H * D] - want).abs().max()) < 2e-5


@pytest.mark.parametrize("M,blk,valid,K", [(300, 24, 17, 40), (700, 136, 130, 264)])
def test_gemm_tn_blocks_drops_the_padding(dev, M, blk, valid, K):
    """od_gemm_tn_blocks: G's columns in blocks of `blk` with `valid` live ones (the packed SwiGLU projection, 1365 of 1408 twice) — one launch over
    the padded width must equal the two launches on the halves it replaces, rows compacted, padding columns (non-zero here) dropped, bias included."""
    g = torch.Generator().manual_seed(77)
    for dtype in DTYPES:
        G, A = mk((M, 2 * blk), g, dev, dtype), mk((M, K), g, dev, dtype)
        dW, db = torch.zeros(2 * valid, K, device=dev), torch.zeros(2 * valid, device=dev)
        ops.gemm_tn(G, A, dW, n_cols=2 * blk, k_cols=K, dbias=db, n_block=blk, n_valid=valid)
        Gf, Af = G.float().cpu(), A.float().cpu()
        ref = torch.cat([Gf[:, :valid].t() @ Af, Gf[:, blk:blk + valid].t() @ Af])
        refb = torch.cat([Gf[:, :valid].sum(0), Gf[:, blk:blk + valid].sum(0)])
        assert rel_l2(dW, ref) < TOL[dtype] and rel_l2(db, refb) < TOL[dtype]


def test_det_api_refuses_what_it_cannot_do(dev):
    """od_det_*: a flush outside every registered range is an error (not a silent no-op), the table holds a fixed number of ranges (20), and clearing it
    switches the mode off (a later kernel uses the plain float atomic again)."""
    from osu_dreamer_amd import _lib, det
    L = _lib.lib()
    try:
        det.force(True)
        ctx = det.context(dev)
        a, b = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
        ctx.register(a)
        with pytest.raises(_lib.HipKernelError):
            L.od_det_flush(b.data_ptr(), b.numel(), ops._stream(b))
        with pytest.raises(_lib.HipKernelError):
            L.od_det_flush(a.data_ptr() + 4 * 32, 64, ops._stream(a))          # runs past the end of the range
        cap = (L.cdll.od_det_table_bytes() - 8) // 24
        keep = [torch.zeros(8, device=dev) for _ in range(cap - 1)]
        for t in keep:
            ctx.register(t)
        with pytest.raises(_lib.HipKernelError):
            ctx.register(torch.zeros(8, device=dev))                           # one more than the table holds
        assert len(ctx.live_ranges()) == cap                                   # ... and the previous table is still the active one
    finally:
        det.force(None)
    G, A = torch.randn(70, 16).to(dev), torch.randn(70, 8).to(dev)
    dW = torch.zeros(16, 8, device=dev)
    ops.gemm_tn(G, A, dW)                                                      # mode off: lands in dW directly
    assert rel_l2(dW, G.cpu().t() @ A.cpu()) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,L,C,ks", [(2, 77, 64, 5), (1, 33, 128, 3), (3, 130, 512, 5), (2, 9, 40, 9), (1, 200, 512, 7)])
def test_rmsnorm_gate_residual_film_dwconv_equals_the_two_kernels(dev, dtype, B, L, C, ks):
    """The fused triple of round 6 (gate + residual of the attention branch, norm + FiLM of the feed-forward branch, the SwiGLU branch's depthwise
    conv: backbone.py:78-86 + swiglu.py:20) against od_rmsnorm_gate_residual_film followed by od_dwconv: identical xo, h2, inverse RMS values
    and conv output (bit for bit in bf16) — runs shorter and longer than a wave's 32 frames, ragged ends, channel counts below a wave's 512; and with h2 = None."""
    g = torch.Generator().manual_seed(40 + L)
    M = B * L
    x, h = mk((M, C), g, dev, dtype), mk((M, C), g, dev, dtype)
    ssg_a, ssg_b = mk((B, 3 * C), g, dev, scale=0.5), mk((B, 3 * C), g, dev, scale=0.5)
    cw, cb = mk((C, ks), g, dev, scale=0.4), mk((C,), g, dev, scale=0.1)
    xo1, h21, y1 = (torch.zeros(M, C, dtype=dtype, device=dev) for _ in range(3))
    ia1, ib1 = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
    ops.rmsnorm_gate_residual_film(x, h, ssg_a, xo1, ia1, ssg_b, None, False, h21, ib1, B, L)
    ops.dwconv(h21, cw, cb, y1, B, L, ks)
    xo2, h22, y2 = (torch.full((M, C), 7.0, dtype=dtype, device=dev) for _ in range(3))
    ia2, ib2 = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
    ops.rmsnorm_gate_residual_film_dwconv(x, h, ssg_a, xo2, ia2, ssg_b, h22, ib2, cw, cb, y2, B, L, ks)
    def same(a, b_):
        # bf16 tensors: bit for bit (on the emulator fp32 too).  fp32 on the GPU: hipcc contracts the two kernels' multiply-adds differently
        # (fma formation), so the fp32 results agree to rounding, not to the bit
        if dtype == torch.bfloat16 or dev.type == "cpu":
            return torch.equal(a.cpu(), b_.cpu())
        return rel_l2(a, b_.cpu()) < 1e-6
    for a, b_ in ((xo1, xo2), (h21, h22), (y1, y2)):
        assert same(a, b_)
    assert rel_l2(ia1, ia2.cpu()) < 1e-6 and rel_l2(ib1, ib2.cpu()) < 1e-6
    y3 = torch.zeros(M, C, dtype=dtype, device=dev)
    xo3 = torch.zeros(M, C, dtype=dtype, device=dev)
    ops.rmsnorm_gate_residual_film_dwconv(x, h, ssg_a, xo3, ia2, ssg_b, None, ib2, cw, cb, y3, B, L, ks)
    assert same(y3, y1) and same(xo3, xo1)
