"""Thin torch-tensor front end over the C ABI (include/osu_dreamer_hip.h).

Each function forwards raw pointers/sizes to one `od_*` entry point on the current HIP
stream.  Tensors are only storage here: no arithmetic is done with torch ops.
Activations are frame-major 2-D tensors [M = B*L, C] (see csrc/od_common.h).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib
from ._lib import OD_ACT_NONE, OD_ACT_SILU, OD_BF16, OD_EPI_NONE, OD_EPI_SILU, OD_F32, OD_F32X3, OD_F32X3W


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return OD_F32
    if dtype == torch.bfloat16:
        return OD_BF16
    if dtype == torch.float16:          # the operand type of the attention core when the host asks for "attention in fp16" (nothing else takes it)
        return _lib.OD_F16
    raise TypeError(f"unsupported compute dtype {dtype}")


class SplitWeight:
    """A GEMM weight packed by `pack_weight(..., split=True)`: fp32-sized storage holding, per row and per 32-element K slab, the 32 bf16
    high halves then the 32 low halves (OD_F32X3W).  Only the fp32-as-3-x-bf16 GEMMs read it."""

    def __init__(self, storage: torch.Tensor):
        self.t = storage


def mm_code(dtype: torch.dtype, x3: bool, w=None) -> int:
    """Compute-type code of a matrix product: fp32 tensors run as three bf16 MFMAs per product when `x3` (OD_F32X3; OD_F32X3W when
    the weight was pre-split at pack time)."""
    if isinstance(w, SplitWeight):
        if not (x3 and dtype == torch.float32):
            raise ValueError("a pre-split weight can only feed the fp32-as-3-x-bf16 product")
        return OD_F32X3W
    return OD_F32X3 if (x3 and dtype == torch.float32) else dt_code(dtype)


def _w(W):
    return W.t if isinstance(W, SplitWeight) else W


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "need a row-major 2-D view"
    return t.stride(0)


def _f32(*ts):
    for t in ts:
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous()), "expected contiguous fp32"


# ---------------------------------------------------------------- GEMMs
def gemm_nt(A, W, bias, C, epilogue=OD_EPI_NONE, accumulate=False, x3=False):
    code, W = mm_code(A.dtype, x3, W), _w(W)
    M, K = A.shape
    N = W.shape[0]
    assert W.shape[1] == K and tuple(C.shape) == (M, N)
    assert A.dtype == W.dtype == C.dtype
    _f32(bias)
    _lib.lib().od_gemm_nt(code, _p(A), _ld(A), _p(W), _ld(W), _p(bias), _p(C), _ld(C), M, N, K,
                          epilogue, int(accumulate), _stream(A))


def gemm_nt_qkrope(A, W, bias, C, wq, wk, table, L, H, hd, eps, x3=False, q_scale=1.0):
    """qkv projection with q/k RMSNorm + RoPE in the epilogue (forward-only): C[:, :2*H*hd] normed + rotated."""
    code, W = mm_code(A.dtype, x3, W), _w(W)
    M, K = A.shape
    N = W.shape[0]
    assert W.shape[1] == K and tuple(C.shape) == (M, N) and A.dtype == W.dtype == C.dtype
    _f32(bias, wq, wk, table)
    _lib.lib().od_gemm_nt_qkrope(code, _p(A), _ld(A), _p(W), _ld(W), _p(bias), _p(C), _ld(C), M, N, K,
                                 _p(wq), _p(wk), _p(table), L, H, hd, eps, q_scale, _stream(A))


def gemm_nt_qkrope_split(A, W, bias, C, qk_out, wq, wk, table, L, H, hd, eps, x3=False, q_scale=1.0):
    """qkv projection for training: C keeps the pre-norm values, qk_out[:, :2*H*hd] gets q/k normed + rotated."""
    code, W = mm_code(A.dtype, x3, W), _w(W)          # a weight pre-split for the fp32-as-3-x-bf16 product (pack_weights(x3=True)) arrives wrapped
    M, K = A.shape
    N = W.shape[0]
    assert W.shape[1] == K and tuple(C.shape) == (M, N) and tuple(qk_out.shape) == (M, 2 * H * hd)
    assert A.dtype == W.dtype == C.dtype
    # qk_out in torch.float16 = "attention in fp16": the roped q, k AND the v columns of C are then written as IEEE half
    # (C stays a bf16 tensor whose last N - 2*H*hd columns hold half bit patterns: view them with .view(torch.float16))
    assert qk_out.dtype == A.dtype or (qk_out.dtype == torch.float16 and A.dtype == torch.bfloat16)
    _f32(bias, wq, wk, table)
    qk_code = _lib.OD_F16 if qk_out.dtype == torch.float16 else (OD_F32 if A.dtype == torch.float32 else code)
    _lib.lib().od_gemm_nt_qkrope_split(code, _p(A), _ld(A), _p(W), _ld(W), _p(bias), _p(C), _ld(C), _p(qk_out),
                                       _ld(qk_out), qk_code, M, N, K, _p(wq), _p(wk), _p(table), L, H, hd, eps, q_scale, _stream(A))


def gemm_tn(G, A, dW, n_cols=None, k_cols=None, dbias=None, n_block=0, n_valid=0):
    """dW[N,K] += G[:, :N]^T A[:, :K] (and dbias[N] += column sums of G);  dW is any fp32 tensor
    viewed as [N, K] rows.  `n_block` / `n_valid`: G's columns come in blocks of n_block of which the first n_valid are live (the padded
    SwiGLU width); dW / dbias are then the UN-padded (N / n_block * n_valid) rows."""
    M = G.shape[0]
    N = n_cols if n_cols is not None else G.shape[1]
    K = k_cols if k_cols is not None else A.shape[1]
    rows = N if not n_block else (N // n_block) * n_valid
    assert not n_block or N % n_block == 0
    assert dW.dtype == torch.float32 and dW.is_contiguous() and dW.numel() == rows * K
    _f32(dbias)
    _lib.lib().od_gemm_tn_blocks(dt_code(G.dtype), _p(G), _ld(G), _p(A), _ld(A), _p(dW), K, _p(dbias), M, N, K, n_block, n_valid, _stream(G))


def colsum(G, out, n_cols=None):
    M = G.shape[0]
    N = n_cols if n_cols is not None else G.shape[1]
    _f32(out)
    _lib.lib().od_colsum(dt_code(G.dtype), _p(G), _ld(G), _p(out), M, N, _stream(G))


def pack_weight(src, dst, transpose=False, row_map=None):
    """dst (compute dtype, zero padded) from the fp32 master `src` viewed as [N, K].  A `SplitWeight` destination receives the
    pre-split (hi, lo) bf16 layout of OD_F32X3W (K padded to a multiple of 32, no transpose)."""
    N = src.shape[0]
    K = src.numel() // N
    _f32(src)
    code = OD_F32X3W if isinstance(dst, SplitWeight) else dt_code(_w(dst).dtype)
    dst = _w(dst)
    if transpose:
        Kp, Np = dst.shape
    else:
        Np, Kp = dst.shape
    assert dst.is_contiguous()
    _lib.lib().od_pack_weight(code, _p(src), N, K, _p(dst), Np, Kp, int(transpose), _p(row_map), _stream(src))


# ---------------------------------------------------------------- per-sample linears (fp32)
def linear_small(x, W, b, out, pre=None, act=OD_ACT_NONE):
    B, K = x.shape
    N = W.shape[0]
    _f32(x, W, b, out, pre)
    _lib.lib().od_linear_small(_p(x), _p(W), _p(b), _p(out), _p(pre), B, N, K, act, _stream(x))


def linear_small_bwd(x, W, pre, dout, dpre, dW, db, dx, accumulate_dx, act=OD_ACT_NONE):
    B, K = x.shape
    N = W.shape[0]
    _f32(x, W, pre, dout, dpre, dW, db, dx)
    _lib.lib().od_linear_small_bwd(_p(x), _p(W), _p(pre), _p(dout), _p(dpre), _p(dW), _p(db), _p(dx),
                                   int(accumulate_dx), B, N, K, act, _stream(x))


# ---------------------------------------------------------------- boundary / layout
def cl_to_frames(src, dst):
    B, C, L = src.shape
    _f32(src)
    _lib.lib().od_cl_to_frames(dt_code(dst.dtype), _p(src), _p(dst), _ld(dst), B, C, L, _stream(src))


def proj_in(xt, W, bias, x):
    B, E, L = xt.shape
    D = x.shape[1]
    _f32(xt, W, bias)
    _lib.lib().od_proj_in(dt_code(x.dtype), _p(xt), _p(W), _p(bias), _p(x), _ld(x), B, E, L, D, _stream(xt))


def proj_in_bwd(xt, dx, dW, db):
    B, E, L = xt.shape
    D = dx.shape[1]
    _f32(xt, dW, db)
    _lib.lib().od_proj_in_bwd(dt_code(dx.dtype), _p(xt), _p(dx), _ld(dx), _p(dW), _p(db), B, E, L, D, _stream(xt))


def silu(x, y):
    assert x.is_contiguous() and y.is_contiguous()
    _lib.lib().od_silu(dt_code(x.dtype), _p(x), _p(y), x.numel(), _stream(x))


def silu_bwd(x, dy, dx):
    assert x.is_contiguous() and dy.is_contiguous() and dx.is_contiguous()
    _lib.lib().od_silu_bwd(dt_code(x.dtype), _p(x), _p(dy), _p(dx), x.numel(), _stream(x))


# ---------------------------------------------------------------- norm / modulation
def rmsnorm_film(x, ssg, cl, cl_bcast, h, inv_rms, B, L, eps=1e-6):
    C = x.shape[1]
    _f32(ssg, inv_rms)
    _lib.lib().od_rmsnorm_film(dt_code(x.dtype), _p(x), _ld(x), _p(ssg), _p(cl), _ld(cl) if cl is not None else 0,
                               int(cl_bcast), _p(h), _ld(h), _p(inv_rms), B, L, C, eps, _stream(x))


def rmsnorm_film_bwd(x, inv_rms, ssg, dh, dres, dssg, B, L):
    C = x.shape[1]
    _f32(ssg, inv_rms, dssg)
    _lib.lib().od_rmsnorm_film_bwd(dt_code(x.dtype), _p(x), _ld(x), _p(inv_rms), _p(ssg), _p(dh), _ld(dh), _p(dres),
                                   _ld(dres), _p(dssg), B, L, C, _stream(x))


def rmsnorm_gate_residual(x, h, ssg, xo, inv_rms, B, L, eps=1e-6):
    C = x.shape[1]
    _f32(ssg, inv_rms)
    _lib.lib().od_rmsnorm_gate_residual(dt_code(x.dtype), _p(x), _ld(x), _p(h), _ld(h), _p(ssg), _p(xo), _ld(xo),
                                        _p(inv_rms), B, L, C, eps, _stream(x))


def rmsnorm_gate_residual_film(x, h, ssg_a, xo, inv_a, ssg_b, cl, cl_bcast, h2, inv_b, B, L, eps=1e-6):
    """xo = x + rms(h) * gate_a;  h2 = rms(xo) * (1 + scale_b) + shift_b (+ cl)  — forward-only fused pair."""
    _f32(ssg_a, ssg_b, inv_a, inv_b)
    _lib.lib().od_rmsnorm_gate_residual_film(dt_code(x.dtype), _p(x), _ld(x), _p(h), _ld(h), _p(ssg_a), _p(xo), _ld(xo),
                                             _p(inv_a), _p(ssg_b), _p(cl), _ld(cl) if cl is not None else 0, int(cl_bcast),
                                             _p(h2), _ld(h2), _p(inv_b), B, L, x.shape[1], eps, _stream(x))


def rmsnorm_gate_residual_film_dwconv(x, h, ssg_a, xo, inv_a, ssg_b, h2, inv_b, conv_w, conv_b, y, B, L, ksize, eps=1e-6):
    """rmsnorm_gate_residual_film (no cl) and the depthwise conv of the SwiGLU branch in one pass: y = dwconv(h2) + b.  h2 may be None."""
    _f32(ssg_a, ssg_b, inv_a, inv_b, conv_w, conv_b)
    assert xo.data_ptr() != x.data_ptr() and xo.data_ptr() != h.data_ptr()
    _lib.lib().od_rmsnorm_gate_residual_film_dwconv(dt_code(x.dtype), _p(x), _ld(x), _p(h), _ld(h), _p(ssg_a), _p(xo), _ld(xo), _p(inv_a),
                                                    _p(ssg_b), _p(h2), _ld(h2) if h2 is not None else 0, _p(inv_b), _p(conv_w), _p(conv_b),
                                                    _p(y), _ld(y), B, L, x.shape[1], ksize, eps, _stream(x))


def rmsnorm_gate_residual_bwd(h, inv_rms, ssg, dy, dh, dssg, B, L):
    C = h.shape[1]
    _f32(ssg, inv_rms, dssg)
    _lib.lib().od_rmsnorm_gate_residual_bwd(dt_code(h.dtype), _p(h), _ld(h), _p(inv_rms), _p(ssg), _p(dy), _ld(dy),
                                            _p(dh), _ld(dh), _p(dssg), B, L, C, _stream(h))


# ---------------------------------------------------------------- attention
def rope_table(table, L, hd):
    _f32(table)
    _lib.lib().od_rope_table(_p(table), L, hd, _stream(table))


def qk_norm_rope(qkv, wq, wk, table, qk_out, B, L, H, hd, eps, q_scale=1.0):
    _f32(wq, wk, table)
    _lib.lib().od_qk_norm_rope(dt_code(qkv.dtype), _p(qkv), _ld(qkv), _p(wq), _p(wk), _p(table), _p(qk_out),
                               _ld(qk_out), B, L, H, hd, eps, q_scale, _stream(qkv))


def qk_norm_rope_bwd(qkv, wq, wk, table, dqk, dqkv, dwq, dwk, B, L, H, hd, eps, q_scale=1.0):
    _f32(wq, wk, table, dwq, dwk)
    _lib.lib().od_qk_norm_rope_bwd(dt_code(qkv.dtype), _p(qkv), _ld(qkv), _p(wq), _p(wk), _p(table), _p(dqk), _ld(dqk),
                                   _p(dqkv), _ld(dqkv), _p(dwq), _p(dwk), B, L, H, hd, eps, q_scale, _stream(qkv))


def flash_attn_fwd(q, k, v, o, lse, B, H, L, hd, scale, x3=False, q_prescaled=False):
    """q, k, v in torch.float16 (o stays bf16): the half-operand kernel ("attention in fp16")."""
    _f32(lse)
    assert q.dtype == k.dtype == v.dtype and (o.dtype == q.dtype or (q.dtype == torch.float16 and o.dtype == torch.bfloat16))
    _lib.lib().od_flash_attn_fwd(mm_code(q.dtype, x3), _p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(o), _ld(o), _p(lse),
                                 B, H, L, hd, scale, int(q_prescaled), _stream(q))


class AttnAux:
    """The side stream + events of the two-stream attention backward (od_attn_aux_create); owned by whoever plans the launches."""

    def __init__(self):
        import ctypes
        self.handle = ctypes.c_void_p()
        _lib.lib().od_attn_aux_create(ctypes.byref(self.handle))

    def close(self):
        if self.handle:
            _lib.lib().od_attn_aux_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def flash_attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, B, H, L, hd, scale, q_prescaled=False, aux=None):
    _f32(lse, delta)
    _lib.lib().od_flash_attn_bwd_aux(dt_code(q.dtype), _p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(o), _ld(o), _p(do),
                                     _ld(do), _p(lse), _p(delta), _p(dq), _ld(dq), _p(dk), _ld(dk), _p(dv), _ld(dv),
                                     B, H, L, hd, scale, int(q_prescaled), aux.handle if aux is not None else None, _stream(q))


class FusedAttnBwdWorkspace:
    """Caller-owned device memory of od_flash_attn_bwd_fused for one (B, H, L): control block + chain flags (zeroed once here; every call
    leaves them zero), the start values (-lse', -delta) and the running dQ tiles.  One instance serves every layer of a step (same stream)."""

    def __init__(self, B, H, L, device, dtype=torch.bfloat16):
        """`dtype` torch.float16: room for the staged half copy of dO as well ("attention in fp16")."""
        import ctypes
        total, zero = ctypes.c_long(), ctypes.c_long()
        _lib.lib().od_flash_attn_bwd_fused_ws_bytes(dt_code(dtype), B, H, L, ctypes.byref(total), ctypes.byref(zero))
        self.shape = (B, H, L)
        self.dtype = dtype
        self.bytes = total.value
        assert zero.value <= self.bytes
        self.zero_bytes = zero.value
        self.buf = torch.empty(self.bytes, dtype=torch.uint8, device=device)
        self.reset()

    def reset(self):
        """Zero the control block and the running tiles (a zero tile carries write number 0): once at creation, and again after a status-3 launch."""
        self.buf[:self.zero_bytes].zero_()

    def err_ptr(self) -> int:
        """Device address of the sticky error word: ops.sqnorm / ops.adamw_ema fold it into the step without a host round trip."""
        return self.buf.data_ptr() + int(_lib.lib().cdll.od_flash_attn_bwd_fused_err_offset())

    def err_view(self) -> torch.Tensor:
        """The sticky error word as an int32[1] tensor (a view of the workspace): what a data-parallel step reduces over the ranks."""
        off = int(_lib.lib().cdll.od_flash_attn_bwd_fused_err_offset())
        return self.buf[off:off + 4].view(torch.int32)

    def status(self) -> int:
        """The sticky error word (0 = every launch processed all of its jobs; 1 / 2 / 3: see the header).  Waits for the current stream."""
        import ctypes
        err = ctypes.c_int(-1)
        _lib.lib().od_flash_attn_bwd_fused_status(_p(self.buf), ctypes.byref(err), _stream(self.buf))
        return err.value


def flash_attn_bwd_fused(q, k, v, o, do, lse, dq, dk, dv, B, H, L, hd, scale, ws: FusedAttnBwdWorkspace, q_prescaled=False):
    """The 5-pass fused attention backward (bf16, head_dim 64): dq, dk, dv from one kernel, dQ summed over key blocks by the L2 chain.
    q, k, v in torch.float16: the half-operand form (o, do, dq, dk, dv stay bf16; the workspace must have been made for float16)."""
    _f32(lse)
    assert ws.shape == (B, H, L), (ws.shape, (B, H, L))
    assert q.dtype == k.dtype == v.dtype and o.dtype == do.dtype == dq.dtype == dk.dtype == dv.dtype == torch.bfloat16
    assert q.dtype == torch.bfloat16 or (q.dtype == torch.float16 and ws.dtype == torch.float16)
    _lib.lib().od_flash_attn_bwd_fused(dt_code(q.dtype), _p(q), _ld(q), _p(k), _ld(k), _p(v), _ld(v), _p(o), _ld(o), _p(do), _ld(do),
                                       _p(lse), _p(dq), _ld(dq), _p(dk), _ld(dk), _p(dv), _ld(dv), B, H, L, hd, scale, int(q_prescaled),
                                       _p(ws.buf), ws.bytes, _stream(q))


# ---------------------------------------------------------------- feed-forward
def dwconv(x, w, bias, y, B, L, ksize):
    C = x.shape[1]
    _f32(w, bias)
    _lib.lib().od_dwconv(dt_code(x.dtype), _p(x), _ld(x), _p(w), _p(bias), _p(y), _ld(y), B, L, C, ksize, _stream(x))


def dwconv_bwd(x, w, dy, dx, dw, db, B, L, ksize):
    C = x.shape[1]
    _f32(w, dw, db)
    _lib.lib().od_dwconv_bwd(dt_code(x.dtype), _p(x), _ld(x), _p(w), _p(dy), _ld(dy), _p(dx), _ld(dx), _p(dw), _p(db),
                             B, L, C, ksize, _stream(x))


def scale_channels(x, scale, B, L):
    """x[(b, l), c] *= scale[b, c] in place (Dropout1d's channel mask; also its backward)."""
    _f32(scale)
    assert scale.shape == (B, x.shape[1]) and scale.is_contiguous()
    _lib.lib().od_scale_channels(dt_code(x.dtype), _p(x), _ld(x), _p(scale), B, L, x.shape[1], _stream(x))


def swiglu_rmsnorm(vg, hh, inv_rms, Hf, Hp, eps=1e-6):
    _f32(inv_rms)
    _lib.lib().od_swiglu_rmsnorm(dt_code(vg.dtype), _p(vg), _ld(vg), _p(hh), _ld(hh), _p(inv_rms), vg.shape[0], Hf, Hp,
                                 eps, _stream(vg))


def swiglu_rmsnorm_bwd(vg, inv_rms, dhh, dvg, Hf, Hp):
    _f32(inv_rms)
    _lib.lib().od_swiglu_rmsnorm_bwd(dt_code(vg.dtype), _p(vg), _ld(vg), _p(inv_rms), _p(dhh), _ld(dhh), _p(dvg),
                                     _ld(dvg), vg.shape[0], Hf, Hp, _stream(vg))


# ---------------------------------------------------------------- heads
def final_norm_proj_out(x, W, bias, v, inv_rms, B, L, eps=1e-6):
    C = x.shape[1]
    E = W.shape[0]
    _f32(W, bias, v, inv_rms)
    _lib.lib().od_final_norm_proj_out(dt_code(x.dtype), _p(x), _ld(x), _p(W), _p(bias), _p(v), _p(inv_rms), B, L, C, E,
                                      eps, _stream(x))


def final_norm_proj_out_bwd(x, inv_rms, W, dv, dx, dW, db, B, L):
    C = x.shape[1]
    E = W.shape[0]
    _f32(W, dv, inv_rms, dW, db)
    _lib.lib().od_final_norm_proj_out_bwd(dt_code(x.dtype), _p(x), _ld(x), _p(inv_rms), _p(W), _p(dv), _p(dx), _ld(dx),
                                          _p(dW), _p(db), B, L, C, E, _stream(x))


def uhead_fwd(xt, w, fsum, U):
    """w: the eight u_head tensors (w0,b0,w1,b1,w3,b3,w4,b4)."""
    B, E, L = xt.shape
    _f32(xt, fsum, *w)
    _lib.lib().od_uhead_fwd(_p(xt), *[_p(t) for t in w], _p(fsum), B, E, L, U, _stream(xt))


def uhead_bwd(xt, w, dfm, g, U):
    B, E, L = xt.shape
    _f32(xt, dfm, *w, *g)
    _lib.lib().od_uhead_bwd(_p(xt), *[_p(t) for t in w], _p(dfm), *[_p(t) for t in g], B, E, L, U, _stream(xt))


def uhead_tail(fsum, mod, w_out, b_out, u, L, u_scale):
    B, U = fsum.shape
    _f32(fsum, mod, w_out, b_out, u)
    _lib.lib().od_uhead_tail(_p(fsum), _p(mod), _p(w_out), _p(b_out), _p(u), B, U, L, u_scale, _stream(fsum))


def uhead_tail_bwd(fsum, mod, w_out, b_out, du, dfm, dmod, dw_out, db_out, L, u_scale):
    B, U = fsum.shape
    _f32(fsum, mod, w_out, b_out, du, dfm, dmod, dw_out, db_out)
    _lib.lib().od_uhead_tail_bwd(_p(fsum), _p(mod), _p(w_out), _p(b_out), _p(du), _p(dfm), _p(dmod), _p(dw_out),
                                 _p(db_out), B, U, L, u_scale, _stream(fsum))


# ---------------------------------------------------------------- loss / sampler
def make_xt(x0, x1, t, xt, dsq):
    B, E, L = x0.shape
    _f32(x0, x1, t, xt, dsq)
    _lib.lib().od_make_xt(_p(x0), _p(x1), _p(t), _p(xt), _p(dsq), B, E, L, _stream(x0))


def loss_grad(xt, x1, u, v, dsq, dv, sums, c0, osl_w, del_w):
    B, E, L = xt.shape
    _f32(xt, x1, u, v, dsq, dv, sums)
    _lib.lib().od_loss_grad(_p(xt), _p(x1), _p(u), _p(v), _p(dsq), _p(dv), _p(sums), B, E, L, c0, osl_w, del_w, _stream(xt))


def loss_finalize(sums, dsq, u, out, du, c0, osl_w, del_w):
    B = u.shape[0]
    _f32(sums, dsq, u, out, du)
    _lib.lib().od_loss_finalize(_p(sums), _p(dsq), _p(u), _p(out), _p(du), B, c0, osl_w, del_w, _stream(u))


def sampler_step(x, u, v, eta):
    B, E, L = x.shape
    _f32(x, u, v, eta)
    _lib.lib().od_sampler_step(_p(x), _p(u), _p(v), _p(eta), B, E, L, _stream(x))


def sampler_eta(u, eta, c0, num_steps):
    _f32(u, eta)
    _lib.lib().od_sampler_eta(_p(u), _p(eta), u.shape[0], c0, num_steps, _stream(u))


# ---------------------------------------------------------------- optimizer
def sqnorm(g, out, status_ptr: int = 0):
    """out[0] += sum g^2; `status_ptr` (a device address, 0 = none): non-zero error word -> out[0] = NaN, on the device."""
    _f32(g, out)
    _lib.lib().od_sqnorm(_p(g), g.numel(), _p(out), status_ptr or None, _stream(g))


def adamw_ema(p, g, m, v, ema, lr, beta1, beta2, eps, weight_decay, step, ema_decay, ema_mode, gnorm_sq, max_norm, status_ptr: int = 0):
    _f32(p, g, m, v, ema, gnorm_sq)
    _lib.lib().od_adamw_ema(_p(p), _p(g), _p(m), _p(v), _p(ema), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                            ema_decay, ema_mode, _p(gnorm_sq), max_norm, status_ptr or None, _stream(p))


def ema_update(ema, p, decay, mode):
    _f32(ema, p)
    _lib.lib().od_ema_update(_p(ema), _p(p), p.numel(), decay, mode, _stream(p))


# ---------------------------------------------------------------- style model
def style_conditioning(labels, rff_w, rff_b, cond_w, cond_b, null_labels, c):
    B, NL = labels.shape
    F, H = cond_w.shape[1], cond_w.shape[2]
    _f32(labels, rff_w, rff_b, cond_w, cond_b, null_labels, c)
    _lib.lib().od_style_conditioning(_p(labels), _p(rff_w), _p(rff_b), _p(cond_w), _p(cond_b), _p(null_labels), _p(c),
                                     B, NL, F, H, _stream(labels))


def rmsnorm_rows(x, gamma, y, eps):
    M, C = x.shape
    _f32(x, gamma, y)
    _lib.lib().od_rmsnorm_rows(_p(x), _p(gamma), _p(y), M, C, eps, _stream(x))


# ---------------------------------------------------------------- latent model, inference path
def spec_features_conv(audio, w1, b1, g1, w2, b2, g2, out, eps=1e-6):
    B, F, L = audio.shape
    _f32(audio, w1, b1, g1, w2, b2, g2)
    _lib.lib().od_spec_features_conv(dt_code(out.dtype), _p(audio), _p(w1), _p(b1), _p(g1), _p(w2), _p(b2), _p(g2),
                                     _p(out), _ld(out), B, F, L, eps, _stream(audio))


def rmsnorm_affine_film(x, gamma, ssg, y, B, L, act=OD_ACT_NONE, eps=1e-6):
    _f32(gamma, ssg)
    _lib.lib().od_rmsnorm_affine_film(dt_code(x.dtype), _p(x), _ld(x), _p(gamma), _p(ssg), _p(y), _ld(y), B, L, x.shape[1],
                                      eps, act, _stream(x))


def rmsnorm_affine_gate_residual(x, h, gamma, ssg, xo, B, L, eps=1e-6):
    _f32(gamma, ssg)
    _lib.lib().od_rmsnorm_affine_gate_residual(dt_code(x.dtype), _p(x), _ld(x), _p(h), _ld(h), _p(gamma), _p(ssg), _p(xo),
                                               _ld(xo), B, L, x.shape[1], eps, _stream(x))


def unet_mixer(x, p, p_bcast, gx, gamma, xo, B, L, eps=1e-6):
    _f32(gamma)
    _lib.lib().od_unet_mixer(dt_code(x.dtype), _p(x), _ld(x), _p(p), _ld(p), int(p_bcast), _p(gx), _ld(gx), _p(gamma),
                             _p(xo), _ld(xo), B, L, x.shape[1], eps, _stream(x))


def unet_down(x, w, bias, y, B, Lo, stride):
    _f32(w, bias)
    _lib.lib().od_unet_down(dt_code(x.dtype), _p(x), _ld(x), _p(w), _p(bias), _p(y), _ld(y), B, Lo, x.shape[1], stride,
                            _stream(x))


def unet_up(x, w, bias, y, B, Li, stride):
    _f32(w, bias)
    _lib.lib().od_unet_up(dt_code(x.dtype), _p(x), _ld(x), _p(w), _p(bias), _p(y), _ld(y), B, Li, x.shape[1], stride,
                          _stream(x))


def chart_head(x, W, bias, out, B, L, n_sigmoid, rms=False, eps=1e-6):
    _f32(W, bias, out)
    N = W.shape[0]
    _lib.lib().od_chart_head(dt_code(x.dtype), _p(x), _ld(x), _p(W), _p(bias), _p(out), B, L, x.shape[1], N, n_sigmoid,
                             int(rms), eps, _stream(x))


def attn_pool(scores, values, out, B, L, heads, hd):
    _f32(out)
    _lib.lib().od_attn_pool(dt_code(scores.dtype), _p(scores), _ld(scores), _p(values), _ld(values), _p(out), B, L, heads, hd,
                            _stream(scores))
