// Per-frame (row) kernels: channel RMS norm + adaLN modulation, gate/residual, SwiGLU
// gate + RMS norm, final norm + proj_out, q/k RMSNorm + RoPE — forward and backward.
// One wavefront owns a frame: channels are contiguous in the frame-major layout, so a
// lane reads 8 consecutive channels (16 B bf16 / 32 B f32) and the per-frame reduction is
// one 64-lane shuffle tree.  HBM-bound; algorithmic bytes are listed in DESIGN.md.
#include "od_common.h"
#include "od_api_internal.h"

namespace {

#ifndef OD_QK_POS
#define OD_QK_POS 1       // q/k norm + RoPE: the position-major kernel where its shape conditions hold (0 = always the generic kernel)
#endif
#ifndef OD_QK_BPW
#define OD_QK_BPW 8       // batch rows a wave of the position-major kernel handles per position
#endif
constexpr int ROWS_PER_WAVE_BWD = 16;  // rows a wave walks in kernels that also reduce over frames

// load a row chunk-wise into registers: lane owns chunks lane, lane+64, ... of 8 channels
template <class T, int NCH>
__device__ __forceinline__ void load_row(const T* row, int C, int lane, float (&v)[NCH][8]) {
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = (lane + 64 * i) * 8;
        if (c < C) od_ld8(row + c, v[i]);
        else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[i][e] = 0.f;
        }
    }
}
template <int NCH>
__device__ __forceinline__ float row_sumsq(const float (&v)[NCH][8]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) s += v[i][e] * v[i][e];
    return od_wave_sum(s);
}

// ------------------------------------------------------------------ rmsnorm + FiLM (+cl)
template <class T, int NCH>
__global__ __launch_bounds__(256) void rmsnorm_film_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ ssg,
                                                           const T* __restrict__ cl, int ldcl, int cl_bcast,
                                                           T* __restrict__ h, int ldh, float* __restrict__ inv_rms,
                                                           int B, int L, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= (long)B * L) return;
    const int b = (int)(m / L), l = (int)(m % L);
    float v[NCH][8];
    load_row<T, NCH>(x + m * ldx, C, lane, v);
    const float inv = rsqrtf(row_sumsq<NCH>(v) / (float)C + eps);
    if (lane == 0 && inv_rms) inv_rms[m] = inv;
    const float* sc = ssg + (size_t)b * 3 * C;
    const T* clrow = cl ? cl + (size_t)(cl_bcast ? l : m) * ldcl : nullptr;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = (lane + 64 * i) * 8;
        if (c >= C) continue;
        float s[8], sh[8], o[8];
        od_ld8(sc + c, s); od_ld8(sc + C + c, sh);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = v[i][e] * inv * (1.f + s[e]) + sh[e];
        if (clrow) {
            float a[8]; od_ld8(clrow + c, a);
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] += a[e];
        }
        od_st8(h + m * ldh + c, o);
    }
}

// backward: dres += inv*(g - xhat*mean(g*xhat)), g = dh*(1+scale); dscale += dh*xhat; dshift += dh
template <class T, int NCH>
__global__ __launch_bounds__(256) void rmsnorm_film_bwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ inv_rms,
                                                               const float* __restrict__ ssg, const T* __restrict__ dh, int lddh,
                                                               T* __restrict__ dres, int lddres, float* __restrict__ dssg,
                                                               int B, int L, int C, const OdDetTable* __restrict__ det) {
    __shared__ float red[2][NCH * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int l0 = (blockIdx.x * 4 + wave) * ROWS_PER_WAVE_BWD;
    float dsc[NCH][8], dsh[NCH][8], scl[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = (lane + 64 * i) * 8;
#pragma unroll
        for (int e = 0; e < 8; e++) { dsc[i][e] = 0.f; dsh[i][e] = 0.f; scl[i][e] = 0.f; }
        if (c < C) od_ld8(ssg + (size_t)b * 3 * C + c, scl[i]);
    }
    for (int r = 0; r < ROWS_PER_WAVE_BWD; r++) {
        const int l = l0 + r;
        if (l >= L) break;
        const long m = (long)b * L + l;
        float xv[NCH][8], gv[NCH][8];
        load_row<T, NCH>(x + m * ldx, C, lane, xv);
        load_row<T, NCH>(dh + m * lddh, C, lane, gv);
        const float inv = inv_rms[m];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; i++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float xh = xv[i][e] * inv;
                dsc[i][e] += gv[i][e] * xh;
                dsh[i][e] += gv[i][e];
                gv[i][e] *= (1.f + scl[i][e]);
                dot += gv[i][e] * xh;
                xv[i][e] = xh;
            }
        dot = od_wave_sum(dot) / (float)C;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = (lane + 64 * i) * 8;
            if (c >= C) continue;
            float o[8];
            od_ld8(dres + m * lddres + c, o);
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] += inv * (gv[i][e] - xv[i][e] * dot);
            od_st8(dres + m * lddres + c, o);
        }
    }
    // cross-wave reduction of the per-(b,c) sums, then one atomic per channel per block
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NCH; i++)
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int idx = (lane + 64 * i) * 8 + e;
                    if (w == 0) { red[0][idx] = dsc[i][e]; red[1][idx] = dsh[i][e]; }
                    else { red[0][idx] += dsc[i][e]; red[1][idx] += dsh[i][e]; }
                }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        od_red_add(det, dssg + (size_t)b * 3 * C + c, red[0][c]);
        od_red_add(det, dssg + (size_t)b * 3 * C + C + c, red[1][c]);
    }
}

// ------------------------------------------------------------------ rmsnorm * gate + residual
template <class T, int NCH>
__global__ __launch_bounds__(256) void rmsnorm_gate_res_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ h, int ldh,
                                                               const float* __restrict__ ssg, T* __restrict__ xo, int ldxo,
                                                               float* __restrict__ inv_rms, int B, int L, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= (long)B * L) return;
    const int b = (int)(m / L);
    float v[NCH][8];
    load_row<T, NCH>(h + m * ldh, C, lane, v);
    const float inv = rsqrtf(row_sumsq<NCH>(v) / (float)C + eps);
    if (lane == 0 && inv_rms) inv_rms[m] = inv;
    const float* gate = ssg + (size_t)b * 3 * C + 2 * C;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = (lane + 64 * i) * 8;
        if (c >= C) continue;
        float g[8], o[8];
        od_ld8(gate + c, g);
        od_ld8(x + m * ldx + c, o);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] += v[i][e] * inv * g[e];
        od_st8(xo + m * ldxo + c, o);
    }
}

// fused pair (forward only): xo = x + rms_norm(h) * gate_a, then h2 = rms_norm(xo) * (1 + scale_b) + shift_b (+ cl)
// — the gate/residual that closes one branch and the norm + FiLM that opens the next (backbone.py:78-86) in
// one pass over the frame.  xo is rounded to T before the second norm, so the result equals the two-kernel path.
template <class T, int NCH>
__global__ __launch_bounds__(256) void rmsnorm_gate_res_film_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ h, int ldh,
                                                                    const float* __restrict__ ssg_a, T* __restrict__ xo, int ldxo,
                                                                    float* __restrict__ inv_a, const float* __restrict__ ssg_b,
                                                                    const T* __restrict__ cl, int ldcl, int cl_bcast,
                                                                    T* __restrict__ h2, int ldh2, float* __restrict__ inv_b,
                                                                    int B, int L, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= (long)B * L) return;
    const int b = (int)(m / L), l = (int)(m % L);
    float v[NCH][8];
    load_row<T, NCH>(h + m * ldh, C, lane, v);
    const float inv = rsqrtf(row_sumsq<NCH>(v) / (float)C + eps);
    if (lane == 0 && inv_a) inv_a[m] = inv;
    const float* gate = ssg_a + (size_t)b * 3 * C + 2 * C;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = (lane + 64 * i) * 8;
        if (c >= C) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[i][e] = 0.f;
            continue;
        }
        float g[8], o[8];
        od_ld8(gate + c, g);
        od_ld8(x + m * ldx + c, o);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] += v[i][e] * inv * g[e];
        od_st8(xo + m * ldxo + c, o);
#pragma unroll
        for (int e = 0; e < 8; e++) v[i][e] = od_round_to<T>(o[e]);
    }
    const float inv2 = rsqrtf(row_sumsq<NCH>(v) / (float)C + eps);
    if (lane == 0 && inv_b) inv_b[m] = inv2;
    const float* sc = ssg_b + (size_t)b * 3 * C;
    const T* clrow = cl ? cl + (size_t)(cl_bcast ? l : m) * ldcl : nullptr;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = (lane + 64 * i) * 8;
        if (c >= C) continue;
        float s2[8], sh[8], o[8];
        od_ld8(sc + c, s2); od_ld8(sc + C + c, sh);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = v[i][e] * inv2 * (1.f + s2[e]) + sh[e];
        if (clrow) {
            float a[8]; od_ld8(clrow + c, a);
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] += a[e];
        }
        od_st8(h2 + m * ldh2 + c, o);
    }
}

// fused triple (forward, round 6): the pair above AND the depthwise Conv1d that opens the SwiGLU branch (swiglu.py:20): a wave walks a RUN of
// consecutive frames of one batch row (a lane = 8 channels, as everywhere here), forms xo / h2 frame by frame and keeps the last KS rows of h2 in
// a register window (the dwconv kernel's sliding window): y[l] = bias + sum_j w[j] h2[l - R + j] leaves as soon as frame l + R exists.  The R
// frames either side of the run are recomputed (not stored): 2 R / RUN more reads of x and h.  h2 is rounded to the tensor type before it
// enters the window, and the taps are summed in dwconv_kernel's order: the results are bit-identical to od_rmsnorm_gate_residual_film followed
// by od_dwconv.  h2 may be NULL (inference: nothing reads it again); training keeps it for the conv's weight gradient.  C <= 512.
template <class T, int KS, int RUN>
__global__ __launch_bounds__(256) void rmsnorm_gate_res_film_dwconv_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ h, int ldh,
                                                                           const float* __restrict__ ssg_a, T* __restrict__ xo, int ldxo,
                                                                           float* __restrict__ inv_a, const float* __restrict__ ssg_b,
                                                                           T* __restrict__ h2, int ldh2, float* __restrict__ inv_b,
                                                                           const float* __restrict__ cw, const float* __restrict__ cb,
                                                                           T* __restrict__ y, int ldy, int B, int L, int C, float eps) {
    constexpr int R = KS / 2;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int l0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RUN;
    if (l0 >= L) return;
    const int c = lane * 8;
    const bool act = c < C;
    float gate[8], sc[8], sh[8], wv[8][KS], bv[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        gate[e] = 0.f; sc[e] = 0.f; sh[e] = 0.f; bv[e] = 0.f;
#pragma unroll
        for (int j = 0; j < KS; j++) wv[e][j] = 0.f;
    }
    if (act) {
        od_ld8(ssg_a + (size_t)b * 3 * C + 2 * C + c, gate);
        od_ld8(ssg_b + (size_t)b * 3 * C + c, sc);
        od_ld8(ssg_b + (size_t)b * 3 * C + C + c, sh);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            bv[k] = cb[c + k];
#pragma unroll
            for (int j = 0; j < KS; j++) wv[k][j] = cw[(size_t)(c + k) * KS + j];
        }
    }
    float win[KS][8];
#pragma unroll
    for (int j = 0; j < KS; j++)
#pragma unroll
        for (int e = 0; e < 8; e++) win[j][e] = 0.f;
    const int lend = l0 + RUN < L ? l0 + RUN : L;                  // frames [l0, lend) are this wave's
    for (int l = l0 - R; l < lend + R; l++) {
#pragma unroll
        for (int e = 0; e < 8; e++) win[KS - 1][e] = 0.f;          // frames outside the sequence: the conv's zero padding
        if (l >= 0 && l < L) {                                     // (wave-uniform)
            const long m = (long)b * L + l;
            const bool own = l >= l0 && l < lend;
            float v[1][8], o[8];
            load_row<T, 1>(h + m * ldh, C, lane, v);
            const float inv = rsqrtf(row_sumsq<1>(v) / (float)C + eps);
            if (act) {
                od_ld8(x + m * ldx + c, o);
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] += v[0][e] * inv * gate[e];
                if (own) od_st8(xo + m * ldxo + c, o);
#pragma unroll
                for (int e = 0; e < 8; e++) v[0][e] = od_round_to<T>(o[e]);
            }
            const float inv2 = rsqrtf(row_sumsq<1>(v) / (float)C + eps);
            if (own && lane == 0) {
                if (inv_a) inv_a[m] = inv;
                if (inv_b) inv_b[m] = inv2;
            }
            if (act) {
                float hv[8];
#pragma unroll
                for (int e = 0; e < 8; e++) hv[e] = v[0][e] * inv2 * (1.f + sc[e]) + sh[e];
                if (own && h2) od_st8(h2 + m * ldh2 + c, hv);
#pragma unroll
                for (int e = 0; e < 8; e++) win[KS - 1][e] = od_round_to<T>(hv[e]);
            }
        }
        const int lc = l - R;                                      // the window now holds frames lc - R .. lc + R
        if (lc >= l0 && lc < lend && act) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float s = bv[k];
#pragma unroll
                for (int j = 0; j < KS; j++) s += wv[k][j] * win[j][k];
                o[k] = s;
            }
            od_st8(y + ((size_t)b * L + lc) * ldy + c, o);
        }
#pragma unroll
        for (int j = 0; j < KS - 1; j++)
#pragma unroll
            for (int e = 0; e < 8; e++) win[j][e] = win[j + 1][e];
    }
}

template <class T, int NCH>
__global__ __launch_bounds__(256) void rmsnorm_gate_res_bwd_kernel(const T* __restrict__ h, int ldh, const float* __restrict__ inv_rms,
                                                                   const float* __restrict__ ssg, const T* __restrict__ dy, int lddy,
                                                                   T* __restrict__ dh, int lddh, float* __restrict__ dssg,
                                                                   int B, int L, int C, const OdDetTable* __restrict__ det) {
    __shared__ float red[NCH * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int l0 = (blockIdx.x * 4 + wave) * ROWS_PER_WAVE_BWD;
    float dg[NCH][8], gate[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = (lane + 64 * i) * 8;
#pragma unroll
        for (int e = 0; e < 8; e++) { dg[i][e] = 0.f; gate[i][e] = 0.f; }
        if (c < C) od_ld8(ssg + (size_t)b * 3 * C + 2 * C + c, gate[i]);
    }
    for (int r = 0; r < ROWS_PER_WAVE_BWD; r++) {
        const int l = l0 + r;
        if (l >= L) break;
        const long m = (long)b * L + l;
        float hv[NCH][8], gv[NCH][8];
        load_row<T, NCH>(h + m * ldh, C, lane, hv);
        load_row<T, NCH>(dy + m * lddy, C, lane, gv);
        const float inv = inv_rms[m];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; i++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float hh = hv[i][e] * inv;
                dg[i][e] += gv[i][e] * hh;
                gv[i][e] *= gate[i][e];
                dot += gv[i][e] * hh;
                hv[i][e] = hh;
            }
        dot = od_wave_sum(dot) / (float)C;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = (lane + 64 * i) * 8;
            if (c >= C) continue;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = inv * (gv[i][e] - hv[i][e] * dot);
            od_st8(dh + m * lddh + c, o);
        }
    }
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NCH; i++)
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int idx = (lane + 64 * i) * 8 + e;
                    if (w == 0) red[idx] = dg[i][e]; else red[idx] += dg[i][e];
                }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < C; c += 256) od_red_add(det, dssg + (size_t)b * 3 * C + 2 * C + c, red[c]);
}

// ------------------------------------------------------------------ SwiGLU gate + RMS norm over Hf
template <class T, int NCH>
__global__ __launch_bounds__(256) void swiglu_rmsnorm_kernel(const T* __restrict__ vg, int ldvg, T* __restrict__ hh, int ldhh,
                                                             float* __restrict__ inv_rms, long M, int Hf, int Hp, float eps) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float v[NCH][8], g[NCH][8];
    load_row<T, NCH>(vg + m * ldvg, Hp, lane, v);
    load_row<T, NCH>(vg + m * ldvg + Hp, Hp, lane, g);
#pragma unroll
    for (int i = 0; i < NCH; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) v[i][e] *= od_silu(g[i][e]);
    const float inv = rsqrtf(row_sumsq<NCH>(v) / (float)Hf + eps);
    if (lane == 0 && inv_rms) inv_rms[m] = inv;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = (lane + 64 * i) * 8;
        if (c >= Hp) continue;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = v[i][e] * inv;
        od_st8(hh + m * ldhh + c, o);
    }
}

template <class T, int NCH>
__global__ __launch_bounds__(256) void swiglu_rmsnorm_bwd_kernel(const T* __restrict__ vg, int ldvg, const float* __restrict__ inv_rms,
                                                                 const T* __restrict__ dhh, int lddhh, T* __restrict__ dvg, int lddvg,
                                                                 long M, int Hf, int Hp) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float v[NCH][8], g[NCH][8], d[NCH][8];
    load_row<T, NCH>(vg + m * ldvg, Hp, lane, v);
    load_row<T, NCH>(vg + m * ldvg + Hp, Hp, lane, g);
    load_row<T, NCH>(dhh + m * lddhh, Hp, lane, d);
    const float inv = inv_rms[m];
    // one sigmoid per element (an exp and a reciprocal, both quarter rate) serves silu in the row's dot product, silu again in dv and its
    // derivative in dg: the kernel is not purely HBM-bound — recomputing them took a third of its time.  g is overwritten by sigmoid(g)
    // and v by v * silu(g) * inv (the normalised output), the raw values being recovered where needed.
    float dot = 0.f;
    float sg[NCH][8], gs[NCH][8];       // silu(g), sigmoid(g)
#pragma unroll
    for (int i = 0; i < NCH; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            gs[i][e] = od_sigmoid(g[i][e]);
            sg[i][e] = g[i][e] * gs[i][e];
            dot += d[i][e] * (v[i][e] * sg[i][e] * inv);
        }
    dot = od_wave_sum(dot) / (float)Hf;
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int c = (lane + 64 * i) * 8;
        if (c >= Hp) continue;
        float ov[8], og[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float sh = v[i][e] * sg[i][e] * inv;
            const float ds = inv * (d[i][e] - sh * dot);
            ov[e] = ds * sg[i][e];
            og[e] = ds * v[i][e] * (gs[i][e] * (1.0f + g[i][e] * (1.0f - gs[i][e])));      // d/dg [g sigmoid(g)]
        }
        od_st8(dvg + m * lddvg + c, ov);
        od_st8(dvg + m * lddvg + Hp + c, og);
    }
}

// ------------------------------------------------------------------ final rms_norm + proj_out (C -> E<=8)
template <class T, int NCH>
__global__ __launch_bounds__(256) void final_proj_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ vout,
                                                         float* __restrict__ inv_rms, int B, int L, int C, int E, float eps) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= (long)B * L) return;
    const int b = (int)(m / L), l = (int)(m % L);
    float v[NCH][8];
    load_row<T, NCH>(x + m * ldx, C, lane, v);
    const float inv = rsqrtf(row_sumsq<NCH>(v) / (float)C + eps);
    if (lane == 0 && inv_rms) inv_rms[m] = inv;
    for (int e = 0; e < E; e++) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = (lane + 64 * i) * 8;
            if (c >= C) continue;
            float w[8]; od_ld8(W + (size_t)e * C + c, w);
#pragma unroll
            for (int k = 0; k < 8; k++) s += w[k] * v[i][k];
        }
        s = od_wave_sum(s) * inv;
        if (lane == 0) vout[((size_t)b * E + e) * L + l] = s + bias[e];
    }
}

template <class T, int NCH>
__global__ __launch_bounds__(256) void final_proj_bwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ inv_rms,
                                                             const float* __restrict__ W, const float* __restrict__ dv,
                                                             T* __restrict__ dx, int lddx, float* __restrict__ dW,
                                                             float* __restrict__ db, int B, int L, int C, int E,
                                                             const OdDetTable* __restrict__ det) {
    constexpr int MAXE = 8;
    __shared__ float red[MAXE * NCH * 512];
    __shared__ float redb[4][MAXE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int l0 = (blockIdx.x * 4 + wave) * ROWS_PER_WAVE_BWD;
    float wv[MAXE][NCH][8], dw[MAXE][NCH][8], dbv[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; e++) {
        dbv[e] = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = (lane + 64 * i) * 8;
#pragma unroll
            for (int k = 0; k < 8; k++) { wv[e][i][k] = 0.f; dw[e][i][k] = 0.f; }
            if (e < E && c < C) od_ld8(W + (size_t)e * C + c, wv[e][i]);
        }
    }
    for (int r = 0; r < ROWS_PER_WAVE_BWD; r++) {
        const int l = l0 + r;
        if (l >= L) break;
        const long m = (long)b * L + l;
        float xv[NCH][8], dn[NCH][8], dve[MAXE];
        load_row<T, NCH>(x + m * ldx, C, lane, xv);
        const float inv = inv_rms[m];
#pragma unroll
        for (int e = 0; e < MAXE; e++) { dve[e] = e < E ? dv[((size_t)b * E + e) * L + l] : 0.f; dbv[e] += dve[e]; }
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; i++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float xh = xv[i][k] * inv;
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < MAXE; e++) { s += dve[e] * wv[e][i][k]; dw[e][i][k] += dve[e] * xh; }
                dn[i][k] = s;
                dot += s * xh;
                xv[i][k] = xh;
            }
        dot = od_wave_sum(dot) / (float)C;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int c = (lane + 64 * i) * 8;
            if (c >= C) continue;
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; k++) o[k] = inv * (dn[i][k] - xv[i][k] * dot);
            od_st8(dx + m * lddx + c, o);
        }
    }
    if (lane == 0)
        for (int e = 0; e < MAXE; e++) redb[wave][e] = dbv[e];
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int e = 0; e < MAXE; e++)
#pragma unroll
                for (int i = 0; i < NCH; i++)
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int idx = e * NCH * 512 + (lane + 64 * i) * 8 + k;
                        if (w == 0) red[idx] = dw[e][i][k]; else red[idx] += dw[e][i][k];
                    }
        }
        __syncthreads();
    }
    for (int e = 0; e < E; e++)
        for (int c = threadIdx.x; c < C; c += 256) od_red_add(det, dW + (size_t)e * C + c, red[e * NCH * 512 + c]);
    if (threadIdx.x < E) od_red_add(det, db + threadIdx.x, redb[0][threadIdx.x] + redb[1][threadIdx.x] + redb[2][threadIdx.x] + redb[3][threadIdx.x]);
}

// ------------------------------------------------------------------ q/k RMSNorm(hd) + RoPE
// lane owns 8 consecutive features of one head; a head spans hd/8 adjacent lanes.
template <class T, class TO = T>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(const T* __restrict__ qkv, int ldqkv, const float* __restrict__ wq,
                                                           const float* __restrict__ wk, const float* __restrict__ table,
                                                           TO* __restrict__ out, int ldo, int B, int L, int H, int hd, float eps,
                                                           float q_scale) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= (long)B * L) return;
    const int l = (int)(m % L);
    const int lph = hd / 8, half = lph / 2;
    const int nchunks = 2 * H * lph;
    const int niter = (nchunks + 63) / 64;
    for (int it = 0; it < niter; it++) {
        const int q = it * 64 + lane;
        const bool act = q < nchunks;
        const int slot = q / lph, c8 = q % lph;
        float v[8];
        if (act) od_ld8(qkv + m * ldqkv + (size_t)slot * hd + c8 * 8, v);
        else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = 0.f;
        }
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) ss += v[e] * v[e];
        for (int msk = 1; msk < lph; msk <<= 1) ss += __shfl_xor(ss, msk);
        const float inv = rsqrtf(ss / (float)hd + eps);
        const float* w = (slot < H ? wq : wk) + c8 * 8;
        float y[8], p[8];
#pragma unroll
        for (int e = 0; e < 8; e++) y[e] = act ? v[e] * inv * w[e] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) p[e] = __shfl_xor(y[e], half);
        if (act) {
            const bool first = c8 < half;
            const int j0 = (c8 % half) * 8;
            const float* tb = table + ((size_t)l * (hd / 2) + j0) * 2;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float cs = tb[2 * e], sn = tb[2 * e + 1];
                o[e] = (first ? (y[e] * cs - p[e] * sn) : (p[e] * sn + y[e] * cs)) * (slot < H ? q_scale : 1.f);
            }
            od_st8(out + m * ldo + (size_t)slot * hd + c8 * 8, o);
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void qk_norm_rope_bwd_kernel(const T* __restrict__ qkv, int ldqkv, const float* __restrict__ wq,
                                                               const float* __restrict__ wk, const float* __restrict__ table,
                                                               const T* __restrict__ dqk, int lddqk, T* __restrict__ dqkv, int lddqkv,
                                                               float* __restrict__ dwq, float* __restrict__ dwk,
                                                               int B, int L, int H, int hd, float eps, float q_scale,
                                                               const OdDetTable* __restrict__ det) {
    __shared__ long long sdw[2][256];          // fixed point: the block's sum does not depend on the order its lanes arrive in
    __shared__ int s_bad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 512; i += 256) sdw[i >> 8][i & 255] = 0;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    const int lph = hd / 8, half = lph / 2;
    const int nchunks = 2 * H * lph;
    const int niter = (nchunks + 63) / 64;
    float aq[8], ak[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { aq[e] = 0.f; ak[e] = 0.f; }
    const long M = (long)B * L;
    const long mbase = ((long)blockIdx.x * 4 + wave) * ROWS_PER_WAVE_BWD;
    for (int r = 0; r < ROWS_PER_WAVE_BWD; r++) {
        const long m = mbase + r;
        if (m >= M) break;
        const int l = (int)(m % L);
        for (int it = 0; it < niter; it++) {
            const int q = it * 64 + lane;
            const bool act = q < nchunks;
            const int slot = q / lph, c8 = q % lph;
            float v[8], d[8];
            if (act) {
                od_ld8(qkv + m * ldqkv + (size_t)slot * hd + c8 * 8, v); od_ld8(dqk + m * lddqk + (size_t)slot * hd + c8 * 8, d);
                if (slot < H) {
#pragma unroll
                    for (int e = 0; e < 8; e++) d[e] *= q_scale;      // the forward scaled its q outputs
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) { v[e] = 0.f; d[e] = 0.f; }
            }
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) ss += v[e] * v[e];
            for (int msk = 1; msk < lph; msk <<= 1) ss += __shfl_xor(ss, msk);
            const float inv = rsqrtf(ss / (float)hd + eps);
            // un-rotate the gradient: dy1 = d1*c + d2*s ; dy2 = -d1*s + d2*c
            float p[8], dy[8];
#pragma unroll
            for (int e = 0; e < 8; e++) p[e] = __shfl_xor(d[e], half);
            const bool first = c8 < half;
            const int j0 = (c8 % (half > 0 ? half : 1)) * 8;
            const float* tb = table + ((size_t)l * (hd / 2) + j0) * 2;
            const float* w = (slot < H ? wq : wk) + c8 * 8;
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float cs = act ? tb[2 * e] : 0.f, sn = act ? tb[2 * e + 1] : 0.f;
                dy[e] = first ? (d[e] * cs + p[e] * sn) : (-p[e] * sn + d[e] * cs);
                const float xh = v[e] * inv;
                const float wv = act ? w[e] : 0.f;
                if (act) { if (slot < H) aq[e] += dy[e] * xh; else ak[e] += dy[e] * xh; }
                dy[e] *= wv;              // d xhat
                dot += dy[e] * xh;
                v[e] = xh;
            }
            for (int msk = 1; msk < lph; msk <<= 1) dot += __shfl_xor(dot, msk);
            dot /= (float)hd;
            if (act) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = inv * (dy[e] - v[e] * dot);
                od_st8(dqkv + m * lddqkv + (size_t)slot * hd + c8 * 8, o);
            }
        }
    }
    const int c8 = lane % lph;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        od_lds_fix_add(&sdw[0][c8 * 8 + e], aq[e], &s_bad);
        od_lds_fix_add(&sdw[1][c8 * 8 + e], ak[e], &s_bad);
    }
    __syncthreads();
    if (threadIdx.x < hd) {
        od_red_add(det, dwq + threadIdx.x, od_lds_unfix(sdw[0][threadIdx.x], s_bad));
        od_red_add(det, dwk + threadIdx.x, od_lds_unfix(sdw[1][threadIdx.x], s_bad));
    }
}

// ---- q/k RMSNorm + RoPE, position-major variant (hd = 8*LPH, H*LPH a multiple of 64) -------------------
// A lane owns 8 consecutive features (one 16-byte piece) and a wave-wide load covers 64 adjacent pieces — 1 KB of one frame, fully
// coalesced — so a frame's 2H heads take NIT = 2*H*LPH/64 loads.  Because H*LPH is a multiple of 64 the lane's place inside its head
// (c8 = lane % LPH) is the same in every one of them: its 8 norm weights (q and k) live in registers for the whole kernel, and a wave
// works on ONE position l for several batch rows (frames b*L + l, b = b0..b1), so the 16 cos/sin values it needs are loaded once per
// position instead of once per piece.  Cross-lane traffic: the LPH-lane sum of squares and the rotary partner (lane ^ LPH/2).
// (The lane-per-head kernel this replaces moved whole heads per lane — 64 distinct cache lines per load instruction — and ran at
// 39 % (backward) / 54 % (forward) of the HBM peak, profiles/r02j_pmc_step.txt.)
template <class T, int LPH, int NIT>
__global__ __launch_bounds__(256) void qk_norm_rope_pos_kernel(const T* __restrict__ qkv, int ldqkv, const float* __restrict__ wq,
                                                               const float* __restrict__ wk, const float* __restrict__ table,
                                                               T* __restrict__ out, int ldo, int B, int L, float eps, float q_scale,
                                                               int bpw, int lpw) {
    constexpr int HD = 8 * LPH, HALFL = LPH / 2;
    const int lane = threadIdx.x & 63, c8 = lane % LPH;
    const int wpl = (B + bpw - 1) / bpw;                          // waves per group of positions
    const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l0 = (int)(gw / wpl) * lpw, b0 = (int)(gw % wpl) * bpw;
    if (l0 >= L) return;
    const int l1 = l0 + lpw < L ? l0 + lpw : L, b1 = b0 + bpw < B ? b0 + bpw : B;
    float w[2][8];
    od_ld8(wq + c8 * 8, w[0]); od_ld8(wk + c8 * 8, w[1]);
#pragma unroll
    for (int e = 0; e < 8; e++) w[0][e] *= q_scale;
    const bool first = c8 < HALFL;
    for (int l = l0; l < l1; l++) {
        float tb[16];                                              // (cos, sin) of features j0 .. j0+7 at position l
        const float* tp = table + ((size_t)l * (HD / 2) + (c8 % HALFL) * 8) * 2;
        od_ld8(tp, *(float(*)[8])&tb[0]); od_ld8(tp + 8, *(float(*)[8])&tb[8]);
        float cs[8], sn[8];
#pragma unroll
        for (int e = 0; e < 8; e++) { cs[e] = tb[2 * e]; sn[e] = first ? -tb[2 * e + 1] : tb[2 * e + 1]; }
        for (int b = b0; b < b1; b++) {
            const long m = (long)b * L + l;
            float v[NIT][8];
#pragma unroll
            for (int it = 0; it < NIT; it++) od_ld8(qkv + m * ldqkv + (size_t)(it * 64 + lane) * 8, v[it]);
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) ss += v[it][e] * v[it][e];
#pragma unroll
                for (int msk = 1; msk < LPH; msk <<= 1) ss += __shfl_xor(ss, msk);
                const float inv = rsqrtf(ss / (float)HD + eps);
                const float* wv = w[it < NIT / 2 ? 0 : 1];
                float y[8], o[8];
#pragma unroll
                for (int e = 0; e < 8; e++) y[e] = v[it][e] * inv * wv[e];
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = y[e] * cs[e] + __shfl_xor(y[e], HALFL) * sn[e];
                od_st8(out + m * ldo + (size_t)(it * 64 + lane) * 8, o);
            }
        }
    }
}

// backward of the above; the per-feature weight gradients accumulate in registers over the wave's frames, meet across the wave's
// heads by shuffles and across the block in LDS: 2*HD global atomics per block.
template <class T, int LPH, int NIT>
__global__ __launch_bounds__(256) void qk_norm_rope_pos_bwd_kernel(const T* __restrict__ qkv, int ldqkv, const float* __restrict__ wq,
                                                                   const float* __restrict__ wk, const float* __restrict__ table,
                                                                   const T* __restrict__ dqk, int lddqk, T* __restrict__ dqkv, int lddqkv,
                                                                   float* __restrict__ dwq, float* __restrict__ dwk,
                                                                   int B, int L, float eps, float q_scale, int bpw, int lpw,
                                                                   const OdDetTable* __restrict__ det) {
    constexpr int HD = 8 * LPH, HALFL = LPH / 2;
    __shared__ long long s_dw[2][HD];          // fixed point (see qk_norm_rope_bwd_kernel)
    __shared__ int s_bad;
    for (int i = threadIdx.x; i < 2 * HD; i += 256) s_dw[i / HD][i % HD] = 0;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, c8 = lane % LPH;
    const int wpl = (B + bpw - 1) / bpw;
    const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l0 = (int)(gw / wpl) * lpw, b0 = (int)(gw % wpl) * bpw;
    const int l1 = l0 + lpw < L ? l0 + lpw : L, b1 = b0 + bpw < B ? b0 + bpw : B;
    float w[2][8], acc[2][8];
    od_ld8(wq + c8 * 8, w[0]); od_ld8(wk + c8 * 8, w[1]);
#pragma unroll
    for (int e = 0; e < 8; e++) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    const bool first = c8 < HALFL;
    for (int l = l0; l < l1; l++) {                               // (empty when l0 >= L: the wave still joins the barrier below)
        float tb[16];
        const float* tp = table + ((size_t)l * (HD / 2) + (c8 % HALFL) * 8) * 2;
        od_ld8(tp, *(float(*)[8])&tb[0]); od_ld8(tp + 8, *(float(*)[8])&tb[8]);
        float cs[8], sn[8];
#pragma unroll
        for (int e = 0; e < 8; e++) { cs[e] = tb[2 * e]; sn[e] = first ? tb[2 * e + 1] : -tb[2 * e + 1]; }   // un-rotation
        for (int b = b0; b < b1; b++) {
            const long m = (long)b * L + l;
            float v[NIT][8], d[NIT][8];
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                od_ld8(qkv + m * ldqkv + (size_t)(it * 64 + lane) * 8, v[it]);
                od_ld8(dqk + m * lddqk + (size_t)(it * 64 + lane) * 8, d[it]);
            }
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int qk = it < NIT / 2 ? 0 : 1;
                const float gs = qk == 0 ? q_scale : 1.f;          // the forward scaled its q outputs
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) ss += v[it][e] * v[it][e];
#pragma unroll
                for (int msk = 1; msk < LPH; msk <<= 1) ss += __shfl_xor(ss, msk);
                const float inv = rsqrtf(ss / (float)HD + eps);
                float dy[8], xh[8], dot = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) d[it][e] *= gs;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float u = d[it][e] * cs[e] + __shfl_xor(d[it][e], HALFL) * sn[e];
                    xh[e] = v[it][e] * inv;
                    acc[qk][e] += u * xh[e];
                    dy[e] = u * w[qk][e];
                    dot += dy[e] * xh[e];
                }
#pragma unroll
                for (int msk = 1; msk < LPH; msk <<= 1) dot += __shfl_xor(dot, msk);
                dot /= (float)HD;
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = inv * (dy[e] - xh[e] * dot);
                od_st8(dqkv + m * lddqkv + (size_t)(it * 64 + lane) * 8, o);
            }
        }
    }
#pragma unroll
    for (int msk = LPH; msk < 64; msk <<= 1)
#pragma unroll
        for (int e = 0; e < 8; e++) { acc[0][e] += __shfl_xor(acc[0][e], msk); acc[1][e] += __shfl_xor(acc[1][e], msk); }
    if (lane < LPH) {
#pragma unroll
        for (int e = 0; e < 8; e++) { od_lds_fix_add(&s_dw[0][c8 * 8 + e], acc[0][e], &s_bad); od_lds_fix_add(&s_dw[1][c8 * 8 + e], acc[1][e], &s_bad); }
    }
    __syncthreads();
    if (threadIdx.x < HD) {
        od_red_add(det, dwq + threadIdx.x, od_lds_unfix(s_dw[0][threadIdx.x], s_bad));
        od_red_add(det, dwk + threadIdx.x, od_lds_unfix(s_dw[1][threadIdx.x], s_bad));
    }
}

__global__ void rope_table_kernel(float* __restrict__ table, int L, int hd) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int half = hd / 2;
    if (i >= (long)L * half) return;
    const int l = (int)(i / half), j = (int)(i % half);
    // inv_freq = 10000^(e), e = (float)(2j) / (float)(-hd) as the reference forms it (attn.py:19), then the fp32 product
    // l * inv_freq (attn.py:21).  The power goes through double and is rounded once: a fast fp32 pow that is 1 ulp off at a
    // HIGH frequency would be 2e-3 rad at position 32768; this way the table never depends on the device's powf (torch's
    // own fp32 pow differs from the correctly rounded value at two LOW frequencies only: <= 8e-6 rad at 32768).
    const float e = (float)(2 * j) / (float)(-hd);
    const float inv_freq = (float)pow(10000.0, (double)e);
    const float ang = (float)l * inv_freq;
    table[2 * i] = cosf(ang);
    table[2 * i + 1] = sinf(ang);
}

inline int nch_for(int C) { return (C + 511) / 512; }

// position-major q/k kernels: a wave takes `bpw` batch rows of `lpw` consecutive positions (about 8 frames per wave)
inline void qk_pos_split(int B, int L, int& bpw, int& lpw) {
    bpw = B < OD_QK_BPW ? B : OD_QK_BPW;
    lpw = OD_QK_BPW / bpw;
    if (lpw < 1) lpw = 1;
    (void)L;
}

}  // namespace

#define DISPATCH_T_NCH(DT, NCHV, CALL)                                          \
    do {                                                                        \
        if ((DT) == OD_BF16) {                                                  \
            typedef bf16_t T_;                                                  \
            if ((NCHV) == 1) { constexpr int N_ = 1; CALL; }                    \
            else if ((NCHV) == 2) { constexpr int N_ = 2; CALL; }               \
            else if ((NCHV) == 3) { constexpr int N_ = 3; CALL; }               \
            else return OD_ERR_UNSUPPORTED;                                     \
        } else if ((DT) == OD_F32) {                                            \
            typedef float T_;                                                   \
            if ((NCHV) == 1) { constexpr int N_ = 1; CALL; }                    \
            else if ((NCHV) == 2) { constexpr int N_ = 2; CALL; }               \
            else if ((NCHV) == 3) { constexpr int N_ = 3; CALL; }               \
            else return OD_ERR_UNSUPPORTED;                                     \
        } else return OD_ERR_ARG;                                               \
    } while (0)

extern "C" int od_rmsnorm_film(int dtype, const void* x, int ldx, const float* ssg, const void* cl, int ldcl, int cl_bcast,
                               void* h, int ldh, float* inv_rms, int B, int L, int C, float eps, void* stream) {
    if (C % 8 || ldx % 8 || ldh % 8 || (cl && ldcl % 8)) return OD_ERR_ALIGN;
    const long M = (long)B * L;
    dim3 grid((unsigned)((M + 3) / 4));
    DISPATCH_T_NCH(dtype, nch_for(C),
        OD_LAUNCH((rmsnorm_film_kernel<T_, N_>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx, ssg, (const T_*)cl, ldcl,
                  cl_bcast, (T_*)h, ldh, inv_rms, B, L, C, eps));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_rmsnorm_film_bwd(int dtype, const void* x, int ldx, const float* inv_rms, const float* ssg, const void* dh,
                                   int lddh, void* dres, int lddres, float* dssg, int B, int L, int C, void* stream) {
    if (C % 8 || ldx % 8 || lddh % 8 || lddres % 8) return OD_ERR_ALIGN;
    dim3 grid((L + 4 * ROWS_PER_WAVE_BWD - 1) / (4 * ROWS_PER_WAVE_BWD), B);
    DISPATCH_T_NCH(dtype, nch_for(C),
        OD_LAUNCH((rmsnorm_film_bwd_kernel<T_, N_>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx, inv_rms, ssg,
                  (const T_*)dh, lddh, (T_*)dres, lddres, dssg, B, L, C, od_det_active()));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_rmsnorm_gate_residual(int dtype, const void* x, int ldx, const void* h, int ldh, const float* ssg, void* xo,
                                        int ldxo, float* inv_rms, int B, int L, int C, float eps, void* stream) {
    if (C % 8 || ldx % 8 || ldh % 8 || ldxo % 8) return OD_ERR_ALIGN;
    const long M = (long)B * L;
    dim3 grid((unsigned)((M + 3) / 4));
    DISPATCH_T_NCH(dtype, nch_for(C),
        OD_LAUNCH((rmsnorm_gate_res_kernel<T_, N_>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx, (const T_*)h, ldh, ssg,
                  (T_*)xo, ldxo, inv_rms, B, L, C, eps));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_rmsnorm_gate_residual_film(int dtype, const void* x, int ldx, const void* h, int ldh, const float* ssg_a, void* xo,
                                             int ldxo, float* inv_a, const float* ssg_b, const void* cl, int ldcl, int cl_bcast,
                                             void* h2, int ldh2, float* inv_b, int B, int L, int C, float eps, void* stream) {
    if (C % 8 || ldx % 8 || ldh % 8 || ldxo % 8 || ldh2 % 8 || (cl && ldcl % 8)) return OD_ERR_ALIGN;
    const long M = (long)B * L;
    dim3 grid((unsigned)((M + 3) / 4));
    DISPATCH_T_NCH(dtype, nch_for(C),
        OD_LAUNCH((rmsnorm_gate_res_film_kernel<T_, N_>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx, (const T_*)h, ldh,
                  ssg_a, (T_*)xo, ldxo, inv_a, ssg_b, (const T_*)cl, ldcl, cl_bcast, (T_*)h2, ldh2, inv_b, B, L, C, eps));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_rmsnorm_gate_residual_film_dwconv(int dtype, const void* x, int ldx, const void* h, int ldh, const float* ssg_a, void* xo,
                                                    int ldxo, float* inv_a, const float* ssg_b, void* h2, int ldh2, float* inv_b,
                                                    const float* conv_w, const float* conv_b, void* y, int ldy, int B, int L, int C,
                                                    int ksize, float eps, void* stream) {
    if (C % 8 || ldx % 8 || ldh % 8 || ldxo % 8 || (h2 && ldh2 % 8) || ldy % 8) return OD_ERR_ALIGN;
    if (C > 512 || (ksize != 3 && ksize != 5 && ksize != 7 && ksize != 9)) return OD_ERR_UNSUPPORTED;
    if (!x || !h || !xo || !y || !conv_w || !conv_b || xo == x || xo == h) return OD_ERR_ARG;       // the halo frames re-read x and h: xo must not alias them
    // a wave walks 32 frames (2 R / 32 of halo re-reads); launches too small to cover the chip that way (the sampler: B L = 4460) take runs of 8
    const bool small = (long)B * ((L + 31) / 32) < 2048;
    const int run = small ? 8 : 32;
    dim3 grid((unsigned)((L + 4 * run - 1) / (4 * run)), (unsigned)B);
#define FD_GO2(T_, KS_, RUN_) OD_LAUNCH((rmsnorm_gate_res_film_dwconv_kernel<T_, KS_, RUN_>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx, \
                   (const T_*)h, ldh, ssg_a, (T_*)xo, ldxo, inv_a, ssg_b, (T_*)h2, ldh2, inv_b, conv_w, conv_b, (T_*)y, ldy, B, L, C, eps)
#define FD_GO1(T_, KS_) do { if (small) FD_GO2(T_, KS_, 8); else FD_GO2(T_, KS_, 32); } while (0)
#define FD_GO(KS_) do { if (dtype == OD_BF16) FD_GO1(bf16_t, KS_); else if (dtype == OD_F32) FD_GO1(float, KS_); else return OD_ERR_UNSUPPORTED; } while (0)
    if (ksize == 5) FD_GO(5); else if (ksize == 3) FD_GO(3); else if (ksize == 7) FD_GO(7); else FD_GO(9);
#undef FD_GO
#undef FD_GO1
#undef FD_GO2
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_rmsnorm_gate_residual_bwd(int dtype, const void* h, int ldh, const float* inv_rms, const float* ssg,
                                            const void* dy, int lddy, void* dh, int lddh, float* dssg, int B, int L, int C,
                                            void* stream) {
    if (C % 8 || ldh % 8 || lddy % 8 || lddh % 8) return OD_ERR_ALIGN;
    dim3 grid((L + 4 * ROWS_PER_WAVE_BWD - 1) / (4 * ROWS_PER_WAVE_BWD), B);
    DISPATCH_T_NCH(dtype, nch_for(C),
        OD_LAUNCH((rmsnorm_gate_res_bwd_kernel<T_, N_>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)h, ldh, inv_rms, ssg,
                  (const T_*)dy, lddy, (T_*)dh, lddh, dssg, B, L, C, od_det_active()));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_swiglu_rmsnorm(int dtype, const void* vg, int ldvg, void* hh, int ldhh, float* inv_rms, int M, int Hf, int Hp,
                                 float eps, void* stream) {
    if (Hp % 8 || ldvg % 8 || ldhh % 8 || Hf > Hp) return OD_ERR_ALIGN;
    dim3 grid((unsigned)((M + 3) / 4));
    DISPATCH_T_NCH(dtype, nch_for(Hp),
        OD_LAUNCH((swiglu_rmsnorm_kernel<T_, N_>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)vg, ldvg, (T_*)hh, ldhh, inv_rms,
                  (long)M, Hf, Hp, eps));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_swiglu_rmsnorm_bwd(int dtype, const void* vg, int ldvg, const float* inv_rms, const void* dhh, int lddhh,
                                     void* dvg, int lddvg, int M, int Hf, int Hp, void* stream) {
    if (Hp % 8 || ldvg % 8 || lddhh % 8 || lddvg % 8 || Hf > Hp) return OD_ERR_ALIGN;
    dim3 grid((unsigned)((M + 3) / 4));
    DISPATCH_T_NCH(dtype, nch_for(Hp),
        OD_LAUNCH((swiglu_rmsnorm_bwd_kernel<T_, N_>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)vg, ldvg, inv_rms,
                  (const T_*)dhh, lddhh, (T_*)dvg, lddvg, (long)M, Hf, Hp));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_final_norm_proj_out(int dtype, const void* x, int ldx, const float* W, const float* bias, float* v,
                                      float* inv_rms, int B, int L, int C, int E, float eps, void* stream) {
    if (C % 8 || ldx % 8) return OD_ERR_ALIGN;
    if (E > 8 || C > 1024) return OD_ERR_UNSUPPORTED;
    const long M = (long)B * L;
    dim3 grid((unsigned)((M + 3) / 4));
    DISPATCH_T_NCH(dtype, nch_for(C),
        OD_LAUNCH((final_proj_kernel<T_, N_>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx, W, bias, v, inv_rms, B, L, C, E, eps));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_final_norm_proj_out_bwd(int dtype, const void* x, int ldx, const float* inv_rms, const float* W,
                                          const float* dv, void* dx, int lddx, float* dW, float* db, int B, int L, int C, int E,
                                          void* stream) {
    if (C % 8 || ldx % 8 || lddx % 8) return OD_ERR_ALIGN;
    if (E > 8 || C > 512) return OD_ERR_UNSUPPORTED;
    dim3 grid((L + 4 * ROWS_PER_WAVE_BWD - 1) / (4 * ROWS_PER_WAVE_BWD), B);
    if (dtype == OD_BF16)
        OD_LAUNCH((final_proj_bwd_kernel<bf16_t, 1>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, inv_rms, W, dv,
                  (bf16_t*)dx, lddx, dW, db, B, L, C, E, od_det_active());
    else
        OD_LAUNCH((final_proj_bwd_kernel<float, 1>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, inv_rms, W, dv,
                  (float*)dx, lddx, dW, db, B, L, C, E, od_det_active());
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_rope_table(float* table, int L, int hd, void* stream) {
    if (hd % 2 || L <= 0) return OD_ERR_ARG;
    const long n = (long)L * (hd / 2);
    OD_LAUNCH(rope_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, L, hd);
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_qk_norm_rope(int dtype, const void* qkv, int ldqkv, const float* wq, const float* wk, const float* table,
                               void* qk_out, int ldo, int B, int L, int H, int hd, float eps, float q_scale, void* stream) {
    if (hd % 16 || hd > 256 || 64 % (hd / 8)) return OD_ERR_UNSUPPORTED;
    if (ldqkv % 8 || ldo % 8) return OD_ERR_ALIGN;
    const long M = (long)B * L;
    const int lph = hd / 8;
    if (OD_QK_POS && dtype != OD_F16 && (lph == 4 || lph == 8) && (H * lph) % 64 == 0 && ldqkv >= 2 * H * hd) {      // position-major kernel
        const int nit = 2 * H * lph / 64;
        if (nit == 2 || nit == 4 || nit == 8) {
            int bpw, lpw; qk_pos_split(B, L, bpw, lpw);
            const long waves = (long)((L + lpw - 1) / lpw) * ((B + bpw - 1) / bpw);
            dim3 g2((unsigned)((waves + 3) / 4));
#define QKP(TT, LPHV, NITV) OD_LAUNCH((qk_norm_rope_pos_kernel<TT, LPHV, NITV>), g2, dim3(256), 0, (hipStream_t)stream, (const TT*)qkv, ldqkv, wq, wk, table, (TT*)qk_out, ldo, B, L, eps, q_scale, bpw, lpw)
#define QKP_T(TT) do { if (lph == 8) { if (nit == 2) QKP(TT, 8, 2); else if (nit == 4) QKP(TT, 8, 4); else QKP(TT, 8, 8); } \
                       else { if (nit == 2) QKP(TT, 4, 2); else if (nit == 4) QKP(TT, 4, 4); else QKP(TT, 4, 8); } } while (0)
            if (dtype == OD_BF16) QKP_T(bf16_t); else QKP_T(float);
#undef QKP_T
#undef QKP
            OD_CHECK_LAUNCH();
            return 0;
        }
    }
    dim3 grid((unsigned)((M + 3) / 4));
    if (dtype == OD_F16)                       // bf16 in, IEEE half out ("attention in fp16")
        OD_LAUNCH((qk_norm_rope_kernel<bf16_t, f16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, ldqkv, wq, wk, table,
                  (f16_t*)qk_out, ldo, B, L, H, hd, eps, q_scale);
    else if (dtype == OD_BF16)
        OD_LAUNCH((qk_norm_rope_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, ldqkv, wq, wk, table,
                  (bf16_t*)qk_out, ldo, B, L, H, hd, eps, q_scale);
    else
        OD_LAUNCH((qk_norm_rope_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)qkv, ldqkv, wq, wk, table,
                  (float*)qk_out, ldo, B, L, H, hd, eps, q_scale);
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_qk_norm_rope_bwd(int dtype, const void* qkv, int ldqkv, const float* wq, const float* wk, const float* table,
                                   const void* dqk, int lddqk, void* dqkv, int lddqkv, float* dwq, float* dwk, int B, int L,
                                   int H, int hd, float eps, float q_scale, void* stream) {
    if (hd % 16 || hd > 256 || 64 % (hd / 8)) return OD_ERR_UNSUPPORTED;
    if (ldqkv % 8 || lddqk % 8 || lddqkv % 8) return OD_ERR_ALIGN;
    const long M = (long)B * L;
    const int lph = hd / 8;
    if (OD_QK_POS && (lph == 4 || lph == 8) && (H * lph) % 64 == 0) {      // position-major kernel
        const int nit = 2 * H * lph / 64;
        if (nit == 2 || nit == 4 || nit == 8) {
            int bpw, lpw; qk_pos_split(B, L, bpw, lpw);
            lpw *= 4;                                              // 4x fewer blocks meeting in the 2*hd global atomics
            const long waves = (long)((L + lpw - 1) / lpw) * ((B + bpw - 1) / bpw);
            dim3 g2((unsigned)((waves + 3) / 4));
#define QKPB(TT, LPHV, NITV) OD_LAUNCH((qk_norm_rope_pos_bwd_kernel<TT, LPHV, NITV>), g2, dim3(256), 0, (hipStream_t)stream, (const TT*)qkv, ldqkv, wq, wk, table, (const TT*)dqk, lddqk, (TT*)dqkv, lddqkv, dwq, dwk, B, L, eps, q_scale, bpw, lpw, od_det_active())
#define QKPB_T(TT) do { if (lph == 8) { if (nit == 2) QKPB(TT, 8, 2); else if (nit == 4) QKPB(TT, 8, 4); else QKPB(TT, 8, 8); } \
                        else { if (nit == 2) QKPB(TT, 4, 2); else if (nit == 4) QKPB(TT, 4, 4); else QKPB(TT, 4, 8); } } while (0)
            if (dtype == OD_BF16) QKPB_T(bf16_t); else QKPB_T(float);
#undef QKPB_T
#undef QKPB
            OD_CHECK_LAUNCH();
            return 0;
        }
    }
    dim3 grid((unsigned)((M + 4 * ROWS_PER_WAVE_BWD - 1) / (4 * ROWS_PER_WAVE_BWD)));
    if (dtype == OD_BF16)
        OD_LAUNCH((qk_norm_rope_bwd_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, ldqkv, wq, wk, table,
                  (const bf16_t*)dqk, lddqk, (bf16_t*)dqkv, lddqkv, dwq, dwk, B, L, H, hd, eps, q_scale, od_det_active());
    else
        OD_LAUNCH((qk_norm_rope_bwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)qkv, ldqkv, wq, wk, table,
                  (const float*)dqk, lddqk, (float*)dqkv, lddqkv, dwq, dwk, B, L, H, hd, eps, q_scale, od_det_active());
    OD_CHECK_LAUNCH();
    return 0;
}
