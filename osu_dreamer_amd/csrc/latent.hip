// Latent model, inference path (osu_dreamer/models/latent/{spec_features,unet,model}.py): the step either
// side of diffusion.sample in LDM.sample.  Frame-major [B*L][C] like the denoiser; C = h_dim = 128 is
// narrower than a wavefront x 8 channels, so here G = C/8 lanes own a frame and a wave walks 64/G frames.
// The SwiGLU body of each block reuses od_dwconv / od_gemm_nt / od_swiglu_rmsnorm.  All HBM-bound row work.
#include "od_common.h"
#include "od_api_internal.h"

namespace {

__device__ __forceinline__ float group_sum(float v, int G) {
    for (int m = G >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float sumsq8(const float (&v)[8]) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) s += v[e] * v[e];
    return s;
}
// frame and channel owned by this lane: G lanes per frame, 256/G frames per block
#define OD_ROW_OF_LANE()                                                        \
    const int G = C >> 3;                                                       \
    const long m = (long)blockIdx.x * (256 / G) + threadIdx.x / G;              \
    const int c = (threadIdx.x % G) * 8;                                        \
    const bool live = m < M;                                                    \
    const long mr = live ? m : M - 1   /* dead lanes shadow the last frame so shuffles stay convergent */

// y = act( rms_norm(x) * gamma * (1 + scale[b]) + shift[b] )       unet.py:50 (norm + FiLM), :51 out_norm,
//                                                                  spec_features.py:27-28 (norm + SiLU)
template <class T>
__global__ __launch_bounds__(256) void rms_affine_film_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                              const float* __restrict__ ssg, T* __restrict__ y, int ldy,
                                                              long M, int L, int C, float eps, int act) {
    OD_ROW_OF_LANE();
    float v[8], g[8], o[8];
    od_ld8(x + mr * ldx + c, v);
    const float inv = rsqrtf(group_sum(sumsq8(v), G) / (float)C + eps);
    od_ld8(gamma + c, g);
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = v[e] * inv * g[e];
    if (ssg) {
        const float* sc = ssg + (size_t)(mr / L) * 3 * C;
        float s[8], sh[8];
        od_ld8(sc + c, s); od_ld8(sc + C + c, sh);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = o[e] * (1.f + s[e]) + sh[e];
    }
    if (act == OD_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = od_silu(o[e]);
    }
    if (live) od_st8(y + m * ldy + c, o);
}

// xo = x + rms_norm(h) * gamma * (1 + gate[b])                      unet.py:28,51
template <class T>
__global__ __launch_bounds__(256) void rms_affine_gate_res_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ h, int ldh,
                                                                  const float* __restrict__ gamma, const float* __restrict__ ssg,
                                                                  T* __restrict__ xo, int ldxo, long M, int L, int C, float eps) {
    OD_ROW_OF_LANE();
    float v[8], g[8], r[8], o[8];
    od_ld8(h + mr * ldh + c, v);
    const float inv = rsqrtf(group_sum(sumsq8(v), G) / (float)C + eps);
    od_ld8(gamma + c, g);
    od_ld8(x + mr * ldx + c, r);
    float gt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ssg) od_ld8(ssg + (size_t)(mr / L) * 3 * C + 2 * C + c, gt);
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = r[e] + v[e] * inv * g[e] * (1.f + gt[e]);
    if (live) od_st8(xo + m * ldxo + c, o);
}

// xo = x + rms_norm(p) * gamma * gx;  p = proj(skip) rows (frame l of batch 0 when the skip is broadcast),
// gx = gate(x) rows                                                 unet.py:117-126 (mixer)
template <class T>
__global__ __launch_bounds__(256) void mixer_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ p, int ldp, int p_bcast,
                                                    const T* __restrict__ gx, int ldg, const float* __restrict__ gamma,
                                                    T* __restrict__ xo, int ldxo, long M, int L, int C, float eps) {
    OD_ROW_OF_LANE();
    float v[8], g[8], r[8], q[8], o[8];
    od_ld8(p + (p_bcast ? mr % L : mr) * ldp + c, v);
    const float inv = rsqrtf(group_sum(sumsq8(v), G) / (float)C + eps);
    od_ld8(gamma + c, g);
    od_ld8(x + mr * ldx + c, r);
    od_ld8(gx + mr * ldg + c, q);
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = r[e] + v[e] * inv * g[e] * q[e];
    if (live) od_st8(xo + m * ldxo + c, o);
}

// y[b][lo][c] = mean_{j<s} ( bias[c] + sum_k w[c][k] x[b][s*lo + j + k - r][c] ),  k = 2r+1 = 1 + 2*(s/2), zero padded
// — depthwise conv then AvgPool1d(s)                                unet.py:58-63
template <class T>
__global__ __launch_bounds__(256) void unet_down_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ y, int ldy, int B, int Lo,
                                                        int C, int s) {
    const int G = C >> 3, r = s / 2, ks = 2 * r + 1, L = Lo * s;
    const long M = (long)B * Lo;
    const long m = (long)blockIdx.x * (256 / G) + threadIdx.x / G;
    const int c = (threadIdx.x % G) * 8;
    if (m >= M) return;
    const int b = (int)(m / Lo), lo = (int)(m % Lo);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // sum_j sum_k w[k] x[s*lo + j + k - r]  =  sum_t x[s*lo - r + t] * (sum of w[k] over k + j = t)
    for (int t = 0; t < s + ks - 1; t++) {
        const int l = s * lo - r + t;
        if (l < 0 || l >= L) continue;
        float v[8];
        od_ld8(x + ((long)b * L + l) * ldx + c, v);
        const int k0 = t - (s - 1) > 0 ? t - (s - 1) : 0, k1 = t < ks - 1 ? t : ks - 1;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float ws = 0.f;
            for (int k = k0; k <= k1; k++) ws += w[(c + e) * ks + k];
            acc[e] += ws * v[e];
        }
    }
    float bb[8], o[8];
    od_ld8(bias + c, bb);
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = acc[e] / (float)s + bb[e];
    od_st8(y + m * ldy + c, o);
}

// y[b][l][c] = bias[c] + sum_k w[c][k] x[b][(l + k - r) / s][c] for 0 <= l + k - r < s*Li
// — nearest Upsample(s) then depthwise conv                          unet.py:80-85
template <class T>
__global__ __launch_bounds__(256) void unet_up_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                      const float* __restrict__ bias, T* __restrict__ y, int ldy, int B, int Li,
                                                      int C, int s) {
    const int G = C >> 3, r = s / 2, ks = 2 * r + 1, L = Li * s;
    const long M = (long)B * L;
    const long m = (long)blockIdx.x * (256 / G) + threadIdx.x / G;
    const int c = (threadIdx.x % G) * 8;
    if (m >= M) return;
    const int b = (int)(m / L), l = (int)(m % L);
    float o[8];
    od_ld8(bias + c, o);
    for (int k = 0; k < ks; k++) {
        const int lu = l + k - r;
        if (lu < 0 || lu >= L) continue;
        float v[8];
        od_ld8(x + ((long)b * Li + lu / s) * ldx + c, v);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] += w[(c + e) * ks + k] * v[e];
    }
    od_st8(y + m * ldy + c, o);
}

// out[b][n][l] = f_n( bias[n] + sum_c W[n][c] x[(b,l)][c] ),  f_n = sigmoid for n < n_sigmoid, identity after;
// with rms != 0 the N outputs of a frame are RMS-normalised (no gain) instead
// — proj_out + the hit-signal sigmoid of decode (latent/model.py:114,127-131); temporal_head (:65-68)
template <class T, int NMAX>
__global__ __launch_bounds__(256) void chart_head_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ out, long M, int L,
                                                         int C, int N, int n_sigmoid, int rms, float eps) {
    OD_ROW_OF_LANE();
    float v[8];
    od_ld8(x + mr * ldx + c, v);
    const int b = (int)(mr / L), l = (int)(mr % L);
    float y[NMAX], ss = 0.f;
#pragma unroll
    for (int n = 0; n < NMAX; n++) {
        y[n] = 0.f;
        if (n >= N) continue;
        float wv[8], s = 0.f;
        od_ld8(W + (size_t)n * C + c, wv);
#pragma unroll
        for (int e = 0; e < 8; e++) s += wv[e] * v[e];
        s = group_sum(s, G) + bias[n];
        if (n < n_sigmoid) s = od_sigmoid(s);
        y[n] = s; ss += s * s;
    }
    const float k = rms ? rsqrtf(ss / (float)N + eps) : 1.f;
#pragma unroll
    for (int n = 0; n < NMAX; n++)
        if (n < N && live && c == 0) out[((size_t)b * N + n) * L + l] = y[n] * k;
}

// AttnPool (latent/model.py:23-36): out[b][h*hd + d] = sum_l softmax_l(scores[(b,l)][h]) * values[(b,l)][h*hd + d].
// One block per (b, h); thread = feature d (hd <= 256), softmax statistics by a block reduction over the frames.
template <class T>
__global__ __launch_bounds__(256) void attn_pool_kernel(const T* __restrict__ scores, int lds_, const T* __restrict__ values, int ldv,
                                                        float* __restrict__ out, int L, int Hh, int hd) {
    __shared__ float red[256];
    __shared__ float s_p[256];
    const int b = blockIdx.y, h = blockIdx.x, t = threadIdx.x;
    const T* sb = scores + (size_t)b * L * lds_ + h;
    float mx = -3.0e38f;
    for (int l = t; l < L; l += 256) mx = fmaxf(mx, od_t<T>::ld(sb + (size_t)l * lds_));
    red[t] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] = fmaxf(red[t], red[t + s]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float acc = 0.f, den = 0.f;
    for (int l0 = 0; l0 < L; l0 += 256) {
        const int l = l0 + t;
        s_p[t] = l < L ? __expf(od_t<T>::ld(sb + (size_t)l * lds_) - mx) : 0.f;
        __syncthreads();
        const int n = L - l0 < 256 ? L - l0 : 256;
        if (t < hd) {
            const T* vb = values + ((size_t)b * L + l0) * ldv + h * hd + t;
            for (int j = 0; j < n; j++) acc += s_p[j] * od_t<T>::ld(vb + (size_t)j * ldv);
        }
        for (int j = 0; j < n; j++) den += s_p[j];       // every thread keeps the same denominator
        __syncthreads();
    }
    if (t < hd) out[(size_t)b * Hh * hd + h * hd + t] = acc / den;
}

// ---- SpecFeatures front end (spec_features.py:17-26): two strided Conv2d over (freq, time), each followed by a
// channel RMS norm (gamma) and SiLU, then 'b c a l -> b (c a) l'.  One block = TL frames; the spectrogram tile
// (with its 2-frame halo and 1-bin zero border) and the first conv's output live in LDS; weights are read with
// wave-uniform indices (scalar loads).
constexpr int SF_TL = 64, SF_F = 72, SF_C1 = 8, SF_A1 = 12, SF_C2 = 32, SF_A2 = 3;
template <class T>
__global__ __launch_bounds__(256) void spec_conv_kernel(const float* __restrict__ audio, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, const float* __restrict__ g1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        const float* __restrict__ g2, T* __restrict__ out, int ldo, int L,
                                                        float eps) {
    __shared__ float s_in[SF_F + 2][SF_TL + 4];            // freq rows -1..72, frames l0-2 .. l0+TL+1
    __shared__ float s_h1[SF_C1][SF_A1 + 2][SF_TL + 2];    // freq rows -1..12, frames l0-1 .. l0+TL
    const int b = blockIdx.y, l0 = blockIdx.x * SF_TL, tid = threadIdx.x;
    const float* ab = audio + (size_t)b * SF_F * L;
    for (int i = tid; i < (SF_F + 2) * (SF_TL + 4); i += 256) {
        const int f = i / (SF_TL + 4) - 1, t = i % (SF_TL + 4), l = l0 - 2 + t;
        s_in[f + 1][t] = (f >= 0 && f < SF_F && l >= 0 && l < L) ? ab[(size_t)f * L + l] : 0.f;
    }
    for (int i = tid; i < SF_C1 * 2 * (SF_TL + 2); i += 256) {   // zero freq border of h1
        const int cc = i / (2 * (SF_TL + 2)), rr = (i / (SF_TL + 2)) % 2, t = i % (SF_TL + 2);
        s_h1[cc][rr ? SF_A1 + 1 : 0][t] = 0.f;
    }
    __syncthreads();
    // conv1 (1 -> 8, kernel (8,3), stride (6,1), pad (1,1)) + rms over the 8 channels + SiLU
    for (int i = tid; i < SF_A1 * (SF_TL + 2); i += 256) {
        const int a = i / (SF_TL + 2), t = i % (SF_TL + 2), l = l0 - 1 + t;
        float acc[SF_C1];
#pragma unroll
        for (int cc = 0; cc < SF_C1; cc++) acc[cc] = b1[cc];
        for (int fi = 0; fi < 8; fi++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const float v = s_in[6 * a + fi][t + j];     // freq 6a + fi - 1 (+1 border), frame l + j - 1
#pragma unroll
                for (int cc = 0; cc < SF_C1; cc++) acc[cc] += w1[(cc * 8 + fi) * 3 + j] * v;
            }
        float ss = 0.f;
#pragma unroll
        for (int cc = 0; cc < SF_C1; cc++) ss += acc[cc] * acc[cc];
        const float inv = rsqrtf(ss / (float)SF_C1 + eps);
        const bool inside = l >= 0 && l < L;                 // conv2 zero-pads h1 in time
#pragma unroll
        for (int cc = 0; cc < SF_C1; cc++) s_h1[cc][a + 1][t] = inside ? od_silu(acc[cc] * inv * g1[cc]) : 0.f;
    }
    __syncthreads();
    // conv2 (8 -> 32, kernel (6,3), stride (4,1), pad (1,1)) + rms over the 32 channels + SiLU
    for (int i = tid; i < SF_A2 * SF_TL; i += 256) {
        const int a = i / SF_TL, t = i % SF_TL, l = l0 + t;
        if (l >= L) continue;
        float acc[SF_C2];
#pragma unroll
        for (int c2 = 0; c2 < SF_C2; c2++) acc[c2] = b2[c2];
#pragma unroll 1
        for (int cc = 0; cc < SF_C1; cc++)
#pragma unroll 1
            for (int fi = 0; fi < 6; fi++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const float v = s_h1[cc][4 * a + fi][t + j];   // freq 4a + fi - 1 (+1 border), frame l + j - 1
#pragma unroll
                    for (int c2 = 0; c2 < SF_C2; c2++) acc[c2] += w2[((c2 * SF_C1 + cc) * 6 + fi) * 3 + j] * v;
                }
        float ss = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < SF_C2; c2++) ss += acc[c2] * acc[c2];
        const float inv = rsqrtf(ss / (float)SF_C2 + eps);
        T* orow = out + ((size_t)b * L + l) * ldo;
#pragma unroll
        for (int c2 = 0; c2 < SF_C2; c2++) od_t<T>::st(orow + c2 * SF_A2 + a, od_silu(acc[c2] * inv * g2[c2]));
    }
}

#define DISPATCH_T(DT, CALL)                                       \
    do {                                                           \
        if ((DT) == OD_BF16) { typedef bf16_t T_; CALL; }          \
        else if ((DT) == OD_F32) { typedef float T_; CALL; }       \
        else return OD_ERR_ARG;                                    \
    } while (0)

inline bool lanes_ok(int C) { return C >= 8 && C <= 512 && (C & (C - 1)) == 0; }   // G = C/8 a power of two <= 64
inline unsigned row_grid(long M, int C) { const int rpb = 256 / (C >> 3); return (unsigned)((M + rpb - 1) / rpb); }

}  // namespace

extern "C" int od_spec_features_conv(int dtype, const float* audio, const float* w1, const float* b1, const float* g1,
                                     const float* w2, const float* b2, const float* g2, void* out, int ldo, int B, int F, int L,
                                     float eps, void* stream) {
    if (F != SF_F) return OD_ERR_UNSUPPORTED;
    if (B <= 0 || L <= 0) return OD_ERR_ARG;
    dim3 grid((L + SF_TL - 1) / SF_TL, B);
    DISPATCH_T(dtype, OD_LAUNCH((spec_conv_kernel<T_>), grid, dim3(256), 0, (hipStream_t)stream, audio, w1, b1, g1, w2, b2, g2,
                                (T_*)out, ldo, L, eps));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_rmsnorm_affine_film(int dtype, const void* x, int ldx, const float* gamma, const float* ssg, void* y, int ldy,
                                      int B, int L, int C, float eps, int act, void* stream) {
    if (!lanes_ok(C) || ldx % 8 || ldy % 8) return OD_ERR_ALIGN;
    const long M = (long)B * L;
    DISPATCH_T(dtype, OD_LAUNCH((rms_affine_film_kernel<T_>), dim3(row_grid(M, C)), dim3(256), 0, (hipStream_t)stream, (const T_*)x,
                                ldx, gamma, ssg, (T_*)y, ldy, M, L, C, eps, act));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_rmsnorm_affine_gate_residual(int dtype, const void* x, int ldx, const void* h, int ldh, const float* gamma,
                                               const float* ssg, void* xo, int ldxo, int B, int L, int C, float eps,
                                               void* stream) {
    if (!lanes_ok(C) || ldx % 8 || ldh % 8 || ldxo % 8) return OD_ERR_ALIGN;
    const long M = (long)B * L;
    DISPATCH_T(dtype, OD_LAUNCH((rms_affine_gate_res_kernel<T_>), dim3(row_grid(M, C)), dim3(256), 0, (hipStream_t)stream,
                                (const T_*)x, ldx, (const T_*)h, ldh, gamma, ssg, (T_*)xo, ldxo, M, L, C, eps));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_unet_mixer(int dtype, const void* x, int ldx, const void* p, int ldp, int p_bcast, const void* gx, int ldg,
                             const float* gamma, void* xo, int ldxo, int B, int L, int C, float eps, void* stream) {
    if (!lanes_ok(C) || ldx % 8 || ldp % 8 || ldg % 8 || ldxo % 8) return OD_ERR_ALIGN;
    const long M = (long)B * L;
    DISPATCH_T(dtype, OD_LAUNCH((mixer_kernel<T_>), dim3(row_grid(M, C)), dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx,
                                (const T_*)p, ldp, p_bcast, (const T_*)gx, ldg, gamma, (T_*)xo, ldxo, M, L, C, eps));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_unet_down(int dtype, const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int B, int Lo,
                            int C, int stride, void* stream) {
    if (!lanes_ok(C) || ldx % 8 || ldy % 8) return OD_ERR_ALIGN;
    if (stride < 1 || stride > 8) return OD_ERR_UNSUPPORTED;
    const long M = (long)B * Lo;
    DISPATCH_T(dtype, OD_LAUNCH((unet_down_kernel<T_>), dim3(row_grid(M, C)), dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx,
                                w, bias, (T_*)y, ldy, B, Lo, C, stride));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_unet_up(int dtype, const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int B, int Li,
                          int C, int stride, void* stream) {
    if (!lanes_ok(C) || ldx % 8 || ldy % 8) return OD_ERR_ALIGN;
    if (stride < 1 || stride > 8) return OD_ERR_UNSUPPORTED;
    const long M = (long)B * Li * stride;
    DISPATCH_T(dtype, OD_LAUNCH((unet_up_kernel<T_>), dim3(row_grid(M, C)), dim3(256), 0, (hipStream_t)stream, (const T_*)x, ldx, w,
                                bias, (T_*)y, ldy, B, Li, C, stride));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_chart_head(int dtype, const void* x, int ldx, const float* W, const float* bias, float* out, int B, int L, int C,
                             int N, int n_sigmoid, int rms, float eps, void* stream) {
    if (!lanes_ok(C) || ldx % 8) return OD_ERR_ALIGN;
    if (N < 1 || N > 16) return OD_ERR_UNSUPPORTED;
    const long M = (long)B * L;
    DISPATCH_T(dtype, OD_LAUNCH((chart_head_kernel<T_, 16>), dim3(row_grid(M, C)), dim3(256), 0, (hipStream_t)stream, (const T_*)x,
                                ldx, W, bias, out, M, L, C, N, n_sigmoid, rms, eps));
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_attn_pool(int dtype, const void* scores, int lds, const void* values, int ldv, float* out, int B, int L, int Hh,
                            int hd, void* stream) {
    if (hd < 1 || hd > 256 || Hh < 1 || L < 1) return OD_ERR_UNSUPPORTED;
    DISPATCH_T(dtype, OD_LAUNCH((attn_pool_kernel<T_>), dim3(Hh, B), dim3(256), 0, (hipStream_t)stream, (const T_*)scores, lds,
                                (const T_*)values, ldv, out, L, Hh, hd));
    OD_CHECK_LAUNCH();
    return 0;
}
