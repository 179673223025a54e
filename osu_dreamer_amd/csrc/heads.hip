// Distance head (u_head), diffusion loss and sampler-step kernels.  All fp32, all on the
// reference's boundary layout (B, E, L): these touch only the 6-channel latent.
//   u_head : models/diffusion/model.py:58-65,99-102   loss : models/diffusion/train.py:78-101
//   sampler: models/diffusion/model.py:131-136
#include "od_common.h"
#include "od_api_internal.h"

namespace {

constexpr int UW = 32;        // frames per window (ext-1)
constexpr int UOWN = 30;      // owned frames per window
constexpr int UWPB = 4;       // windows a block walks before flushing its gradient accumulators (8: 1262 us, 4: 1116, 2: 1095 at the bench shape)
constexpr int MAXU = 64, MAXE = 8;

__device__ __forceinline__ float red32(float v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

struct UHeadW {
    const float *w0, *b0, *w1, *b1, *w3, *b3, *w4, *b4;
};

struct UHeadSmem {
    float xts[MAXE][UW + 4];
    float z0s[MAXE][UW + 2];
    float z1s[MAXU][UW + 2];
    float a1s[MAXU][UW + 2];
    float z3s[MAXU][UW + 1];   // later reused for dz3
    float g4s[MAXU][UW + 1];   // dz4, later dz1
    float w4s[MAXU][MAXU + 1];
};

// forward pass of one window into LDS; returns nothing.  l0 = first owned frame.
__device__ __forceinline__ void uhead_window_fwd(UHeadSmem& S, const float* __restrict__ xt, const UHeadW& P, int b, int l0,
                                                 int E, int L, int U) {
    const int t = threadIdx.x;
    for (int idx = t; idx < E * (UW + 4); idx += 256) {
        const int e = idx / (UW + 4), ix = idx % (UW + 4);
        const int f = l0 - 3 + ix;
        S.xts[e][ix] = (f >= 0 && f < L) ? xt[((size_t)b * E + e) * L + f] : 0.f;
    }
    __syncthreads();
    for (int idx = t; idx < E * (UW + 2); idx += 256) {
        const int e = idx / (UW + 2), i2 = idx % (UW + 2);
        S.z0s[e][i2] = P.b0[e] + P.w0[e * 3 + 0] * S.xts[e][i2] + P.w0[e * 3 + 1] * S.xts[e][i2 + 1] + P.w0[e * 3 + 2] * S.xts[e][i2 + 2];
    }
    __syncthreads();
    for (int idx = t; idx < U * (UW + 2); idx += 256) {
        const int c = idx / (UW + 2), i2 = idx % (UW + 2);
        const int f = l0 - 2 + i2;
        float z = P.b1[c];
        for (int e = 0; e < E; e++) z += P.w1[c * E + e] * S.z0s[e][i2];
        S.z1s[c][i2] = z;
        S.a1s[c][i2] = (f >= 0 && f < L) ? od_silu(z) : 0.f;
    }
    __syncthreads();
    for (int idx = t; idx < U * UW; idx += 256) {
        const int c = idx / UW, i = idx % UW;
        S.z3s[c][i] = P.b3[c] + P.w3[c * 3 + 0] * S.a1s[c][i] + P.w3[c * 3 + 1] * S.a1s[c][i + 1] + P.w3[c * 3 + 2] * S.a1s[c][i + 2];
    }
    __syncthreads();
}

__device__ __forceinline__ void uhead_load_w4(UHeadSmem& S, const UHeadW& P, int U) {
    for (int idx = threadIdx.x; idx < U * U; idx += 256) S.w4s[idx / U][idx % U] = P.w4[idx];
    __syncthreads();
}

__global__ __launch_bounds__(256) void uhead_fwd_kernel(const float* __restrict__ xt, UHeadW P, float* __restrict__ fsum,
                                                        int E, int L, int U, int nwin, int wpb, const OdDetTable* __restrict__ det) {
    __shared__ UHeadSmem S;
    const int b = blockIdx.y, t = threadIdx.x;
    const int i = t & 31, grp = t >> 5, cpg = U / 8;
    uhead_load_w4(S, P, U);
    float acc[MAXU / 8];
#pragma unroll
    for (int q = 0; q < MAXU / 8; q++) acc[q] = 0.f;
    for (int wi = 0; wi < wpb; wi++) {
        const int win = blockIdx.x * wpb + wi;
        if (win >= nwin) break;
        const int l0 = win * UOWN;
        uhead_window_fwd(S, xt, P, b, l0, E, L, U);
        const int f = l0 - 1 + i;
        const bool own = (i >= 1 && i < UW - 1 && f < L);
#pragma unroll
        for (int q = 0; q < MAXU / 8; q++) {
            if (q < cpg) {
                const int c = grp * cpg + q;
                float z = P.b4[c];
                for (int cc = 0; cc < U; cc++) z += S.w4s[c][cc] * S.z3s[cc][i];
                if (own) acc[q] += od_silu(z);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < MAXU / 8; q++) {
        const float s = red32(acc[q]);
        if (q < cpg && i == 0) od_red_add(det, fsum + (size_t)b * U + grp * cpg + q, s);
    }
}

struct UHeadG {
    float *dw0, *db0, *dw1, *db1, *dw3, *db3, *dw4, *db4;
};

// (the LDS sums below are each owned by ONE lane — channel c belongs to lane 0 of group c / cpg — so their order is fixed; only the
// block-to-global step depends on the order blocks arrive in)
__global__ __launch_bounds__(256) void uhead_bwd_kernel(const float* __restrict__ xt, UHeadW P, const float* __restrict__ dfm,
                                                        UHeadG G, int E, int L, int U, int nwin, const OdDetTable* __restrict__ det) {
    __shared__ UHeadSmem S;
    __shared__ float sdw3[MAXU][3], sdb3[MAXU], sdb4[MAXU], sdw1[MAXU][MAXE], sdb1[MAXU], sdw0[MAXE][3], sdb0[MAXE];
    const int b = blockIdx.y, t = threadIdx.x;
    const int i = t & 31, grp = t >> 5, cpg = U / 8;
    for (int idx = t; idx < MAXU; idx += 256) {
        sdb3[idx] = 0.f; sdb4[idx] = 0.f; sdb1[idx] = 0.f;
        for (int j = 0; j < 3; j++) sdw3[idx][j] = 0.f;
        for (int e = 0; e < MAXE; e++) sdw1[idx][e] = 0.f;
    }
    if (t < MAXE) { sdb0[t] = 0.f; for (int j = 0; j < 3; j++) sdw0[t][j] = 0.f; }
    uhead_load_w4(S, P, U);
    float adw4[16];
#pragma unroll
    for (int q = 0; q < 16; q++) adw4[q] = 0.f;
    const float invL = 1.0f / (float)L;

    for (int wi = 0; wi < UWPB; wi++) {
        const int win = blockIdx.x * UWPB + wi;
        if (win >= nwin) break;
        const int l0 = win * UOWN;
        uhead_window_fwd(S, xt, P, b, l0, E, L, U);
        const int f = l0 - 1 + i;
        const bool inr = (f >= 0 && f < L);
        const bool own = (i >= 1 && i < UW - 1 && f < L);
        // dz4 for every ext-1 frame
#pragma unroll
        for (int q = 0; q < MAXU / 8; q++) {
            if (q < cpg) {
                const int c = grp * cpg + q;
                float z = P.b4[c];
                for (int cc = 0; cc < U; cc++) z += S.w4s[c][cc] * S.z3s[cc][i];
                const float g = inr ? dfm[(size_t)b * U + c] * invL * od_silu_grad(z) : 0.f;
                S.g4s[c][i] = g;
                const float s = red32(own ? g : 0.f);
                if (i == 0) atomicAdd(&sdb4[c], s);
            }
        }
        __syncthreads();
        // dw4[c][c'] += sum_{owned i} dz4[c][i] * z3[c'][i]
        {
            const int iend = (L - (l0 - 1)) < (UW - 1) ? (L - (l0 - 1)) : (UW - 1);
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int o = t + 256 * q;
                if (o < U * U) {
                    const int c = o / U, cc = o % U;
                    float s = 0.f;
                    for (int ii = 1; ii < iend; ii++) s += S.g4s[c][ii] * S.z3s[cc][ii];
                    adw4[q] += s;
                }
            }
        }
        __syncthreads();
        // dz3[c'][i] = sum_c w4[c][c'] dz4[c][i]  -> overwrite z3s
        float dz3v[MAXU / 8];
#pragma unroll
        for (int q = 0; q < MAXU / 8; q++) {
            dz3v[q] = 0.f;
            if (q < cpg) {
                const int cc = grp * cpg + q;
                float s = 0.f;
                for (int c = 0; c < U; c++) s += S.w4s[c][cc] * S.g4s[c][i];
                dz3v[q] = s;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MAXU / 8; q++)
            if (q < cpg) S.z3s[grp * cpg + q][i] = dz3v[q];
        __syncthreads();
        // dw3, db3 over owned frames; da1 -> dz1 (into g4s) for owned frames
#pragma unroll
        for (int q = 0; q < MAXU / 8; q++) {
            if (q < cpg) {
                const int c = grp * cpg + q;
                const float g = own ? S.z3s[c][i] : 0.f;
                float s = red32(g);
                if (i == 0) atomicAdd(&sdb3[c], s);
                for (int j = 0; j < 3; j++) {
                    s = red32(g * S.a1s[c][i + j]);
                    if (i == 0) atomicAdd(&sdw3[c][j], s);
                }
                float dz1 = 0.f;
                if (own) {
                    float da1 = 0.f;
                    for (int j = 0; j < 3; j++) da1 += P.w3[c * 3 + j] * S.z3s[c][i - j + 1];
                    dz1 = da1 * od_silu_grad(S.z1s[c][i + 1]);
                }
                S.g4s[c][i] = dz1;
                s = red32(dz1);
                if (i == 0) atomicAdd(&sdb1[c], s);
                for (int e = 0; e < E; e++) {
                    s = red32(dz1 * S.z0s[e][i + 1]);
                    if (i == 0) atomicAdd(&sdw1[c][e], s);
                }
            }
        }
        __syncthreads();
        // dz0[e][i] = sum_c w1[c][e] dz1[c][i];  dw0, db0
        {
            const bool ev = grp < E;          // predicate, not a branch: both halves of a wave shuffle together
            const int e = ev ? grp : 0;
            float dz0 = 0.f;
            if (ev)
                for (int c = 0; c < U; c++) dz0 += P.w1[c * E + e] * S.g4s[c][i];
            float s = red32(dz0);
            if (ev && i == 0) atomicAdd(&sdb0[e], s);
            for (int j = 0; j < 3; j++) {
                s = red32(dz0 * S.xts[e][i + 1 + j]);
                if (ev && i == 0) atomicAdd(&sdw0[e][j], s);
            }
        }
        __syncthreads();
    }
    // flush
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int o = t + 256 * q;
        if (o < U * U) od_red_add(det, G.dw4 + o, adw4[q]);
    }
    __syncthreads();
    for (int c = t; c < U; c += 256) {
        od_red_add(det, G.db4 + c, sdb4[c]);
        od_red_add(det, G.db3 + c, sdb3[c]);
        od_red_add(det, G.db1 + c, sdb1[c]);
        for (int j = 0; j < 3; j++) od_red_add(det, G.dw3 + c * 3 + j, sdw3[c][j]);
        for (int e = 0; e < E; e++) od_red_add(det, G.dw1 + c * E + e, sdw1[c][e]);
    }
    if (t < E) {
        od_red_add(det, G.db0 + t, sdb0[t]);
        for (int j = 0; j < 3; j++) od_red_add(det, G.dw0 + t * 3 + j, sdw0[t][j]);
    }
}

// tail: f = fsum/L; fm = f*(1+mod[0:U]) + mod[U:2U]; y = w.fm + b; u = u_scale*softplus(y)
__global__ void uhead_tail_kernel(const float* __restrict__ fsum, const float* __restrict__ mod, const float* __restrict__ w,
                                  const float* __restrict__ bo, float* __restrict__ u, int B, int U, int L, float u_scale) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int c = lane; c < U; c += 64) {
        const float f = fsum[(size_t)b * U + c] / (float)L;
        s += w[c] * (f * (1.f + mod[(size_t)b * 2 * U + c]) + mod[(size_t)b * 2 * U + U + c]);
    }
    s = od_wave_sum(s);
    if (lane == 0) {
        const float y = s + bo[0];
        const float sp = y > 20.f ? y : log1pf(__expf(y));
        u[b] = u_scale * sp;
    }
}
__global__ void uhead_tail_bwd_kernel(const float* __restrict__ fsum, const float* __restrict__ mod, const float* __restrict__ w,
                                      const float* __restrict__ bo, const float* __restrict__ du, float* __restrict__ dfm,
                                      float* __restrict__ dmod, float* __restrict__ dw, float* __restrict__ dbo,
                                      int B, int U, int L, float u_scale, const OdDetTable* __restrict__ det) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int c = lane; c < U; c += 64) {
        const float f = fsum[(size_t)b * U + c] / (float)L;
        s += w[c] * (f * (1.f + mod[(size_t)b * 2 * U + c]) + mod[(size_t)b * 2 * U + U + c]);
    }
    s = od_wave_sum(s);
    const float y = s + bo[0];
    const float dy = du[b] * u_scale * od_sigmoid(y);   // d softplus = sigmoid
    for (int c = lane; c < U; c += 64) {
        const float f = fsum[(size_t)b * U + c] / (float)L;
        const float sc = mod[(size_t)b * 2 * U + c], sh = mod[(size_t)b * 2 * U + U + c];
        const float fm = f * (1.f + sc) + sh;
        od_red_add(det, dw + c, dy * fm);
        const float dfmv = dy * w[c];
        dfm[(size_t)b * U + c] = dfmv * (1.f + sc);      // gradient wrt the mean-pooled feature f
        dmod[(size_t)b * 2 * U + c] = dfmv * f;
        dmod[(size_t)b * 2 * U + U + c] = dfmv;
    }
    if (lane == 0) od_red_add(det, dbo, dy);
}

// ------------------------------------------------------------ loss
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = od_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void make_xt_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                      const float* __restrict__ t, float* __restrict__ xt,
                                                      float* __restrict__ dsq, int EL, int L, const OdDetTable* __restrict__ det) {
    __shared__ float sh[4];
    const int b = blockIdx.y;
    const float w = t[b];
    float acc = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < EL; i += gridDim.x * 256) {
        const size_t o = (size_t)b * EL + i;
        const float a = x0[o], e = x1[o];
        // torch.lerp: w < 0.5 ? a + w*(e-a) : e - (e-a)*(1-w)
        const float v = w < 0.5f ? a + w * (e - a) : e - (e - a) * (1.f - w);
        xt[o] = v;
        const float d = v - e;
        acc += d * d;
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) od_red_add(det, dsq + b, acc / (float)L);
}

// sums[b][0] = S1 = fds(xt - u v, x1), [1] = S2 = fds(v, vt), [2] = dS1/du ; dv = dLoss/dv
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ xt, const float* __restrict__ x1,
                                                        const float* __restrict__ u, const float* __restrict__ v,
                                                        const float* __restrict__ dsq, float* __restrict__ dv,
                                                        float* __restrict__ sums, int B, int EL, int L, float c0,
                                                        float osl_w, float del_w, const OdDetTable* __restrict__ det) {
    __shared__ float sh[4];
    const int b = blockIdx.y;
    const float ub = u[b], den = dsq[b] + c0, ut = sqrtf(den);
    const float k1 = osl_w / ((float)B * den), k2 = del_w / (float)B, invL2 = 2.f / (float)L;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < EL; i += gridDim.x * 256) {
        const size_t o = (size_t)b * EL + i;
        const float xv = xt[o], e = x1[o], vv = v[o];
        const float r1 = xv - ub * vv - e;          // denoised - x1
        const float r2 = vv - (xv - e) / ut;        // v - v_target
        s1 += r1 * r1; s2 += r2 * r2; s3 += r1 * (-vv);
        dv[o] = k1 * invL2 * r1 * (-ub) + k2 * invL2 * r2;
    }
    s1 = block_sum_256(s1, sh); s2 = block_sum_256(s2, sh); s3 = block_sum_256(s3, sh);
    if (threadIdx.x == 0) {
        od_red_add(det, sums + b * 3 + 0, s1 / (float)L);
        od_red_add(det, sums + b * 3 + 1, s2 / (float)L);
        od_red_add(det, sums + b * 3 + 2, s3 * invL2);
    }
}

__global__ void loss_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ dsq, const float* __restrict__ u,
                                     float* __restrict__ out, float* __restrict__ du, int B, float c0, float osl_w, float del_w) {
    const int lane = threadIdx.x;
    float osl = 0.f, del = 0.f, mape = 0.f;
    for (int b = lane; b < B; b += 64) {
        const float den = dsq[b] + c0, ut = sqrtf(den);
        osl += sums[b * 3 + 0] / den;
        del += sums[b * 3 + 1];
        mape += fabsf((u[b] - ut) / ut);
        du[b] = osl_w / ((float)B * den) * sums[b * 3 + 2];
    }
    osl = od_wave_sum(osl) / (float)B; del = od_wave_sum(del) / (float)B; mape = od_wave_sum(mape) / (float)B;
    if (lane == 0) { out[0] = osl_w * osl + del_w * del; out[1] = osl; out[2] = del; out[3] = mape; }
}

__global__ __launch_bounds__(256) void sampler_step_kernel(float* __restrict__ x, const float* __restrict__ u,
                                                           const float* __restrict__ v, const float* __restrict__ eta, int EL) {
    const int b = blockIdx.y;
    const float k = eta[0] * u[b];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < EL; i += gridDim.x * 256) {
        const size_t o = (size_t)b * EL + i;
        x[o] = x[o] - k * v[o];
    }
}

__global__ void sampler_eta_kernel(const float* __restrict__ u, float* __restrict__ eta, int B, float c0, int num_steps) {
    const int lane = threadIdx.x;
    float s = 0.f;
    for (int b = lane; b < B; b += 64) s += u[b];
    s = od_wave_sum(s) / (float)B;
    if (lane == 0) {
        const float r = sqrtf(c0);
        const float u0 = s > r + 1e-6f ? s : r + 1e-6f;
        eta[0] = 1.f - powf(r / u0, 1.f / (float)num_steps);
        eta[1] = s;
    }
}

}  // namespace

extern "C" int od_uhead_fwd(const float* xt, const float* w0, const float* b0, const float* w1, const float* b1, const float* w3,
                            const float* b3, const float* w4, const float* b4, float* fsum, int B, int E, int L, int U,
                            void* stream) {
    if (U > MAXU || U % 8 || E > MAXE) return OD_ERR_UNSUPPORTED;
    const int nwin = (L + UOWN - 1) / UOWN;
    UHeadW P{w0, b0, w1, b1, w3, b3, w4, b4};
    // windows a block walks: up to UWPB for long sequences, fewer when that would leave CUs idle (sampler: L ~ 1e3)
    int wpb = (int)((long)nwin * B / 512);
    wpb = wpb < 1 ? 1 : (wpb > UWPB ? UWPB : wpb);
    OD_LAUNCH(uhead_fwd_kernel, dim3((nwin + wpb - 1) / wpb, B), dim3(256), 0, (hipStream_t)stream, xt, P, fsum, E, L, U, nwin, wpb, od_det_active());
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_uhead_bwd(const float* xt, const float* w0, const float* b0, const float* w1, const float* b1, const float* w3,
                            const float* b3, const float* w4, const float* b4, const float* dfm, float* dw0, float* db0,
                            float* dw1, float* db1, float* dw3, float* db3, float* dw4, float* db4, int B, int E, int L, int U,
                            void* stream) {
    if (U > MAXU || U % 8 || E > MAXE) return OD_ERR_UNSUPPORTED;
    const int nwin = (L + UOWN - 1) / UOWN;
    UHeadW P{w0, b0, w1, b1, w3, b3, w4, b4};
    UHeadG G{dw0, db0, dw1, db1, dw3, db3, dw4, db4};
    OD_LAUNCH(uhead_bwd_kernel, dim3((nwin + UWPB - 1) / UWPB, B), dim3(256), 0, (hipStream_t)stream, xt, P, dfm, G, E, L, U, nwin, od_det_active());
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_uhead_tail(const float* fsum, const float* mod, const float* w_out, const float* b_out, float* u, int B, int U,
                             int L, float u_scale, void* stream) {
    OD_LAUNCH(uhead_tail_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, fsum, mod, w_out, b_out, u, B, U, L, u_scale);
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_uhead_tail_bwd(const float* fsum, const float* mod, const float* w_out, const float* b_out, const float* du,
                                 float* dfm, float* dmod, float* dw_out, float* db_out, int B, int U, int L, float u_scale,
                                 void* stream) {
    OD_LAUNCH(uhead_tail_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, fsum, mod, w_out, b_out, du, dfm, dmod, dw_out,
              db_out, B, U, L, u_scale, od_det_active());
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_make_xt(const float* x0, const float* x1, const float* t, float* xt, float* dsq, int B, int E, int L, void* stream) {
    const int EL = E * L;
    int gx = (EL + 255) / 256; if (gx > 256) gx = 256;
    OD_LAUNCH(make_xt_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x0, x1, t, xt, dsq, EL, L, od_det_active());
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_loss_grad(const float* xt, const float* x1, const float* u, const float* v, const float* dsq, float* dv,
                            float* sums, int B, int E, int L, float c0, float osl_w, float del_w, void* stream) {
    const int EL = E * L;
    int gx = (EL + 255) / 256; if (gx > 256) gx = 256;
    OD_LAUNCH(loss_grad_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, xt, x1, u, v, dsq, dv, sums, B, EL, L, c0, osl_w, del_w, od_det_active());
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_loss_finalize(const float* sums, const float* dsq, const float* u, float* out, float* du, int B, float c0,
                                float osl_w, float del_w, void* stream) {
    OD_LAUNCH(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, dsq, u, out, du, B, c0, osl_w, del_w);
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_sampler_step(float* x, const float* u, const float* v, const float* eta, int B, int E, int L, void* stream) {
    const int EL = E * L;
    int gx = (EL + 255) / 256; if (gx > 256) gx = 256;
    OD_LAUNCH(sampler_step_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, u, v, eta, EL);
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_sampler_eta(const float* u, float* eta, int B, float c0, int num_steps, void* stream) {
    OD_LAUNCH(sampler_eta_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, u, eta, B, c0, num_steps);
    OD_CHECK_LAUNCH();
    return 0;
}
