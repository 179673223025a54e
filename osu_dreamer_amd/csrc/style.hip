// Style-model sampler pieces that are not already covered by the denoiser's entry points
// (osu_dreamer/models/style/model.py:73-100).  Everything is fp32 on [B][H] vectors (B = diffs, H = 256).
#include "od_common.h"
#include "od_api_internal.h"

namespace {

// c[b][h] = sum_n ( labels[b][n] < 0 ? null[n][h] : cond_b[n][h] + sum_f rff(b,n,f) * cond_w[n][f][h] ),
// rff(b,n,f) = scale * cos(labels[b][n]/10 * W[f] + bias[f])           — style/model.py:73-80
__global__ __launch_bounds__(256) void style_cond_kernel(const float* __restrict__ labels, const float* __restrict__ rffW,
                                                         const float* __restrict__ rffb, const float* __restrict__ cw,
                                                         const float* __restrict__ cb, const float* __restrict__ nul,
                                                         float* __restrict__ c, int NL, int F, int H, float scale) {
    OD_DYN_SMEM(smem_raw);
    float* s_rff = (float*)smem_raw;       // [NL][F]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < NL * F; i += 256) {
        const int n = i / F, f = i % F;
        s_rff[i] = scale * cosf(labels[b * NL + n] / 10.0f * rffW[f] + rffb[f]);
    }
    __syncthreads();
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    float acc = 0.f;
    for (int n = 0; n < NL; n++) {
        if (labels[b * NL + n] < 0.f) { acc += nul[n * H + h]; continue; }
        float s = cb[n * H + h];
        for (int f = 0; f < F; f++) s += s_rff[n * F + f] * cw[((size_t)n * F + f) * H + h];
        acc += s;
    }
    c[(size_t)b * H + h] = acc;
}

// y[m] = x[m] * rsqrt(mean(x[m]^2) + eps) (* gamma)        — nn.RMSNorm / rms_norm on row vectors
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           float* __restrict__ y, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = x[(size_t)m * C + c]; ss += v * v; }
    const float inv = rsqrtf(od_wave_sum(ss) / (float)C + eps);
    for (int c = lane; c < C; c += 64) y[(size_t)m * C + c] = x[(size_t)m * C + c] * inv * (gamma ? gamma[c] : 1.f);
}

}  // namespace

extern "C" int od_style_conditioning(const float* labels, const float* rff_w, const float* rff_b, const float* cond_w,
                                     const float* cond_b, const float* null_labels, float* c, int B, int NL, int F, int H,
                                     void* stream) {
    if (NL * F * 4 > 60000) return OD_ERR_UNSUPPORTED;
    const float scale = sqrtf(2.0f / (float)F);
    OD_LAUNCH(style_cond_kernel, dim3((H + 255) / 256, B), dim3(256), NL * F * sizeof(float), (hipStream_t)stream, labels, rff_w, rff_b,
              cond_w, cond_b, null_labels, c, NL, F, H, scale);
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_rmsnorm_rows(const float* x, const float* gamma, float* y, int M, int C, float eps, void* stream) {
    OD_LAUNCH(rmsnorm_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, y, M, C, eps);
    OD_CHECK_LAUNCH();
    return 0;
}
