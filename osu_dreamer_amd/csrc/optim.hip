// Fused optimizer pass over the flat parameter arena, plus hipGraph helpers and misc ABI.
//   grad-norm clip (Lightning gradient_clip_val, models/diffusion/model.yml:39)
//   -> AdamW (models/diffusion/train.py:110-118, torch defaults)
//   -> EMA  (train.py:67,125-126: first update copies, then lerp by 1-decay)
// One read of g/p/m/v/ema and one write of p/m/v/ema per element: 36 B/param, HBM-bound.
#include "od_common.h"
#include "od_api_internal.h"

namespace {

// `status` (optional): a device-side error word of an earlier kernel of the step (the fused attention backward's, attn_bwd_fused.hip).  Non-zero
// poisons the norm with NaN — no host round trip, checked in EVERY step.
// The sum itself is order-free (always: the clip coefficient multiplies every gradient, so a norm that differs in its last bit from run to run
// would make the whole step differ): every block leaves its partial in a slot, the LAST block to finish adds the slots up in index order.
// (Process-global scratch: one process drives one GPU, and the optimizer's norm is one launch at a time.)
constexpr int SQ_MAX_BLOCKS = 2048;
__device__ float g_sq_part[SQ_MAX_BLOCKS];
__device__ unsigned g_sq_done;
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long n, float* __restrict__ out,
                                                     const int* __restrict__ status) {
    __shared__ float sh[256];
    __shared__ int s_last;
    float s = 0.f;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(g + i * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    s = od_wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        g_sq_part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
#if !defined(OD_EMU)
        __threadfence();
#endif
        s_last = atomicAdd(&g_sq_done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
#if !defined(OD_EMU)
    __threadfence();
#endif
    float t = 0.f;                                        // fixed order: thread t sums slots t, t + 256, ...; then a fixed tree
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += ((volatile float*)g_sq_part)[i];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float tot = sh[0];
        if (status && status[0] != 0) tot = __builtin_nanf("");
        out[0] += tot;
        g_sq_done = 0;
    }
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ ema, long n, float lr,
                                                        float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                                        float ema_decay, int ema_mode, const float* __restrict__ gnorm_sq,
                                                        float max_norm, const int* __restrict__ status) {
    if (status && status[0] != 0) return;        // gradients of a failed launch: the step is skipped (the host raises at its next check)
    float clip = 1.f;
    if (max_norm > 0.f && gnorm_sq) {
        const float c = max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f);
        clip = c < 1.f ? c : 1.f;
    }
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * clip;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (ema_mode == 1) ema[i] = pi;
        else if (ema_mode == 2) { const float e = ema[i]; ema[i] = e + (1.f - ema_decay) * (pi - e); }
    }
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, long n, float decay, int mode) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float pi = p[i];
        if (mode == 1) ema[i] = pi;
        else { const float e = ema[i]; ema[i] = e + (1.f - decay) * (pi - e); }
    }
}

}  // namespace

extern "C" int od_ema_update(float* ema, const float* p, long n, float ema_decay, int ema_mode, void* stream) {
    if (n <= 0 || (ema_mode != 1 && ema_mode != 2)) return OD_ERR_ARG;
    int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
    OD_LAUNCH(ema_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ema, p, n, ema_decay, ema_mode);
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_sqnorm(const float* g, long n, float* out, const int* status, void* stream) {
    int blocks = (int)((n / 4 + 255) / 256); if (blocks > SQ_MAX_BLOCKS) blocks = SQ_MAX_BLOCKS; if (blocks < 1) blocks = 1;
#if !defined(OD_EMU)
    // the arrival counter is re-armed ON THE LAUNCH STREAM in front of every launch: a launch that was aborted (or killed by a fault) cannot
    // leave it non-zero for the next one, which would then never see a last block and leave the norm at 0 — clip off, status lost.
    // The scratch is process-global: launches on DIFFERENT streams at the same time are not supported (one optimizer norm at a time).
    static unsigned* done_addr = [] { void* p = nullptr; return hipGetSymbolAddress(&p, HIP_SYMBOL(g_sq_done)) == hipSuccess ? (unsigned*)p : (unsigned*)nullptr; }();
    if (!done_addr || hipMemsetAsync(done_addr, 0, sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return OD_ERR_ARG;
#endif
    OD_LAUNCH(sqnorm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, out, status);
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, long n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, float ema_decay, int ema_mode, const float* gnorm_sq,
                            float max_norm, const int* status, void* stream) {
    if (step < 1 || n <= 0) return OD_ERR_ARG;
    if (ema_mode != 0 && !ema) return OD_ERR_ARG;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
    OD_LAUNCH(adamw_ema_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, n, lr, beta1, beta2, eps,
              weight_decay, bc1, bc2_sqrt, ema_decay, ema_mode, gnorm_sq, max_norm, status);
    OD_CHECK_LAUNCH();
    return 0;
}

extern "C" int od_version(void) { return 100; }

extern "C" const char* od_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case OD_ERR_ARG: return "invalid argument";
        case OD_ERR_ALIGN: return "leading dimension / size not aligned to the kernel's vector width";
        case OD_ERR_UNSUPPORTED: return "shape outside the compiled kernel set";
        case OD_ERR_COMM: return "RCCL call failed (or librccl could not be bound)";
        default: break;
    }
#if !defined(OD_EMU)
    if (code <= -1000) return hipGetErrorString((hipError_t)(-(code + 1000)));
#endif
    return "unknown error";
}

// ---- hipGraph capture of a launch sequence (the sampler loop) ---------------------------
#if !defined(OD_EMU)
extern "C" int od_graph_begin(void* stream) {
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    return e == hipSuccess ? 0 : -(int)e - 1000;
}
extern "C" int od_graph_end(void* stream, void** graph_exec_out) {
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
    if (e != hipSuccess) return -(int)e - 1000;
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) return -(int)e - 1000;
    *graph_exec_out = (void*)exec;
    return 0;
}
extern "C" int od_graph_launch(void* graph_exec, void* stream) {
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    return e == hipSuccess ? 0 : -(int)e - 1000;
}
extern "C" int od_graph_destroy(void* graph_exec) {
    hipError_t e = hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    return e == hipSuccess ? 0 : -(int)e - 1000;
}
#else
extern "C" int od_graph_begin(void*) { return OD_ERR_UNSUPPORTED; }
extern "C" int od_graph_end(void*, void**) { return OD_ERR_UNSUPPORTED; }
extern "C" int od_graph_launch(void*, void*) { return OD_ERR_UNSUPPORTED; }
extern "C" int od_graph_destroy(void*) { return OD_ERR_UNSUPPORTED; }
#endif
