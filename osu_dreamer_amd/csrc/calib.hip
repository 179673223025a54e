// Bare-MFMA calibration of a box: what dense bf16 rate does THIS board sustain under its power cap right now?  bench.py runs it for ~2 s beside
// every measurement, so that a step time can be read against the box it was taken on (the pool's boxes differ by +-5 % at the cap).
// One wave per SIMD, 4 waves per workgroup, one workgroup per CU; 64 independent 16x16x32 accumulators; pseudo-random operands (the power an
// MFMA draws depends on how its operand bits toggle); nothing but MFMAs in the loop.
#include "od_common.h"
#include "od_api_internal.h"

#if !defined(OD_EMU)
namespace {
__global__ __launch_bounds__(256, 1) void mfma_calib_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    s16x8 fa[8], fb[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const unsigned h = (unsigned)(lane * 64 + i * 8 + e) * 2654435761u;
            fa[i][e] = (short)(0x3f00 + ((h >> 7) & 0xff) + ((h >> 3) & 0x8000));
            fb[i][e] = (short)(0x3f00 + ((h >> 17) & 0xff) + ((h >> 11) & 0x8000));
        }
    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; i++) acc[i] = (f32x4)(0.f);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int n = 0; n < 64; n++) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[n]) : "v"(fa[n >> 3]), "v"(fb[n & 7]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 64; i++) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace
#endif

// One launch of the calibration kernel: `iters` x 64 MFMAs per wave, 4 waves on each of the device's CUs; *flops_out = the FLOPs it executes.
// `scratch` = device memory of at least 4 * 256 * (number of CUs) bytes.  The caller times it (HIP events on `stream`).
extern "C" int od_mfma_calibrate(float* scratch, int iters, double* flops_out, void* stream) {
    if (!scratch || iters <= 0 || !flops_out) return OD_ERR_ARG;
#if defined(OD_EMU)
    (void)stream;
    return OD_ERR_UNSUPPORTED;
#else
    const int cus = od_num_cus();
    // 140 KiB of (unused) LDS per workgroup keeps it at one workgroup per CU
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)mfma_calib_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    if (attr_rc != hipSuccess) return -(int)attr_rc - 1000;
    hipLaunchKernelGGL(mfma_calib_kernel, dim3(cus), dim3(256), 140 * 1024, (hipStream_t)stream, scratch, iters);
    OD_CHECK_LAUNCH();
    *flops_out = (double)cus * 4.0 * (double)iters * 64.0 * 16384.0;
    return 0;
#endif
}
