// Do LDS reads overlap with MFMA work on a gfx950 SIMD?  2 waves per SIMD, hipEvent timing.
//  mode 0: 16 MFMA per wave-iteration       mode 1: 6 ds_read_b128 per wave-iteration
//  mode 2: 16 MFMA + 6 ds_read_b128 interleaved (the 256x256 GEMM's ratio: 24 reads per 64 MFMAs)
//  mode 3: 16 MFMA + 12 ds_read_b64_tr_b16 interleaved (the TN GEMM's ratio: 48 per 64)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define ITERS 2048
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 16384; i += 512) ((float*)lds)[i] = i * 1e-6f;
    __syncthreads();
    s16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    f32x4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (f32x4)(0.f);
    s16x8 f[6];
    for (int i = 0; i < 6; i++) f[i] = a;
    const int lane = threadIdx.x & 63;
    int off = ((threadIdx.x * 16) & 8191);
    for (int it = 0; it < ITERS; it++) {
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 6; i++) f[i] = *(const s16x8*)(lds + ((off + i * 8192 + it * 1024) & 65535 & ~15));
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + ((off + i * 8192 + it * 1024) & 65535 & ~15)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + ((off + i * 8192 + it * 1024 + 512) & 65535 & ~15)));
                f[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
        if (MODE != 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(MODE >= 2 ? f[i % 6] : a, b, acc[i & 7], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 6; i++) a += f[i];
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    s += a[0] + lane;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 512>>>(out); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, 512>>>(out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %8.3f ms   per SIMD per round of 2 wave-iterations: %7.1f ns\n", name, ms, ms * 1e6 / ITERS);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0>("2 waves x 16 MFMA", out);
    run<1>("2 waves x 6 ds_read_b128", out);
    run<2>("2 waves x (16 MFMA fed by 6 ds_read_b128)", out);
    run<3>("2 waves x (16 MFMA fed by 12 ds_read_b64_tr_b16)", out);
    return 0;
}
