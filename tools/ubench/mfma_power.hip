// Sustained MFMA rate at the board's power limit, by MFMA shape: ./mfma_power 16|32 [seconds]
//   16: v_mfma_f32_16x16x32_bf16, 64 independent accumulators;  32: v_mfma_f32_32x32x16_bf16, 16 independent accumulators (256 AGPRs both).
// One wave per SIMD, 4 waves per workgroup, one workgroup per CU, nothing but MFMAs in the loop.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63;
    s16x8 fa[8], fb[8];
    // pseudo-random bf16 operands in [-2, 2): the power an MFMA draws depends on how its operand bits toggle
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const unsigned h = (unsigned)(lane * 64 + i * 8 + e) * 2654435761u;
            fa[i][e] = (short)(0x3f00 + ((h >> 7) & 0xff) + ((h >> 3) & 0x8000));
            fb[i][e] = (short)(0x3f00 + ((h >> 17) & 0xff) + ((h >> 11) & 0x8000));
        }
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[64];
#pragma unroll
        for (int i = 0; i < 64; i++) acc[i] = (f32x4)(0.f);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int n = 0; n < 64; n++) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[n]) : "v"(fa[n >> 3]), "v"(fb[n & 7]));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 64; i++) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[16];
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = (f32x16)(0.f);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 2; r++)                      // 32 MFMAs of 32x32x16 = the FLOPs of 64 of 16x16x32
#pragma unroll
                for (int n = 0; n < 16; n++) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[n]) : "v"(fa[(n >> 2) + 4 * r]), "v"(fb[(n & 3) + 4 * r]));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][15];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 16;
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    float* out; hipMalloc(&out, 256 * 256 * 4);
    const int iters = 20000;
    hipFuncSetAttribute((const void*)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    auto launch = [&]() { if (shape == 16) k<16><<<256, 256, 131072>>>(out, iters); else k<32><<<256, 256, 131072>>>(out, iters); };
    launch(); hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    double el = 0;
    while (el < secs) {
        for (int i = 0; i < 10; i++) launch();
        hipDeviceSynchronize();
        n += 10;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("shape %d: %d launches in %.2f s: %.0f TF/s\n", shape, n, el, (double)n * 256 * 4 * iters * 64 * 16384 / el / 1e12);
    return 0;
}
