// The attention-backward instruction mix (per MFMA: 0.5 v_exp_f32, 0.5 v_mul_f32, 0.5 v_cvt_pk_bf16_f32, ~0.19 ds_read_b128, 0.25 ds_read_b64_tr_b16)
// with no dependencies, barriers or global loads, at ONE wave per SIMD (64 accumulators) and at TWO (32 accumulators each): the ceiling of either
// organisation for this mix.    ./mix2_power 1|2 [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int NACC, int OCC>
__global__ __launch_bounds__(256, OCC) void k(float* out, int iters) {
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) ((unsigned*)smem)[i] = 0x3f803c00u + i * 2654435761u % 0x7fffu;
    __syncthreads();
    s16x8 fa[8], fb[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { fa[i] = *(const s16x8*)(smem + ((lane * 8 + i) * 16) % 16384); fb[i] = *(const s16x8*)(smem + 16384 + ((lane * 8 + i) * 16) % 16384); }
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (f32x4)(0.f);
    float e[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
    unsigned cv[4] = {0, 0, 0, 0};
    s16x8 lr = fa[0];
    s16x4 tr = {0, 0, 0, 0};
    const unsigned laddr = (unsigned)(lane * 16);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int n = 0; n < NACC; n++) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[n]) : "v"(fa[(n >> 3) & 7]), "v"(fb[n & 7]));
            if ((n & 1) == 0) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(e[(n >> 1) & 7]));
            if ((n & 1) == 1) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(e[(n >> 1) & 7]) : "v"(e[((n >> 1) + 1) & 7]));
            if ((n & 1) == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(cv[(n >> 1) & 3]) : "v"(e[(n >> 1) & 7]), "v"(e[((n >> 1) + 1) & 7]));
            if (n % 5 == 1 && n / 5 < 12) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lr) : "v"(laddr), "i"((n / 5) * 1024));
            if ((n & 3) == 3) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(tr) : "v"(laddr), "i"((n >> 2) * 1024));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; i++) s += e[i];
    s += (float)(cv[0] + cv[1] + cv[2] + cv[3]) + (float)lr[0] + (float)tr[0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int occ = argc > 1 ? atoi(argv[1]) : 1;
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    float* out; hipMalloc(&out, 512 * 256 * 4);
    const int iters = 10000;
    hipFuncSetAttribute((const void*)k<64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k<32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    auto launch = [&]() { if (occ == 1) k<64, 1><<<256, 256, 131072>>>(out, iters); else k<32, 2><<<512, 256, 65536>>>(out, 2 * iters); };
    launch(); hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0; double el = 0;
    while (el < secs) {
        for (int i = 0; i < 10; i++) launch();
        hipDeviceSynchronize();
        n += 10;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // both configurations execute 256 CUs x 4 SIMDs x iters x 64 MFMAs per launch... the two-wave one twice that (512 workgroups x 2 iters x 32)
    printf("%d wave(s) per SIMD: %.0f TF/s executed (bare MFMAs: ~2400)\n", occ, (double)n * 256 * 4 * iters * 64 * 16384 * (occ == 1 ? 1 : 2) / el / 1e12);
    return 0;
}
