// Sustained rate at the power limit of a 64-MFMA loop with one class of filler added: ./mix_power MODE [seconds]
//   0 none | 1 +32 v_exp_f32 | 2 +32 v_cvt_pk_bf16_f32 | 3 +16 v_pk_mul_f32 | 4 +16 ds_read_b128 | 5 +32 ds_read_b64_tr_b16 | 6 = 1+2+3 | 7 = 1+2+3+4+5
//   8 +32 v_exp_f16 | 9 +32 v_fma_f32 | 10 the attention-backward mix without packed math: 32 exp, 32 v_mul, 32 cvt, 12 ds_read_b128, 16 tr reads
//   11 = 10 with two fillers behind every MFMA that has any (bunched)
// One wave per SIMD, fillers between MFMAs (one per two MFMAs), random-ish operand data.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768; i += 256) ((unsigned*)smem)[i] = 0x3f803c00u + i * 2654435761u % 0x7fffu;
    __syncthreads();
    s16x8 fa[8], fb[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { fa[i] = *(const s16x8*)(smem + ((lane * 8 + i) * 16)); fb[i] = *(const s16x8*)(smem + 32768 + ((lane * 8 + i) * 16)); }
    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; i++) acc[i] = (f32x4)(0.f);
    float e[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
    f32x2 pm[4] = {{1.01f, 0.99f}, {1.02f, 0.98f}, {1.03f, 0.97f}, {1.04f, 0.96f}};
    unsigned cv[4] = {0, 0, 0, 0};
    s16x8 lr = fa[0];
    s16x4 tr = {0, 0, 0, 0};
    const unsigned laddr = (unsigned)(lane * 16);
    constexpr bool EXP = MODE == 1 || MODE == 6 || MODE == 7, CVT = MODE == 2 || MODE == 6 || MODE == 7, MUL = MODE == 3 || MODE == 6 || MODE == 7,
                   LDS = MODE == 4 || MODE == 7, TRR = MODE == 5 || MODE == 7;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int n = 0; n < 64; n++) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[n]) : "v"(fa[n >> 3]), "v"(fb[n & 7]));
            if (EXP && (n & 1) == 0) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(e[(n >> 1) & 7]));
            if (MODE == 8 && (n & 1) == 0) asm volatile("v_exp_f16_e32 %0, %0" : "+v"(e[(n >> 1) & 7]));
            if (MODE == 9 && (n & 1) == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(e[(n >> 1) & 7]) : "v"(e[((n >> 1) + 1) & 7]), "v"(e[((n >> 1) + 2) & 7]));
            if (CVT && (n & 1) == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(cv[(n >> 1) & 3]) : "v"(e[(n >> 1) & 7]), "v"(e[((n >> 1) + 1) & 7]));
            if (MUL && (n & 3) == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pm[(n >> 2) & 3]) : "v"(pm[((n >> 2) + 1) & 3]));
            if (LDS && (n & 3) == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lr) : "v"(laddr), "i"((n >> 2) * 1024));
            if (TRR && (n & 1) == 1) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(tr) : "v"(laddr), "i"((n >> 1) * 1024));
            if (MODE == 10 || MODE == 11) {
                // 124 fillers over 64 MFMAs: ~2 per MFMA
                if ((n & 1) == 0) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(e[(n >> 1) & 7]));
                if ((n & 1) == 1) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(e[(n >> 1) & 7]) : "v"(e[((n >> 1) + 1) & 7]));
                if ((n & 1) == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(cv[(n >> 1) & 3]) : "v"(e[(n >> 1) & 7]), "v"(e[((n >> 1) + 1) & 7]));
                if (n % 5 == 1 && n / 5 < 12) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lr) : "v"(laddr), "i"((n / 5) * 1024));
                if ((n & 3) == 3) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(tr) : "v"(laddr), "i"((n >> 2) * 1024));
            }
        }
        if (LDS || TRR || MODE == 10 || MODE == 11) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 64; i++) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; i++) s += e[i];
    s += pm[0][0] + pm[1][1] + pm[2][0] + pm[3][1] + (float)(cv[0] + cv[1] + cv[2] + cv[3]) + (float)lr[0] + (float)tr[0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(float* out, double secs) {
    const int iters = 10000;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    k<MODE><<<256, 256, 131072>>>(out, iters); hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0; double el = 0;
    while (el < secs) {
        for (int i = 0; i < 10; i++) k<MODE><<<256, 256, 131072>>>(out, iters);
        hipDeviceSynchronize();
        n += 10;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("mode %d: %.0f TF/s  (%.1f ns per 64-MFMA iteration)\n", MODE, (double)n * 256 * 4 * iters * 64 * 16384 / el / 1e12, el / n / iters * 1e9);
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    float* out; hipMalloc(&out, 256 * 256 * 4);
    switch (mode) {
        case 0: run<0>(out, secs); break; case 1: run<1>(out, secs); break; case 2: run<2>(out, secs); break; case 3: run<3>(out, secs); break;
        case 4: run<4>(out, secs); break; case 5: run<5>(out, secs); break; case 6: run<6>(out, secs); break; case 7: run<7>(out, secs); break;
        case 8: run<8>(out, secs); break; case 9: run<9>(out, secs); break; case 10: run<10>(out, secs); break;
    }
    return 0;
}
