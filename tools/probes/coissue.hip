// Does VALU work overlap with MFMA work on a gfx950 SIMD?  Two waves per SIMD (512 threads/block, 256 blocks, 1 block/CU
// forced by 64 KiB of LDS... not needed: 8 waves/CU = 2 per SIMD), timed with hipEvents; reports SIMD-time per instruction at
// the measured duration.  Modes: MFMA only / VALU only / even waves MFMA + odd waves VALU / both interleaved in each wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define ITERS 4096

template <int MODE, int VOP>   // VOP: 0 = v_fma_f32, 1 = v_exp_f32, 2 = v_pk_mul_f32, 3 = v_cvt_pk_bf16_f32
__global__ __launch_bounds__(512) void k(float* out) {
    const int wave = threadIdx.x >> 6;
    s16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    f32x4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (f32x4)(0.f);
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 1e-3f + i * 0.1f;
    const float c = 0.999f, d = 0.001f;
    const bool do_m = MODE == 0 || MODE == 3 || (MODE == 2 && (wave & 1) == 0);
    const bool do_v = MODE == 1 || MODE == 3 || (MODE == 2 && (wave & 1) == 1);
    for (int it = 0; it < ITERS; it++) {
        if (do_m && !do_v) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
        } else if (do_v && !do_m) {
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (VOP == 0) v[i] = fmaf(v[i], c, d);
                    if (VOP == 1) v[i] = __builtin_amdgcn_exp2f(v[i]);
                    if (VOP == 2 && (i & 1) == 0) { typedef float f2 __attribute__((ext_vector_type(2))); f2 t = {v[i], v[i + 1]}; t = t * c; v[i] = t[0]; v[i + 1] = t[1]; }
                    if (VOP == 3 && (i & 1) == 0) { typedef __bf16 b2 __attribute__((ext_vector_type(2))); typedef float f2 __attribute__((ext_vector_type(2))); f2 t = {v[i], v[i + 1]}; b2 q = __builtin_convertvector(t, b2); v[i] = __builtin_bit_cast(float, q); }
                }
        } else if (do_m && do_v) {   // 8 MFMA + 24 VALU interleaved 1 : 3
#pragma unroll
            for (int i = 0; i < 8; i++) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const int j = (i * 3 + r) & 7;
                    if (VOP == 0) v[j] = fmaf(v[j], c, d);
                    if (VOP == 1) v[j] = __builtin_amdgcn_exp2f(v[j]);
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int VOP> void run(const char* name, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, VOP><<<256, 512>>>(out); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, VOP><<<256, 512>>>(out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves; per wave per iteration: 8 MFMA and/or 24 VALU
    printf("%-44s %8.3f ms   per SIMD per iteration-pair: %7.1f ns  (= %6.0f cycles @2.4GHz)\n", name, ms, ms * 1e6 / ITERS, ms * 1e6 / ITERS * 2.4);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0, 0>("2 waves x 8 MFMA", out);
    run<1, 0>("2 waves x 24 v_fma", out);
    run<1, 1>("2 waves x 24 v_exp", out);
    run<1, 2>("2 waves x 12 v_pk_mul", out);
    run<1, 3>("2 waves x 12 v_cvt_pk_bf16", out);
    run<2, 0>("wave A 8 MFMA | wave B 24 v_fma", out);
    run<2, 1>("wave A 8 MFMA | wave B 24 v_exp", out);
    run<3, 0>("2 waves x (8 MFMA + 24 v_fma interleaved)", out);
    run<3, 1>("2 waves x (8 MFMA + 24 v_exp interleaved)", out);
    return 0;
}
