// Throughput probe: cycles per wave64 instruction for the VALU ops in the attention softmax
// (one wave per SIMD and 2 waves per SIMD), via s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int OP>
__global__ void k(float* out, long long* cyc) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    float c = 1.0001f, d = 0.5f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 256; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
                if (OP == 1) a[i] = fmaf(a[i], c, d);
                if (OP == 2) a[i] = fmaxf(a[i], c);
                if (OP == 3) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {a[i], a[(i + 1) & 7]}; v = v * c; a[i] = v[0]; a[(i + 1) & 7] = v[1]; }
                if (OP == 4) { typedef __bf16 b2 __attribute__((ext_vector_type(2))); typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {a[i], a[(i + 1) & 7]}; b2 b = __builtin_convertvector(v, b2); a[i] += __builtin_bit_cast(float, b); }
                if (OP == 5) a[i] = a[i] + c;
            }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP> void run(const char* name, float* out, long long* cyc) {
    for (int waves : {4, 8, 16}) {
        k<OP><<<256, waves * 64>>>(out, cyc);
        hipDeviceSynchronize();
        long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        double per = (double)h / (256.0 * REP);
        printf("%-14s waves/CU=%2d  cycles(memtime)/instr/wave=%.2f  => per-SIMD issue interval = %.2f\n", name, waves, per, per / (waves / 4.0));
    }
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    run<0>("v_exp_f32", out, cyc); run<1>("v_fma_f32", out, cyc); run<2>("v_max_f32", out, cyc);
    run<3>("v_pk_mul_f32", out, cyc); run<4>("cvt_pk_bf16+add", out, cyc); run<5>("v_add_f32", out, cyc);
    return 0;
}
