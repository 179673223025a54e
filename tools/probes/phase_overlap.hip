// Does a wave's MFMA phase overlap ANOTHER wave's VALU phase on the same SIMD?  Each wave loops over
//   [NM MFMAs (32x32x16, CH independent accumulator chains)] then [NV VALU ops (3/8 v_exp_f32, 5/8 v_add_f32)]
// — the shape of an attention tile (scores -> softmax -> PV) — either as two coarse phases (how hipcc lays the tile out) or
// finely interleaved (one MFMA, then NV/NM VALU ops).  1, 2, 3 waves per SIMD, one workgroup per CU; the waves of a SIMD are
// de-phased by a per-wave start delay so that coarse phases CAN overlap if the hardware allows it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define ITERS 2048

template <int NM, int CH, int NV, bool FINE>
__global__ __launch_bounds__(768) void k(float* out, long long* cyc) {
    extern __shared__ unsigned char smem[];
    s16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (short)(0x3f80 + (threadIdx.x & 3)); b[i] = (short)(0x3f00 + (i & 1)); }
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (f32x16)(0.f);
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 1e-3f + i * 0.1f;
    const float c = 0.999f;
    const int wave = threadIdx.x >> 6;
    // de-phase the waves that share a SIMD (waves w, w+4, w+8): burn half an iteration of VALU per step
    for (int d = 0; d < (wave >> 2) * (NV / 2 + NM * 4); d++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[d & 7]) : "v"(c));
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
        if (FINE) {
#pragma unroll
            for (int i = 0; i < NM; i++) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i % CH]) : "v"(a), "v"(b));
#pragma unroll
                for (int r = 0; r < NV / NM; r++) {
                    const int j = i * (NV / NM) + r;
                    if (j % 8 < 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
                    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(c));
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NM; i++) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i % CH]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < NV; j++) {
                if (j % 8 < 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(c));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][15];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * 768 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * wave] = t0; cyc[2 * wave + 1] = t1; }
}
static float* g_out; static long long* g_cyc;
template <int NM, int CH, int NV, bool FINE> void run(const char* name) {
    auto kern = k<NM, CH, NV, FINE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int threads : {256, 512, 768}) {
        kern<<<256, threads, 96 * 1024>>>(g_out, g_cyc); hipDeviceSynchronize();
        kern<<<256, threads, 96 * 1024>>>(g_out, g_cyc); hipDeviceSynchronize();
        long long hh[24]; hipMemcpy(hh, g_cyc, sizeof(hh), hipMemcpyDeviceToHost);
        const int nw = threads / 64, wps = threads / 256;
        // per SIMD: the slowest of its waves sets the time; iterations done by the SIMD = wps * ITERS
        double worst = 0;
        for (int w = 0; w < nw; w++) { const double d = (double)(hh[2 * w + 1] - hh[2 * w]); worst = d > worst ? d : worst; }
        const double per = worst / ITERS / wps;
        const double mfma = NM * 32.0, valu = NV * (3 * 8.5 + 5 * 4.5) / 8.0;
        printf("%-46s waves/SIMD=%d  cycles per (NM mfma + NV valu) per SIMD = %7.1f   [mfma pipe %5.0f | valu issue ~%5.0f | sum %5.0f]\n", name, wps, per, mfma,
               valu, mfma + valu);
    }
}
int main() {
    hipMalloc(&g_out, 256 * 768 * 4); hipMalloc(&g_cyc, 24 * 8);
    run<16, 2, 64, false>("coarse: 16 mfma (2 chains) then 64 valu");
    run<16, 4, 64, false>("coarse: 16 mfma (4 chains) then 64 valu");
    run<16, 2, 64, true>("fine:   16 x (1 mfma (2 chains) + 4 valu)");
    run<16, 4, 64, true>("fine:   16 x (1 mfma (4 chains) + 4 valu)");
    run<16, 2, 96, false>("coarse: 16 mfma (2 chains) then 96 valu");
    run<16, 2, 96, true>("fine:   16 x (1 mfma (2 chains) + 6 valu)");
    run<16, 4, 96, true>("fine:   16 x (1 mfma (4 chains) + 6 valu)");
    run<16, 1, 64, false>("coarse: 16 mfma (1 chain) then 64 valu");
    run<16, 1, 64, true>("fine:   16 x (1 mfma (1 chain) + 4 valu)");
    return 0;
}
