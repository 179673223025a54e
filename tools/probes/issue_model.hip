// Issue-model probe for a gfx950 SIMD: how many ordinary instructions hide under one MFMA?
//
// Replaces the inconclusive round-1 probe (coissue.hip): that one had no single-wave control and its
// "wave A MFMA | wave B VALU" mode put the two kinds on different SIMDs (a workgroup's waves go to the
// SIMDs in the cyclic order 0,2,1,3, so even waves share SIMDs 0/1 and odd waves SIMDs 2/3).
//
// Every stream here is pinned with `asm volatile` statements (hipcc keeps volatile asm in program order),
// one workgroup per CU (96 KiB of LDS), 1 or 2 waves per SIMD, timed with s_memtime (shader cycles) and
// HIP events.  Per MFMA the loop issues NF filler instructions of one kind; the table gives cycles per MFMA.
//   shapes: v_mfma_f32_16x16x32_bf16 (4 passes = 16 cycles) and v_mfma_f32_32x32x16_bf16 (8 passes = 32 cycles)
//   fillers: v_fma_f32, v_exp_f32, v_cvt_pk_bf16_f32, v_add_f32, v_max_f32, ds_read_b128, s_nop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define ITERS 8192

enum { F_FMA = 0, F_EXP, F_CVT, F_ADD, F_MAX, F_DSR, F_NOP };

template <int FOP>
__device__ __forceinline__ void filler(float& v, float& w, float c, unsigned lds_addr, u32x4& sink) {
    if (FOP == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(c));
    if (FOP == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
    if (FOP == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v) : "v"(w));
    if (FOP == F_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(c));
    if (FOP == F_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v) : "v"(c));
    if (FOP == F_DSR) asm volatile("ds_read_b128 %0, %1" : "=v"(sink) : "v"(lds_addr));
    if (FOP == F_NOP) asm volatile("s_nop 0");
}

// MODE 0: MFMA + NF fillers per MFMA;  MODE 1: fillers only (NF per slot, no MFMA);  MODE 2: MFMAs only
template <int SHAPE, int NF, int FOP, int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    s16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (short)(0x3f80 + (threadIdx.x & 3)); b[i] = (short)(0x3f00 + (i & 1)); }
    f32x4 acc4[8];
    f32x16 acc16[4];
    for (int i = 0; i < 8; i++) acc4[i] = (f32x4)(0.f);
    for (int i = 0; i < 4; i++) acc16[i] = (f32x16)(0.f);
    float v[8], w[8];
    for (int i = 0; i < 8; i++) { v[i] = threadIdx.x * 1e-3f + i * 0.1f; w[i] = v[i] * 0.5f; }
    const float c = 0.999f;
    const unsigned lds_addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    u32x4 sink = (u32x4)(0);
    ((u32x4*)smem)[threadIdx.x] = (u32x4)(threadIdx.x);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE != 1) {
                if (SHAPE == 0)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[i]) : "v"(a), "v"(b));
                else
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc16[i & 3]) : "v"(a), "v"(b));
            }
            if (MODE != 2) {
#pragma unroll
                for (int r = 0; r < NF; r++) filler<FOP>(v[(i * NF + r) & 7], w[(i * NF + r) & 7], c, lds_addr, sink);
            }
        }
        if (FOP == F_DSR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc4[i][0] + acc4[i][3] + v[i] + w[i];
    for (int i = 0; i < 4; i++) s += acc16[i][0] + acc16[i][15];
    s += (float)sink[0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    // every wave of block 0 records its own start and end: the SIMD-level cost is the span over the waves that share it
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

static float* g_out; static long long* g_cyc;   // g_cyc: (t0, t1) per wave of block 0
template <int SHAPE, int NF, int FOP, int MODE> void run(const char* name) {
    auto kern = k<SHAPE, NF, FOP, MODE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int threads : {256, 512}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        kern<<<256, threads, 96 * 1024>>>(g_out, g_cyc); hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<<<256, threads, 96 * 1024>>>(g_out, g_cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long hh[16]; hipMemcpy(hh, g_cyc, 16 * 8, hipMemcpyDeviceToHost);
        const int nw = threads / 64;
        long long tmin = hh[0], tmax = hh[1], own_max = 0, own_min = 1LL << 62;
        for (int w = 0; w < nw; w++) {
            tmin = hh[2 * w] < tmin ? hh[2 * w] : tmin; tmax = hh[2 * w + 1] > tmax ? hh[2 * w + 1] : tmax;
            const long long d = hh[2 * w + 1] - hh[2 * w];
            own_max = d > own_max ? d : own_max; own_min = d < own_min ? d : own_min;
        }
        const long long h = tmax - tmin;                     // span over the block's waves
        const double slots = (double)ITERS * 8;             // MFMA slots per wave
        const int wps = threads / 256;                       // waves per SIMD
        (void)own_max;
        // per-SIMD cost of one slot (= one MFMA + NF fillers of each resident wave, divided by waves)
        printf("%-34s waves/SIMD=%d  fastest wave cyc/slot=%7.2f  span cyc/slot/SIMD=%7.2f  ns/slot/SIMD=%6.2f  (clock %.2f GHz)\n", name, wps,
               own_min / slots, h / slots / wps, ms * 1e6 / slots / wps, h / (ms * 1e6));
    }
}

#define SWEEP(SHAPE, FOP, FNAME)                                                    \
    run<SHAPE, 1, FOP, 0>("mfma" #SHAPE " + 1 " FNAME);                             \
    run<SHAPE, 2, FOP, 0>("mfma" #SHAPE " + 2 " FNAME);                             \
    run<SHAPE, 3, FOP, 0>("mfma" #SHAPE " + 3 " FNAME);                             \
    run<SHAPE, 4, FOP, 0>("mfma" #SHAPE " + 4 " FNAME);                             \
    run<SHAPE, 6, FOP, 0>("mfma" #SHAPE " + 6 " FNAME);                             \
    run<SHAPE, 8, FOP, 0>("mfma" #SHAPE " + 8 " FNAME);

int main() {
    hipMalloc(&g_out, 256 * 512 * 4); hipMalloc(&g_cyc, 16 * 8);
    printf("# shape 0 = v_mfma_f32_16x16x32_bf16, shape 1 = v_mfma_f32_32x32x16_bf16; a 'slot' = one MFMA (+ its fillers)\n");
    run<0, 0, F_FMA, 2>("mfma0 alone");
    run<1, 0, F_FMA, 2>("mfma1 alone");
    run<0, 4, F_FMA, 1>("4 v_fma alone");
    run<0, 4, F_EXP, 1>("4 v_exp alone");
    run<0, 4, F_CVT, 1>("4 v_cvt_pk_bf16 alone");
    run<0, 4, F_ADD, 1>("4 v_add alone");
    run<0, 4, F_MAX, 1>("4 v_max alone");
    run<0, 4, F_DSR, 1>("4 ds_read_b128 alone");
    run<0, 4, F_NOP, 1>("4 s_nop alone");
    SWEEP(0, F_FMA, "v_fma")
    SWEEP(1, F_FMA, "v_fma")
    SWEEP(0, F_EXP, "v_exp")
    SWEEP(1, F_EXP, "v_exp")
    SWEEP(0, F_CVT, "v_cvt_pk")
    SWEEP(1, F_CVT, "v_cvt_pk")
    SWEEP(0, F_DSR, "ds_read_b128")
    SWEEP(1, F_DSR, "ds_read_b128")
    SWEEP(0, F_NOP, "s_nop")
    return 0;
}
