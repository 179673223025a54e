// What does an LDS-DMA instruction (global_load_lds_dwordx4, 1 KiB per wave) cost next to MFMA work on gfx950?
// 2 waves per SIMD, hipEvent timing.  Per wave-iteration: NM MFMAs and ND DMA instructions (source rows L2-resident),
// double-buffered through LDS with a block barrier per iteration like the GEMM main loops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define ITERS 2048
template <int NM, int ND, bool BARRIER>
__global__ __launch_bounds__(512) void k(const unsigned char* __restrict__ src, float* out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][8 * 8 * 1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    s16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    f32x4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (f32x4)(0.f);
    const unsigned char* g = src + (size_t)threadIdx.x * 16;      // every block reads the same 512 KiB: L2 hits
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int d = 0; d < ND; d++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)((it * ND + d) & 63) * 8192),
                                             (__attribute__((address_space(3))) void*)(lds[it & 1] + (wave * 8 + d) * 1024 + lane * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NM; i++) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 7], 0, 0, 0);
        if (BARRIER) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    s += lds[0][threadIdx.x];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NM, int ND, bool BARRIER> void run(const char* name, const unsigned char* src, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NM, ND, BARRIER><<<256, 512>>>(src, out); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NM, ND, BARRIER><<<256, 512>>>(src, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %8.3f ms   per SIMD per round of 2 wave-iterations: %7.1f ns\n", name, ms, ms * 1e6 / ITERS);
}
int main() {
    unsigned char* src; float* out;
    hipMalloc(&src, (size_t)64 * 256 * 8192 + (1 << 22)); hipMemset(src, 1, (size_t)64 * 256 * 8192 + (1 << 22));
    hipMalloc(&out, 256 * 512 * 4);
    run<32, 0, true>("32 MFMA, no DMA, barrier", src, out);
    run<32, 4, true>("32 MFMA + 4 DMA (256x256 GEMM ratio), barrier", src, out);
    run<32, 8, true>("32 MFMA + 8 DMA (128x128 GEMM ratio), barrier", src, out);
    run<0, 8, true>("8 DMA only, barrier", src, out);
    run<32, 8, false>("32 MFMA + 8 DMA, no barrier", src, out);
    run<0, 8, false>("8 DMA only, no barrier", src, out);
    return 0;
}
