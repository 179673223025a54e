// Round-trip time of one LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB per wave) from issue to s_waitcnt vmcnt(0), in shader
// clocks (s_memtime) — from an L2-resident array and from an HBM stream, unloaded (1 wave per CU, one piece in flight) and loaded
// (8 waves per CU, each keeping U pieces in flight: the timed piece is issued behind them).  Feeds the "32 KiB outstanding per CU /
// latency" model of the fetch-bound GEMM (profiles/r02l_gemm_fetch_bound.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(512) void k(const unsigned char* __restrict__ src, size_t per_block, size_t span, int iters, long long* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned char* base = src + (size_t)blockIdx.x * per_block;
    u32x4 srd;
    {
        const unsigned long long a = (unsigned long long)base;
        srd[0] = __builtin_amdgcn_readfirstlane((unsigned)a); srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        srd[2] = 0xffffffffu; srd[3] = 0x00020000u;
    }
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * (U + 1) * 1024);
    size_t p = wave;
    const size_t npieces = span / 1024;
    long long total = 0;
    const unsigned zero = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < U; u++) {                       // background pieces
            const unsigned vo = (unsigned)((p % npieces) * 1024 + lane * 16), m0v = lds_wave + u * 1024;
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(srd), "s"(zero), "s"(m0v) : "memory", "m0");
            p += nw;
        }
        const unsigned vo = (unsigned)((p % npieces) * 1024 + lane * 16), m0v = lds_wave + U * 1024;
        p += nw;
        const long long t0 = __builtin_readcyclecounter();
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(srd), "s"(zero), "s"(m0v) : "memory", "m0");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t1 = __builtin_readcyclecounter();
        total += t1 - t0;
    }
    if (lane == 0) out[blockIdx.x * 8 + wave] = total / iters;
}

template <int U> void run(const char* name, const unsigned char* src, size_t per_block, size_t span, int threads, long long* out) {
    auto kern = k<U>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 2000;
    hipMemset(out, 0, 256 * 8 * 8);
    kern<<<256, threads, 8 * (U + 1) * 1024>>>(src, per_block, span, iters, out); hipDeviceSynchronize();
    kern<<<256, threads, 8 * (U + 1) * 1024>>>(src, per_block, span, iters, out); hipDeviceSynchronize();
    long long h[256 * 8]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; long long mx = 0; int n = 0;
    for (int b = 0; b < 256; b++) for (int w = 0; w < threads / 64; w++) { s += h[b * 8 + w]; mx = h[b * 8 + w] > mx ? h[b * 8 + w] : mx; n++; }
    printf("%-46s waves/CU=%d pieces in flight/wave=%2d   mean %7.0f clk   max %7lld clk\n", name, threads / 64, U + 1, s / n, mx);
}

int main() {
    const size_t total = (size_t)4 << 30;
    unsigned char* src; long long* out;
    hipMalloc(&src, total); hipMemset(src, 1, total); hipMalloc(&out, 256 * 8 * 8);
    printf("# shader clocks (__builtin_readcyclecounter)\n");
    run<0>("L2-resident (every CU re-reads 2 MiB)", src, 0, (size_t)2 << 20, 64, out);
    run<0>("L2-resident", src, 0, (size_t)2 << 20, 512, out);
    run<3>("L2-resident", src, 0, (size_t)2 << 20, 512, out);
    run<7>("L2-resident", src, 0, (size_t)2 << 20, 512, out);
    run<0>("HBM stream (16 MiB per CU)", src, (size_t)16 << 20, (size_t)16 << 20, 64, out);
    run<0>("HBM stream", src, (size_t)16 << 20, (size_t)16 << 20, 512, out);
    run<3>("HBM stream", src, (size_t)16 << 20, (size_t)16 << 20, 512, out);
    run<7>("HBM stream", src, (size_t)16 << 20, (size_t)16 << 20, 512, out);
    return 0;
}
