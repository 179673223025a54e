// What does a workgroup barrier cost a kernel with ONE wave per SIMD?  4 waves, 128 KiB of LDS (one workgroup per CU), a loop of
// 64 independent 16x16x32 bf16 MFMAs per iteration; variants: no barrier, s_barrier every 64 / 128 MFMAs, with / without LDS reads.
//   hipcc --offload-arch=gfx950 -O3 -o barrier_cost barrier_cost.hip && ./barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// bit 0: barrier every 64 MFMAs; bit 1: 16 ds_read_b128 per 64 MFMAs; bit 2: barrier every 128 only (with bit 0);
// bit 3: 8 LDS-DMA pieces (1 KiB each, L2-resident source) per 64 MFMAs, one per 8 MFMAs, with vmcnt(16) before the barrier point;
// bit 4: the same 8 pieces bunched (one per 2 MFMAs in the first 16);  bit 5: split form (s_mov m0 / MFMA / buffer_load) instead of s_mov, s_nop, load
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters, const unsigned char* src) {
    extern __shared__ unsigned char smem[];
    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; i++) acc[i] = (f32x4)(0.f);
    s16x8 fa[8], fb[8];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 8; i++) { fa[i] = (s16x8)((short)(lane + i)); fb[i] = (s16x8)((short)(lane * 3 + i)); }
    const unsigned char* base = smem + (threadIdx.x & 255) * 16;
    u32x4 srd;
    {
        const unsigned long long a = (unsigned long long)src;
        srd[0] = __builtin_amdgcn_readfirstlane((unsigned)a); srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        srd[2] = 1u << 20; srd[3] = 0x00020000u;
    }
    const unsigned voff = lane * 16;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * 8192;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int n = 0; n < 64; n++) {
            if ((MODE & 2) && n % 4 == 0) {
                const s16x8 v = *(const s16x8*)(base + (n / 4) * 4096 + (it & 1) * 65536);
                if (n / 4 < 8) fa[n / 4] = v; else fb[n / 4 - 8] = v;
            }
            const bool d = (MODE & 8) ? ((MODE & 16) ? (n % 2 == 1 && n < 16) : (n % 8 == 1)) : false;
            const int q = (MODE & 16) ? n / 2 : n / 8;
            const unsigned dst = ldsb + (unsigned)q * 1024u + (unsigned)(it & 3) * 32768u, so = (unsigned)(wave * 8 + q) * 1024u + (unsigned)(it & 15) * 32768u;
            if (d && (MODE & 32)) asm volatile("s_mov_b32 m0, %0" ::"s"(dst) : "memory", "m0");
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[n]) : "v"(fa[n >> 3]), "v"(fb[n & 7]));
            if (d) {
                if (MODE & 32) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(so) : "memory", "m0");
                else asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(so), "s"(dst) : "memory", "m0");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE & 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if ((MODE & 1) && (!(MODE & 4) || (it & 1))) asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 64; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, float* out, long long* cyc, int iters, const unsigned char* src) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 256, 131072>>>(out, cyc, iters, src);
    hipEventRecord(e0);
    k<MODE><<<256, 256, 131072>>>(out, cyc, iters, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= 256;
    printf("%-44s %8.1f us  clock64/iter %7.1f  (us*1e3/iter %6.2f ns)  TF/s %6.0f\n", name, ms * 1e3, avg / iters, ms * 1e6 / iters,
           256.0 * 4 * iters * 64 * 16384 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    unsigned char* src; hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    run<0>("64 MFMA", out, cyc, iters, src);
    run<1>("64 MFMA + barrier", out, cyc, iters, src);
    run<2>("64 MFMA + 16 ds_read_b128", out, cyc, iters, src);
    run<3>("64 MFMA + 16 ds_read_b128 + barrier", out, cyc, iters, src);
    run<8>("64 MFMA + 8 DMA spread", out, cyc, iters, src);
    run<8 + 32>("64 MFMA + 8 DMA spread, split form", out, cyc, iters, src);
    run<8 + 16>("64 MFMA + 8 DMA bunched", out, cyc, iters, src);
    run<8 + 16 + 32>("64 MFMA + 8 DMA bunched, split form", out, cyc, iters, src);
    run<8 + 1>("64 MFMA + 8 DMA spread + barrier", out, cyc, iters, src);
    run<8 + 32 + 1>("64 MFMA + 8 DMA spread split + barrier", out, cyc, iters, src);
    run<8 + 32 + 2>("64 MFMA + 8 DMA spread split + 16 reads", out, cyc, iters, src);
    run<8 + 32 + 2 + 1>("64 MFMA + 8 DMA spread split + 16 reads + barrier", out, cyc, iters, src);
    run<8 + 16 + 32 + 2 + 1>("64 MFMA + 8 DMA bunched split + 16 reads + barrier", out, cyc, iters, src);
    return 0;
}
