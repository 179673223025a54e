// The 4-wave GEMM loop skeleton with REAL operand streams: per 64 MFMAs a wave issues 16 ds_read_b128 and 8 LDS-DMA pieces (1 KiB), into a
// ring of four 32-KiB stages, vmcnt(16) + s_barrier per slab.  Waves 0, 1 stream rows of a large matrix A (every workgroup its own 256 rows:
// HBM), waves 2, 3 rows of a small matrix W (L2-resident).  Variants: piece shape 16 rows x 64 B (a 32-deep slab per stage) or
// 8 rows x 128 B (whole cache lines), fetch on / off, barrier on / off.
//   hipcc --offload-arch=gfx950 -O3 -o stream_cost stream_cost.hip && ./stream_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE bit 0: barrier; bit 1: 128-byte-row pieces (else 64-byte rows); bit 2: dead descriptor (no fetch); bit 3: no vmcnt wait
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int slabs, const unsigned char* A, const unsigned char* W, int ldb) {
    extern __shared__ unsigned char smem[];
    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; i++) acc[i] = (f32x4)(0.f);
    s16x8 fa[8], fb[8];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 8; i++) { fa[i] = (s16x8)((short)(lane + i)); fb[i] = (s16x8)((short)(lane * 3 + i)); }
    const unsigned char* base = smem + (threadIdx.x & 255) * 16;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool isw = wave >= 2;
    u32x4 srd;
    {
        const unsigned long long a = (unsigned long long)(isw ? W : A + (size_t)blockIdx.x * 256 * ldb);
        srd[0] = __builtin_amdgcn_readfirstlane((unsigned)a); srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        srd[2] = (MODE & 4) ? 0u : (unsigned)(256 * ldb); srd[3] = 0x00020000u;
    }
    // 64-byte rows: piece = 16 rows x 64 B, 8 pieces per wave and slab (128 rows), slab s at byte 64 s of the row
    // 128-byte rows: piece = 8 rows x 128 B, 16 pieces per wave and PAIR of slabs, issued 8 per slab
    const unsigned voff = (MODE & 2) ? (unsigned)(((wave & 1) * 128 + (lane >> 3)) * ldb + (lane & 7) * 16)
                                     : (unsigned)(((wave & 1) * 128 + (lane >> 2)) * ldb + (lane & 3) * 16);
    const unsigned pstride = (MODE & 2) ? 8u * ldb : 16u * ldb;
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * 8192;
    const long long t0 = clock64();
    for (int s = 0; s < slabs; s++) {
        if (!(MODE & 8)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if (MODE & 1) asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int n = 0; n < 64; n++) {
            if (n % 3 == 0 && n / 3 < 16) {
                const s16x8 v = *(const s16x8*)(base + (n / 3) * 4096 + (s & 1) * 65536);
                if (n / 3 < 8) fa[n / 3] = v; else fb[n / 3 - 8] = v;
            }
            const bool d = n % 8 == 1;
            const int q = n / 8;
            const unsigned dst = ldsb + (unsigned)q * 1024u + (unsigned)(s & 3) * 32768u;
            const unsigned so = (MODE & 2) ? (unsigned)(s >> 1) * 128u + (unsigned)((s & 1) * 8 + q) * pstride : (unsigned)s * 64u + (unsigned)q * pstride;
            if (d) asm volatile("s_mov_b32 m0, %0" ::"s"(dst) : "memory", "m0");
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[n]) : "v"(fa[n >> 3]), "v"(fb[n & 7]));
            if (d) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(so) : "memory", "m0");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = clock64();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 64; i++) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, float* out, long long* cyc, int slabs, const unsigned char* A, const unsigned char* W, int ldb, int blocks) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256, 131072>>>(out, cyc, slabs, A, W, ldb);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256, 131072>>>(out, cyc, slabs, A, W, ldb);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= blocks;
    printf("%-52s %8.1f us  clock64/slab %7.1f  TF/s %6.0f\n", name, ms * 1e3, avg / slabs, (double)blocks * 4 * slabs * 64 * 16384 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; long long* cyc;
    const int blocks = 2048, ldb = 6144, slabs = 96;          // A: 2048 x 256 rows x 6144 B = 3.2 GB; each block streams 256 rows x 96 x 64 B
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    unsigned char *A, *W;
    hipMalloc(&A, (size_t)blocks * 256 * ldb); hipMemset(A, 0, (size_t)blocks * 256 * ldb);
    hipMalloc(&W, (size_t)256 * ldb); hipMemset(W, 0, (size_t)256 * ldb);
    run<4 + 1>("no fetch, barrier", out, cyc, slabs, A, W, ldb, blocks);
    run<0 + 1>("64-B rows, barrier", out, cyc, slabs, A, W, ldb, blocks);
    run<0 + 0>("64-B rows, no barrier", out, cyc, slabs, A, W, ldb, blocks);
    run<2 + 1>("128-B rows, barrier", out, cyc, slabs, A, W, ldb, blocks);
    run<2 + 0>("128-B rows, no barrier", out, cyc, slabs, A, W, ldb, blocks);
    run<2 + 1 + 8>("128-B rows, barrier, no vmcnt wait", out, cyc, slabs, A, W, ldb, blocks);
    return 0;
}
