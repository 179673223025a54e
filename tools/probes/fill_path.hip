// Which path fills LDS fastest?  One workgroup of 8 waves per CU streams 16-byte-per-lane pieces (64 lanes x 16 B = 1 KiB) from
// an L2-resident array (2 MiB, re-read by every CU: the W operand of a GEMM) or from a per-CU HBM stream (the A operand):
//   mode 0: LDS-DMA               buffer_load_dwordx4 ... lds          (what the GEMM / attention kernels use)
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128               (register staging)
//   mode 2: global_load_dwordx4 -> VGPR only                            (the load path alone)
//   mode 3: even waves as mode 0, odd waves as mode 1                   (do the two paths add up?)
// U pieces are in flight per wave (U = 4, 8, 16).  Piece shape: 16 rows x 64 B at a row stride of LD bytes (GEMM k-slab) or 1 KiB
// contiguous.  Output: GB/s per CU and for the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int U>
__global__ __launch_bounds__(512) void k(const unsigned char* __restrict__ src, size_t per_block, size_t span, int row_stride, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* base = src + (size_t)blockIdx.x * per_block;
    // a wave's piece p: row_stride == 0 -> contiguous 1 KiB at p * 1024; else 16 rows x 64 B: row (p*16 + lane/4), chunk lane%4
    const size_t lane_off = row_stride ? (size_t)(lane >> 2) * row_stride + (lane & 3) * 16 : (size_t)lane * 16;
    const size_t piece_step = row_stride ? (size_t)16 * row_stride : 1024;
    u32x4 acc = (u32x4)(0);
    u32x4 srd;
    {
        const unsigned long long a = (unsigned long long)base;
        srd[0] = __builtin_amdgcn_readfirstlane((unsigned)a); srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        srd[2] = 0xffffffffu; srd[3] = 0x00020000u;
    }
    const unsigned lds_wave = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * U * 1024;
    size_t p = wave;                                   // pieces interleaved over the 8 waves
    const size_t npieces = span / piece_step;
    for (int it = 0; it < iters; it++) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t off = (p % npieces) * piece_step + lane_off;
            if (MODE == 0 || (MODE == 3 && !(wave & 1))) {
                const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_wave + u * 1024), vo = (unsigned)off;
                const unsigned zero = 0; asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(srd), "s"(zero), "s"(m0v) : "memory", "m0");
            } else {
                v[u] = *(const u32x4*)(base + off);
            }
            p += 8;
        }
        if (MODE == 0 || (MODE == 3 && !(wave & 1))) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int u = 0; u < U; u++) {                      // asm: a plain store here is dead (overwritten next iteration) and was removed
                const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * U * 1024 + u * 1024 + lane * 16;
                asm volatile("ds_write_b128 %0, %1" ::"v"(la), "v"(v[u]) : "memory");
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) acc ^= v[u];
        }
    }
    __syncthreads();
    if (MODE != 2) acc = *(const u32x4*)(smem + threadIdx.x * 16);
    out[blockIdx.x * 512 + threadIdx.x] = (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
}

template <int MODE, int U> void run(const char* name, const unsigned char* src, size_t per_block, size_t span, int row_stride, float* out) {
    auto kern = k<MODE, U>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 2048 / U * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, 512, 8 * U * 1024>>>(src, per_block, span, row_stride, iters, out); hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<256, 512, 8 * U * 1024>>>(src, per_block, span, row_stride, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * 8 * iters * U * 1024;
    printf("%-52s U=%2d  %7.1f GB/s per CU   %6.2f TB/s chip\n", name, U, bytes / 256 / ms / 1e6, bytes / ms / 1e9);
}

#define ALL(MODE, NAME, ...) run<MODE, 4>(NAME, __VA_ARGS__); run<MODE, 8>(NAME, __VA_ARGS__); run<MODE, 16>(NAME, __VA_ARGS__);

int main() {
    const size_t total = (size_t)4 << 30;
    unsigned char* src; float* out;
    hipMalloc(&src, total); hipMemset(src, 1, total); hipMalloc(&out, 256 * 512 * 4);
    printf("# L2-resident: every CU re-reads the same 2 MiB (per_block = 0)\n");
    ALL(0, "L2  contiguous 1 KiB     LDS-DMA", src, 0, (size_t)2 << 20, 0, out)
    ALL(1, "L2  contiguous 1 KiB     load + ds_write_b128", src, 0, (size_t)2 << 20, 0, out)
    ALL(2, "L2  contiguous 1 KiB     load only", src, 0, (size_t)2 << 20, 0, out)
    ALL(3, "L2  contiguous 1 KiB     half DMA, half load + ds_write", src, 0, (size_t)2 << 20, 0, out)
    ALL(0, "L2  16 rows x 64 B / 1 KiB stride   LDS-DMA", src, 0, (size_t)2 << 20, 1024, out)
    ALL(1, "L2  16 rows x 64 B / 1 KiB stride   load + ds_write", src, 0, (size_t)2 << 20, 1024, out)
    ALL(2, "L2  16 rows x 64 B / 1 KiB stride   load only", src, 0, (size_t)2 << 20, 1024, out)
    ALL(0, "L2  16 rows x 64 B / 6 KiB stride   LDS-DMA", src, 0, (size_t)3 << 20, 6144, out)
    ALL(2, "L2  16 rows x 64 B / 6 KiB stride   load only", src, 0, (size_t)3 << 20, 6144, out)
    ALL(1, "L2  16 rows x 64 B / 6 KiB stride   load + ds_write", src, 0, (size_t)3 << 20, 6144, out)
    ALL(3, "L2  16 rows x 64 B / 6 KiB stride   half DMA, half load + ds_write", src, 0, (size_t)3 << 20, 6144, out)
    printf("# HBM stream: each CU walks its own 16 MiB\n");
    ALL(0, "HBM contiguous 1 KiB     LDS-DMA", src, (size_t)16 << 20, (size_t)16 << 20, 0, out)
    ALL(1, "HBM contiguous 1 KiB     load + ds_write_b128", src, (size_t)16 << 20, (size_t)16 << 20, 0, out)
    ALL(2, "HBM contiguous 1 KiB     load only", src, (size_t)16 << 20, (size_t)16 << 20, 0, out)
    ALL(0, "HBM 16 rows x 64 B / 6 KiB stride   LDS-DMA", src, (size_t)16 << 20, (size_t)12 << 20, 6144, out)
    ALL(2, "HBM 16 rows x 64 B / 6 KiB stride   load only", src, (size_t)16 << 20, (size_t)12 << 20, 6144, out)
    return 0;
}
