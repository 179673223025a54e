// Probe: throughput of fp32 atomicAdd tiles into an L2-resident accumulator, mimicking a fused
// attention backward that accumulates dQ (64x64 fp32 per (q-tile, key-block) pair) with atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
// grid: (key blocks, bh).  each block walks all q tiles, adding a 64x64 tile (row stride ld) per q tile.
__global__ void k(float* dq, int L, int ld, int nq, int mode) {
    const int bh = blockIdx.y;
    float* base = dq + (size_t)bh * L * ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int qt = 0; qt < nq; qt++) {
        // 256 threads: each wave covers 16 rows; lane -> (row = lane>>2 .. ), 16 floats per lane
        float* tile = base + (size_t)qt * 64 * ld;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            int idx = (i * 256 + threadIdx.x);        // 0..4095
            int row = idx >> 6, col = idx & 63;
            if (mode == 0) atomicAdd(tile + row * ld + col, 1.0f);
            else unsafeAtomicAdd(tile + row * ld + col, 1.0f);
        }
    }
}
int main() {
    const int L = 8192, ld = 64, BH = 512, nq = L / 64, nkb = L / 128;
    float* dq; hipMalloc(&dq, (size_t)BH * L * ld * 4); hipMemset(dq, 0, (size_t)BH * L * ld * 4);
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<dim3(nkb, 64), 256>>>(dq, L, ld, nq, mode); hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<dim3(nkb, BH), 256>>>(dq, L, ld, nq, mode);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double n = (double)nkb * BH * nq * 4096;
        printf("mode %d: %.3f ms for %.3g atomics -> %.1f G atomics/s\n", mode, ms, n, n / ms / 1e6);
    }
    return 0;
}
