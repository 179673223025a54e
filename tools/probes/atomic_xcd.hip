// Probe: fp32 atomic-add rate for a dQ-style accumulation when ALL writers of an address run on ONE XCD
// (block id -> XCD = id & 7, as the attention kernels already place a (batch, head)), by atomic scope:
//   agent scope     -> global_atomic_add_f32 ... sc1   (coherent across XCDs: executed memory-side)
//   workgroup scope -> global_atomic_add_f32           (executed in the XCD's own L2)
// and for writers spread over all XCDs (agent scope only is correct there).  Checks the sums and reports the
// XCC_ID each block really ran on, so "same XCD" is verified, not assumed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SCOPE, bool SAME_XCD, int PK>
__global__ void k(float* dq, int L, int ld, int nq, int nkb, int BH, unsigned* xcc_bad) {
    int bh, kb;
    if (SAME_XCD) {
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        bh = (slot / nkb) * 8 + xcd; kb = slot % nkb;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (threadIdx.x == 0 && (xcc & 0xf) != (unsigned)xcd) atomicAdd(xcc_bad, 1u);
    } else {
        bh = blockIdx.x / nkb; kb = blockIdx.x % nkb;
    }
    if (bh >= BH) return;
    (void)kb;
    float* base = dq + (size_t)bh * L * ld;
    for (int qt = 0; qt < nq; qt++) {
        float* tile = base + (size_t)qt * 64 * ld;
        if (PK == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int idx = i * 256 + threadIdx.x, row = idx >> 6, col = idx & 63;
                __hip_atomic_fetch_add(tile + row * ld + col, 1.0f, __ATOMIC_RELAXED, SCOPE);
            }
        } else {       // packed bf16 pairs: half the instructions
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int idx = i * 256 + threadIdx.x, row = idx >> 5, col = (idx & 31) * 2;
                typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                bf2 v = {(__bf16)1.0f, (__bf16)1.0f};
                bf2* p = (bf2*)((unsigned short*)tile + row * ld + col);
                asm volatile("global_atomic_pk_add_bf16 %0, %1, off" :: "v"(p), "v"(v) : "memory");
            }
        }
    }
}

template <int SCOPE, bool SAME_XCD, int PK> void run(const char* name) {
    const int L = 8192, ld = 64, BH = 512, nq = L / 64, nkb = L / 128;
    float* dq; unsigned* bad;
    hipMalloc(&dq, (size_t)BH * L * ld * 4); hipMalloc(&bad, 4);
    hipMemset(dq, 0, (size_t)BH * L * ld * 4); hipMemset(bad, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<SCOPE, SAME_XCD, PK><<<nkb * BH, 256>>>(dq, L, ld, nq, nkb, BH, bad);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h(64 * 64); unsigned hb;
    hipMemcpy(h.data(), dq + (size_t)37 * L * ld + 64 * ld * 5, 64 * 64 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    int wrong = 0;
    if (PK == 0) for (float v : h) wrong += v != (float)nkb;
    const double n = (double)nkb * BH * nq * 4096;
    printf("%-52s %8.3f ms  %7.1f G elements/s  wrong(sample)=%d  blocks off their XCD=%u\n", name, ms, n / ms / 1e6, wrong, hb);
    hipFree(dq); hipFree(bad);
}
int main() {
    run<__HIP_MEMORY_SCOPE_AGENT, false, 0>("f32, agent scope, writers on all XCDs");
    run<__HIP_MEMORY_SCOPE_AGENT, true, 0>("f32, agent scope, writers on one XCD");
    run<__HIP_MEMORY_SCOPE_WORKGROUP, true, 0>("f32, workgroup scope (L2-local), writers on one XCD");
    run<__HIP_MEMORY_SCOPE_WORKGROUP, false, 0>("f32, workgroup scope, writers on all XCDs (WRONG by design)");
    run<__HIP_MEMORY_SCOPE_AGENT, true, 1>("pk bf16, writers on one XCD");
    return 0;
}
