// Deterministic accumulation: the host side of od_common.h's OdDetTable (see there).  The caller owns every byte: the fp32 destinations, their
// 64-bit shadows (zero when registered) and the 512 bytes of device memory the table is copied to.
#include "od_common.h"
#include "od_api_internal.h"

namespace {
OdDetTable g_host_table = {0, 0, {}};
const OdDetTable* g_active = nullptr;

__global__ __launch_bounds__(256) void det_flush_kernel(float* __restrict__ dst, long long* __restrict__ shadow, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long long s = shadow[i];
        if (s != 0) { dst[i] += od_unfix(s); shadow[i] = 0; }
    }
}
}  // namespace

const OdDetTable* od_det_active() { return g_active; }

extern "C" int od_det_clear(void) {
    g_host_table.n = 0;
    g_active = nullptr;
    return 0;
}

extern "C" int od_det_register(const float* base, long count, void* shadow_i64) {
    if (!base || count <= 0 || !shadow_i64) return OD_ERR_ARG;
    if (g_host_table.n >= (int)(sizeof(g_host_table.r) / sizeof(g_host_table.r[0]))) return OD_ERR_UNSUPPORTED;
    g_host_table.r[g_host_table.n++] = OdDetRange{base, (long long)count, (long long*)shadow_i64};
    return 0;
}

extern "C" int od_det_enable(void* table_dev, void* stream) {
    if (!table_dev) { g_active = nullptr; return 0; }
#if defined(OD_EMU)
    (void)stream;
    *(OdDetTable*)table_dev = g_host_table;
#else
    hipError_t e = hipMemcpyAsync(table_dev, &g_host_table, sizeof(g_host_table), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);      // the host copy may change right after
    if (e != hipSuccess) return -(int)e - 1000;
#endif
    g_active = (const OdDetTable*)table_dev;
    return 0;
}

extern "C" int od_det_table_bytes(void) { return (int)sizeof(OdDetTable); }

extern "C" int od_det_flush(float* dst, long count, void* stream) {
    if (!dst || count <= 0) return OD_ERR_ARG;
    for (int i = 0; i < g_host_table.n; i++) {
        const OdDetRange& r = g_host_table.r[i];
        const long long off = dst - r.base;
        if (off >= 0 && off + count <= r.count) {
            int blocks = (int)((count + 255) / 256); if (blocks > 4096) blocks = 4096;
            OD_LAUNCH(det_flush_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dst, r.shadow + off, count);
            OD_CHECK_LAUNCH();
            return 0;
        }
    }
    return OD_ERR_ARG;      // not (wholly) inside a registered range
}
