// Internal: every kernel translation unit sees the public C ABI it implements.
#pragma once
#include "../../include/osu_dreamer_hip.h"

// integer tunable from the environment (A/B runs: one process per setting); read once by the caller (`static const int`)
#include <stdlib.h>
inline int od_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// Compute units of the CURRENT device, for the persistent kernels' grids (one workgroup per CU); cached per device id.  The emulator has none:
// it runs the persistent grid at 16 workgroups (two per "XCD" of the tile order).
#if defined(OD_EMU)
inline int od_num_cus() { return 16; }
#else
inline int od_num_cus() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cache[dev]) {
        int n = 0;
        cache[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cache[dev];
}
#endif

// The active table of od_det_* (device memory; NULL = plain float atomics): every launcher of a kernel that accumulates with atomics
// hands it to the kernel.  One process drives one GPU: process-global.
struct OdDetTable;
const OdDetTable* od_det_active();
