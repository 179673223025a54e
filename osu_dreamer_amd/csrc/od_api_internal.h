// Internal: every kernel translation unit sees the public C ABI it implements.
#pragma once
#include "../../include/osu_dreamer_hip.h"

// compute units of the current device (persistent kernels size their grids with it); the CPU test build pretends 16
#if defined(OD_EMU)
inline int od_num_cus() { return 16; }
#else
inline int od_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        n = v >= 8 ? v : 8;
    }
    return n;
}
#endif
