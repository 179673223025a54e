// Build identity of libosudreamer_hip.so (host-only translation unit, recompiled by every build: csrc/build.sh passes the hash of the
// kernel sources it is linking, csrc/source_sha.sh).
#include "../../include/osu_dreamer_hip.h"
#ifndef OD_SRC_SHA
#define OD_SRC_SHA "unknown"
#endif
extern "C" const char* od_build_source_sha(void) { return OD_SRC_SHA; }
