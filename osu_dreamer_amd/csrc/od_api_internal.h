// Internal: every kernel translation unit sees the public C ABI it implements.
#pragma once
#include "../../include/osu_dreamer_hip.h"

// integer tunable from the environment (A/B runs: one process per setting); read once by the caller (`static const int`)
#include <stdlib.h>
inline int od_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
