// Internal: every kernel translation unit sees the public C ABI it implements.
#pragma once
#include "../../include/osu_dreamer_hip.h"
