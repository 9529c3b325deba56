// Scene statistics (reference: include/stats.h) with 64-bit counters filled from the device counters
// (rtx_counters) -- the reference's atomic<int> overflows at >= 1024^2 on the 250k mesh.
#pragma once
#include <cstdint>

namespace stats {
inline uint64_t rayTriTests = 0;
inline uint64_t accelStructTests = 0;
inline uint64_t triCopiesCount = 0;
inline uint64_t meshCount = 0;
inline uint64_t acCount = 0;
inline uint64_t raysCasted = 0;
void printStats();
void reset();
}
