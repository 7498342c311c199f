// Optional per-launch HIP-event profiler (bench.py's roofline leg).  Disabled by default: zero cost.
#pragma once
#include <hip/hip_runtime.h>

namespace mrisr {

bool prof_enabled();
struct ProfScope {
    int idx = -1;
    hipStream_t st;
    ProfScope(const char* name, double flops, double bytes, hipStream_t s);
    ~ProfScope();
};

}  // namespace mrisr
