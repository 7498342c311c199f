// Optional per-launch HIP-event profiler (bench.py's roofline leg).  Disabled by default: zero cost.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace mrisr {

bool prof_enabled();
bool prof_shapes();                          // MRISR_PROF_SHAPES=1: GEMM scopes are named per shape
const char* prof_intern(const std::string& s);  // stable storage for dynamic scope names
struct ProfScope {
    int idx = -1;
    hipStream_t st;
    ProfScope(const char* name, double flops, double bytes, hipStream_t s);
    ~ProfScope();
};

}  // namespace mrisr
