// Per-launch HIP events on the launch stream, aggregated per kernel class; reported as JSON through the C ABI.
#include "prof.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/mrisr.h"

namespace mrisr {

struct ProfRec {
    const char* name;
    double flops, bytes;
    hipEvent_t e0, e1;
};
static bool g_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

bool prof_enabled() { return g_on; }
bool prof_shapes() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("MRISR_PROF_SHAPES");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}
const char* prof_intern(const std::string& s) {
    static std::map<std::string, int> pool;
    return pool.emplace(s, 0).first->first.c_str();
}

static hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(const char* name, double flops, double bytes, hipStream_t s) : st(s) {
    if (!g_on) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;  // never inside a capture
    ProfRec r{name, flops, bytes, get_event(), get_event()};
    (void)hipEventRecord(r.e0, s);
    idx = (int)g_recs.size();
    g_recs.push_back(r);
}
ProfScope::~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(g_recs[idx].e1, st);
}

}  // namespace mrisr

namespace mrisr { int mrisr_prof_enable_internal(int on) { g_on = on != 0; return 0; } }
using namespace mrisr;

extern "C" {

int mrisr_prof_enable(int on) {
    g_on = on != 0;
    return 0;
}
int mrisr_prof_reset(void) {
    (void)hipDeviceSynchronize();
    for (auto& r : g_recs) {
        g_pool.push_back(r.e0);
        g_pool.push_back(r.e1);
    }
    g_recs.clear();
    return 0;
}
// JSON: {"<class>": {"launches": n, "ms": total, "flops": total, "bytes": total}, ...}; returns bytes written
int mrisr_prof_report(char* buf, int cap) {
    (void)hipDeviceSynchronize();
    struct Agg { long long n = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        Agg& a = agg[r.name];
        a.n += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    std::string s = "{";
    bool first = true;
    for (auto& kv : agg) {
        char line[320];
        snprintf(line, sizeof(line), "%s\"%s\": {\"launches\": %lld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
                 first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
        s += line;
        first = false;
    }
    s += "}";
    if ((int)s.size() + 1 > cap) return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

}  // extern "C"
