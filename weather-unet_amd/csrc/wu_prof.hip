// Error string, version, and the launch-bracketing instrumentation used by bench.py's roofline leg.
#include <vector>

#include "wu_common.h"

thread_local char g_wu_err[256] = {0};

extern "C" const char* wu_last_error(void) { return g_wu_err; }
extern "C" int wu_version(void) { return 1; }

namespace {
struct ProfState {
    int family = 0;
    int used = 0;
    bool open = false;     // a pre() without its post()
    std::vector<hipEvent_t> ev;  // 2 per launch
    double flops = 0, bytes = 0;
};
ProfState g_prof;
}  // namespace

void wu_prof_pre(int family, hipStream_t s) {
    if (!g_prof.family || !(g_prof.family == WU_FAM_ALL || g_prof.family == family)) return;
    if ((size_t)(2 * g_prof.used + 2) > g_prof.ev.size()) return;
    hipEventRecord(g_prof.ev[2 * g_prof.used], s);
    g_prof.open = true;
}

void wu_prof_post(int family, hipStream_t s, double flops, double bytes) {
    if (!g_prof.open || !(g_prof.family == WU_FAM_ALL || g_prof.family == family)) return;
    hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s);
    g_prof.used++;
    g_prof.flops += flops;
    g_prof.bytes += bytes;
    g_prof.open = false;
}

extern "C" int wu_prof_begin(int family, int max_launches) {
    WU_REQUIRE(max_launches > 0 && max_launches <= (1 << 20), "prof_begin: bad max_launches");
    for (hipEvent_t e : g_prof.ev) hipEventDestroy(e);
    g_prof.ev.assign(2 * (size_t)max_launches, nullptr);
    for (auto& e : g_prof.ev)
        if (hipEventCreate(&e) != hipSuccess) WU_FAIL(-2, "prof_begin: hipEventCreate failed");
    g_prof.family = family;
    g_prof.used = 0;
    g_prof.flops = g_prof.bytes = 0;
    g_prof.open = false;
    return 0;
}

extern "C" int wu_prof_end(int* launches, double* total_ms, double* total_flops, double* total_bytes) {
    double ms = 0;
    for (int i = 0; i < g_prof.used; ++i) {
        hipEventSynchronize(g_prof.ev[2 * i + 1]);
        float t = 0;
        hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
        ms += t;
    }
    if (launches) *launches = g_prof.used;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = g_prof.flops;
    if (total_bytes) *total_bytes = g_prof.bytes;
    g_prof.family = 0;
    for (hipEvent_t e : g_prof.ev) hipEventDestroy(e);
    g_prof.ev.clear();
    return 0;
}
