// Error string, version, and the launch-bracketing instrumentation used by bench.py's roofline leg.
#include <vector>

#include "wu_common.h"

thread_local char g_wu_err[256] = {0};

extern "C" const char* wu_last_error(void) { return g_wu_err; }
extern "C" int wu_version(void) { return 2; }

// Compute units of the current device, read once (the persistent conv / weight-gradient grids launch one workgroup per CU).
// Without a usable device (the CPU-only build container: workspace sizing in the ABI tests) the MI355X figure is assumed.
int wu_num_cus() {
    if (g_wu_opt[WU_OPT_GRID_CUS] > 0) return g_wu_opt[WU_OPT_GRID_CUS];   // experiment: persistent grids sized for a CU-masked stream
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else { (void)hipGetLastError(); cus = 256; }
    }
    return cus;
}
extern "C" int wu_cu_count(void) { return wu_num_cus(); }

// Experiment support (scratch/ab_cumask.py): a stream confined to the compute units whose bits are set in `mask` (`words` x 32 bits).
extern "C" int wu_stream_create_cu_mask(const unsigned* mask, int words, void** stream_out) {
    WU_REQUIRE(mask && words > 0 && stream_out, "stream_create_cu_mask: bad args");
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (e != hipSuccess) WU_FAIL((int)e, "hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    *stream_out = (void*)s;
    return 0;
}
extern "C" int wu_stream_destroy(void* stream) {
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? 0 : -1;
}

// Ordering events between two streams of ONE device: no timing, no system-scope fence.  torch's wait_stream() records a default-flag
// event: every record is then a system-scope release (write-back towards the host) on the recording stream -- 14 of them per training
// step on the critical path of the fused backward (one per weight-gradient hand-off).  Kernels on another stream of the same device
// need device-scope visibility only: a timing-less event with a device-scope release.
extern "C" int wu_event_create(void** event_out) {
    WU_REQUIRE(event_out, "event_create: bad args");
    hipEvent_t e = nullptr;
    // hipEventReleaseToDevice: the DOCUMENTED device-scope release at the record (round 4, advisor).  Round 3 used
    // hipEventDisableSystemFence, which HIP documents for timing events only -- what it leaves of the release is an implementation detail.
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToDevice);
    if (rc != hipSuccess) WU_FAIL((int)rc, "hipEventCreateWithFlags: %s", hipGetErrorString(rc));
    *event_out = (void*)e;
    return 0;
}
extern "C" int wu_event_destroy(void* event) { return hipEventDestroy((hipEvent_t)event) == hipSuccess ? 0 : -1; }
// `waiter` will not run work enqueued after this call before everything enqueued on `producer` so far has finished
extern "C" int wu_stream_order_after(void* waiter, void* producer, void* event) {
    WU_REQUIRE(event, "stream_order_after: bad args");
    hipError_t rc = hipEventRecord((hipEvent_t)event, (hipStream_t)producer);
    if (rc == hipSuccess) rc = hipStreamWaitEvent((hipStream_t)waiter, (hipEvent_t)event, 0);
    if (rc != hipSuccess) WU_FAIL((int)rc, "stream_order_after: %s", hipGetErrorString(rc));
    return 0;
}

// tuning switches (A/B benchmarking of kernel variants inside one process; defaults are the production choices)
int g_wu_opt[16] = {/*CONV_V2*/ 1, /*CONV_PERSISTENT*/ 1, /*WGRAD_V2*/ 1, /*CONV_SMALL*/ 1, /*WGRAD_DMA_INTERLEAVE*/ 1, /*C3_ROWS*/ 0, /*CONV_PRIO*/ 1, /*CONV_STRIDED*/ 1, /*ADAIN_BWD_MARCH*/ 1, /*ADAIN_FWD_MARCH*/ 1, /*GRID_CUS*/ 0, /*CONV_W_RESIDENT*/ 1, /*PW_TILE*/ 0, /*IMG3_TILED*/ 1, /*S2_DGRAD_PARITY*/ 1, /*PW3: D = 2, eight waves, from 128 tiles*/ 2 + 8 + (128 << 5)};
void* g_wu_dbg_ptr = nullptr;   // diagnostic stamp buffer (256 workgroups x 8 waves x 8 u64), NULL in production
extern "C" int wu_set_debug_buffer(void* p) { g_wu_dbg_ptr = p; return 0; }
extern "C" int wu_set_option(int key, int value) {
    WU_REQUIRE(key >= 0 && key < 16, "set_option: bad key %d", key);
    g_wu_opt[key] = value;
    return 0;
}

namespace {
constexpr int kMaxFam = 8;
struct Rec { int family; double flops, bytes; };
struct ProfState {
    unsigned mask = 0;
    int used = 0;
    bool open = false;     // a pre() without its post()
    std::vector<hipEvent_t> ev;  // 2 per launch
    std::vector<Rec> rec;
};
ProfState g_prof;
inline bool on(int family) { return g_prof.mask & (1u << family); }
}  // namespace

void wu_prof_pre(int family, hipStream_t s) {
    if (!g_prof.mask || family <= 0 || family >= kMaxFam || !on(family)) return;
    if ((size_t)(2 * g_prof.used + 2) > g_prof.ev.size()) return;
    (void)hipEventRecord(g_prof.ev[2 * g_prof.used], s);
    g_prof.open = true;
}

void wu_prof_post(int family, hipStream_t s, double flops, double bytes) {
    if (!g_prof.open || !on(family)) return;
    (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s);
    g_prof.rec[g_prof.used] = Rec{family, flops, bytes};
    g_prof.used++;
    g_prof.open = false;
}

extern "C" int wu_prof_begin(unsigned family_mask, int max_launches) {
    WU_REQUIRE(max_launches > 0 && max_launches <= (1 << 20), "prof_begin: bad max_launches");
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.assign(2 * (size_t)max_launches, nullptr);
    g_prof.rec.assign((size_t)max_launches, Rec{0, 0, 0});
    for (auto& e : g_prof.ev)
        // timing events: hipEventDisableSystemFence is HIP's flag for events "only being used to measure timing" -- a default event makes
        // its record a system-scope fence (cache write-back and invalidation, which also slow the work BEHIND it).  Same-box A/B of the
        // default bench line against --no-roofline: 0.18 ms of overhead per 8.3 ms step with default events, 0.15 with
        // hipEventReleaseToDevice, 0.07 with this flag (profiles/r03_roofline_overhead.txt)
        if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) WU_FAIL(-2, "prof_begin: hipEventCreate failed");
    g_prof.mask = family_mask;
    g_prof.used = 0;
    g_prof.open = false;
    return 0;
}

// Per-family totals of the bracketed launches so far (synchronises on their events).
extern "C" int wu_prof_query(int family, int* launches, double* total_ms, double* total_flops, double* total_bytes) {
    double ms = 0, fl = 0, by = 0;
    int n = 0;
    for (int i = 0; i < g_prof.used; ++i) {
        if (g_prof.rec[i].family != family) continue;
        (void)hipEventSynchronize(g_prof.ev[2 * i + 1]);
        float t = 0;
        (void)hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
        ms += t; fl += g_prof.rec[i].flops; by += g_prof.rec[i].bytes; ++n;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (total_bytes) *total_bytes = by;
    return 0;
}

extern "C" int wu_prof_end(void) {
    g_prof.mask = 0;
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
    g_prof.rec.clear();
    g_prof.used = 0;
    return 0;
}
