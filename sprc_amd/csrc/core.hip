// core.hip -- version, error reporting, CU-partition streams, the per-launch profiler
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

#include "common.hpp"

namespace sprc {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- CU-partition streams (sprc.h: sprc_stream_create_partition) -------------------------------------
static std::mutex g_part_mu;
static std::vector<std::pair<hipStream_t, int>> g_parts;     // a handful of entries: linear search under a mutex
static int device_cus() {
    constexpr int MAXD = 64;
    static int n[MAXD] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= MAXD) dev = 0;
    if (n[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n[dev] = prop.multiProcessorCount;
        if (n[dev] <= 0) n[dev] = 256;
    }
    return n[dev];
}
int stream_cus(hipStream_t st) {
    {
        std::lock_guard<std::mutex> lk(g_part_mu);
        for (auto& e : g_parts)
            if (e.first == st) return e.second;
    }
    return device_cus();
}

// ---- profiler: event pairs recorded on the launch stream ------------------------------------------
struct ProfRec { int cls; hipEvent_t a, b; double flops, bytes, exec_flops; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static std::mutex g_prof_mu;

static hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(int cls, hipStream_t stream, double flops, double bytes, double exec_flops) : slot(-1), st(stream) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{cls, take_event(), take_event(), flops, bytes, exec_flops < 0.0 ? flops : exec_flops};
    (void)hipEventRecord(r.a, st);
    slot = (int)g_recs.size();
    g_recs.push_back(r);
}
ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_recs[slot].b, st);
}
}  // namespace sprc

extern "C" int sprc_prof_enable(int on) {                 // 1 = start afresh, 0 = pause (records kept for collect), 2 = resume
    using namespace sprc;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (on == 1) {
        for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
        g_recs.clear();
    }
    g_prof_on = on != 0;
    return SPRC_OK;
}

extern "C" int sprc_prof_collect(sprc_prof_entry* out) {
    using namespace sprc;
    SPRC_REQUIRE(out != nullptr, "sprc_prof_collect: null output");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < SPRC_K_COUNT; ++i) out[i] = sprc_prof_entry{0.0, 0.0, 0.0, 0, 0.0, 0.0};
    // launches of one class may overlap in time when the library pipelines two streams: `ms` is the plain sum of launch
    // durations, `busy_ms` the length of the UNION of their [start, end] intervals (time with >= 1 launch of the class
    // executing); on a single stream the two agree.
    std::vector<std::pair<float, float>> iv[SPRC_K_COUNT];
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) { set_error("sprc_prof_collect: event sync failed"); return SPRC_ELAUNCH; }
        float ms = 0.f, t0 = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        (void)hipEventElapsedTime(&t0, g_recs.front().a, r.a);
        out[r.cls].ms += ms; out[r.cls].flops += r.flops; out[r.cls].bytes += r.bytes; out[r.cls].launches += 1; out[r.cls].exec_flops += r.exec_flops;
        iv[r.cls].push_back({t0, t0 + ms});
    }
    for (int c = 0; c < SPRC_K_COUNT; ++c) {
        std::sort(iv[c].begin(), iv[c].end());
        float lo = 0.f, hi = 0.f;
        bool open = false;
        for (auto& x : iv[c]) {
            if (open && x.first <= hi) { hi = std::max(hi, x.second); continue; }
            if (open) out[c].busy_ms += hi - lo;
            lo = x.first; hi = x.second; open = true;
        }
        if (open) out[c].busy_ms += hi - lo;
    }
    return SPRC_OK;
}

extern "C" int sprc_stream_create_partition(int32_t part, int32_t nparts, sprc_stream* out) {
    using namespace sprc;
    SPRC_REQUIRE(out != nullptr && nparts >= 1 && nparts <= 4 && part >= 0 && part < nparts, "sprc_stream_create_partition: part %d of %d", part, nparts);
    const int ncu = device_cus();
    SPRC_REQUIRE(ncu % (8 * nparts) == 0 && ncu <= 1024, "sprc_stream_create_partition: %d CUs do not split into %d partitions per XCD", ncu, nparts);
    // mask bit i = CU (i / 8) of XCD (i % 8) (the KFD distributes the bits round-robin over the XCCs, then over a chiplet's shader engines):
    // partition `part` takes the CUs whose index inside their XCD is congruent to it -- the same share of every XCD and of every XCD's L2
    uint32_t mask[32] = {0};
    int n = 0;
    const char* le = getenv("SPRC_PART_LAYOUT");     // probe switch (tools/cumask_probe.py): 0 = per-XCD shares (default), 1 = whole XCDs
    const int layout = le ? atoi(le) : 0;            // (bit i % 8), 2 = alternate bits, 3 = contiguous bit ranges
    for (int i = 0; i < ncu; ++i) {
        const int owner = layout == 1 ? (i % 8) * nparts / 8 : layout == 2 ? i % nparts : layout == 3 ? (int)((int64_t)i * nparts / ncu) : (i / 8) % nparts;
        if (owner == part) { mask[i >> 5] |= 1u << (i & 31); ++n; }
    }
    hipStream_t st = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)((ncu + 31) / 32), mask);
    if (e != hipSuccess) { set_error("sprc_stream_create_partition: hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e)); return SPRC_EUNSUPPORTED; }
    {
        std::lock_guard<std::mutex> lk(g_part_mu);
        g_parts.push_back({st, n});
    }
    *out = (sprc_stream)st;
    return SPRC_OK;
}
extern "C" int sprc_stream_destroy(sprc_stream s) {
    using namespace sprc;
    {
        std::lock_guard<std::mutex> lk(g_part_mu);
        for (size_t i = 0; i < g_parts.size(); ++i)
            if (g_parts[i].first == (hipStream_t)s) { g_parts.erase(g_parts.begin() + i); break; }
    }
    return hipStreamDestroy((hipStream_t)s) == hipSuccess ? SPRC_OK : SPRC_ELAUNCH;
}
extern "C" int sprc_stream_cus(sprc_stream s) { return sprc::stream_cus((hipStream_t)s); }

extern "C" int sprc_version(void) { return SPRC_ABI_VERSION; }
extern "C" const char* sprc_last_error(void) { return sprc::g_err; }
