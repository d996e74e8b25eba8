// core.hip -- version, error reporting
#include <stdarg.h>

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

extern "C" int sprc_version(void) { return SPRC_ABI_VERSION; }
extern "C" const char* sprc_last_error(void) { return sprc::g_err; }
