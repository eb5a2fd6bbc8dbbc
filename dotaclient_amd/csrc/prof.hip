// Optional in-library timing with HIP events on the caller's stream (used by bench.py for the
// roofline line; off by default and then completely inert).
//
// A "region" brackets one kernel launch (or one GEMM call) with two events recorded on the same
// stream the kernel is launched on, and carries the algorithmic flops / bytes of that launch.
// dc_profile_report() synchronises and sums per region name.
#include "../../include/dotaclient_hip.h"
#include "kernels.h"
#include <map>
#include <string>
#include <vector>

namespace dc {

struct ProfRec { hipEvent_t a, b; double flops, bytes; int name_id; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<std::string> g_names;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

bool prof_enabled() { return g_prof_on; }

int prof_begin(const char* name, double flops, double bytes, hipStream_t s) {
    if (!g_prof_on) return -1;
    int id = -1;
    for (size_t i = 0; i < g_names.size(); ++i) if (g_names[i] == name) { id = (int)i; break; }
    if (id < 0) { g_names.push_back(name); id = (int)g_names.size() - 1; }
    ProfRec r{get_event(), get_event(), flops, bytes, id};
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}

void prof_end(int h, hipStream_t s) {
    if (h < 0 || !g_prof_on) return;
    (void)hipEventRecord(g_recs[h].b, s);
}

}  // namespace dc

extern "C" {

int dc_profile_enable(int on) {
    dc::g_prof_on = on != 0;
    return 0;
}

// Writes up to max_regions entries: names (64 bytes each, NUL-terminated), launches, total_ms,
// flops, bytes (sums).  Returns the number of regions and clears the records.
int dc_profile_report(char* names, int64_t* launches, double* total_ms, double* flops, double* bytes, int max_regions) {
    using namespace dc;
    const int n = (int)g_names.size();
    std::vector<int64_t> cnt(n, 0);
    std::vector<double> ms(n, 0.0), fl(n, 0.0), by(n, 0.0);
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.b);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        cnt[r.name_id] += 1; ms[r.name_id] += t; fl[r.name_id] += r.flops; by[r.name_id] += r.bytes;
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    g_recs.clear();
    int out = 0;
    for (int i = 0; i < n && out < max_regions; ++i) {
        if (cnt[i] == 0) continue;
        snprintf(names + 64 * out, 64, "%s", g_names[i].c_str());
        launches[out] = cnt[i]; total_ms[out] = ms[i]; flops[out] = fl[i]; bytes[out] = by[i];
        ++out;
    }
    return out;
}

}  // extern "C"
