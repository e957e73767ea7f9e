// Per-kernel timing with HIP events on the launch stream (see cgs_internal.h).
#include <mutex>
#include <vector>
#include "cgs_internal.h"

int g_cgs_prof_on = 0;

namespace {
struct Pair { hipEvent_t a, b; };
std::mutex g_mu;
std::vector<Pair> g_pending[CGS_PROF_COUNT];
std::vector<Pair> g_pool;
hipEvent_t g_open[CGS_PROF_COUNT];
double g_ms[CGS_PROF_COUNT];
int64_t g_launches[CGS_PROF_COUNT];
const char *const kNames[CGS_PROF_COUNT] = {
    "filter", "preprocess", "depth_sort", "offsets_scan", "emit_pairs", "tile_sort", "ranges",
    "blend_fwd", "blend_bwd", "preprocess_bwd", "expand_fwd", "expand_bwd", "rate_fwd", "rate_bwd",
    "mlp_fwd", "mlp_bwd", "mlp_wgrad", "ctx_fwd", "ctx_bwd", "loss_fwd", "loss_bwd",
    "level_mlp_fwd", "level_mlp_bwd", "level_mlp_wgrad"};

Pair get_pair() {
    if (!g_pool.empty()) { Pair p = g_pool.back(); g_pool.pop_back(); return p; }
    Pair p;
    hipEventCreate(&p.a);
    hipEventCreate(&p.b);
    return p;
}
thread_local Pair t_cur[CGS_PROF_COUNT];
}  // namespace

void cgs_prof_begin(int id, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    t_cur[id] = get_pair();
    hipEventRecord(t_cur[id].a, stream);
}

void cgs_prof_end(int id, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    hipEventRecord(t_cur[id].b, stream);
    g_pending[id].push_back(t_cur[id]);
}

static void drain_locked() {
    for (int id = 0; id < CGS_PROF_COUNT; ++id) {
        for (Pair &p : g_pending[id]) {
            hipEventSynchronize(p.b);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { g_ms[id] += ms; g_launches[id] += 1; }
            g_pool.push_back(p);
        }
        g_pending[id].clear();
    }
}

extern "C" int cgs_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    for (int id = 0; id < CGS_PROF_COUNT; ++id) { g_ms[id] = 0.0; g_launches[id] = 0; }
    g_cgs_prof_on = on ? 1 : 0;
    return CGS_OK;
}

extern "C" int cgs_prof_count(void) { return CGS_PROF_COUNT; }

extern "C" const char *cgs_prof_name(int id) { return (id >= 0 && id < CGS_PROF_COUNT) ? kNames[id] : ""; }

extern "C" int cgs_prof_read(int id, double *total_ms, int64_t *launches) {
    if (id < 0 || id >= CGS_PROF_COUNT || !total_ms || !launches) { cgs_set_error("prof_read: bad args"); return CGS_ERR_ARG; }
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    *total_ms = g_ms[id];
    *launches = g_launches[id];
    return CGS_OK;
}
