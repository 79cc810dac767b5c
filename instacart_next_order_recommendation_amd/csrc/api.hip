// api.hip — diagnostics half of the C ABI: thread-local error text, version string and the
// hipEvent kernel timers that bench.py reads for its roofline line.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include "common.h"

namespace icrec {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

static std::mutex g_attr_mu;
static std::set<std::pair<const void*, int>> g_attr_done;

int ensure_dynamic_lds(const void* kernel, int bytes) {
    int dev = 0;
    ICREC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_attr_mu);
    if (g_attr_done.count({kernel, dev})) return ICREC_OK;
    ICREC_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    g_attr_done.insert({kernel, dev});
    return ICREC_OK;
}

static bool g_timing = false;
static std::mutex g_tmu;
struct Pending {
    hipEvent_t a, b;
};
static std::vector<Pending> g_pending[T_NSLOTS];
static TimingSlot g_slots[T_NSLOTS];

bool timing_on() { return g_timing; }

ScopedTimer::ScopedTimer(int slot_, hipStream_t s) : slot(slot_), stream(s) {
    if (!g_timing) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
    hipEventRecord(a, stream);
}
ScopedTimer::~ScopedTimer() {
    if (!a || !b) return;
    hipEventRecord(b, stream);
    std::lock_guard<std::mutex> lk(g_tmu);
    g_pending[slot].push_back({a, b});
}

static void resolve(int slot) {
    for (auto& p : g_pending[slot]) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_slots[slot].total_ms += ms;
            g_slots[slot].n += 1;
        }
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    g_pending[slot].clear();
}

}  // namespace icrec

using namespace icrec;

extern "C" {

const char* icrec_last_error(void) { return g_err; }

const char* icrec_version(void) { return "icrec 0.1 (gfx950; exact-f32 and f16x3 MFMA)"; }

int icrec_timing_enable(int on) {
    g_timing = on != 0;
    return ICREC_OK;
}

int icrec_timing_reset(void) {
    std::lock_guard<std::mutex> lk(g_tmu);
    for (int s = 0; s < T_NSLOTS; ++s) {
        resolve(s);
        g_slots[s] = TimingSlot();
    }
    return ICREC_OK;
}

int icrec_timing_query(int which, double* avg_ms, int64_t* n_launches) {
    ICREC_REQUIRE(which >= 0 && which < T_NSLOTS && avg_ms && n_launches, "icrec_timing_query: bad argument");
    std::lock_guard<std::mutex> lk(g_tmu);
    resolve(which);
    *n_launches = g_slots[which].n;
    *avg_ms = g_slots[which].n ? g_slots[which].total_ms / (double)g_slots[which].n : 0.0;
    return ICREC_OK;
}

}  // extern "C"
