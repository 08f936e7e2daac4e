// tuning.hip -- see tuning.h.  Host code only.
#include "tuning.h"

#include <atomic>
#include <cstring>

namespace srs {
namespace tuning {

static std::atomic<int64_t> g_val[N_TUNABLES];
static const char *const kNames[N_TUNABLES] = {"msm_sort", "msm_l0", "msm_wide", "msm_wide_min", "msm_slots", "msm_slot_log", "msm_expect_ovf",
                                               "msm_quad_max", "commit_chunks", "pg_f_eval", "pg_g_fft", "jit_always", "no_jit"};
static struct Init {
    Init() { for (auto &v : g_val) v.store(UNSET, std::memory_order_relaxed); }
} g_init;

int64_t get(Id id) { return g_val[id].load(std::memory_order_relaxed); }
const char *name_of(int id) { return id >= 0 && id < N_TUNABLES ? kNames[id] : nullptr; }
static int find(const char *name) {
    if (!name) return -1;
    for (int i = 0; i < N_TUNABLES; ++i)
        if (std::strcmp(kNames[i], name) == 0) return i;
    return -1;
}
int set(const char *name, int64_t value) {
    const int i = find(name);
    if (i < 0) return -1;
    g_val[i].store(value, std::memory_order_relaxed);
    return 0;
}
int64_t get_by_name(const char *name, int *found) {
    const int i = find(name);
    if (found) *found = i >= 0;
    return i < 0 ? UNSET : get((Id)i);
}
void reset() { for (auto &v : g_val) v.store(UNSET, std::memory_order_relaxed); }

}  // namespace tuning
}  // namespace srs
