// prof.hip -- see prof.h
#include "prof.h"

namespace srs {
namespace prof {

State &state() {
    static State s;
    return s;
}

void collect() {
    State &s = state();
    if (!s.on && s.pending.empty()) return;
    std::lock_guard<std::mutex> lk(s.mu);
    for (auto &p : s.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            Stat &st = s.stats[p.name];
            st.total_ms += ms;
            st.launches += 1;
            st.units += p.units;
        }
        s.pool[p.device].push_back(p.e0);
        s.pool[p.device].push_back(p.e1);
    }
    s.pending.clear();
}

void reset() {
    collect();
    State &s = state();
    std::lock_guard<std::mutex> lk(s.mu);
    s.stats.clear();
    s.seen.clear();
}

void sampling(uint32_t every) {
    State &s = state();
    std::lock_guard<std::mutex> lk(s.mu);
    s.sample_every = every ? every : 1;
}

void enable(bool on) {
    collect();
    state().on = on;
}

bool get(const char *name, Stat &out) {
    collect();
    State &s = state();
    std::lock_guard<std::mutex> lk(s.mu);
    auto it = s.stats.find(name);
    if (it == s.stats.end()) return false;
    out = it->second;
    return true;
}

}  // namespace prof
}  // namespace srs
