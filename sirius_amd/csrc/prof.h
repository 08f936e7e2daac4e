// prof.h -- optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default: when disabled a Scope costs one branch.
#pragma once
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "devrt.h"

namespace srs {
namespace prof {

struct Stat {
    double total_ms = 0;
    uint64_t launches = 0;
    uint64_t units = 0;
};
struct Pending {
    const char *name;
    hipEvent_t e0, e1;
    uint64_t units;
};
struct State {
    bool on = false;
    std::mutex mu;
    std::map<std::string, Stat> stats;
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
};
State &state();

inline bool enabled() { return state().on; }

// records e0 now and e1 at destruction, both on `st`
struct Scope {
    bool live = false;
    Pending p;
    hipStream_t st;
    Scope(const char *name, hipStream_t stream, uint64_t units) : st(stream) {
        State &s = state();
        if (!s.on) return;
        std::lock_guard<std::mutex> lk(s.mu);
        auto take = [&]() {
            hipEvent_t e;
            if (!s.pool.empty()) { e = s.pool.back(); s.pool.pop_back(); return e; }
            if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            return e;
        };
        p.name = name;
        p.units = units;
        p.e0 = take();
        p.e1 = take();
        if (!p.e0 || !p.e1) return;
        (void)hipEventRecord(p.e0, st);
        live = true;
    }
    ~Scope() {
        if (!live) return;
        (void)hipEventRecord(p.e1, st);
        State &s = state();
        std::lock_guard<std::mutex> lk(s.mu);
        s.pending.push_back(p);
    }
};

// call after the stream has been synchronised: folds finished event pairs into the statistics
void collect();
void reset();
void enable(bool on);
bool get(const char *name, Stat &out);

}  // namespace prof
}  // namespace srs
