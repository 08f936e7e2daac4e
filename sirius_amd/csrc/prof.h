// prof.h -- optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default: when disabled a Scope costs one branch.
#pragma once
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "devrt.h"
#if !defined(SRS_EMU)
#include <hip/hip_ext.h>
#endif

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
    int device = 0;                 // events belong to the device they were created on (r06: one process may drive several)
};
struct State {
    bool on = false;
    uint32_t sample_every = 1;      // SRS_LAUNCH_TIMED binds events to every n-th launch of a name only (sampling(): an event-bracketed launch costs ~8 us of idle device)
    std::map<std::string, uint64_t> seen;      // launches per name since reset(), sampled or not
    std::mutex mu;
    std::map<std::string, Stat> stats;
    std::vector<Pending> pending;
    std::map<int, std::vector<hipEvent_t>> pool;      // per device
};
inline int current_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
}
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
        p.device = current_device();
        std::vector<hipEvent_t> &pool = s.pool[p.device];
        auto take = [&]() {
            hipEvent_t e;
            if (!pool.empty()) { e = pool.back(); pool.pop_back(); return e; }
            if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            return e;
        };
        p.name = name;
        p.units = units;
        p.e0 = take();
        p.e1 = take();
        if (!p.e0 || !p.e1) {
            if (p.e0) pool.push_back(p.e0);
            if (p.e1) pool.push_back(p.e1);
            return;
        }
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

// The same timing for ONE kernel launch without event packets around it: the two events are bound to the kernel's own dispatch
// (hipExtLaunchKernelGGL), so the stream carries no extra markers -- a Scope's hipEventRecord pair costs ~9 us of idle device before
// and after the kernel it brackets (r03 kernel traces: the only gaps inside an MSM chunk were the two around k_accum0).
struct KernelEvents {
    bool live = false;
    Pending p;
    KernelEvents(const char *name, uint64_t units) {
        State &s = state();
        if (!s.on) return;
        std::lock_guard<std::mutex> lk(s.mu);
        if (s.sample_every > 1 && (s.seen[name]++ % s.sample_every) != 0) return;      // not this launch
        p.device = current_device();
        std::vector<hipEvent_t> &pool = s.pool[p.device];
        auto take = [&]() {
            hipEvent_t e;
            if (!pool.empty()) { e = pool.back(); pool.pop_back(); return e; }
            if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            return e;
        };
        p.name = name;
        p.units = units;
        p.e0 = take();
        p.e1 = take();
        live = p.e0 && p.e1;
        if (!live) {                              // the one event that could be had goes back to the pool
            if (p.e0) pool.push_back(p.e0);
            if (p.e1) pool.push_back(p.e1);
        }
    }
    void launched() {
        State &s = state();
        std::lock_guard<std::mutex> lk(s.mu);
        s.pending.push_back(p);
    }
};
#if defined(SRS_EMU)
#define SRS_LAUNCH_TIMED(name, units, kernel, grid, block, smem, stream, ...) SRS_LAUNCH(kernel, grid, block, smem, stream, __VA_ARGS__)
#else
#define SRS_LAUNCH_TIMED(name, units, kernel, grid, block, smem, stream, ...)                                                        \
    do {                                                                                                                            \
        ::srs::prof::KernelEvents ke_(name, units);                                                                                 \
        if (ke_.live) {                                                                                                             \
            hipExtLaunchKernelGGL(kernel, dim3 grid, dim3 block, (uint32_t)(smem), (hipStream_t)(stream), ke_.p.e0, ke_.p.e1, 0u,   \
                                  __VA_ARGS__);                                                                                     \
            ke_.launched();                                                                                                         \
        } else {                                                                                                                    \
            SRS_LAUNCH(kernel, grid, block, smem, stream, __VA_ARGS__);                                                             \
        }                                                                                                                           \
    } while (0)
#endif

// call after the stream has been synchronised: folds finished event pairs into the statistics
void collect();
void reset();
void enable(bool on);
void sampling(uint32_t every);      // 1 = every timed launch (default)
bool get(const char *name, Stat &out);

}  // namespace prof
}  // namespace srs
