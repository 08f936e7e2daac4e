// hipemu.h -- a tiny CPU emulator of the HIP execution model.  TEST INFRASTRUCTURE ONLY.
//
// There is no GPU in the development container and GPU minutes are rationed, so the kernel
// *logic* (indexing, LDS protocols, barriers, wave shuffles, atomics) of sirius_amd/csrc/*.hip is
// additionally compiled with g++ against this header (-DSRS_EMU) and executed on the host by
// tests/emu/.  It is never built into, linked with or loaded by the product library:
// libsirius_amd.so is compiled by hipcc for gfx950 only and has no CPU path.
//
// Model: blocks run one after another inside one OS thread; the threads of a block are ucontext
// fibers, so __syncthreads and wave shuffles have true barrier semantics (a fiber that reaches a
// barrier yields until every live thread of the block / lane of the wave has arrived);
// `__shared__` is a `static thread_local` (one block alive at a time per host worker).  Wave = 64 lanes.
// Large grids are spread over a few host threads (blocks are independent; global atomics are real atomics).
#pragma once
#include <ucontext.h>
#include <thread>
#include <cstdlib>
#include <mutex>
#include <condition_variable>
#include <functional>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __restrict__
#define __launch_bounds__(...)
#define __constant__ static

struct alignas(16) uint4 {
    unsigned x, y, z, w;
};
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ {
    unsigned x, y, z;
};
inline thread_local uint3_ threadIdx, blockIdx;
inline thread_local dim3 blockDim, gridDim;
static constexpr int warpSize = 64;

namespace hipemu {
enum Wait { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3, WAIT_QUAD = 4 };
struct Fiber {
    ucontext_t ctx;
    std::unique_ptr<char[]> stack;
    int state = RUNNABLE;
    uint3_ tid;
};
struct BlockCtx {
    std::vector<Fiber> fibers;
    std::vector<uint64_t> wave_slots;   // 64 x 8-byte exchange slots per wave
    std::vector<uint32_t> quad_slots;   // 8 dwords per lane: quad exchange of a whole field element
    std::vector<uint32_t> bulk_slots;   // 32 dwords per lane: wave exchange of a whole XYZZ point
    unsigned nthreads = 0;
    unsigned cur = 0;                   // running fiber
    ucontext_t sched;
    void (*entry)(void *) = nullptr;
    void *entry_arg = nullptr;
};
inline thread_local BlockCtx g_ctx;
inline thread_local unsigned t_lin = 0;   // linear thread id of the running fiber
inline constexpr size_t kStack = 512 * 1024;

inline void yield_as(int wait) {
    Fiber &f = g_ctx.fibers[g_ctx.cur];
    if (wait == WAIT_QUAD) {
        // last lane of the quad to arrive releases the others and keeps running (no scheduler round needed)
        unsigned lo = g_ctx.cur & ~3u, hi = std::min(g_ctx.nthreads, lo + 4);
        bool all = true;
        for (unsigned t = lo; t < hi; ++t)
            if (t != g_ctx.cur && g_ctx.fibers[t].state != WAIT_QUAD && g_ctx.fibers[t].state != DONE) all = false;
        if (all) {
            for (unsigned t = lo; t < hi; ++t)
                if (g_ctx.fibers[t].state == WAIT_QUAD) g_ctx.fibers[t].state = RUNNABLE;
            return;
        }
    }
    f.state = wait;
    swapcontext(&f.ctx, &g_ctx.sched);
}
inline void trampoline() {
    g_ctx.entry(g_ctx.entry_arg);
    g_ctx.fibers[g_ctx.cur].state = DONE;
    swapcontext(&g_ctx.fibers[g_ctx.cur].ctx, &g_ctx.sched);
}
// run one block to completion
inline void run_block() {
    unsigned n = g_ctx.nthreads;
    for (unsigned t = 0; t < n; ++t) {
        Fiber &f = g_ctx.fibers[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack.get();
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
        f.state = RUNNABLE;
    }
    for (;;) {
        unsigned done = 0, progressed = 0;
        for (unsigned qb = 0; qb < n; qb += 4) {
            // run a quad until none of its lanes is runnable: quad exchanges complete here, without global rounds
            for (bool again = true; again;) {
                again = false;
                for (unsigned t = qb; t < std::min(n, qb + 4); ++t) {
                    Fiber &f = g_ctx.fibers[t];
                    if (f.state != RUNNABLE) continue;
                    g_ctx.cur = t;
                    t_lin = t;
                    threadIdx = f.tid;
                    swapcontext(&g_ctx.sched, &f.ctx);
                    ++progressed;
                }
                for (unsigned t = qb; t < std::min(n, qb + 4); ++t) again = again || g_ctx.fibers[t].state == RUNNABLE;
            }
        }
        for (unsigned t = 0; t < n; ++t) done += g_ctx.fibers[t].state == DONE;
        if (done == n) break;
        // release barriers whose participants have all arrived
        unsigned live = 0, at_block = 0;
        for (unsigned t = 0; t < n; ++t) {
            int s = g_ctx.fibers[t].state;
            if (s != DONE) ++live;
            if (s == WAIT_BLOCK) ++at_block;
        }
        bool released = false;
        if (live && at_block == live) {
            for (unsigned t = 0; t < n; ++t)
                if (g_ctx.fibers[t].state == WAIT_BLOCK) g_ctx.fibers[t].state = RUNNABLE;
            released = true;
        }
        for (unsigned w = 0; w * 64 < n; ++w) {
            unsigned lo = w * 64, hi = std::min(n, lo + 64), wl = 0, ww = 0;
            for (unsigned t = lo; t < hi; ++t) {
                int s = g_ctx.fibers[t].state;
                if (s != DONE) ++wl;
                if (s == WAIT_WAVE) ++ww;
            }
            if (wl && ww == wl) {
                for (unsigned t = lo; t < hi; ++t)
                    if (g_ctx.fibers[t].state == WAIT_WAVE) g_ctx.fibers[t].state = RUNNABLE;
                released = true;
            }
        }
        for (unsigned qd = 0; qd * 4 < n; ++qd) {               // quad exchanges (DPP quad_perm): 4 adjacent lanes
            unsigned lo = qd * 4, hi = std::min(n, lo + 4), ql = 0, qw = 0;
            for (unsigned t = lo; t < hi; ++t) {
                int s = g_ctx.fibers[t].state;
                if (s != DONE) ++ql;
                if (s == WAIT_QUAD) ++qw;
            }
            // a DPP read involves exactly the lanes executing it: on the device a quad is either wholly inside a
            // branch or wholly outside (callers keep quads convergent), so "all lanes of the quad that are waiting
            // at a quad exchange and none runnable" is the release condition
            bool any_runnable = false;
            for (unsigned t = lo; t < hi; ++t) any_runnable = any_runnable || g_ctx.fibers[t].state == RUNNABLE;
            if (qw && !any_runnable && qw == ql) {
                for (unsigned t = lo; t < hi; ++t)
                    if (g_ctx.fibers[t].state == WAIT_QUAD) g_ctx.fibers[t].state = RUNNABLE;
                released = true;
            }
        }
        if (!released && !progressed) {
            std::fprintf(stderr, "hipemu: barrier deadlock (divergent __syncthreads / shuffle)\n");
            std::abort();
        }
    }
}
}  // namespace hipemu

inline void __syncthreads() { hipemu::yield_as(hipemu::WAIT_BLOCK); }
inline void __threadfence() {}
inline void __threadfence_block() {}

template <class T>
inline T __emu_wave_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shuffle payload <= 8 bytes");
    using namespace hipemu;
    unsigned wave = t_lin / 64, lane = t_lin % 64;
    uint64_t raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    g_ctx.wave_slots[wave * 64 + lane] = raw;
    yield_as(WAIT_WAVE);
    unsigned wave_n = std::min(64u, g_ctx.nthreads - wave * 64);
    uint64_t got = (src_lane >= 0 && (unsigned)src_lane < wave_n) ? g_ctx.wave_slots[wave * 64 + src_lane] : raw;
    yield_as(WAIT_WAVE);
    T o;
    std::memcpy(&o, &got, sizeof(T));
    return o;
}
// value of lane (quad base + k) for the 4 lanes of a quad; only the quad has to be convergent
inline uint32_t __emu_quad_bcast(uint32_t v, int k) {
    using namespace hipemu;
    unsigned wave = t_lin / 64, lane = t_lin % 64;
    g_ctx.wave_slots[wave * 64 + lane] = v;
    yield_as(WAIT_QUAD);
    uint64_t got = g_ctx.wave_slots[wave * 64 + (lane & ~3u) + (unsigned)k];
    yield_as(WAIT_QUAD);
    return (uint32_t)got;
}
// __shfl_down of a 128-byte object in one wave barrier pair (32 dword shuffles would cost 64 fiber switches)
inline void __emu_shfl_down_bulk128(const void *in, void *out, unsigned d, int width) {
    using namespace hipemu;
    unsigned wave = t_lin / 64, lane = t_lin % 64;
    std::memcpy(&g_ctx.bulk_slots[(size_t)t_lin * 48], in, 128);
    yield_as(WAIT_WAVE);
    unsigned wave_n = std::min(64u, g_ctx.nthreads - wave * 64);
    int src = (int)lane + (int)d;
    if (src / width != (int)lane / width || (unsigned)src >= wave_n) src = (int)lane;
    uint32_t tmp[32];
    std::memcpy(tmp, &g_ctx.bulk_slots[(size_t)(wave * 64 + (unsigned)src) * 48], 128);
    yield_as(WAIT_WAVE);
    std::memcpy(out, tmp, 128);
}
// the same for up to 48 dwords (a point in 9-limb coordinates is 36)
inline void __emu_shfl_down_bulk(const void *in, void *out, unsigned d, int width, unsigned nwords) {
    using namespace hipemu;
    unsigned wave = t_lin / 64, lane = t_lin % 64;
    std::memcpy(&g_ctx.bulk_slots[(size_t)t_lin * 48], in, nwords * 4);
    yield_as(WAIT_WAVE);
    unsigned wave_n = std::min(64u, g_ctx.nthreads - wave * 64);
    int src = (int)lane + (int)d;
    if (src / width != (int)lane / width || (unsigned)src >= wave_n) src = (int)lane;
    uint32_t tmp[48];
    std::memcpy(tmp, &g_ctx.bulk_slots[(size_t)(wave * 64 + (unsigned)src) * 48], nwords * 4);
    yield_as(WAIT_WAVE);
    std::memcpy(out, tmp, nwords * 4);
}
inline void __emu_quad_bcast_n(const uint32_t *v, int k, uint32_t *out, unsigned nwords) {
    using namespace hipemu;
    std::memcpy(&g_ctx.quad_slots[(size_t)t_lin * 12], v, nwords * 4);
    yield_as(WAIT_QUAD);
    uint32_t tmp[12];
    std::memcpy(tmp, &g_ctx.quad_slots[(size_t)((t_lin & ~3u) + (unsigned)k) * 12], nwords * 4);
    yield_as(WAIT_QUAD);
    std::memcpy(out, tmp, nwords * 4);
}
// 8 dwords at once (one field element): a single barrier pair instead of 8
inline void __emu_quad_bcast8(const uint32_t *v, int k, uint32_t *out) {
    using namespace hipemu;
    std::memcpy(&g_ctx.quad_slots[(size_t)t_lin * 12], v, 32);
    yield_as(WAIT_QUAD);
    uint32_t tmp[8];
    std::memcpy(tmp, &g_ctx.quad_slots[(size_t)((t_lin & ~3u) + (unsigned)k) * 12], 32);
    yield_as(WAIT_QUAD);
    std::memcpy(out, tmp, 32);
}
template <class T>
inline T __shfl(T v, int lane, int width = 64) {
    int self = hipemu::t_lin % 64;
    return __emu_wave_exchange(v, (self / width) * width + (lane % width));
}
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    int self = hipemu::t_lin % 64;
    return __emu_wave_exchange(v, self ^ mask);
}
template <class T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
    int self = hipemu::t_lin % 64;
    int src = self + (int)d;
    if (src / width != self / width) src = self;
    return __emu_wave_exchange(v, src);
}
template <class T>
inline T __shfl_up(T v, unsigned d, int width = 64) {
    int self = hipemu::t_lin % 64;
    int src = self - (int)d;
    if (src < 0 || src / width != self / width) src = self;
    return __emu_wave_exchange(v, src);
}
inline unsigned long long __ballot(int pred) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) {
        int p = __emu_wave_exchange(pred ? 1 : 0, l);
        unsigned wave = hipemu::t_lin / 64;
        unsigned wave_n = std::min(64u, hipemu::g_ctx.nthreads - wave * 64);
        if ((unsigned)l < wave_n && p) m |= 1ull << l;
    }
    return m;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline unsigned __brev(unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i);
    return r;
}
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }

template <class T>
inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T>
inline T atomicSub(T *p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_SEQ_CST); }
template <class T>
inline T atomicMax(T *p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
template <class T>
inline T atomicMin(T *p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
template <class T>
inline T atomicExch(T *p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
template <class T>
inline T atomicCAS(T *p, T cmp, T v) {
    __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return cmp;
}

// ---------------------------------------------------------------- runtime API subset
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorPeerAccessAlreadyEnabled = 704 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline const char *hipGetErrorString(hipError_t) { return "hipemu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) {
    *p = std::aligned_alloc(256, (n + 255) / 256 * 256 + 256);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <class T>
inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; };
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { a->type = hipMemoryTypeUnregistered; return hipSuccess; }   // the emulator's "host" is pageable: the launch-first order runs
inline hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind,
                                   hipStream_t = nullptr) {
    for (size_t r = 0; r < height; ++r) std::memmove((char *)d + r * dpitch, (const char *)s + r * spitch, width);
    return hipSuccess;
}

inline hipError_t hipMemset2DAsync(void *d, size_t pitch, int value, size_t width, size_t height, hipStream_t = nullptr) {
    for (size_t r = 0; r < height; ++r) std::memset((char *)d + r * pitch, value, width);
    return hipSuccess;
}

namespace hipemu {
// Run kernel(args...) over grid x block: the threads of a block are fibers; the blocks of a large grid are dealt to a few
// host threads (every piece of emulator state is thread_local), a small grid runs on the calling thread.
template <class K, class... A>
inline void run_blocks(K kernel, dim3 grid, dim3 block, size_t first, size_t stride, A... args) {
    unsigned nthreads = block.x * block.y * block.z;
    blockDim = block;
    gridDim = grid;
    g_ctx.nthreads = nthreads;
    if (g_ctx.fibers.size() < nthreads) {
        size_t old = g_ctx.fibers.size();
        g_ctx.fibers.resize(nthreads);
        for (size_t t = old; t < nthreads; ++t) g_ctx.fibers[t].stack.reset(new char[kStack]);
    }
    for (unsigned t = 0; t < nthreads; ++t) {
        g_ctx.fibers[t].tid.x = t % block.x;
        g_ctx.fibers[t].tid.y = (t / block.x) % block.y;
        g_ctx.fibers[t].tid.z = t / (block.x * block.y);
    }
    g_ctx.wave_slots.assign((size_t)((nthreads + 63) / 64) * 64, 0);
    g_ctx.quad_slots.assign((size_t)((nthreads + 63) / 64) * 64 * 12, 0);
    g_ctx.bulk_slots.assign((size_t)((nthreads + 63) / 64) * 64 * 48, 0);
    auto body = [&]() { kernel(args...); };
    using B = decltype(body);
    g_ctx.entry = [](void *p) { (*static_cast<B *>(p))(); };
    g_ctx.entry_arg = &body;
    size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    for (size_t b = first; b < nblocks; b += stride) {
        blockIdx.x = (unsigned)(b % grid.x);
        blockIdx.y = (unsigned)((b / grid.x) % grid.y);
        blockIdx.z = (unsigned)(b / ((size_t)grid.x * grid.y));
        run_block();
    }
}
inline unsigned host_workers() {
    static const unsigned n = [] {
        const char *e = std::getenv("HIPEMU_THREADS");
        unsigned v = e ? (unsigned)std::atoi(e) : std::thread::hardware_concurrency();
        return v < 1 ? 1u : (v > 16 ? 16u : v);
    }();
    return n;
}
// persistent host workers (their fiber stacks live as long as the process): one job at a time; a launch that finds the pool
// busy (another host thread is launching) simply runs on its own thread
struct Pool {
    std::mutex job_mu;                      // held by the launching thread for the whole job
    std::mutex mu;
    std::condition_variable cv, done_cv;
    std::vector<std::thread> threads;
    std::function<void(unsigned)> job;
    unsigned long gen = 0;
    unsigned pending = 0;
    bool quit = false;
    void start(unsigned n) {
        for (unsigned w = 1; w < n; ++w)
            threads.emplace_back([this, w]() {
                unsigned long seen = 0;
                for (;;) {
                    std::function<void(unsigned)> j;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return quit || gen != seen; });
                        if (quit) return;
                        seen = gen;
                        j = job;
                    }
                    j(w);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) done_cv.notify_all();
                    }
                }
            });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv.notify_all();
        for (auto &t : threads) t.join();
    }
};
inline Pool &pool() {
    static Pool p;
    static std::once_flag once;
    std::call_once(once, [] { p.start(host_workers()); });
    return p;
}
template <class K, class... A>
inline void launch(K kernel, dim3 grid, dim3 block, size_t dyn_smem, A... args) {
    (void)dyn_smem;
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    const size_t work = nblocks * block.x * block.y * block.z;
    const unsigned W = host_workers();
    if (W <= 1 || nblocks < 2 || work < 4096) {
        run_blocks(kernel, grid, block, 0, 1, args...);
        return;
    }
    Pool &P = pool();
    std::unique_lock<std::mutex> job_lk(P.job_mu, std::try_to_lock);
    if (!job_lk.owns_lock()) {
        run_blocks(kernel, grid, block, 0, 1, args...);
        return;
    }
    {
        std::lock_guard<std::mutex> lk(P.mu);
        P.job = [=](unsigned w) { run_blocks(kernel, grid, block, w, W, args...); };
        P.pending = W - 1;
        ++P.gen;
    }
    P.cv.notify_all();
    run_blocks(kernel, grid, block, 0, W, args...);
    std::unique_lock<std::mutex> lk(P.mu);
    P.done_cv.wait(lk, [&] { return P.pending == 0; });
}
}  // namespace hipemu
