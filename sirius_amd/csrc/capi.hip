// capi.hip -- the extern "C" surface declared in include/sirius_amd.h.
#include "../../include/sirius_amd.h"

#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <set>
#include <thread>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <atomic>
#include <chrono>
#include <string>
#include <vector>

#include "curve.cuh"
#include "decider.h"
#include "devrt.h"
#include "msm.h"
#include "ntt.h"
#include "poseidon.h"
#include "prof.h"
#include "rowprog.h"
#include "tuning.h"

namespace srs {
static thread_local std::string g_err;

// SRS_HOST_TRACE=1: host-clock checkpoints of the composite entries (where the time BETWEEN kernels goes), printed to stderr per call
struct HostTrace {
    const char *name;
    bool on;
    std::vector<std::pair<const char *, double>> pts;
    static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    explicit HostTrace(const char *n) : name(n) {
        static const bool enabled = [] { const char *e = std::getenv("SRS_HOST_TRACE"); return e && e[0] == '1'; }();
        on = enabled;
        if (on) pts.emplace_back("enter", now());
    }
    void mark(const char *what) { if (on) pts.emplace_back(what, now()); }
    ~HostTrace() {
        if (!on || pts.size() < 2) return;
        std::string line = std::string("[srs host trace] ") + name + ":";
        for (size_t i = 1; i < pts.size(); ++i) line += " " + std::string(pts[i].first) + " +" + std::to_string((long)(pts[i].second - pts[i - 1].second)) + "us";
        line += " | total " + std::to_string((long)(pts.back().second - pts.front().second)) + "us, entered at " + std::to_string((long long)pts.front().second) + "\n";
        fputs(line.c_str(), stderr);
    }
};
// device scratch of the handle-less entry points (folds, lookup h/g) when they are given HOST operands: grow-only, one
// per calling thread, so that a fold does not pay a hipMalloc + hipFree (an implicit device synchronisation) per call
static thread_local srs::ThreadArena g_scratch;
void set_error(const std::string &msg) { g_err = msg; }
const char *get_error() { return g_err.c_str(); }
}  // namespace srs

using namespace srs;

static_assert(sizeof(srs_fe) == sizeof(fe_t) && sizeof(srs_affine) == sizeof(affine_t), "ABI layout");

// One logical shard of a multi-device key (srs_ck_create_multi): its stripes of the window table on `device`, a stream and a
// staging arena there, and a host thread bound to that device which runs the shard's part of every commit.
struct CkShard {
    int device = 0;
    msm::Key key;
    Arena staging;
    hipStream_t stream = nullptr;
    // streamed commits on a multi-device key (r05, multi_commit_streamed): the shard's stripes come up over ITS link in chunks that overlap ITS
    // MSM; `full` = a vector-sized landing buffer on the shard's device in which only the shard's stripes are ever filled
    hipStream_t copy_stream = nullptr, peer_stream = nullptr;
    std::vector<hipEvent_t> events;
    hipEvent_t peer_done = nullptr;
    bool peer_pending = false;      // peer_done has been recorded: the forwarded stripes of the previous commit may still be read from `full`
    fe_t *full = nullptr;
    size_t full_cap = 0;
    uint64_t stat_h2d_bytes = 0, stat_peer_bytes = 0, stat_streamed = 0;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    std::function<void()> job;      // set by post(), cleared by the worker
    bool has_job = false, quit = false;
    int rc = 0;
    std::string err;

    void loop() {
        (void)hipSetDevice(device);
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return has_job || quit; });
                if (quit && !has_job) return;
                fn = job;
            }
            int r = 0;
            std::string e;
            try {
                fn();
            } catch (const DeviceError &de) {
                r = de.rc;
                e = get_error();
            } catch (const std::exception &ex) {
                r = SRS_ERR_DEVICE;
                e = ex.what();
            } catch (...) {
                r = SRS_ERR_DEVICE;
                e = "unknown exception in a key shard";
            }
            std::lock_guard<std::mutex> lk(mu);
            rc = r;
            err = e;
            has_job = false;
            done_cv.notify_all();
        }
    }
    void post(std::function<void()> fn) {
        if constexpr (rt::kEmulated) {
            // the CPU logic emulator keeps one global execution context (tests/emu/hipemu.h): shard jobs run inline, one by one
            rc = 0;
            try {
                fn();
            } catch (const DeviceError &de) {
                rc = de.rc;
                err = get_error();
            }
            return;
        }
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(fn);
        has_job = true;
        rc = 0;
        cv.notify_one();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        done_cv.wait(lk, [&] { return !has_job; });
        return rc;
    }
};

struct srs_ck {
    // One commit at a time per key handle: the scratch arena, the landing slots, the running buckets of a chunked commit, the copy
    // stream and the shard workers' job slots are per-handle state.  Calls from several host threads on ONE handle serialise here
    // (recursive: srs_commit_upload* end in srs_commit); distinct handles run concurrently.
    std::recursive_mutex mu;
    msm::Key key;       // the key of a single-device handle; for a multi-device handle only curve / global_len are meaningful
    Arena staging;      // H2D staging of host scalars
    hipStream_t copy_stream = nullptr;       // srs_commit_upload: uploads run here, the MSMs on the caller's stream
    std::vector<hipEvent_t> events;
    std::vector<std::unique_ptr<CkShard>> shards;      // non-empty: multi-device key
};

struct srs_poseidon {
    poseidon::Hash *h = nullptr;
};

struct srs_sparse {
    decider::Sparse *m = nullptr;
    Arena io;
};

struct srs_structure {
    rowprog::Structure *s = nullptr;
    Arena io;           // staged witnesses / cross-term vectors
    // rows a sharded rank reads beyond its own stripes (srs_structure_fold_sharded), as a device list per reference_compat value;
    // valid for the (rank, world) it was built for
    uint32_t *halo_dev[2] = {nullptr, nullptr};
    size_t halo_n[2] = {0, 0};
    uint32_t halo_rank = 0, halo_world = 0;
};

namespace {

// Persistent host workers for the few independent scalar multiplications of an instance fold (srs_point_lincomb):
// spawning a std::thread per point cost ~30 us each, more than the 80 us scalar multiplication it ran.
class HostPool {
public:
    static HostPool &get() {
        static HostPool *p = new HostPool();      // never destroyed: workers are detached and outlive static teardown
        return *p;
    }
    // runs fn(0..n-1); the caller takes part; returns when all are done
    void parallel_for(size_t n, const std::function<void(size_t)> &fn) {
        if (n <= 1 || workers_ == 0) {
            for (size_t i = 0; i < n; ++i) fn(i);
            return;
        }
        std::unique_lock<std::mutex> call(call_mu_);     // one parallel_for at a time
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            total_ = n;
            next_ = 0;
            pending_ = n;
            ++generation_;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
        total_ = next_ = 0;
    }

private:
    HostPool() {
        unsigned hw = std::thread::hardware_concurrency();
        workers_ = hw > 1 ? std::min(7u, hw - 1) : 0;
        for (unsigned i = 0; i < workers_; ++i) std::thread([this] { loop(); }).detach();
    }
    // every index is claimed under the mutex (a handful of items per call: contention is irrelevant, and a worker that
    // wakes up late can never take an index of the wrong call)
    void drain() {
        for (;;) {
            const std::function<void(size_t)> *fn;
            size_t i;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (next_ >= total_) return;
                i = next_++;
                fn = fn_;
            }
            (*fn)(i);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_cv_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
            }
            drain();
        }
    }
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t total_ = 0, next_ = 0, pending_ = 0;
    uint64_t generation_ = 0;
    unsigned workers_ = 0;
};

int fail(int rc, const std::string &msg) {
    set_error(msg);
    return rc;
}

template <class Fn>
int guarded(Fn &&fn) {
    try {
        return fn();
    } catch (const DeviceError &e) {
        return e.rc;
    } catch (const std::bad_alloc &) {
        return fail(SRS_ERR_DEVICE, "host allocation failed");
    } catch (const std::exception &e) {
        return fail(SRS_ERR_DEVICE, e.what());
    }
}

bool valid_curve(int c) { return c == SRS_CURVE_BN256 || c == SRS_CURVE_GRUMPKIN; }
bool valid_field(int f) { return f == SRS_FIELD_FR || f == SRS_FIELD_FQ; }

size_t shard_count(size_t n, uint32_t rank, uint32_t world) {
    if (world == 1) return n;
    const size_t S = (size_t)1 << msm::STRIPE_LOG;
    size_t full = n >> msm::STRIPE_LOG, rem = n & (S - 1);
    size_t cnt = (full / world) * S;
    if (rank < full % world) cnt += S;
    if (rank == full % world) cnt += rem;
    return cnt;
}

// XYZZ -> affine for a batch with ONE field inversion (Montgomery's trick); identities pass through
template <class C>
void to_affine_batch_t(const xyzz_t *in, affine_t *out, size_t n) {
    using F = typename C::F;
    std::vector<fe_t> pre(n);
    fe_t acc = F::one();
    for (size_t i = 0; i < n; ++i) {
        pre[i] = acc;
        if (!Ec<C>::is_identity(in[i])) acc = F::mul(acc, F::mul(in[i].zz, in[i].zzz));
    }
    fe_t inv = F::inv(acc);
    for (size_t i = n; i-- > 0;) {
        if (Ec<C>::is_identity(in[i])) { out[i] = Ec<C>::affine_identity(); continue; }
        fe_t d = F::mul(inv, pre[i]);                       // 1 / (zz * zzz)
        inv = F::mul(inv, F::mul(in[i].zz, in[i].zzz));
        out[i].x = F::mul(in[i].x, F::mul(d, in[i].zzz));
        out[i].y = F::mul(in[i].y, F::mul(d, in[i].zz));
    }
}
void to_affine_batch(int curve, const xyzz_t *in, affine_t *out, size_t n) {
    if (curve == SRS_CURVE_BN256) to_affine_batch_t<Bn256>(in, out, n); else to_affine_batch_t<Grumpkin>(in, out, n);
}

bool g_device_ok = false;
int g_device = -1;        // the process's device (one process per GPU); bound by the first srs_init
// r06: a host thread may bind ITSELF to another device (srs_init_thread): one process, one thread per GPU, every thread with its own
// handles -- sharded keys and structures (srs_ck_create_sharded, srs_structure_set_shard) then put commits, leaves, cross terms and folds
// on every device of a node from a single process; the partial sums are added by the caller (no collective needed inside a process).
static thread_local int t_device = -1;
int home_device() { return t_device >= 0 ? t_device : (g_device < 0 ? 0 : g_device); }

// HIP's current device is per host thread: a caller's worker thread starts on device 0.  Every entry point rebinds the
// calling thread to its device (srs_init_thread) or the process's (srs_init), so that handles created on one thread work from any
// other thread bound to the same device.
int ensure_device() {
    if (!g_device_ok && t_device < 0) return srs_init(-1);
    const int want = t_device >= 0 ? t_device : g_device;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != want) {
        if (hipSetDevice(want) != hipSuccess) return fail(SRS_ERR_DEVICE, "hipSetDevice failed");
    }
    return SRS_OK;
}

// ---- multi-device keys --------------------------------------------------------------------------------------------
int physical_device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    return n;
}

// copies the block-cyclic stripes of shard `d` (stripe s with s % world == d) of src[0 .. n) compactly to dst (a buffer on
// the shard's device); src is host memory or memory of the process's device.  One strided copy for the whole stripes
// plus one for the tail.  Returns the number of elements copied (= shard_count(n, d, world)).
size_t copy_stripes(fe_t *dst, const fe_t *src, size_t n, uint32_t d, uint32_t world, bool src_is_device, hipStream_t st) {
    const size_t S = (size_t)1 << msm::STRIPE_LOG;
    const hipMemcpyKind kind = src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (world == 1) {
        if (n) SRS_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(fe_t), kind, st));
        return n;
    }
    const size_t full = n >> msm::STRIPE_LOG, rem = n & (S - 1);
    const size_t mine = full / world + (d < full % world ? 1 : 0);          // whole stripes of this shard
    if (mine)
        SRS_HIP_CHECK(hipMemcpy2DAsync(dst, S * sizeof(fe_t), src + (size_t)d * S, (size_t)world * S * sizeof(fe_t),
                                       S * sizeof(fe_t), mine, kind, st));
    size_t cnt = mine * S;
    if (rem && d == full % world) {
        SRS_HIP_CHECK(hipMemcpyAsync(dst + cnt, src + full * S, rem * sizeof(fe_t), kind, st));
        cnt += rem;
    }
    return cnt;
}

template <class C>
void sum_partials_t(const std::vector<std::vector<xyzz_t>> &parts, size_t batch, xyzz_t *out) {
    for (size_t m = 0; m < batch; ++m) {
        xyzz_t a = Ec<C>::identity();
        for (const auto &p : parts) a = Ec<C>::add(a, p[m]);
        out[m] = a;
    }
}

// commit through a multi-device key: every shard thread stages ITS stripes of every vector on its device (from the caller's
// host buffer, or with a peer copy from the process's device) and runs its partial MSMs; the partial sums are added here.
// ready: event recorded on the caller's stream after the producer of device-resident scalars (or nullptr).
int multi_commit(srs_ck *ck, const srs_fe *const *scalars, const size_t *n, size_t batch, int space, int repr, hipEvent_t ready,
                 xyzz_t *res) {
    const uint32_t world = (uint32_t)ck->shards.size();
    std::vector<std::vector<xyzz_t>> parts(world, std::vector<xyzz_t>(batch));
    for (uint32_t d = 0; d < world; ++d) {
        CkShard *sh = ck->shards[d].get();
        sh->post([=, &parts]() {
            size_t total = 256;
            std::vector<uint32_t> nloc(batch);
            for (size_t m = 0; m < batch; ++m) {
                nloc[m] = (uint32_t)shard_count(n[m], d, world);
                total += Arena::pad(((size_t)nloc[m] + 1) * sizeof(fe_t));
            }
            sh->staging.reserve(total);
            sh->staging.reset();
            if (ready) SRS_HIP_CHECK(hipStreamWaitEvent(sh->stream, ready, 0));
            std::vector<const fe_t *> dptr(batch);
            for (size_t m = 0; m < batch; ++m) {
                fe_t *dst = sh->staging.take<fe_t>((size_t)nloc[m] + 1);
                copy_stripes(dst, reinterpret_cast<const fe_t *>(scalars[m]), n[m], d, world, space == SRS_SPACE_DEVICE, sh->stream);
                dptr[m] = dst;
            }
            msm::run(sh->key, dptr.data(), nloc.data(), (uint32_t)batch, repr == SRS_REPR_MONT, sh->stream, parts[d].data());
        });
    }
    int rc = SRS_OK;
    std::string err;
    for (uint32_t d = 0; d < world; ++d) {
        int r = ck->shards[d]->wait();
        if (r && !rc) { rc = r; err = ck->shards[d]->err; }
    }
    if (rc) return fail(rc, "multi-device commit: " + err);
    if (ck->key.curve == SRS_CURVE_BN256) sum_partials_t<Bn256>(parts, batch, res); else sum_partials_t<Grumpkin>(parts, batch, res);
    return SRS_OK;
}

void free_shards(srs_ck *ck) {
    for (auto &sp : ck->shards) {
        CkShard *sh = sp.get();
        if (sh->worker.joinable()) {
            {
                std::lock_guard<std::mutex> lk(sh->mu);
                sh->quit = true;
            }
            sh->cv.notify_one();
            sh->worker.join();
        }
        (void)hipSetDevice(sh->device);
        if (sh->key.table) (void)hipFree(sh->key.table);
        msm::release(sh->key);
        sh->staging.release();
        if (sh->full) (void)hipFree(sh->full);
        for (hipEvent_t e : sh->events) (void)hipEventDestroy(e);
        if (sh->peer_done) (void)hipEventDestroy(sh->peer_done);
        if (sh->copy_stream) (void)hipStreamDestroy(sh->copy_stream);
        if (sh->peer_stream) (void)hipStreamDestroy(sh->peer_stream);
        if (sh->stream) (void)hipStreamDestroy(sh->stream);
    }
    ck->shards.clear();
}

// builds the shards of a multi-device key: `fill(shard)` must leave table[0 .. key.len) (window 0) on the shard's device
template <class Fill>
int create_multi(int curve, size_t len, int n_devices, Fill fill, srs_ck **out) {
    const int phys = physical_device_count();
    if (phys <= 0) return fail(SRS_ERR_DEVICE, "no HIP device visible: libsirius_amd has no CPU path");
    const uint32_t world = n_devices > 0 ? (uint32_t)n_devices : (uint32_t)phys;
    if (world > 64) return fail(SRS_ERR_INVALID, "srs_ck_create_multi: more than 64 shards");
    const int home = home_device();
    srs_ck *ck = new srs_ck();
    ck->key.curve = curve;
    ck->key.global_len = len;
    ck->key.len = 0;
    try {
        for (uint32_t d = 0; d < world; ++d) {
            // the shard joins ck->shards BEFORE anything is allocated for it: a failure below (hipMalloc of a large table, the
            // fill, build_table) then reaches free_shards through srs_ck_free and releases its table, stream and arena
            ck->shards.emplace_back(new CkShard());
            CkShard *sh = ck->shards.back().get();
            sh->device = (home + (int)d) % phys;               // shard 0 on the process's device
            sh->key.curve = curve;
            sh->key.global_len = len;
            sh->key.rank = d;
            sh->key.world = world;
            sh->key.len = shard_count(len, d, world);
            sh->key.compact_scalars = true;                    // the shard is handed ITS scalars, already gathered
            SRS_HIP_CHECK(hipSetDevice(sh->device));
            if (sh->device != home) {        // stripes of device-resident scalars come, and streamed stripes leave, by peer copies: they need the access
                int can = 0;
                SRS_HIP_CHECK(hipDeviceCanAccessPeer(&can, sh->device, home));
                hipError_t pe = can ? hipDeviceEnablePeerAccess(home, 0) : hipErrorInvalidDevice;
                if (pe == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); pe = hipSuccess; }
                if (pe != hipSuccess) {
                    (void)hipGetLastError();
                    set_error("srs_ck_create_multi: device " + std::to_string(sh->device) + " has no peer access to device " + std::to_string(home) +
                              " (multi-device keys copy stripes between their devices: xGMI or PCIe P2P is required)");
                    throw DeviceError{SRS_ERR_DEVICE};
                }
            }
            SRS_HIP_CHECK(hipStreamCreateWithFlags(&sh->stream, hipStreamNonBlocking));
            if (sh->key.len) {
                SRS_HIP_CHECK(hipMalloc((void **)&sh->key.table, sh->key.len * msm::NWIN * sizeof(affine_t)));
                fill(*sh);
                msm::build_table(sh->key, sh->stream);
            }
        }
        SRS_HIP_CHECK(hipSetDevice(home));
        if constexpr (!rt::kEmulated) {
            for (auto &sp : ck->shards) {
                CkShard *sh = sp.get();
                sh->worker = std::thread([sh] { sh->loop(); });
            }
        }
    } catch (...) {
        (void)hipSetDevice(home);
        srs_ck_free(ck);
        throw;
    }
    *out = ck;
    return SRS_OK;
}

}  // namespace

extern "C" {

const char *srs_last_error(void) { return get_error(); }
const char *srs_version(void) { return "sirius_amd 0.1.0 (gfx950)"; }

int srs_tuning_set(const char *name, int64_t value) {
    if (tuning::set(name, value) != 0) return fail(SRS_ERR_INVALID, std::string("srs_tuning_set: unknown tunable '") + (name ? name : "(null)") + "'");
    return SRS_OK;
}
int srs_tuning_get(const char *name, int64_t *value) {
    int found = 0;
    const int64_t v = tuning::get_by_name(name, &found);
    if (!found || !value) return fail(SRS_ERR_INVALID, "srs_tuning_get: unknown tunable or null output");
    *value = v;
    return SRS_OK;
}
void srs_tuning_reset(void) { tuning::reset(); }
const char *srs_tuning_name(int index) { return tuning::name_of(index); }

int srs_init(int device_ordinal) {
    return guarded([&]() -> int {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
            return fail(SRS_ERR_DEVICE, "no HIP device visible: libsirius_amd has no CPU path");
        int dev = device_ordinal;
        if (dev < 0) SRS_HIP_CHECK(hipGetDevice(&dev));
        if (dev >= count) return fail(SRS_ERR_INVALID, "device ordinal out of range");
        SRS_HIP_CHECK(hipSetDevice(dev));
        std::string arch;
        if (!rt::device_arch(dev, arch)) return fail(SRS_ERR_DEVICE, "hipGetDeviceProperties failed");
        if (arch.compare(0, 6, "gfx950") != 0)
            return fail(SRS_ERR_DEVICE, "device is " + arch + ", kernels are built for gfx950 only");
        g_device = dev;
        g_device_ok = true;
        return SRS_OK;
    });
}

int srs_init_thread(int device_ordinal) {
    return guarded([&]() -> int {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
            return fail(SRS_ERR_DEVICE, "no HIP device visible: libsirius_amd has no CPU path");
        if (device_ordinal < 0) {            // unbind: back to the process's device
            t_device = -1;
            return SRS_OK;
        }
        if (device_ordinal >= count) return fail(SRS_ERR_INVALID, "device ordinal out of range");
        SRS_HIP_CHECK(hipSetDevice(device_ordinal));
        std::string arch;
        if (!rt::device_arch(device_ordinal, arch)) return fail(SRS_ERR_DEVICE, "hipGetDeviceProperties failed");
        if (arch.compare(0, 6, "gfx950") != 0)
            return fail(SRS_ERR_DEVICE, "device is " + arch + ", kernels are built for gfx950 only");
        t_device = device_ordinal;
        return SRS_OK;
    });
}

int srs_scalar_field_of(int curve) {
    if (!valid_curve(curve)) return -1;
    return curve == SRS_CURVE_BN256 ? SRS_FIELD_FR : SRS_FIELD_FQ;
}

int srs_layout_selftest(int field, const srs_fe *one, const srs_fe *two) {
    if (!valid_field(field) || !one || !two) return fail(SRS_ERR_INVALID, "srs_layout_selftest: bad argument");
    fe_t o, t, eo, et;
    std::memcpy(&o, one, 32);
    std::memcpy(&t, two, 32);
    if (field == SRS_FIELD_FR) { eo = Fr::one(); et = Fr::dbl(eo); } else { eo = Fq::one(); et = Fq::dbl(eo); }
    if (std::memcmp(&o, &eo, 32) != 0 || std::memcmp(&t, &et, 32) != 0)
        return fail(SRS_ERR_LAYOUT, "field elements are not 4x64 little-endian Montgomery (R = 2^256)");
    return SRS_OK;
}

int srs_layout_selftest_point(int curve, const srs_affine *generator) {
    if (!valid_curve(curve) || !generator) return fail(SRS_ERR_INVALID, "srs_layout_selftest_point: bad argument");
    affine_t g, e;
    std::memcpy(&g, generator, 64);
    if (curve == SRS_CURVE_BN256) {
        e.x = Fq::one();
        e.y = Fq::dbl(Fq::one());                                   // bn256 G1 generator (1, 2)
    } else {
        // grumpkin generator (1, sqrt(-16)), y = 17631683881184975370165255887551781615748388533673675138860
        fe_t y;
        const uint32_t yc[8] = {0x823f272cu, 0x833fc48du, 0xf1181294u, 0x2d270d45u, 0x06a45d63u, 0xcf135e75u, 0x00000002u, 0u};
        for (int j = 0; j < 8; ++j) y.v[j] = yc[j];
        e.x = Fr::one();
        e.y = Fr::to_mont(y);
    }
    if (std::memcmp(&g, &e, 64) != 0)
        return fail(SRS_ERR_LAYOUT, "the generator is not x || y in 4x64 little-endian Montgomery form (R = 2^256)");
    return SRS_OK;
}

int srs_dev_alloc(size_t bytes, void **out) {
    if (!out) return fail(SRS_ERR_INVALID, "srs_dev_alloc: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        void *p = nullptr;
        SRS_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1));
        *out = p;
        return SRS_OK;
    });
}
void srs_dev_free(void *p) {
    if (p && ensure_device() == SRS_OK) (void)hipFree(p);
}
int srs_host_alloc(size_t bytes, void **out) {
    if (!out) return fail(SRS_ERR_INVALID, "srs_host_alloc: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        void *p = nullptr;
        SRS_HIP_CHECK(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
        *out = p;
        return SRS_OK;
    });
}
void srs_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}
int srs_upload(void *dst_dev, const void *src_host, size_t bytes, void *stream) {
    if (bytes && (!dst_dev || !src_host)) return fail(SRS_ERR_INVALID, "srs_upload: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        if (bytes) SRS_HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
        return SRS_OK;
    });
}
int srs_download(void *dst_host, const void *src_dev, size_t bytes, void *stream) {
    if (bytes && (!dst_host || !src_dev)) return fail(SRS_ERR_INVALID, "srs_download: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        if (bytes) SRS_HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
        SRS_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        return SRS_OK;
    });
}

int srs_ck_create_sharded(int curve, const srs_affine *bases, size_t len, int space, uint32_t rank,
                          uint32_t world, srs_ck **out) {
    if (!valid_curve(curve) || !out || (!bases && len) || world == 0 || rank >= world)
        return fail(SRS_ERR_INVALID, "srs_ck_create: bad argument");
    if (len > ((size_t)1 << 27)) return fail(SRS_ERR_INVALID, "srs_ck_create: key longer than 2^27 bases");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        srs_ck *ck = new srs_ck();
        ck->key.curve = curve;
        ck->key.global_len = len;
        ck->key.rank = rank;
        ck->key.world = world;
        ck->key.len = shard_count(len, rank, world);
        const size_t n = ck->key.len;
        try {
            if (n) {
                SRS_HIP_CHECK(hipMalloc((void **)&ck->key.table, n * msm::NWIN * sizeof(affine_t)));
                const hipMemcpyKind kind = space == SRS_SPACE_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
                if (world == 1) {
                    SRS_HIP_CHECK(hipMemcpy(ck->key.table, bases, n * sizeof(affine_t), kind));
                } else {
                    const size_t S = (size_t)1 << msm::STRIPE_LOG;
                    size_t local = 0;
                    for (size_t s = rank; s * S < len; s += world) {
                        size_t cnt = std::min(S, len - s * S);
                        SRS_HIP_CHECK(hipMemcpy(ck->key.table + local, bases + s * S, cnt * sizeof(affine_t), kind));
                        local += cnt;
                    }
                }
                msm::build_table(ck->key, nullptr);
            }
        } catch (...) {
            srs_ck_free(ck);
            throw;
        }
        *out = ck;
        return SRS_OK;
    });
}

int srs_ck_setup_synthetic(int curve, size_t len, uint64_t seed, uint32_t rank, uint32_t world, srs_ck **out) {
    if (!valid_curve(curve) || !out || world == 0 || rank >= world) return fail(SRS_ERR_INVALID, "srs_ck_setup_synthetic: bad argument");
    if (len > ((size_t)1 << 27)) return fail(SRS_ERR_INVALID, "srs_ck_setup_synthetic: key longer than 2^27 bases");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        srs_ck *ck = new srs_ck();
        ck->key.curve = curve;
        ck->key.global_len = len;
        ck->key.rank = rank;
        ck->key.world = world;
        ck->key.len = shard_count(len, rank, world);
        try {
            if (ck->key.len) {
                SRS_HIP_CHECK(hipMalloc((void **)&ck->key.table, ck->key.len * msm::NWIN * sizeof(affine_t)));
                msm::generate_bases(ck->key, seed, nullptr);
                msm::build_table(ck->key, nullptr);
            }
        } catch (...) {
            srs_ck_free(ck);
            throw;
        }
        *out = ck;
        return SRS_OK;
    });
}

static int read_shard_bases(const msm::Key &key, hipStream_t st, srs_affine *out) {       // key.len points, this shard's window 0
    if (!key.len) return SRS_OK;
    affine_t *tmp = nullptr;
    SRS_HIP_CHECK(hipMalloc((void **)&tmp, key.len * sizeof(affine_t)));
    try {
        msm::read_bases(key, tmp, st);
        SRS_HIP_CHECK(hipMemcpyAsync(out, tmp, key.len * sizeof(affine_t), hipMemcpyDeviceToHost, st));
        SRS_HIP_CHECK(hipStreamSynchronize(st));
    } catch (...) {
        (void)hipFree(tmp);
        throw;
    }
    (void)hipFree(tmp);
    return SRS_OK;
}

int srs_ck_get_bases(const srs_ck *ck, srs_affine *out) {
    if (!ck || ((ck->key.len || !ck->shards.empty()) && !out)) return fail(SRS_ERR_INVALID, "srs_ck_get_bases: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    if (!ck->shards.empty()) {           // multi-device key: the whole key, stripes gathered from the shards
        return guarded([&]() -> int {
            const size_t S = (size_t)1 << msm::STRIPE_LOG, len = ck->key.global_len;
            const uint32_t world = (uint32_t)ck->shards.size();
            for (uint32_t d = 0; d < world; ++d) {
                CkShard *sh = ck->shards[d].get();
                std::vector<srs_affine> loc(sh->key.len);
                SRS_HIP_CHECK(hipSetDevice(sh->device));
                read_shard_bases(sh->key, sh->stream, loc.data());
                size_t at = 0;
                for (size_t st = d; st * S < len; st += world) {
                    size_t cnt = std::min(S, len - st * S);
                    std::memcpy(out + st * S, loc.data() + at, cnt * sizeof(srs_affine));
                    at += cnt;
                }
            }
            return ensure_device();
        });
    }
    return guarded([&]() -> int {
        if (ck->key.len) {       // the table lives in the multiplier's internal Montgomery form: convert window 0 back
            affine_t *tmp = nullptr;
            SRS_HIP_CHECK(hipMalloc((void **)&tmp, ck->key.len * sizeof(affine_t)));
            try {
                msm::read_bases(ck->key, tmp, nullptr);
                SRS_HIP_CHECK(hipMemcpy(out, tmp, ck->key.len * sizeof(affine_t), hipMemcpyDeviceToHost));
            } catch (...) {
                (void)hipFree(tmp);
                throw;
            }
            (void)hipFree(tmp);
        }
        return SRS_OK;
    });
}
size_t srs_ck_local_len(const srs_ck *ck) { return ck ? (ck->shards.empty() ? ck->key.len : ck->key.global_len) : 0; }

int srs_ck_load_file(int curve, const char *path, size_t k, uint32_t rank, uint32_t world, srs_ck **out) {
    if (!valid_curve(curve) || !path || !out) return fail(SRS_ERR_INVALID, "srs_ck_load_file: bad argument");
    if (k > 27) return fail(SRS_ERR_INVALID, "srs_ck_load_file: key longer than 2^27 bases");      // srs_ck_create's limit, before any allocation
    const size_t len = (size_t)1 << k;
    std::vector<srs_affine> host;
    int rc = guarded([&]() -> int {
        host.resize(len);
        FILE *f = std::fopen(path, "rb");
        if (!f) return fail(SRS_ERR_IO, std::string("srs_ck_load_file: cannot open ") + path);
        size_t got = std::fread(host.data(), sizeof(srs_affine), len, f);
        std::fclose(f);
        if (got != len) return fail(SRS_ERR_IO, "srs_ck_load_file: failed to fill whole buffer");      // read_exact
        return SRS_OK;
    });
    if (rc) return rc;
    srs_ck *ck = nullptr;
    rc = srs_ck_create_sharded(curve, host.data(), len, SRS_SPACE_HOST, rank, world, &ck);
    if (rc) return rc;
    size_t bad = 0;
    rc = srs_ck_count_off_curve(ck, &bad);
    if (rc == SRS_OK && bad) rc = fail(SRS_ERR_INVALID_DATA, "Wrong file in cache, some ptr out of curve");
    if (rc) { srs_ck_free(ck); return rc; }
    *out = ck;
    return SRS_OK;
}

int srs_ck_save_file(const srs_ck *ck, const char *path) {
    if (!ck || !path) return fail(SRS_ERR_INVALID, "srs_ck_save_file: bad argument");
    if (ck->key.world != 1) return fail(SRS_ERR_INVALID, "srs_ck_save_file: sharded key");
    std::vector<srs_affine> host(srs_ck_local_len(ck));
    int rc = srs_ck_get_bases(ck, host.data());
    if (rc) return rc;
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(SRS_ERR_IO, std::string("srs_ck_save_file: cannot create ") + path);
    size_t put = std::fwrite(host.data(), sizeof(srs_affine), host.size(), f);
    int cl = std::fclose(f);
    if (put != host.size() || cl != 0) return fail(SRS_ERR_IO, "srs_ck_save_file: short write");
    return SRS_OK;
}

int srs_ck_count_off_curve(const srs_ck *ck, size_t *bad) {
    if (!ck || !bad) return fail(SRS_ERR_INVALID, "srs_ck_count_off_curve: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        if (ck->shards.empty()) {
            *bad = msm::count_off_curve(ck->key, nullptr);
            return SRS_OK;
        }
        size_t total = 0;
        for (auto &sp : ck->shards) {
            SRS_HIP_CHECK(hipSetDevice(sp->device));
            total += msm::count_off_curve(sp->key, sp->stream);
        }
        *bad = total;
        return ensure_device();
    });
}

int srs_ck_create(int curve, const srs_affine *bases, size_t len, int space, srs_ck **out) {
    return srs_ck_create_sharded(curve, bases, len, space, 0, 1, out);
}

int srs_ck_create_multi(int curve, const srs_affine *bases, size_t len, int space, int n_devices, srs_ck **out) {
    if (!valid_curve(curve) || !out || (!bases && len) || n_devices < 0) return fail(SRS_ERR_INVALID, "srs_ck_create_multi: bad argument");
    if (len > ((size_t)1 << 27)) return fail(SRS_ERR_INVALID, "srs_ck_create_multi: key longer than 2^27 bases");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        return create_multi(curve, len, n_devices, [&](CkShard &sh) {
            const size_t S = (size_t)1 << msm::STRIPE_LOG;
            const hipMemcpyKind kind = space == SRS_SPACE_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
            size_t local = 0;
            for (size_t st = sh.key.rank; st * S < len; st += sh.key.world) {
                size_t cnt = std::min(S, len - st * S);
                SRS_HIP_CHECK(hipMemcpyAsync(sh.key.table + local, bases + st * S, cnt * sizeof(affine_t), kind, sh.stream));
                local += cnt;
            }
        }, out);
    });
}
int srs_ck_setup_synthetic_multi(int curve, size_t len, uint64_t seed, int n_devices, srs_ck **out) {
    if (!valid_curve(curve) || !out || n_devices < 0) return fail(SRS_ERR_INVALID, "srs_ck_setup_synthetic_multi: bad argument");
    if (len > ((size_t)1 << 27)) return fail(SRS_ERR_INVALID, "srs_ck_setup_synthetic_multi: key longer than 2^27 bases");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        return create_multi(curve, len, n_devices, [&](CkShard &sh) { msm::generate_bases(sh.key, seed, sh.stream); }, out);
    });
}
int srs_ck_num_shards(const srs_ck *ck) { return ck ? (ck->shards.empty() ? 1 : (int)ck->shards.size()) : 0; }
int srs_ck_msm_stats(const srs_ck *ck, uint64_t *out) {
    if (!ck || !out) return fail(SRS_ERR_INVALID, "srs_ck_msm_stats: bad argument");
    for (int i = 0; i < 4; ++i) out[i] = 0;
    auto add = [&](const msm::Key &k) {
        out[0] += k.stat_slot_sets;
        out[1] += k.stat_hot_sets;
        out[2] += k.stat_redo;
        out[3] += k.stat_other_sets;
    };
    if (ck->shards.empty()) add(ck->key);
    for (const auto &sh : ck->shards) add(sh->key);
    return SRS_OK;
}

int srs_ck_shard_stats(const srs_ck *ck, int shard, uint64_t *out) {
    if (!ck || !out || shard < 0) return fail(SRS_ERR_INVALID, "srs_ck_shard_stats: bad argument");
    for (int i = 0; i < 4; ++i) out[i] = 0;
    if (ck->shards.empty()) {
        if (shard != 0) return fail(SRS_ERR_INVALID, "srs_ck_shard_stats: no such shard");
        out[3] = (uint64_t)home_device();
        return SRS_OK;
    }
    if ((size_t)shard >= ck->shards.size()) return fail(SRS_ERR_INVALID, "srs_ck_shard_stats: no such shard");
    const CkShard *sh = ck->shards[shard].get();
    out[0] = sh->stat_h2d_bytes;
    out[1] = sh->stat_peer_bytes;
    out[2] = sh->stat_streamed;
    out[3] = (uint64_t)sh->device;
    return SRS_OK;
}

int srs_ck_has_wide_table(const srs_ck *ck) {
    if (!ck) return 0;
    if (ck->shards.empty()) return ck->key.table_w != nullptr ? 1 : 0;
    for (const auto &sh : ck->shards)
        if (sh->key.len && !sh->key.table_w) return 0;
    return 1;
}

void srs_ck_free(srs_ck *ck) {
    if (!ck) return;
    if (!ck->shards.empty()) {
        free_shards(ck);
        (void)ensure_device();
    }
    for (hipEvent_t e : ck->events) (void)hipEventDestroy(e);
    if (ck->copy_stream) (void)hipStreamDestroy(ck->copy_stream);
    if (ck->key.table) (void)hipFree(ck->key.table);
    msm::release(ck->key);
    ck->staging.release();
    delete ck;
}

size_t srs_ck_len(const srs_ck *ck) { return ck ? ck->key.global_len : 0; }

int srs_commit_batch(srs_ck *ck, const srs_fe *const *scalars, const size_t *n, size_t batch, int space,
                     int repr, void *stream, srs_affine *out) {
    if (!ck || !out || (batch && (!scalars || !n))) return fail(SRS_ERR_INVALID, "srs_commit: bad argument");
    if (batch == 0) return SRS_OK;
    std::lock_guard<std::recursive_mutex> key_lock(ck->mu);
    for (size_t m = 0; m < batch; ++m) {
        if (n[m] > ck->key.global_len)
            return fail(SRS_ERR_TOO_LONG_INPUT, "Can't commit too long input: input len: " + std::to_string(n[m]) +
                                                    ", but limit is " + std::to_string(ck->key.global_len));
        if (n[m] && !scalars[m]) return fail(SRS_ERR_INVALID, "srs_commit: null scalar vector");
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (!ck->shards.empty()) {
        return guarded([&]() -> int {
            hipStream_t st = (hipStream_t)stream;
            hipEvent_t ready = nullptr;
            if (space == SRS_SPACE_DEVICE) {           // the shards' streams wait for whatever produced the scalars on `stream`
                if (ck->events.empty()) { hipEvent_t e; SRS_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ck->events.push_back(e); }
                ready = ck->events[0];
                SRS_HIP_CHECK(hipEventRecord(ready, st));
            }
            std::vector<xyzz_t> res(batch);
            int mrc = multi_commit(ck, scalars, n, batch, space, repr, ready, res.data());
            if (mrc) return mrc;
            std::vector<affine_t> aff(batch);
            to_affine_batch(ck->key.curve, res.data(), aff.data(), batch);
            std::memcpy(out, aff.data(), batch * sizeof(affine_t));
            return SRS_OK;
        });
    }
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        const uint32_t world = ck->key.world, rank = ck->key.rank;
        std::vector<uint32_t> nloc(batch);
        std::vector<const fe_t *> dptr(batch);
        for (size_t m = 0; m < batch; ++m) nloc[m] = (uint32_t)shard_count(n[m], rank, world);
        if (space == SRS_SPACE_DEVICE) {
            for (size_t m = 0; m < batch; ++m) dptr[m] = reinterpret_cast<const fe_t *>(scalars[m]);
        } else {
            // stage the FULL vectors (the kernels pick this rank's stripes)
            size_t total = 0;
            for (size_t m = 0; m < batch; ++m) total += Arena::pad(n[m] * sizeof(fe_t));
            ck->staging.reserve(total + 256);
            ck->staging.reset();
            for (size_t m = 0; m < batch; ++m) {
                fe_t *d = ck->staging.take<fe_t>(n[m] ? n[m] : 1);
                if (n[m]) SRS_HIP_CHECK(hipMemcpyAsync(d, scalars[m], n[m] * sizeof(fe_t), hipMemcpyHostToDevice, st));
                dptr[m] = d;
            }
        }
        std::vector<xyzz_t> res(batch);
        msm::run(ck->key, dptr.data(), nloc.data(), (uint32_t)batch, repr == SRS_REPR_MONT, st, res.data());
        std::vector<affine_t> aff(batch);
        to_affine_batch(ck->key.curve, res.data(), aff.data(), batch);
        std::memcpy(out, aff.data(), batch * sizeof(affine_t));
        return SRS_OK;
    });
}

int srs_commit(srs_ck *ck, const srs_fe *scalars, size_t n, int space, int repr, void *stream, srs_affine *out) {
    const srs_fe *v[1] = {scalars};
    size_t nn[1] = {n};
    return srs_commit_batch(ck, v, nn, 1, space, repr, stream, out);
}

namespace {
// a piece of the vector being committed: `len` elements from host memory land at element offset `off`; whatever lies between
// pieces is zero (the padding of util::concatenate_with_padding)
struct Seg {
    const fe_t *src;
    size_t off, len;
};

// true when `p` is ordinary (not page-locked, not registered) host memory
static bool host_is_pageable(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();                    // unregistered host memory is reported as an error: that IS the pageable case
        return true;
    }
    return attr.type == hipMemoryTypeUnregistered;
}

// uploads dst[a, b) = pieces + zero padding, on stream `cs`
void upload_range(fe_t *dst, const std::vector<Seg> &segs, size_t a, size_t b, hipStream_t cs) {
    size_t at = a;
    for (const Seg &sg : segs) {
        const size_t lo = std::max(a, sg.off), hi = std::min(b, sg.off + sg.len);
        if (lo >= hi) continue;
        if (lo > at) SRS_HIP_CHECK(hipMemsetAsync(dst + at, 0, (lo - at) * sizeof(fe_t), cs));
        SRS_HIP_CHECK(hipMemcpyAsync(dst + lo, sg.src + (lo - sg.off), (hi - lo) * sizeof(fe_t), hipMemcpyHostToDevice, cs));
        at = hi;
    }
    if (b > at) SRS_HIP_CHECK(hipMemsetAsync(dst + at, 0, (b - at) * sizeof(fe_t), cs));
}

// The same for ONE RANK of a block-cyclic sharding: uploads rank R's stripes (2^STRIPE_LOG elements; stripe s belongs to rank s % W) of the global
// range [a, b) of the concatenation -- pieces + zero padding up to the next piece / n -- to their global positions in dst.  Per piece and kind
// (data / padding) the whole stripes are ONE strided copy (or strided memset), partial stripes at the ends of a piece one small operation
// each: a dozen operations per chunk whatever its size.  Pieces may start anywhere (stripes are global).  -> data elements copied.
size_t upload_stripes(fe_t *dst, const std::vector<Seg> &segs, size_t n, size_t a, size_t b, uint32_t R, uint32_t W, hipStream_t cs) {
    const size_t SL = (size_t)1 << msm::STRIPE_LOG, pitch = (size_t)W * SL * sizeof(fe_t);
    size_t copied = 0;
    auto over = [&](size_t lo, size_t hi, const fe_t *src /* element at global offset lo, or nullptr: zeros */) {
        lo = std::max(lo, a);
        const size_t lo0 = lo;
        hi = std::min(hi, b);
        if (lo >= hi) return;
        auto op1 = [&](size_t g, size_t len) {
            if (src) { SRS_HIP_CHECK(hipMemcpyAsync(dst + g, src + (g - lo0), len * sizeof(fe_t), hipMemcpyHostToDevice, cs)); copied += len; }
            else SRS_HIP_CHECK(hipMemsetAsync(dst + g, 0, len * sizeof(fe_t), cs));
        };
        size_t s_lo = lo / SL;
        const size_t s_last = (hi - 1) / SL;
        if (s_lo == s_last) {                              // inside one stripe
            if (s_lo % W == R) op1(lo, hi - lo);
            return;
        }
        if (lo % SL) {                                     // partial first stripe
            if (s_lo % W == R) op1(lo, (s_lo + 1) * SL - lo);
            ++s_lo;
        }
        const size_t full_end = hi / SL;                   // stripes [s_lo, full_end) are whole
        if (hi % SL && full_end % W == R) op1(full_end * SL, hi - full_end * SL);          // partial last stripe
        if (s_lo < full_end) {
            const size_t s0 = s_lo + ((R + W - s_lo % W) % W);
            if (s0 < full_end) {
                const size_t rows = (full_end - s0 + W - 1) / W, g = s0 * SL;
                if (src) {
                    SRS_HIP_CHECK(hipMemcpy2DAsync(dst + g, pitch, src + (g - lo0), pitch, SL * sizeof(fe_t), rows, hipMemcpyHostToDevice, cs));
                    copied += rows * SL;
                } else {
                    SRS_HIP_CHECK(hipMemset2DAsync(dst + g, pitch, 0, SL * sizeof(fe_t), rows, cs));
                }
            }
        }
    };
    size_t at = 0;                                         // everything before the first piece is padding too
    for (size_t k = 0; k < segs.size(); ++k) {
        const Seg &sg = segs[k];
        if (sg.off > at) over(at, sg.off, nullptr);
        // (the data pointer handed to `over` must point at the element of global offset max(sg.off, a))
        const size_t d_lo = std::max(sg.off, a);
        if (sg.len && d_lo < sg.off + sg.len) over(sg.off, sg.off + sg.len, sg.src + (d_lo - sg.off));
        at = std::max(at, sg.off + sg.len);
    }
    if (n > at) over(at, n, nullptr);
    return copied;
}

// the streamed commit of srs_commit_upload / srs_commit_upload_columns (single-device key): chunk j goes up on the key's copy
// stream while the MSM of chunk j - 1 runs on the caller's stream; `cuts` = chunk boundaries (element offsets, first 0, last n)
// A key sharded over processes (world > 1; chunk boundaries multiples of world * 2^10): every chunk brings up and accumulates only THIS rank's
// block-cyclic stripes of its range -- the same overlap of upload and MSM, 1 / world of both per rank (r05: from any list of pieces, so the
// column form streams too: upload_stripes).
// The stream resources of whoever runs the commit -- a single-device key handle, or one shard of a multi-device key -- and, for a
// shard, where its stripes are forwarded once they are in its HBM (the device copy the caller asked for lives on the process's device).
struct StreamRes {
    msm::Key &key;
    hipStream_t &copy_stream;
    std::vector<hipEvent_t> &events;
    uint64_t *h2d_bytes = nullptr;        // statistics: bytes this commit brought up from host memory
    // peer forwarding (multi-device keys): every uploaded chunk's stripes are copied from `dst` (this device) to the same positions of
    // `peer_dst` (the process's device) on `peer_stream`, behind the chunk's upload
    fe_t *peer_dst = nullptr;
    hipStream_t peer_stream = nullptr;
    uint64_t *peer_bytes = nullptr;
};

// -> the commitment as an XYZZ point (the caller normalises: a multi-device key adds its shards' partial sums first)
int commit_streamed_core(StreamRes ck_, const std::vector<Seg> &segs, size_t n, const std::vector<size_t> &cut, fe_t *dst, int repr,
                         hipStream_t st, xyzz_t *sum_out) {
    StreamRes *ck = &ck_;
    HostTrace ht("commit_streamed");
    const size_t chunks = cut.size() - 1;
    const uint32_t W = ck->key.world, R = ck->key.rank;
    const size_t SL = (size_t)1 << msm::STRIPE_LOG;
    auto local = [&](size_t a, size_t b) { return shard_count(b, R, W) - shard_count(a, R, W); };     // this rank's elements of [a, b)
    size_t per = 0;
    for (size_t j = 0; j < chunks; ++j) per = std::max(per, local(cut[j], cut[j + 1]));
    if (!ck->copy_stream) SRS_HIP_CHECK(hipStreamCreateWithFlags(&ck->copy_stream, hipStreamNonBlocking));
    while (ck->events.size() < chunks + 1) {
        hipEvent_t e;
        SRS_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ck->events.push_back(e);
    }
    // the copy stream must not overwrite dst while earlier work on the caller's stream still reads it
    SRS_HIP_CHECK(hipEventRecord(ck->events[chunks], st));
    SRS_HIP_CHECK(hipStreamWaitEvent(ck->copy_stream, ck->events[chunks], 0));
    std::vector<bool> launched(chunks, false);
    // every chunk on the 16-bit windows: the chunks fold their buckets into one running set and only the last one is reduced
    bool fold = chunks > 1;
    for (size_t j = 0; j < chunks; ++j) fold = fold && local(cut[j], cut[j + 1]) > 0 && msm::may_fold(ck->key, (uint32_t)local(cut[j], cut[j + 1]));
    msm::reserve(ck->key, (uint32_t)per, 1);
    auto upload = [&](size_t j) {
        if (W == 1) {
            upload_range(dst, segs, cut[j], cut[j + 1], ck->copy_stream);
        } else {                             // the rank's stripes of [cut_j, cut_j+1): strided copies per piece, strided memsets for the padding
            const size_t up = upload_stripes(dst, segs, n, cut[j], cut[j + 1], R, W, ck->copy_stream);
            if (ck->h2d_bytes) *ck->h2d_bytes += up * sizeof(fe_t);
        }
        SRS_HIP_CHECK(hipEventRecord(ck->events[j], ck->copy_stream));
        if (ck->h2d_bytes && W == 1) *ck->h2d_bytes += (cut[j + 1] - cut[j]) * sizeof(fe_t);
        if (ck->peer_dst && W > 1) {         // the same stripes, from this device's landing buffer to the process's device copy
            SRS_HIP_CHECK(hipStreamWaitEvent(ck->peer_stream, ck->events[j], 0));
            const size_t len = cut[j + 1] - cut[j], full = len / SL;
            const size_t mine = full > R ? (full - R + W - 1) / W : 0;
            const fe_t *d = dst + cut[j];
            fe_t *pd = ck->peer_dst + cut[j];
            if (mine)
                SRS_HIP_CHECK(hipMemcpy2DAsync(pd + R * SL, W * SL * sizeof(fe_t), d + R * SL, W * SL * sizeof(fe_t), SL * sizeof(fe_t), mine,
                                               hipMemcpyDefault, ck->peer_stream));      // (peer access was checked when the key was made)
            if (len % SL && full % W == R)
                SRS_HIP_CHECK(hipMemcpyAsync(pd + full * SL, d + full * SL, (len % SL) * sizeof(fe_t), hipMemcpyDefault, ck->peer_stream));
            if (ck->peer_bytes) *ck->peer_bytes += local(cut[j], cut[j + 1]) * sizeof(fe_t);
        }
    };
    auto launch = [&](size_t j) {
        const fe_t *ptr = dst + cut[j];
        // sharded: nn local scalars of the chunk, read through the stripe map relative to ptr (cut_j is a multiple of world * 2^10);
        // their bases start at local index cut_j / world
        const uint32_t nn = (uint32_t)local(cut[j], cut[j + 1]), base = (uint32_t)(W == 1 ? cut[j] : cut[j] / W);
        if (nn == 0) {                       // a rank without a stripe in this chunk (ragged end)
            SRS_HIP_CHECK(hipStreamWaitEvent(st, ck->events[j], 0));
            return;
        }
        SRS_HIP_CHECK(hipStreamWaitEvent(st, ck->events[j], 0));
        const msm::Fold f = !fold ? msm::FOLD_NONE : (j == 0 ? msm::FOLD_FIRST : (j + 1 == chunks ? msm::FOLD_LAST : msm::FOLD_MIDDLE));
        launched[j] = msm::enqueue(ck->key, &ptr, &nn, &base, 1, repr == SRS_REPR_MONT, st, (uint32_t)j, f);
    };
    // Page-locked source: the uploads are asynchronous -- upload j + 1 is queued before the launches of MSM j, the copy engine never waits
    // for the host.  PAGEABLE source (a Rust Vec<F>: the runtime stages it through its own pinned buffers and hipMemcpyAsync returns only
    // when the chunk has been staged): the launches of MSM j go first, so that the device accumulates chunk j WHILE the host is held in
    // the copy of chunk j + 1 (r06: with the r05 order MSM j could only be queued after upload j + 1 had returned -- the device ran one
    // chunk behind the link, +1.3 ms on the k = 20 step, bench.py secondary.pageable_witness)
    const bool pageable = !segs.empty() && segs[0].len && host_is_pageable(segs[0].src);
    upload(0);
    for (size_t j = 0; j < chunks; ++j) {
        if (pageable) {
            launch(j);
            if (j + 1 < chunks) upload(j + 1);
        } else {
            if (j + 1 < chunks) upload(j + 1);
            launch(j);
        }
    }
    ht.mark("enqueued");
    SRS_HIP_CHECK(hipStreamSynchronize(st));
    SRS_HIP_CHECK(hipGetLastError());
    ht.mark("device done");
    // slot mode: a set that met hot buckets without its overflow kernels (msm.h: overflow_missed) left an incomplete sum -- the whole
    // vector is in HBM by now, so the commit is simply run again, device-resident, with the prediction switched
    bool missed = false;
    std::vector<uint32_t> used_slots;
    for (size_t j = 0; j < chunks; ++j) {
        if (!launched[j]) continue;
        used_slots.push_back((uint32_t)j);
        missed = missed || msm::overflow_missed(ck->key, (uint32_t)j);
    }
    msm::note_commit(ck->key, used_slots.data(), (uint32_t)used_slots.size(), local(0, n));   // + the scalars: the density the next commit's cuts follow
    xyzz_t redo;
    if (missed) {
        ++ck->key.stat_redo;
        const fe_t *whole = dst;
        const uint32_t n_local = (uint32_t)local(0, n);
        msm::run(ck->key, &whole, &n_local, 1, repr == SRS_REPR_MONT, st, &redo);
        ht.mark("overflow redo");
    }
    auto go = [&](auto tag) {
        using C = decltype(tag);
        xyzz_t acc = Ec<C>::identity(), part;
        for (size_t j = fold ? chunks - 1 : 0; j < chunks && !missed; ++j) {          // folded chunks: the last set holds the whole sum
            msm::finish(ck->key, 1, (uint32_t)j, launched[j], &part);
            acc = Ec<C>::add(acc, part);
        }
        if (missed) acc = redo;
        *sum_out = acc;
    };
    if (ck->key.curve == SRS_CURVE_BN256) go(Bn256{}); else go(Grumpkin{});
    ht.mark("host finish");
    prof::collect();
    ht.mark("prof");
    return SRS_OK;
}

// single-device (or process-sharded) key handle: the commit on the handle's own resources, normalised
int commit_streamed(srs_ck *ck, const std::vector<Seg> &segs, size_t n, const std::vector<size_t> &cut, fe_t *dst, int repr,
                    hipStream_t st, srs_affine *out) {
    xyzz_t sum;
    StreamRes r{ck->key, ck->copy_stream, ck->events};
    int rc = commit_streamed_core(r, segs, n, cut, dst, repr, st, &sum);
    if (rc) return rc;
    affine_t a;
    to_affine_batch(ck->key.curve, &sum, &a, 1);
    std::memcpy(out, &a, sizeof(a));
    return SRS_OK;
}

// bucket additions per scalar of the key's last streamed commit (0: none yet)
double key_density(const msm::Key &k) {
    return k.last_scalars ? (double)k.last_entries / (double)k.last_scalars : 0.0;
}

// chunk boundaries of a streamed commit of n elements; `align`: boundaries are multiples of it (columns are never split)
// `n_eff`: the scalars one device accumulates (n / world on a sharded key): what the number of chunks is chosen from
// `density`: bucket additions (non-zero 16-bit digits) per scalar of the key's previous streamed commit, 0 = unknown.  The default cuts are
// tuned for a commit whose accumulation takes about as long per byte as the upload (>= ~5 additions per scalar: the bench witness has
// 7.2); a witness of 0 / 1 / small values (~2.3) accumulates 2-3x faster than it uploads -- the device WAITS for every
// chunk, the commit ends one chunk's work after the last byte, and the schedule that fits is many even chunks with a SMALL last one
// (r05, profiles/r05_ab_survey_cuts.txt).
std::vector<size_t> commit_cuts(size_t n, size_t align, size_t n_eff = 0, double density = 0.0) {
    const size_t want = (size_t)std::max<int64_t>(0, tuning::get_or(tuning::COMMIT_CHUNKS, 0));      // tests: n equal chunks
    if (!n_eff) n_eff = n;
    size_t chunks = want ? want : std::min<size_t>(4, std::max<size_t>(1, n_eff >> 20));
    chunks = std::min<size_t>(chunks, msm::LANDING_SLOTS);
    auto up = [&](size_t x) { return std::min(n, (x + align - 1) / align * align); };
    std::vector<size_t> cut(1, 0);
    if (!want && chunks < 4 && n_eff >= ((size_t)1 << 19)) {       // 0.5 - 4 M scalars: a quarter first, so that 3/4 of the upload hides behind its MSM
        const size_t c = up(n / 4);
        if (c > 0 && c < n) cut.push_back(c);
    } else if (want || chunks < 4) {     // equal pieces
        const size_t per = up((n + chunks - 1) / chunks);
        for (size_t a = per; a < n; a += per) cut.push_back(a);
    } else {                             // a SHORT first chunk (its upload is the only one nothing overlaps), growing ones after it
        // (measured on the 12 * 2^20 witness, profiles/r02_commit_cuts.txt: six chunks growing ~1.5x beat four -- the MSM left to do
        // after the last byte has arrived is what counts once the per-chunk fixed cost is down to ~0.4 ms)
        // r03: re-tuned twice as the per-chunk costs fell (profiles/r03_ab_accum0_variants.txt, r03_ab_commit_cuts.txt).  With the
        // faster accumulation the chip keeps up with the uploads until the last chunk, so what counts is the MSM left once the last
        // byte has arrived: seven chunks, the last one 26 % -- was {0.045, 0.15, 0.32, 0.53, 0.77}, then {0.024, 0.089, 0.208, 0.399, 0.677}
        // r04 (slot mode: no accumulation levels per chunk, ~0.15 ms of fixed cost per chunk): the chip now WAITED for the uploads of the
        // steeply growing middle chunks (~0.9 ms per commit in the kernel trace) -- nine chunks, each upload no longer than the chunk before
        // it takes to accumulate: 12.0 -> 11.6 ms per step (profiles/r04_ab_slots_cuts.txt).  Later in r04, with the per-chunk cost down to
        // ~0.14 ms and accumulation about as fast per byte as the upload, the no-wait condition  upload(j + 1) <= cost(j)  means chunks
        // growing LINEARLY: ten of them, 11.2 -> 11.0 ms (same file, last section)
        std::vector<double> frac = {0.02, 0.058, 0.115, 0.19, 0.285, 0.40, 0.535, 0.69, 0.86};
        if (density > 0.0 && density < 4.5)            // upload-bound regime: eleven even chunks, the last one 4 %
            frac = {0.03, 0.12, 0.24, 0.36, 0.48, 0.60, 0.72, 0.82, 0.90, 0.96};
        for (double f : frac) {
            const size_t c = up((size_t)(f * (double)n));
            if (c > cut.back() && c < n && cut.size() < msm::LANDING_SLOTS) cut.push_back(c);    // one landing slot per chunk
        }
    }
    cut.push_back(n);
    return cut;
}

// srs_commit_upload on a MULTI-DEVICE key (r05; VERDICT r04 "missing" 2): every shard brings up ITS block-cyclic stripes of the host
// vector over ITS link -- n * 32 / shards bytes each, once -- in chunks whose upload overlaps the shard's own MSM of the chunks already in
// its HBM (commit_streamed_core, the process-sharded form: the stripes land at their global positions of a vector-sized buffer on the
// shard's device).  The device copy the caller asked for lives on the process's device: shard 0 (same device) uploads straight into it, the
// other shards forward their stripes to it with peer copies behind their uploads (xGMI on a multi-GPU node) -- the whole witness never
// crosses ONE link, and nothing is uploaded twice.  The caller's stream waits for the forwarded stripes; the partial sums are added here.
int multi_commit_streamed(srs_ck *ck, const std::vector<Seg> &segs, size_t n, fe_t *dev_copy, int repr, hipStream_t st, srs_affine *out) {
    const uint32_t world = (uint32_t)ck->shards.size();
    const int home = home_device();
    const size_t align = (size_t)world << msm::STRIPE_LOG;
    const std::vector<size_t> cut = commit_cuts(n, align, n / world, key_density(ck->shards[0]->key));
    std::vector<xyzz_t> parts(world);
    hipEvent_t ready = nullptr;
    if (dev_copy) {                              // nothing may land in the device copy while earlier work on the caller's stream still reads it
        if (ck->events.empty()) { hipEvent_t e; SRS_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ck->events.push_back(e); }
        ready = ck->events[0];
        SRS_HIP_CHECK(hipEventRecord(ready, st));
    }
    for (uint32_t d = 0; d < world; ++d) {
        CkShard *sh = ck->shards[d].get();
        sh->post([=, &parts, &segs, &cut]() {
            const bool direct = dev_copy && sh->device == home && d == 0;          // shard 0 lands in the caller's device copy itself
            fe_t *dst = dev_copy;
            if (!direct) {
                if (sh->full_cap < n) {
                    if (sh->full) SRS_HIP_CHECK(hipFree(sh->full));
                    sh->full = nullptr;
                    sh->full_cap = 0;
                    SRS_HIP_CHECK(hipMalloc((void **)&sh->full, n * sizeof(fe_t)));
                    sh->full_cap = n;
                }
                dst = sh->full;
            }
            // `full` is reused by every commit: this commit's uploads must not overwrite stripes the PREVIOUS commit's peer copies still read
            // (the core orders its copy stream behind sh->stream)
            if (!direct && sh->peer_pending) SRS_HIP_CHECK(hipStreamWaitEvent(sh->stream, sh->peer_done, 0));
            if (dev_copy && !direct) {
                if (!sh->peer_stream) SRS_HIP_CHECK(hipStreamCreateWithFlags(&sh->peer_stream, hipStreamNonBlocking));
                if (!sh->peer_done) SRS_HIP_CHECK(hipEventCreateWithFlags(&sh->peer_done, hipEventDisableTiming));
                SRS_HIP_CHECK(hipStreamWaitEvent(sh->peer_stream, ready, 0));
            } else if (direct) {
                SRS_HIP_CHECK(hipStreamWaitEvent(sh->stream, ready, 0));           // (the core orders its copy stream behind sh->stream)
            }
            StreamRes r{sh->key, sh->copy_stream, sh->events};
            r.h2d_bytes = &sh->stat_h2d_bytes;
            if (dev_copy && !direct) {
                r.peer_dst = dev_copy;
                r.peer_stream = sh->peer_stream;
                r.peer_bytes = &sh->stat_peer_bytes;
            }
            // the shard's kernels read ITS stripes at their global positions (the process-sharded layout), not a gathered vector
            struct Restore { msm::Key &k; bool v; ~Restore() { k.compact_scalars = v; } } restore{sh->key, sh->key.compact_scalars};
            sh->key.compact_scalars = false;
            ++sh->stat_streamed;
            int rc = commit_streamed_core(r, segs, n, cut, dst, repr, sh->stream, &parts[d]);
            if (rc) throw DeviceError{rc};
            if (dev_copy && !direct) {
                SRS_HIP_CHECK(hipEventRecord(sh->peer_done, sh->peer_stream));
                sh->peer_pending = true;
            }
        });
    }
    int rc = SRS_OK;
    std::string err;
    for (uint32_t d = 0; d < world; ++d) {
        int r = ck->shards[d]->wait();
        if (r && !rc) { rc = r; err = ck->shards[d]->err; }
    }
    if (rc) return fail(rc, "multi-device streamed commit: " + err);
    for (uint32_t d = 0; d < world && dev_copy; ++d) {
        CkShard *sh = ck->shards[d].get();
        const bool direct = sh->device == home && d == 0;
        if (!direct && sh->peer_done) SRS_HIP_CHECK(hipStreamWaitEvent(st, sh->peer_done, 0));   // the caller's stream sees the whole device copy
    }
    xyzz_t sum;
    std::vector<std::vector<xyzz_t>> pv(world, std::vector<xyzz_t>(1));
    for (uint32_t d = 0; d < world; ++d) pv[d][0] = parts[d];
    if (ck->key.curve == SRS_CURVE_BN256) sum_partials_t<Bn256>(pv, 1, &sum); else sum_partials_t<Grumpkin>(pv, 1, &sum);
    affine_t a;
    to_affine_batch(ck->key.curve, &sum, &a, 1);
    std::memcpy(out, &a, sizeof(a));
    return SRS_OK;
}
}  // namespace

int srs_commit_upload(srs_ck *ck, const srs_fe *scalars_host, size_t n, srs_fe *dev_copy, int repr, void *stream, srs_affine *out) {
    if (!ck || !out || (n && !scalars_host)) return fail(SRS_ERR_INVALID, "srs_commit_upload: bad argument");
    if (n > ck->key.global_len)
        return fail(SRS_ERR_TOO_LONG_INPUT, "Can't commit too long input: input len: " + std::to_string(n) + ", but limit is " +
                                                std::to_string(ck->key.global_len));
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> key_lock(ck->mu);
    hipStream_t st = (hipStream_t)stream;
    if (ck->shards.empty() && ck->key.world > 1 && n && dev_copy) {
        // sharded (process-per-GPU) key: the rank's kernels -- its partial MSM, its rows of the cross terms, its tiles of the
        // ProtoGalaxy leaves -- read only the rank's block-cyclic stripes of any vector, so only those go up: n * 32 / world
        // bytes over this GPU's link, in chunks whose upload overlaps the MSM of the previous chunk (commit_streamed); the other
        // stripes of dev_copy are left as they are
        return guarded([&]() -> int {
            const size_t align = (size_t)ck->key.world << msm::STRIPE_LOG;
            std::vector<Seg> segs(1, Seg{reinterpret_cast<const fe_t *>(scalars_host), 0, n});
            return commit_streamed(ck, segs, n, commit_cuts(n, align, n / ck->key.world, key_density(ck->key)), reinterpret_cast<fe_t *>(dev_copy), repr, st, out);
        });
    }
    if (!ck->shards.empty() && n >= ((size_t)ck->shards.size() << (msm::STRIPE_LOG + 1))) {
        // multi-device key: every shard streams its stripes over its own link, overlapped with its MSM; the device copy is assembled on the
        // process's device by peer copies (multi_commit_streamed)
        return guarded([&]() -> int {
            std::vector<Seg> segs(1, Seg{reinterpret_cast<const fe_t *>(scalars_host), 0, n});
            return multi_commit_streamed(ck, segs, n, reinterpret_cast<fe_t *>(dev_copy), repr, st, out);
        });
    }
    if (!ck->shards.empty() || ck->key.world != 1 || n == 0) {
        // (short vectors on a multi-device key) every shard pulls its stripes from the host buffer over its own link; sharded (process-per-GPU)
        // key without a device copy: the kernels pick this rank's stripes out of the full vector.
        if (dev_copy && n) {
            rc = guarded([&]() -> int {
                SRS_HIP_CHECK(hipMemcpyAsync(dev_copy, scalars_host, n * sizeof(fe_t), hipMemcpyHostToDevice, st));
                return SRS_OK;
            });
            if (rc) return rc;
        }
        if (ck->shards.empty() && dev_copy) return srs_commit(ck, dev_copy, n, SRS_SPACE_DEVICE, repr, stream, out);
        return srs_commit(ck, scalars_host, n, SRS_SPACE_HOST, repr, stream, out);
    }
    return guarded([&]() -> int {
        fe_t *dst = reinterpret_cast<fe_t *>(dev_copy);
        if (!dst) {
            ck->staging.reserve(Arena::pad(n * sizeof(fe_t)) + 256);
            ck->staging.reset();
            dst = ck->staging.take<fe_t>(n);
        }
        std::vector<Seg> segs(1, Seg{reinterpret_cast<const fe_t *>(scalars_host), 0, n});
        return commit_streamed(ck, segs, n, commit_cuts(n, 1024, 0, key_density(ck->key)), dst, repr, st, out);
    });
}

size_t srs_concat_len(const size_t *lens, size_t n_columns, size_t pad_size) {
    size_t total = 0;
    for (size_t c = 0; c < n_columns; ++c) total += std::max(lens ? lens[c] : 0, pad_size);
    return total;
}

int srs_concat_with_padding(srs_fe *out_dev, const srs_fe *const *columns_host, const size_t *lens, size_t n_columns, size_t pad_size,
                            void *stream) {
    if ((n_columns && (!columns_host || !lens)) || (srs_concat_len(lens, n_columns, pad_size) && !out_dev))
        return fail(SRS_ERR_INVALID, "srs_concat_with_padding: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        std::vector<Seg> segs;
        size_t off = 0;
        for (size_t c = 0; c < n_columns; ++c) {
            if (lens[c] && !columns_host[c]) return fail(SRS_ERR_INVALID, "srs_concat_with_padding: null column");
            segs.push_back(Seg{reinterpret_cast<const fe_t *>(columns_host[c]), off, lens[c]});
            off += std::max(lens[c], pad_size);
        }
        upload_range(reinterpret_cast<fe_t *>(out_dev), segs, 0, off, (hipStream_t)stream);
        return SRS_OK;
    });
}

int srs_commit_upload_columns(srs_ck *ck, const srs_fe *const *columns_host, const size_t *lens, size_t n_columns, size_t pad_size,
                              srs_fe *dev_copy, int repr, void *stream, srs_affine *out) {
    if (!ck || !out || (n_columns && (!columns_host || !lens))) return fail(SRS_ERR_INVALID, "srs_commit_upload_columns: bad argument");
    const size_t n = srs_concat_len(lens, n_columns, pad_size);
    if (n > ck->key.global_len)
        return fail(SRS_ERR_TOO_LONG_INPUT, "Can't commit too long input: input len: " + std::to_string(n) + ", but limit is " +
                                                std::to_string(ck->key.global_len));
    for (size_t c = 0; c < n_columns; ++c)
        if (lens[c] && !columns_host[c]) return fail(SRS_ERR_INVALID, "srs_commit_upload_columns: null column");
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> key_lock(ck->mu);
    hipStream_t st = (hipStream_t)stream;
    return guarded([&]() -> int {
        std::vector<Seg> segs;
        size_t off = 0;
        bool uniform = true;                 // every column no longer than the pad: chunk boundaries fall between columns
        for (size_t c = 0; c < n_columns; ++c) {
            segs.push_back(Seg{reinterpret_cast<const fe_t *>(columns_host[c]), off, lens[c]});
            uniform = uniform && lens[c] <= pad_size;
            off += std::max(lens[c], pad_size);
        }
        fe_t *dst = reinterpret_cast<fe_t *>(dev_copy);
        if (!ck->shards.empty() && n >= ((size_t)ck->shards.size() << (msm::STRIPE_LOG + 1)))
            // multi-device key (r05): every shard streams ITS stripes of the columns (and of their zero padding) over its own link
            return multi_commit_streamed(ck, segs, n, dst, repr, st, out);
        const bool sharded_streamed = ck->shards.empty() && ck->key.world > 1 && n >= ((size_t)ck->key.world << (msm::STRIPE_LOG + 1));
        const bool streamed = ck->shards.empty() && (ck->key.world == 1 || sharded_streamed) && n != 0;
        if (!dst) {
            ck->staging.reserve(Arena::pad((n + 1) * sizeof(fe_t)) + 256);
            ck->staging.reset();
            dst = ck->staging.take<fe_t>(n + 1);
        }
        if (!streamed) {                     // (short vectors on multi-device / sharded keys) assemble in HBM, then the ordinary commit
            upload_range(dst, segs, 0, n, st);
            return srs_commit(ck, reinterpret_cast<const srs_fe *>(dst), n, SRS_SPACE_DEVICE, repr, stream, out);
        }
        if (sharded_streamed)                // process-sharded key (r05): the rank's stripes of the columns, chunk boundaries on world * 2^10
            return commit_streamed(ck, segs, n, commit_cuts(n, (size_t)ck->key.world << msm::STRIPE_LOG, n / ck->key.world, key_density(ck->key)), dst,
                                   repr, st, out);
        const size_t align = (uniform && pad_size >= 1024) ? pad_size : 1024;
        std::vector<size_t> cut = commit_cuts(n, align, 0, key_density(ck->key));
        return commit_streamed(ck, segs, n, cut, dst, repr, st, out);
    });
}

int srs_point_sum(int curve, const srs_affine *points, size_t n, srs_affine *out) {
    if (!valid_curve(curve) || !out || (n && !points)) return fail(SRS_ERR_INVALID, "srs_point_sum: bad argument");
    auto go = [&](auto tag) {
        using C = decltype(tag);
        xyzz_t acc = Ec<C>::identity();
        for (size_t i = 0; i < n; ++i) {
            affine_t p;
            std::memcpy(&p, &points[i], sizeof(p));
            acc = Ec<C>::madd(acc, p);
        }
        affine_t a = Ec<C>::to_affine(acc);
        std::memcpy(out, &a, sizeof(a));
    };
    if (curve == SRS_CURVE_BN256) go(Bn256{}); else go(Grumpkin{});
    return SRS_OK;
}

int srs_point_mul(int curve, const srs_fe *scalar, int repr, const srs_affine *p, srs_affine *out) {
    if (!valid_curve(curve) || !scalar || !p || !out) return fail(SRS_ERR_INVALID, "srs_point_mul: bad argument");
    auto go = [&](auto tag) {
        using C = decltype(tag);
        fe_t s;
        affine_t P;
        std::memcpy(&s, scalar, 32);
        std::memcpy(&P, p, 64);
        if (repr == SRS_REPR_MONT) s = C::S::from_mont(s);
        affine_t a = Ec<C>::to_affine(Ec<C>::mul_canon(s.v, P));
        std::memcpy(out, &a, sizeof(a));
    };
    if (curve == SRS_CURVE_BN256) go(Bn256{}); else go(Grumpkin{});
    return SRS_OK;
}

int srs_ntt_batch(int field, srs_fe *a, size_t n, size_t stride, size_t batch, int inverse, int coset, int space,
                  void *stream) {
    if (!valid_field(field)) return fail(SRS_ERR_INVALID, "srs_ntt: unknown field");
    if (n == 0 || (n & (n - 1))) return fail(SRS_ERR_NOT_POW2, "srs_ntt: length is not a power of two");
    uint32_t log_n = 0;
    while (((size_t)1 << log_n) < n) ++log_n;
    if (field != SRS_FIELD_FR) return fail(SRS_ERR_INVALID, "srs_ntt: only bn256::Fr is 2-adic enough (Fq has S = 1)");
    if (log_n > ntt::FR_S)
        return fail(SRS_ERR_K_TOO_LARGE, "k=" + std::to_string(log_n) + " should no larger than F::S=" + std::to_string(ntt::FR_S));
    if (batch == 0) return SRS_OK;
    if (!a || (batch > 1 && stride < n)) return fail(SRS_ERR_INVALID, "srs_ntt: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        if (space == SRS_SPACE_DEVICE) {
            ntt::run(reinterpret_cast<fe_t *>(a), log_n, stride, (uint32_t)batch, inverse != 0, coset != 0, st);
            SRS_HIP_CHECK(hipStreamSynchronize(st));
        } else {
            fe_t *d = nullptr;
            size_t total = (batch - 1) * stride + n;
            SRS_HIP_CHECK(hipMalloc((void **)&d, total * sizeof(fe_t)));
            try {
                SRS_HIP_CHECK(hipMemcpyAsync(d, a, total * sizeof(fe_t), hipMemcpyHostToDevice, st));
                ntt::run(d, log_n, stride, (uint32_t)batch, inverse != 0, coset != 0, st);
                SRS_HIP_CHECK(hipMemcpyAsync(a, d, total * sizeof(fe_t), hipMemcpyDeviceToHost, st));
                SRS_HIP_CHECK(hipStreamSynchronize(st));
            } catch (...) {
                (void)hipFree(d);
                throw;
            }
            SRS_HIP_CHECK(hipFree(d));
        }
        SRS_HIP_CHECK(hipGetLastError());
        return SRS_OK;
    });
}

int srs_ntt_set_max_radix_bits(int bits) {
    int b = bits < 4 ? 4 : (bits > 8 ? 8 : bits);
    ntt::set_max_radix_bits((uint32_t)b);
    return b;
}

int srs_ntt(int field, srs_fe *a, size_t n, int inverse, int coset, int space, void *stream) {
    return srs_ntt_batch(field, a, n, n, 1, inverse, coset, space, stream);
}

// acc + sum scalars[i] * points[i]; inputs already validated
static void point_lincomb_impl(int curve, const srs_affine *acc, const srs_affine *points, const srs_fe *scalars, size_t n, int repr,
                               srs_affine *out) {
    auto go = [&](auto tag) {
        using C = decltype(tag);
        std::vector<xyzz_t> part(n);
        auto work = [&](size_t i) {
            fe_t s;
            affine_t P;
            std::memcpy(&s, &scalars[i], 32);
            std::memcpy(&P, &points[i], 64);
            if (repr == SRS_REPR_MONT) s = C::S::from_mont(s);
            part[i] = Ec<C>::mul_canon(s.v, P);
        };
        HostPool::get().parallel_for(n, work);           // the d scalar-muls are independent
        xyzz_t a = Ec<C>::identity();
        if (acc) {
            affine_t A;
            std::memcpy(&A, acc, 64);
            a = Ec<C>::from_affine(A);
        }
        for (size_t i = 0; i < n; ++i) a = Ec<C>::add(a, part[i]);
        affine_t r = Ec<C>::to_affine(a);
        std::memcpy(out, &r, sizeof(r));
    };
    if (curve == SRS_CURVE_BN256) go(Bn256{}); else go(Grumpkin{});
}

int srs_fe_powers(int field, const srs_fe *r, size_t n, srs_fe *out) {
    if (!valid_field(field) || !r || (n && !out)) return fail(SRS_ERR_INVALID, "srs_fe_powers: bad argument");
    fe_t x, acc;
    std::memcpy(&x, r, 32);
    acc = x;
    for (size_t i = 0; i < n; ++i) {
        std::memcpy(&out[i], &acc, 32);
        acc = field == SRS_FIELD_FR ? Fr::mul(acc, x) : Fq::mul(acc, x);
    }
    return SRS_OK;
}

int srs_point_lincomb(int curve, const srs_affine *acc, const srs_affine *points, const srs_fe *scalars, size_t n, int repr,
                      srs_affine *out) {
    if (!valid_curve(curve) || !out || (n && (!points || !scalars))) return fail(SRS_ERR_INVALID, "srs_point_lincomb: bad argument");
    point_lincomb_impl(curve, acc, points, scalars, n, repr, out);
    return SRS_OK;
}

// The instance fold is host work that nothing on the device waits for: run behind the caller's back on one background
// thread (FIFO; it borrows the HostPool for the independent scalar multiplications) while the caller enqueues the next
// commitment.  Inputs are copied at submission; only `out` has to stay valid until srs_job_wait.
namespace {
class AsyncJobs {
public:
    static AsyncJobs &get() {
        static AsyncJobs *q = new AsyncJobs();    // never destroyed (detached worker)
        return *q;
    }
    uint64_t submit(std::function<void()> fn) {
        std::lock_guard<std::mutex> lk(mu_);
        const uint64_t id = ++last_id_;
        queue_.emplace_back(id, std::move(fn));
        open_.insert(id);
        cv_.notify_one();
        return id;
    }
    // -1: unknown (or already waited-for) job; otherwise the job's result code (0 = it ran to completion)
    int wait(uint64_t id, std::string &err) {
        std::unique_lock<std::mutex> lk(mu_);
        if (!open_.count(id)) return -1;
        done_cv_.wait(lk, [&] { return done_.count(id) != 0; });
        const int rc = done_[id].first;
        err = done_[id].second;
        done_.erase(id);
        open_.erase(id);
        return rc;
    }

private:
    AsyncJobs() { std::thread([this] { loop(); }).detach(); }
    void loop() {
        for (;;) {
            std::pair<uint64_t, std::function<void()>> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return !queue_.empty(); });
                job = std::move(queue_.front());
                queue_.pop_front();
            }
            int rc = SRS_OK;
            std::string err;
            try {                                  // an exception must neither escape the detached thread nor leave the job open
                job.second();
            } catch (const std::exception &e) {
                rc = SRS_ERR_DEVICE;
                err = e.what();
            } catch (...) {
                rc = SRS_ERR_DEVICE;
                err = "unknown exception in an asynchronous job";
            }
            std::lock_guard<std::mutex> lk(mu_);
            done_[job.first] = std::make_pair(rc, err);
            done_cv_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::deque<std::pair<uint64_t, std::function<void()>>> queue_;
    std::set<uint64_t> open_;
    std::map<uint64_t, std::pair<int, std::string>> done_;
    uint64_t last_id_ = 0;
};
}  // namespace

int srs_point_lincomb_async(int curve, const srs_affine *acc, const srs_affine *points, const srs_fe *scalars, size_t n, int repr,
                            srs_affine *out, uint64_t *job) {
    if (!valid_curve(curve) || !out || !job || (n && (!points || !scalars)))
        return fail(SRS_ERR_INVALID, "srs_point_lincomb_async: bad argument");
    struct Args {
        bool has_acc;
        srs_affine acc;
        std::vector<srs_affine> points;
        std::vector<srs_fe> scalars;
    };
    auto a = std::make_shared<Args>();
    a->has_acc = acc != nullptr;
    if (acc) a->acc = *acc;
    a->points.assign(points, points + n);
    a->scalars.assign(scalars, scalars + n);
    *job = AsyncJobs::get().submit([=] {
        point_lincomb_impl(curve, a->has_acc ? &a->acc : nullptr, a->points.data(), a->scalars.data(), n, repr, out);
    });
    return SRS_OK;
}

int srs_job_wait(uint64_t job) {
    std::string err;
    const int rc = AsyncJobs::get().wait(job, err);
    if (rc < 0) return fail(SRS_ERR_INVALID, "srs_job_wait: unknown job");
    if (rc) return fail(rc, "srs_job_wait: the job failed: " + err);
    return SRS_OK;
}

void srs_profile_enable(int on) { prof::enable(on != 0); }
void srs_profile_reset(void) { prof::reset(); }
void srs_profile_sampling(unsigned every) { prof::sampling(every); }
int srs_profile_get(const char *name, double *total_ms, uint64_t *launches, uint64_t *units) {
    prof::Stat st;
    if (!name || !prof::get(name, st)) return SRS_ERR_INVALID;
    if (total_ms) *total_ms = st.total_ms;
    if (launches) *launches = st.launches;
    if (units) *units = st.units;
    return SRS_OK;
}

// ------------------------------------------------------------------ row programs
static int structure_create_impl(const char *who, int field, uint32_t k, size_t num_selectors, size_t num_fixed, size_t num_advice,
                                 const uint8_t *const *selectors, const srs_fe *const *fixed, int space, const uint64_t *gates,
                                 size_t gates_words, size_t num_gates, size_t num_lookups, int has_vector_lookup,
                                 const uint64_t *lookup_exprs, size_t lookup_words, srs_structure **out) {
    if (!valid_field(field) || !out || k > 30 || (num_selectors && !selectors) || (num_fixed && !fixed) || (gates_words && !gates) ||
        (num_lookups && (!lookup_exprs || !lookup_words)))
        return fail(SRS_ERR_INVALID, std::string(who) + ": bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        std::string err;
        int crc = 0;
        rowprog::Structure *s = rowprog::create(field, k, num_selectors, num_fixed, num_advice, selectors,
                                                reinterpret_cast<const fe_t *const *>(fixed), space == SRS_SPACE_DEVICE, gates,
                                                gates_words, num_gates, num_lookups, has_vector_lookup != 0, lookup_exprs,
                                                lookup_words, crc, err);
        if (!s) return fail(crc ? crc : SRS_ERR_INVALID, std::string(who) + ": " + err);
        srs_structure *S = new srs_structure();
        S->s = s;
        *out = S;
        return SRS_OK;
    });
}
int srs_structure_create(int field, uint32_t k, size_t num_selectors, size_t num_fixed, size_t num_advice,
                         const uint8_t *const *selectors, const srs_fe *const *fixed, int space, const uint64_t *gates,
                         size_t gates_words, size_t num_gates, srs_structure **out) {
    return structure_create_impl("srs_structure_create", field, k, num_selectors, num_fixed, num_advice, selectors, fixed, space, gates,
                                 gates_words, num_gates, 0, 0, nullptr, 0, out);
}
int srs_structure_create_lookup(int field, uint32_t k, size_t num_selectors, size_t num_fixed, size_t num_advice,
                                const uint8_t *const *selectors, const srs_fe *const *fixed, int space, const uint64_t *gates,
                                size_t gates_words, size_t num_gates, size_t num_lookups, int has_vector_lookup,
                                const uint64_t *lookup_exprs, size_t lookup_words, srs_structure **out) {
    return structure_create_impl("srs_structure_create_lookup", field, k, num_selectors, num_fixed, num_advice, selectors, fixed, space,
                                 gates, gates_words, num_gates, num_lookups, has_vector_lookup, lookup_exprs, lookup_words, out);
}

void srs_structure_free(srs_structure *S) {
    if (!S) return;
    rowprog::destroy(S->s);
    S->io.release();
    for (int i = 0; i < 2; ++i)
        if (S->halo_dev[i]) (void)hipFree(S->halo_dev[i]);
    delete S;
}
int srs_structure_set_shard(srs_structure *S, uint32_t rank, uint32_t world) {
    if (!S || world == 0 || rank >= world) return fail(SRS_ERR_INVALID, "srs_structure_set_shard: bad argument");
    rowprog::set_shard(S->s, rank, world);
    return SRS_OK;
}
// rows of ONE column a sharded rank's kernels read beyond its own stripes: the rotation halo of every stripe, and row 0 (+ rotations) when
// every ProtoGalaxy leaf sits at row 0 (reference_compat, src/plonk/mod.rs:714).  Sorted, unique.
static std::vector<size_t> shard_halo_rows(const srs_structure *S, int reference_compat) {
    const uint32_t world = rowprog::shard_world(S->s), rank = rowprog::shard_rank(S->s);
    const size_t rows = rowprog::rows(S->s), SL = (size_t)1 << rowprog::ROW_STRIPE_LOG;
    int32_t lo = 0, hi = 0;
    rowprog::rotation_range(S->s, &lo, &hi);
    auto own = [&](size_t row) { return (row >> rowprog::ROW_STRIPE_LOG) % world == rank; };
    std::vector<size_t> need;
    auto add_row = [&](int64_t r) {
        const size_t row = (size_t)(((r % (int64_t)rows) + (int64_t)rows) % (int64_t)rows);
        if (!own(row)) need.push_back(row);
    };
    if (lo < 0 || hi > 0) {
        for (size_t s = rank; s < rows / SL; s += world) {
            for (int64_t r = lo; r < 0; ++r) add_row((int64_t)(s * SL) + r);
            for (int64_t r = 0; r < hi; ++r) add_row((int64_t)((s + 1) * SL) + r);
        }
    }
    if (reference_compat)
        for (int64_t r = lo; r <= hi; ++r) add_row(r);
    std::sort(need.begin(), need.end());
    need.erase(std::unique(need.begin(), need.end()), need.end());
    return need;
}

// ProtoGalaxy::fold_witness / RelaxedPlonkWitness::fold of a row-sharded structure's accumulator: the rank's stripes AND the halo rows
int srs_structure_fold_sharded(srs_structure *S, srs_fe *out, const srs_fe *const *W, const srs_fe *coefs, size_t J, int reference_compat,
                               void *stream) {
    if (!S || !out || !W || !coefs || J == 0) return fail(SRS_ERR_INVALID, "srs_structure_fold_sharded: bad argument");
    const uint32_t world = rowprog::shard_world(S->s), rank = rowprog::shard_rank(S->s), k = rowprog::log_rows(S->s);
    const size_t rows = rowprog::rows(S->s), cols = rowprog::num_witness_columns(S->s), SL = (size_t)1 << rowprog::ROW_STRIPE_LOG;
    const int field = rowprog::field(S->s);
    if (world > 1 && (k < rowprog::ROW_STRIPE_LOG || (rows / SL) % world != 0))
        return fail(SRS_ERR_INVALID, "srs_structure_fold_sharded: 2^k / 2^10 is not a multiple of the world size -- the row stripes of the "
                                     "columns do not coincide with the key's stripes; fold the whole vector (srs_fold_lincomb)");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        std::string err;
        int erc = rowprog::lincomb(field, reinterpret_cast<fe_t *>(out), reinterpret_cast<const fe_t *const *>(W),
                                   reinterpret_cast<const fe_t *>(coefs), J, rows * cols, st, err, rank, world);
        if (erc) return fail(erc, "srs_structure_fold_sharded: " + err);
        if (world <= 1) return SRS_OK;
        const int c = reference_compat ? 1 : 0;
        if (S->halo_rank != rank || S->halo_world != world) {           // the shard changed: both lists are stale
            for (int i = 0; i < 2; ++i) {
                if (S->halo_dev[i]) (void)hipFree(S->halo_dev[i]);
                S->halo_dev[i] = nullptr;
                S->halo_n[i] = 0;
            }
            S->halo_rank = rank;
            S->halo_world = world;
        }
        if (!S->halo_dev[c]) {
            const std::vector<size_t> need = shard_halo_rows(S, reference_compat);
            std::vector<uint32_t> r32(need.begin(), need.end());
            SRS_HIP_CHECK(hipMalloc((void **)&S->halo_dev[c], (r32.size() + 1) * sizeof(uint32_t)));
            if (!r32.empty()) SRS_HIP_CHECK(hipMemcpy(S->halo_dev[c], r32.data(), r32.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            S->halo_n[c] = r32.size();
        }
        erc = rowprog::lincomb_rows(field, reinterpret_cast<fe_t *>(out), reinterpret_cast<const fe_t *const *>(W),
                                    reinterpret_cast<const fe_t *>(coefs), J, S->halo_dev[c], S->halo_n[c], cols, rows, st, err);
        if (erc) return fail(erc, "srs_structure_fold_sharded: " + err);
        SRS_HIP_CHECK(hipGetLastError());
        return SRS_OK;
    });
}

int srs_structure_upload_shard_halo(const srs_structure *S, const srs_fe *witness_host, srs_fe *dev_copy, size_t n, int reference_compat,
                                    void *stream) {
    if (!S || !witness_host || !dev_copy) return fail(SRS_ERR_INVALID, "srs_structure_upload_shard_halo: bad argument");
    const uint32_t world = rowprog::shard_world(S->s), rank = rowprog::shard_rank(S->s), k = rowprog::log_rows(S->s);
    const size_t rows = rowprog::rows(S->s), cols = rowprog::num_witness_columns(S->s);
    if (n != rows * cols) return fail(SRS_ERR_INVALID, "srs_structure_upload_shard_halo: n is not num_witness_columns * 2^k");
    if (world <= 1) return SRS_OK;
    const size_t SL = (size_t)1 << rowprog::ROW_STRIPE_LOG;
    if (k < rowprog::ROW_STRIPE_LOG || (rows / SL) % world != 0)
        return fail(SRS_ERR_INVALID, "srs_structure_upload_shard_halo: 2^k / 2^10 is not a multiple of the world size -- the row stripes of the "
                                     "columns do not coincide with the key's stripes; upload the whole witness instead");
    int rc = ensure_device();
    if (rc) return rc;
    // row r of column c is element c * 2^k + r of the witness
    std::vector<std::pair<size_t, size_t>> runs;        // [first, last) row intervals to upload, per column
    const std::vector<size_t> need = shard_halo_rows(S, reference_compat);
    for (size_t i = 0; i < need.size();) {
        size_t j = i + 1;
        while (j < need.size() && need[j] == need[j - 1] + 1) ++j;
        runs.emplace_back(need[i], need[j - 1] + 1);
        i = j;
    }
    return guarded([&]() -> int {
        const fe_t *src = reinterpret_cast<const fe_t *>(witness_host);
        fe_t *dst = reinterpret_cast<fe_t *>(dev_copy);
        for (auto &r : runs)
            SRS_HIP_CHECK(hipMemcpy2DAsync(dst + r.first, rows * sizeof(fe_t), src + r.first, rows * sizeof(fe_t), (r.second - r.first) * sizeof(fe_t),
                                           cols, hipMemcpyHostToDevice, (hipStream_t)stream));
        return SRS_OK;
    });
}
size_t srs_structure_num_cross_terms(const srs_structure *S) { return S ? rowprog::degree(S->s) : 0; }
size_t srs_structure_num_challenges(const srs_structure *S) { return S ? rowprog::num_challenges(S->s) : 0; }
size_t srs_structure_num_witness_columns(const srs_structure *S) { return S ? rowprog::num_witness_columns(S->s) : 0; }
int srs_jit_selfcheck(size_t *code_bytes, char *log, size_t log_cap) {
    if constexpr (rt::kEmulated) return fail(SRS_ERR_INVALID, "srs_jit_selfcheck: not part of the emulator build");
    std::string text;
    const bool ok = rowprog::jit_selfcheck(code_bytes, text);
    if (log && log_cap) {
        const size_t n = std::min(text.size(), log_cap - 1);
        std::memcpy(log, text.data(), n);
        log[n] = 0;
    }
    return ok ? SRS_OK : fail(SRS_ERR_INVALID, "srs_jit_selfcheck: hiprtc rejected the emitted form: " + text.substr(0, 400));
}

int srs_structure_kernel_kind(const srs_structure *S, int which) {
    if (!S || which < 0 || which > 2) return -1;
    return rowprog::kernel_kind(S->s, which);
}

size_t srs_structure_program_source(srs_structure *S, int which, char *buf, size_t cap, uint64_t *fingerprint, int *spec_id) {
    if (!S) return 0;
    std::string src;
    rowprog::spec_source(S->s, which, fingerprint, spec_id, src);
    if (buf && cap) {
        size_t n = std::min(src.size(), cap - 1);
        std::memcpy(buf, src.data(), n);
        buf[n] = 0;
    }
    return src.size();
}

static int cross_terms_impl(srs_structure *S, srs_ck *ck, const srs_fe *W1, const srs_fe *W2, const srs_fe *challenges,
                            size_t n_challenges, int space, void *stream, srs_fe *const *T_out, srs_affine *commits_out) {
    if (!S || !W1 || !W2 || (n_challenges && !challenges)) return fail(SRS_ERR_INVALID, "srs_cross_terms: bad argument");
    if (ck && !commits_out) return fail(SRS_ERR_INVALID, "srs_commit_cross_terms: commits_out is NULL");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        rowprog::Structure *s = S->s;
        const size_t d = rowprog::degree(s), rows = rowprog::rows(s), wlen = rowprog::num_witness_columns(s) * rows;
        if (d == 0) return SRS_OK;
        if (ck && rows > ck->key.global_len)
            return fail(SRS_ERR_TOO_LONG_INPUT, "Can't commit too long input: input len: " + std::to_string(rows) +
                                                    ", but limit is " + std::to_string(ck->key.global_len));
        const bool host = space != SRS_SPACE_DEVICE;
        const bool own_T = host || !T_out;
        size_t need = 1024;
        if (host) need += 2 * Arena::pad(wlen * sizeof(fe_t));
        if (own_T) need += d * Arena::pad(rows * sizeof(fe_t));
        S->io.reserve(need);
        S->io.reset();
        const fe_t *dW1 = reinterpret_cast<const fe_t *>(W1), *dW2 = reinterpret_cast<const fe_t *>(W2);
        if (host) {
            fe_t *a = S->io.take<fe_t>(wlen ? wlen : 1), *b = S->io.take<fe_t>(wlen ? wlen : 1);
            SRS_HIP_CHECK(hipMemcpyAsync(a, W1, wlen * sizeof(fe_t), hipMemcpyHostToDevice, st));
            SRS_HIP_CHECK(hipMemcpyAsync(b, W2, wlen * sizeof(fe_t), hipMemcpyHostToDevice, st));
            dW1 = a;
            dW2 = b;
        }
        std::vector<fe_t *> dT(d);
        for (size_t k = 0; k < d; ++k) dT[k] = own_T ? S->io.take<fe_t>(rows) : reinterpret_cast<fe_t *>(T_out[k]);
        if (rowprog::shard_world(s) > 1)                 // rows of the other ranks' stripes are not evaluated: they read as zero, in the
            for (size_t k = 0; k < d; ++k)               // library's staging and in caller-provided device vectors alike (the error fold
                SRS_HIP_CHECK(hipMemsetAsync(dT[k], 0, rows * sizeof(fe_t), st));      // then leaves E unchanged outside the local stripes)
        std::string err;
        // with a key, the batched MSM below runs on the same stream and ends with a synchronisation
        int erc = rowprog::evaluate(s, 0, dW1, dW2, reinterpret_cast<const fe_t *>(challenges), n_challenges, dT.data(), st, err, ck == nullptr);
        if (erc) return fail(erc, "srs_cross_terms: " + err);
        if (ck) {
            std::vector<const srs_fe *> v(d);
            std::vector<size_t> nn(d, rows);
            for (size_t k = 0; k < d; ++k) v[k] = reinterpret_cast<const srs_fe *>(dT[k]);
            int crc = srs_commit_batch(ck, v.data(), nn.data(), d, SRS_SPACE_DEVICE, SRS_REPR_MONT, stream, commits_out);
            if (crc) return crc;
        }
        if (host && T_out) {
            for (size_t k = 0; k < d; ++k) SRS_HIP_CHECK(hipMemcpyAsync(T_out[k], dT[k], rows * sizeof(fe_t), hipMemcpyDeviceToHost, st));
            SRS_HIP_CHECK(hipStreamSynchronize(st));
        }
        return SRS_OK;
    });
}

int srs_cross_terms(srs_structure *S, const srs_fe *W1, const srs_fe *W2, const srs_fe *challenges, size_t n_challenges,
                    int space, void *stream, srs_fe *const *T_out) {
    if (!T_out) return fail(SRS_ERR_INVALID, "srs_cross_terms: T_out is NULL");
    return cross_terms_impl(S, nullptr, W1, W2, challenges, n_challenges, space, stream, T_out, nullptr);
}
int srs_commit_cross_terms(srs_structure *S, srs_ck *ck, const srs_fe *W1, const srs_fe *W2, const srs_fe *challenges,
                           size_t n_challenges, int space, void *stream, srs_fe *const *T_out, srs_affine *commits_out) {
    if (!ck) return fail(SRS_ERR_INVALID, "srs_commit_cross_terms: ck is NULL");
    return cross_terms_impl(S, ck, W1, W2, challenges, n_challenges, space, stream, T_out, commits_out);
}

int srs_eval_gates(srs_structure *S, int homogeneous, const srs_fe *W, const srs_fe *challenges, size_t n_challenges,
                   int space, void *stream, srs_fe *out) {
    if (!S || !W || !out || (n_challenges && !challenges)) return fail(SRS_ERR_INVALID, "srs_eval_gates: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        rowprog::Structure *s = S->s;
        const size_t rows = rowprog::rows(s), wlen = rowprog::num_witness_columns(s) * rows;
        const bool host = space != SRS_SPACE_DEVICE;
        const fe_t *dW = reinterpret_cast<const fe_t *>(W);
        fe_t *dO = reinterpret_cast<fe_t *>(out);
        if (host) {
            S->io.reserve(Arena::pad(wlen * sizeof(fe_t)) + Arena::pad(rows * sizeof(fe_t)) + 1024);
            S->io.reset();
            fe_t *a = S->io.take<fe_t>(wlen ? wlen : 1);
            SRS_HIP_CHECK(hipMemcpyAsync(a, W, wlen * sizeof(fe_t), hipMemcpyHostToDevice, st));
            dW = a;
            dO = S->io.take<fe_t>(rows);
        }
        std::string err;
        fe_t *outs[1] = {dO};
        int erc = rowprog::evaluate(s, homogeneous ? 2 : 1, dW, nullptr, reinterpret_cast<const fe_t *>(challenges), n_challenges, outs, st, err);
        if (erc) return fail(erc, "srs_eval_gates: " + err);
        if (host) {
            SRS_HIP_CHECK(hipMemcpyAsync(out, dO, rows * sizeof(fe_t), hipMemcpyDeviceToHost, st));
            SRS_HIP_CHECK(hipStreamSynchronize(st));
        }
        return SRS_OK;
    });
}

int srs_is_sat_gates(srs_structure *S, int homogeneous, const srs_fe *W, const srs_fe *challenges, size_t n_challenges,
                     const srs_fe *E, int space, void *stream, size_t *mismatch_count) {
    if (!S || !W || !mismatch_count || (n_challenges && !challenges) || (homogeneous && !E) || (!homogeneous && E))
        return fail(SRS_ERR_INVALID, "srs_is_sat_gates: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        rowprog::Structure *s = S->s;
        const size_t rows = rowprog::rows(s), wlen = rowprog::num_witness_columns(s) * rows;
        const bool host = space != SRS_SPACE_DEVICE;
        S->io.reserve(Arena::pad((wlen + 1) * sizeof(fe_t)) + 2 * Arena::pad(rows * sizeof(fe_t)) + 1024);
        S->io.reset();
        const fe_t *dW = reinterpret_cast<const fe_t *>(W), *dE = reinterpret_cast<const fe_t *>(E);
        if (host) {
            fe_t *a = S->io.take<fe_t>(wlen + 1);
            SRS_HIP_CHECK(hipMemcpyAsync(a, W, wlen * sizeof(fe_t), hipMemcpyHostToDevice, st));
            dW = a;
            if (E) {
                fe_t *e = S->io.take<fe_t>(rows);
                SRS_HIP_CHECK(hipMemcpyAsync(e, E, rows * sizeof(fe_t), hipMemcpyHostToDevice, st));
                dE = e;
            }
        }
        fe_t *vals = S->io.take<fe_t>(rows);
        fe_t *outs[1] = {vals};
        std::string err;
        int erc = rowprog::evaluate(s, homogeneous ? 2 : 1, dW, nullptr, reinterpret_cast<const fe_t *>(challenges), n_challenges, outs, st, err);
        if (erc) return fail(erc, "srs_is_sat_gates: " + err);
        *mismatch_count = rowprog::count_mismatch(vals, dE, rows, st);
        return SRS_OK;
    });
}

// ------------------------------------------------------------------ off-circuit random oracle (host code)
int srs_poseidon_new(int field, size_t t, size_t rate, size_t r_f, size_t r_p, srs_poseidon **out) {
    if (!valid_field(field) || !out) return fail(SRS_ERR_INVALID, "srs_poseidon_new: bad argument");
    return guarded([&]() -> int {
        std::string err;
        poseidon::Hash *h = poseidon::create(field, t, rate, r_f, r_p, err);
        if (!h) return fail(SRS_ERR_INVALID, "srs_poseidon_new: " + err);
        srs_poseidon *H = new srs_poseidon();
        H->h = h;
        *out = H;
        return SRS_OK;
    });
}
void srs_poseidon_free(srs_poseidon *H) {
    if (!H) return;
    delete H->h;
    delete H;
}
void srs_poseidon_reset(srs_poseidon *H) {
    if (!H) return;
    H->h->buf.clear();
    H->h->state.clear();
    H->h->done = 0;
}
int srs_poseidon_absorb_field(srs_poseidon *H, const srs_fe *v, size_t n) {
    if (!H || (n && !v)) return fail(SRS_ERR_INVALID, "srs_poseidon_absorb_field: bad argument");
    return guarded([&]() -> int {
        poseidon::absorb(*H->h, reinterpret_cast<const fe_t *>(v), n);
        return SRS_OK;
    });
}
int srs_poseidon_absorb_point(srs_poseidon *H, int curve, const srs_affine *p) {
    if (!H || !valid_curve(curve) || !p) return fail(SRS_ERR_INVALID, "srs_poseidon_absorb_point: bad argument");
    // coordinates live in the curve's base field: bn256 -> Fq (1), grumpkin -> Fr (0); util::fe_to_fe between different
    // fields (poseidon_hash.rs:130-133) is not provided
    if (H->h->field != (curve == SRS_CURVE_BN256 ? SRS_FIELD_FQ : SRS_FIELD_FR))
        return fail(SRS_ERR_INVALID, "srs_poseidon_absorb_point: the oracle's field is not the curve's base field");
    return guarded([&]() -> int {
        poseidon::absorb(*H->h, reinterpret_cast<const fe_t *>(p), 2);       // (x, y); the identity is (0, 0) (:137-139)
        return SRS_OK;
    });
}
int srs_poseidon_squeeze(srs_poseidon *H, size_t num_bits, int out_field, srs_fe *out) {
    if (!H || !out || !valid_field(out_field)) return fail(SRS_ERR_INVALID, "srs_poseidon_squeeze: bad argument");
    return guarded([&]() -> int {
        std::string err;
        fe_t o;
        if (!poseidon::squeeze(*H->h, num_bits, out_field, o, err)) return fail(SRS_ERR_INVALID, "srs_poseidon_squeeze: " + err);
        std::memcpy(out, &o, 32);
        return SRS_OK;
    });
}
int srs_poseidon_squeeze_device(srs_poseidon *H, size_t num_bits, int out_field, srs_fe *out, double *kernel_ms) {
    if (!H || !out || !valid_field(out_field)) return fail(SRS_ERR_INVALID, "srs_poseidon_squeeze_device: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        std::string err;
        fe_t o;
        if (!poseidon::squeeze_device(*H->h, num_bits, out_field, o, kernel_ms, err))
            return fail(SRS_ERR_INVALID, "srs_poseidon_squeeze_device: " + err);
        std::memcpy(out, &o, 32);
        return SRS_OK;
    });
}

// ------------------------------------------------------------------ deciders
int srs_sparse_create(int field, size_t n, const uint64_t *rows, const uint64_t *cols, const srs_fe *values, size_t nnz, srs_sparse **out) {
    if (!valid_field(field) || !out || (nnz && (!rows || !cols || !values))) return fail(SRS_ERR_INVALID, "srs_sparse_create: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        std::string err;
        int crc = 0;
        decider::Sparse *m = decider::create(field, n, rows, cols, reinterpret_cast<const fe_t *>(values), nnz, crc, err);
        if (!m) return fail(crc ? crc : SRS_ERR_INVALID, "srs_sparse_create: " + err);
        srs_sparse *M = new srs_sparse();
        M->m = m;
        *out = M;
        return SRS_OK;
    });
}
void srs_sparse_free(srs_sparse *M) {
    if (!M) return;
    decider::destroy(M->m);
    M->io.release();
    delete M;
}
static int sparse_apply(srs_sparse *M, const srs_fe *Z, int space, void *stream, srs_fe *y, size_t *mismatch_count) {
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        const size_t n = decider::dim(M->m);
        const bool host = space != SRS_SPACE_DEVICE;
        const fe_t *dZ = reinterpret_cast<const fe_t *>(Z);
        fe_t *dY = reinterpret_cast<fe_t *>(y);
        if (host) {
            M->io.reserve(2 * Arena::pad((n + 1) * sizeof(fe_t)) + 1024);
            M->io.reset();
            fe_t *a = M->io.take<fe_t>(n + 1);
            SRS_HIP_CHECK(hipMemcpyAsync(a, Z, n * sizeof(fe_t), hipMemcpyHostToDevice, st));
            dZ = a;
            if (y) dY = M->io.take<fe_t>(n + 1);
        }
        if (mismatch_count) *mismatch_count = decider::permutation_mismatches(M->m, dZ, st);
        else decider::matvec(M->m, dZ, dY, st);
        if (host && y) {
            SRS_HIP_CHECK(hipMemcpyAsync(y, dY, n * sizeof(fe_t), hipMemcpyDeviceToHost, st));
            SRS_HIP_CHECK(hipStreamSynchronize(st));
        }
        return SRS_OK;
    });
}
int srs_sparse_matvec(srs_sparse *M, const srs_fe *Z, int space, void *stream, srs_fe *y) {
    if (!M || (decider::dim(M->m) && (!Z || !y))) return fail(SRS_ERR_INVALID, "srs_sparse_matvec: bad argument");
    return sparse_apply(M, Z, space, stream, y, nullptr);
}
int srs_is_sat_permutation(srs_sparse *M, const srs_fe *Z, int space, void *stream, size_t *mismatch_count) {
    if (!M || !mismatch_count || (decider::dim(M->m) && !Z)) return fail(SRS_ERR_INVALID, "srs_is_sat_permutation: bad argument");
    return sparse_apply(M, Z, space, stream, nullptr, mismatch_count);
}
int srs_is_sat_witness_commit(srs_ck *ck, const srs_fe *const *W, const size_t *n, size_t n_rounds, const srs_affine *W_commitments,
                              const srs_fe *E, size_t n_E, const srs_affine *E_commitment, int space, void *stream,
                              size_t *w_mismatch_count, int *e_mismatch) {
    if (!ck || !w_mismatch_count || (n_rounds && (!W || !n || !W_commitments)) || (E && (!E_commitment || !e_mismatch)))
        return fail(SRS_ERR_INVALID, "srs_is_sat_witness_commit: bad argument");
    if (ck->shards.empty() && ck->key.world > 1)       // a rank's partial sum never equals the full commitment
        return fail(SRS_ERR_INVALID, "srs_is_sat_witness_commit: key sharded across processes (combine the partial commitments with "
                                     "srs_point_sum and compare on the caller's side, or use a multi-device key)");
    std::vector<const srs_fe *> v(W, W + n_rounds);
    std::vector<size_t> nn(n, n + n_rounds);
    if (E) { v.push_back(E); nn.push_back(n_E); }
    std::vector<srs_affine> got(v.size());
    if (!v.empty()) {
        int rc = srs_commit_batch(ck, v.data(), nn.data(), v.size(), space, SRS_REPR_MONT, stream, got.data());
        if (rc) return rc;
    }
    size_t bad = 0;
    for (size_t i = 0; i < n_rounds; ++i) bad += std::memcmp(&got[i], &W_commitments[i], sizeof(srs_affine)) != 0;
    *w_mismatch_count = bad;
    if (e_mismatch) *e_mismatch = E ? (std::memcmp(&got[n_rounds], E_commitment, sizeof(srs_affine)) != 0) : 0;
    return SRS_OK;
}

// ------------------------------------------------------------------ lookup arguments
int srs_lookup_coeff_1(srs_structure *S, const srs_fe *advice, const srs_fe *r, int space, void *stream, srs_fe *const *ls,
                       srs_fe *const *ts, srs_fe *const *ms) {
    if (!S || !advice || !r || !ls || !ts || !ms) return fail(SRS_ERR_INVALID, "srs_lookup_coeff_1: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        rowprog::Structure *s = S->s;
        const size_t L = rowprog::num_lookups(s), rows = rowprog::rows(s), alen = rowprog::num_advice(s) * rows;
        if (!L) return fail(SRS_ERR_INVALID, "srs_lookup_coeff_1: structure has no lookup arguments");
        const bool host = space != SRS_SPACE_DEVICE;
        std::vector<fe_t *> dl(L), dt(L), dm(L);
        const fe_t *dA = reinterpret_cast<const fe_t *>(advice);
        if (host) {
            S->io.reserve(Arena::pad((alen + 1) * sizeof(fe_t)) + 3 * L * Arena::pad(rows * sizeof(fe_t)) + 1024);
            S->io.reset();
            fe_t *a = S->io.take<fe_t>(alen + 1);
            SRS_HIP_CHECK(hipMemcpyAsync(a, advice, alen * sizeof(fe_t), hipMemcpyHostToDevice, st));
            dA = a;
            for (size_t i = 0; i < L; ++i) { dl[i] = S->io.take<fe_t>(rows); dt[i] = S->io.take<fe_t>(rows); dm[i] = S->io.take<fe_t>(rows); }
        } else {
            for (size_t i = 0; i < L; ++i) {
                dl[i] = reinterpret_cast<fe_t *>(ls[i]); dt[i] = reinterpret_cast<fe_t *>(ts[i]); dm[i] = reinterpret_cast<fe_t *>(ms[i]);
            }
        }
        fe_t rr;
        std::memcpy(&rr, r, 32);
        std::string err;
        int erc = rowprog::lookup_coeff_1(s, dA, rr, dl.data(), dt.data(), dm.data(), st, err);
        if (erc) return fail(erc, "srs_lookup_coeff_1: " + err);
        if (host) {
            for (size_t i = 0; i < L; ++i) {
                SRS_HIP_CHECK(hipMemcpyAsync(ls[i], dl[i], rows * sizeof(fe_t), hipMemcpyDeviceToHost, st));
                SRS_HIP_CHECK(hipMemcpyAsync(ts[i], dt[i], rows * sizeof(fe_t), hipMemcpyDeviceToHost, st));
                SRS_HIP_CHECK(hipMemcpyAsync(ms[i], dm[i], rows * sizeof(fe_t), hipMemcpyDeviceToHost, st));
            }
            SRS_HIP_CHECK(hipStreamSynchronize(st));
        }
        return SRS_OK;
    });
}

int srs_lookup_coeff_2(int field, const srs_fe *l, const srs_fe *t, const srs_fe *m, const srs_fe *r, size_t n, int space,
                       void *stream, srs_fe *h, srs_fe *g) {
    if (!valid_field(field) || !r || n > 0xFFFFFFFFull || (n && (!l || !t || !m || !h || !g)))
        return fail(SRS_ERR_INVALID, "srs_lookup_coeff_2: bad argument");
    if (!n) return SRS_OK;
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        fe_t rr;
        std::memcpy(&rr, r, 32);
        if (space == SRS_SPACE_DEVICE) {
            rowprog::lookup_coeff_2(field, reinterpret_cast<const fe_t *>(l), reinterpret_cast<const fe_t *>(t),
                                    reinterpret_cast<const fe_t *>(m), rr, n, reinterpret_cast<fe_t *>(h), reinterpret_cast<fe_t *>(g), st);
            SRS_HIP_CHECK(hipStreamSynchronize(st));
        } else {
            fe_t *buf = nullptr;
            SRS_HIP_CHECK(hipMalloc((void **)&buf, 5 * n * sizeof(fe_t)));
            try {
                SRS_HIP_CHECK(hipMemcpyAsync(buf, l, n * sizeof(fe_t), hipMemcpyHostToDevice, st));
                SRS_HIP_CHECK(hipMemcpyAsync(buf + n, t, n * sizeof(fe_t), hipMemcpyHostToDevice, st));
                SRS_HIP_CHECK(hipMemcpyAsync(buf + 2 * n, m, n * sizeof(fe_t), hipMemcpyHostToDevice, st));
                rowprog::lookup_coeff_2(field, buf, buf + n, buf + 2 * n, rr, n, buf + 3 * n, buf + 4 * n, st);
                SRS_HIP_CHECK(hipMemcpyAsync(h, buf + 3 * n, n * sizeof(fe_t), hipMemcpyDeviceToHost, st));
                SRS_HIP_CHECK(hipMemcpyAsync(g, buf + 4 * n, n * sizeof(fe_t), hipMemcpyDeviceToHost, st));
                SRS_HIP_CHECK(hipStreamSynchronize(st));
            } catch (...) { (void)hipFree(buf); throw; }
            (void)hipFree(buf);
        }
        SRS_HIP_CHECK(hipGetLastError());
        prof::collect();
        return SRS_OK;
    });
}

int srs_batch_invert_assigned(int field, const srs_fe *numerators, const srs_fe *denominators, const uint8_t *has_denominator,
                              size_t n, int space, void *stream, srs_fe *out) {
    if (!valid_field(field) || n > 0xFFFFFFFFull || (n && (!numerators || !denominators || !out)))
        return fail(SRS_ERR_INVALID, "srs_batch_invert_assigned: bad argument");
    if (!n) return SRS_OK;
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        if (space == SRS_SPACE_DEVICE) {
            rowprog::assigned_invert(field, reinterpret_cast<const fe_t *>(numerators), reinterpret_cast<const fe_t *>(denominators),
                                     has_denominator, n, reinterpret_cast<fe_t *>(out), st);
            SRS_HIP_CHECK(hipStreamSynchronize(st));
        } else {
            g_scratch.reserve(3 * Arena::pad(n * sizeof(fe_t)) + Arena::pad(n) + 256);
            g_scratch.reset();
            fe_t *a = g_scratch.take<fe_t>(n), *b = g_scratch.take<fe_t>(n), *o = g_scratch.take<fe_t>(n);
            uint8_t *h = nullptr;
            SRS_HIP_CHECK(hipMemcpyAsync(a, numerators, n * sizeof(fe_t), hipMemcpyHostToDevice, st));
            SRS_HIP_CHECK(hipMemcpyAsync(b, denominators, n * sizeof(fe_t), hipMemcpyHostToDevice, st));
            if (has_denominator) {
                h = g_scratch.take<uint8_t>(n);
                SRS_HIP_CHECK(hipMemcpyAsync(h, has_denominator, n, hipMemcpyHostToDevice, st));
            }
            rowprog::assigned_invert(field, a, b, h, n, o, st);
            SRS_HIP_CHECK(hipMemcpyAsync(out, o, n * sizeof(fe_t), hipMemcpyDeviceToHost, st));
            SRS_HIP_CHECK(hipStreamSynchronize(st));
        }
        SRS_HIP_CHECK(hipGetLastError());
        return SRS_OK;
    });
}

int srs_is_sat_log_derivative(srs_structure *S, const srs_fe *W, int space, void *stream, size_t *mismatch_count) {
    if (!S || !W || !mismatch_count) return fail(SRS_ERR_INVALID, "srs_is_sat_log_derivative: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        rowprog::Structure *s = S->s;
        const size_t wlen = rowprog::num_witness_columns(s) * rowprog::rows(s);
        const fe_t *dW = reinterpret_cast<const fe_t *>(W);
        if (space != SRS_SPACE_DEVICE && rowprog::num_lookups(s)) {
            S->io.reserve(Arena::pad((wlen + 1) * sizeof(fe_t)) + 1024);
            S->io.reset();
            fe_t *a = S->io.take<fe_t>(wlen + 1);
            SRS_HIP_CHECK(hipMemcpyAsync(a, W, wlen * sizeof(fe_t), hipMemcpyHostToDevice, st));
            dW = a;
        }
        *mismatch_count = rowprog::log_derivative_mismatches(s, dW, st);
        return SRS_OK;
    });
}

// ------------------------------------------------------------------ folds
int srs_fold_witness(int field, srs_fe *out, const srs_fe *w1, const srs_fe *w2, const srs_fe *r, size_t n, int space, void *stream) {
    if (!valid_field(field) || !r || (n && (!out || !w1 || !w2))) return fail(SRS_ERR_INVALID, "srs_fold_witness: bad argument");
    if (n == 0) return SRS_OK;
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        fe_t rr;
        std::memcpy(&rr, r, 32);
        if (space == SRS_SPACE_DEVICE) {
            // device-resident operands: stream-ordered (the result is ready in `stream` order, like any library kernel)
            rowprog::fold_w(field, reinterpret_cast<fe_t *>(out), reinterpret_cast<const fe_t *>(w1), reinterpret_cast<const fe_t *>(w2), rr, n, st);
        } else {
            g_scratch.reserve(2 * Arena::pad(n * sizeof(fe_t)) + 256);
            g_scratch.reset();
            fe_t *a = g_scratch.take<fe_t>(n), *b = g_scratch.take<fe_t>(n);
            SRS_HIP_CHECK(hipMemcpyAsync(a, w1, n * sizeof(fe_t), hipMemcpyHostToDevice, st));
            SRS_HIP_CHECK(hipMemcpyAsync(b, w2, n * sizeof(fe_t), hipMemcpyHostToDevice, st));
            rowprog::fold_w(field, a, a, b, rr, n, st);
            SRS_HIP_CHECK(hipMemcpyAsync(out, a, n * sizeof(fe_t), hipMemcpyDeviceToHost, st));
            SRS_HIP_CHECK(hipStreamSynchronize(st));
        }
        SRS_HIP_CHECK(hipGetLastError());
        return SRS_OK;
    });
}

int srs_fold_error(int field, srs_fe *out, const srs_fe *e, const srs_fe *const *T, size_t n_terms, const srs_fe *r, size_t n,
                   int space, void *stream) {
    if (!valid_field(field) || !r || (n && (!out || !e)) || (n_terms && !T)) return fail(SRS_ERR_INVALID, "srs_fold_error: bad argument");
    if (n == 0) return SRS_OK;
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        fe_t rr;
        std::memcpy(&rr, r, 32);
        std::string err;
        if (space == SRS_SPACE_DEVICE) {
            int erc = rowprog::fold_e(field, reinterpret_cast<fe_t *>(out), reinterpret_cast<const fe_t *>(e),
                                      reinterpret_cast<const fe_t *const *>(T), n_terms, rr, n, st, err);
            if (erc) return fail(erc, "srs_fold_error: " + err);   // stream-ordered, see srs_fold_witness
        } else {
            g_scratch.reserve((n_terms + 1) * Arena::pad(n * sizeof(fe_t)) + 256);
            g_scratch.reset();
            fe_t *buf = g_scratch.take<fe_t>(n);
            SRS_HIP_CHECK(hipMemcpyAsync(buf, e, n * sizeof(fe_t), hipMemcpyHostToDevice, st));
            std::vector<const fe_t *> tp(n_terms);
            for (size_t k = 0; k < n_terms; ++k) {
                fe_t *t = g_scratch.take<fe_t>(n);
                SRS_HIP_CHECK(hipMemcpyAsync(t, T[k], n * sizeof(fe_t), hipMemcpyHostToDevice, st));
                tp[k] = t;
            }
            int erc = rowprog::fold_e(field, buf, buf, tp.data(), n_terms, rr, n, st, err);
            if (erc) return fail(erc, "srs_fold_error: " + err);
            SRS_HIP_CHECK(hipMemcpyAsync(out, buf, n * sizeof(fe_t), hipMemcpyDeviceToHost, st));
            SRS_HIP_CHECK(hipStreamSynchronize(st));
        }
        SRS_HIP_CHECK(hipGetLastError());
        return SRS_OK;
    });
}

// ------------------------------------------------------------------ ProtoGalaxy
int srs_pg_context_new(const srs_structure *S, size_t traces_len, srs_pg_context *out) {
    if (!S || !out) return fail(SRS_ERR_INVALID, "srs_pg_context_new: bad argument");
    if (((traces_len + 1) & traces_len) != 0) return fail(SRS_ERR_INVALID, "instances_to_fold must be a power of two");   // poly/mod.rs:225
    rowprog::PgSizes z;
    if (!rowprog::pg_sizes(S->s, traces_len, z)) return fail(SRS_ERR_INVALID, "structure has no gates");
    out->count_of_evaluation_with_padding = z.count_with_padding;
    out->betas_count = z.betas_count;
    out->fft_points_count_F = z.points_F;
    out->fft_points_count_G = z.points_G;
    out->instances_to_fold = z.instances_to_fold;
    out->lagrange_domain = z.lagrange_domain;
    out->fft_log_domain_size_K = z.log_domain_K;
    return SRS_OK;
}

static int pg_impl(srs_structure *S, int mode, const srs_fe *weights, size_t n_weights, const srs_fe *delta, const srs_fe *const *W,
                   const srs_fe *const *challenges, size_t n_challenges, size_t J, int space, int compat, void *stream, srs_fe *out) {
    if (!S || !weights || !W || !out || (n_challenges && !challenges)) return fail(SRS_ERR_INVALID, "srs_pg_*: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        rowprog::Structure *s = S->s;
        const size_t wlen = rowprog::num_witness_columns(s) * rowprog::rows(s);
        std::vector<const fe_t *> dW(J);
        if (space == SRS_SPACE_DEVICE) {
            for (size_t j = 0; j < J; ++j) dW[j] = reinterpret_cast<const fe_t *>(W[j]);
        } else {
            S->io.reserve(J * Arena::pad((wlen + 1) * sizeof(fe_t)) + 1024);
            S->io.reset();
            for (size_t j = 0; j < J; ++j) {
                fe_t *d = S->io.take<fe_t>(wlen + 1);
                SRS_HIP_CHECK(hipMemcpyAsync(d, W[j], wlen * sizeof(fe_t), hipMemcpyHostToDevice, st));
                dW[j] = d;
            }
        }
        std::vector<const fe_t *> ch(J);
        for (size_t j = 0; j < J; ++j) ch[j] = n_challenges ? reinterpret_cast<const fe_t *>(challenges[j]) : nullptr;
        std::string err;
        size_t n_out = 0;
        int erc = rowprog::pg_sum(s, mode, dW.data(), ch.data(), n_challenges, J, reinterpret_cast<const fe_t *>(weights), n_weights,
                                  reinterpret_cast<const fe_t *>(delta), compat, st, reinterpret_cast<fe_t *>(out), &n_out, err);
        if (erc) return fail(erc, "srs_pg: " + err);
        return SRS_OK;
    });
}

int srs_pg_compute_F(srs_structure *S, const srs_fe *betas, size_t n_betas, const srs_fe *delta, const srs_fe *W,
                     const srs_fe *challenges, size_t n_challenges, int space, int reference_compat, void *stream, srs_fe *poly_F) {
    if (!delta) return fail(SRS_ERR_INVALID, "srs_pg_compute_F: delta is NULL");
    const srs_fe *Ws[1] = {W};
    const srs_fe *chs[1] = {challenges};
    return pg_impl(S, 0, betas, n_betas, delta, Ws, chs, n_challenges, 1, space, reference_compat, stream, poly_F);
}
int srs_pg_compute_G(srs_structure *S, const srs_fe *betas_stroke, size_t n_betas, const srs_fe *const *W,
                     const srs_fe *const *challenges, size_t n_challenges, size_t n_instances, int space, int reference_compat,
                     void *stream, srs_fe *poly_G) {
    if (n_instances < 2) return fail(SRS_ERR_INVALID, "You can't fold 0 traces");                // poly/mod.rs:27
    if (n_instances & (n_instances - 1)) return fail(SRS_ERR_INVALID, "instances_to_fold must be a power of two");
    return pg_impl(S, 1, betas_stroke, n_betas, nullptr, W, challenges, n_challenges, n_instances, space, reference_compat, stream, poly_G);
}
int srs_pg_evaluate_e(srs_structure *S, const srs_fe *betas, size_t n_betas, const srs_fe *W, const srs_fe *challenges,
                      size_t n_challenges, int space, int reference_compat, void *stream, srs_fe *e) {
    const srs_fe *Ws[1] = {W};
    const srs_fe *chs[1] = {challenges};
    return pg_impl(S, 2, betas, n_betas, nullptr, Ws, chs, n_challenges, 1, space, reference_compat, stream, e);
}

int srs_pg_compute_K_from_G(const srs_fe *poly_G, size_t n_G, const srs_fe *poly_F_in_alpha, size_t instances_to_fold,
                            uint32_t fft_log_domain_size_K, void *stream, srs_fe *poly_K) {
    if (!poly_G || !poly_F_in_alpha || !poly_K || instances_to_fold == 0) return fail(SRS_ERR_INVALID, "srs_pg_compute_K_from_G: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        fe_t fa;
        std::memcpy(&fa, poly_F_in_alpha, 32);
        std::string err;
        int erc = rowprog::pg_K_from_G(reinterpret_cast<const fe_t *>(poly_G), n_G, fa, instances_to_fold, fft_log_domain_size_K,
                                       (hipStream_t)stream, reinterpret_cast<fe_t *>(poly_K), err);
        if (erc) return fail(erc, "srs_pg_compute_K_from_G: " + err);
        return SRS_OK;
    });
}

int srs_pg_beta_stroke(const srs_fe *betas, size_t n, const srs_fe *alpha, const srs_fe *delta, srs_fe *out) {
    if ((n && (!betas || !out)) || !alpha || !delta) return fail(SRS_ERR_INVALID, "srs_pg_beta_stroke: bad argument");
    fe_t a, d;
    std::memcpy(&a, alpha, 32);
    std::memcpy(&d, delta, 32);
    for (size_t i = 0; i < n; ++i) {                   // BetaStrokeIter: next = beta[i] + alpha * delta^(2^i)   (poly/mod.rs:449-462)
        fe_t b;
        std::memcpy(&b, &betas[i], 32);
        b = Fr::add(b, Fr::mul(a, d));
        std::memcpy(&out[i], &b, 32);
        d = Fr::sqr(d);
    }
    return SRS_OK;
}

int srs_lagrange_eval(const srs_fe *X, uint32_t log_n, srs_fe *out) {
    if (!X || !out || log_n > ntt::FR_S) return fail(SRS_ERR_INVALID, "srs_lagrange_eval: bad argument");
    fe_t x;
    std::memcpy(&x, X, 32);
    std::vector<fe_t> v = rowprog::lagrange_eval(x, log_n);
    std::memcpy(out, v.data(), v.size() * sizeof(fe_t));
    return SRS_OK;
}
int srs_poly_eval(const srs_fe *coeffs, size_t n, const srs_fe *x, srs_fe *out) {
    if ((n && !coeffs) || !x || !out) return fail(SRS_ERR_INVALID, "srs_poly_eval: bad argument");
    fe_t xx;
    std::memcpy(&xx, x, 32);
    fe_t r = rowprog::poly_eval(reinterpret_cast<const fe_t *>(coeffs), n, xx);
    std::memcpy(out, &r, 32);
    return SRS_OK;
}
int srs_pg_calculate_e(const srs_fe *poly_F, size_t n_F, const srs_fe *poly_K, size_t n_K, const srs_fe *gamma,
                       const srs_fe *alpha, uint32_t log_n, srs_fe *out) {
    if (!poly_F || !poly_K || !gamma || !alpha || !out) return fail(SRS_ERR_INVALID, "srs_pg_calculate_e: bad argument");
    fe_t g, a;
    std::memcpy(&g, gamma, 32);
    std::memcpy(&a, alpha, 32);
    fe_t l0 = rowprog::lagrange_eval(g, log_n)[0];
    fe_t fa = rowprog::poly_eval(reinterpret_cast<const fe_t *>(poly_F), n_F, a);
    fe_t z = Fr::sub(Fr::pow_u64(g, (uint64_t)1 << log_n), Fr::one());      // eval_vanish_polynomial, lagrange.rs:83-85
    fe_t kg = rowprog::poly_eval(reinterpret_cast<const fe_t *>(poly_K), n_K, g);
    fe_t r = Fr::add(Fr::mul(fa, l0), Fr::mul(z, kg));
    std::memcpy(out, &r, 32);
    return SRS_OK;
}

// ProtoGalaxy::prove (src/nifs/protogalaxy/mod.rs:400-481) as ONE call: the five polynomial / fold steps and the two challenges
// between them without a round trip through the caller per step.
int srs_pg_prove(srs_structure *S, srs_poseidon *ro, const srs_fe *betas, size_t n_betas, const srs_fe *delta,
                 const srs_fe *const *W, const srs_fe *const *challenges, size_t n_challenges, size_t n_instances, int reference_compat,
                 void *stream, srs_fe *alpha_gamma, srs_fe *poly_F, srs_fe *poly_K, srs_fe *betas_stroke, srs_fe *e, srs_fe *lagrange,
                 srs_fe *W_folded) {
    if (!S || !betas || !delta || !W || !alpha_gamma || !poly_F || !poly_K || !betas_stroke || !e || !lagrange ||
        (n_challenges && !challenges))
        return fail(SRS_ERR_INVALID, "srs_pg_prove: bad argument");
    if (n_instances < 2) return fail(SRS_ERR_INVALID, "You can't fold 0 traces");                // poly/mod.rs:27
    if (n_instances & (n_instances - 1)) return fail(SRS_ERR_INVALID, "instances_to_fold must be a power of two");
    if (ro && ro->h->field != SRS_FIELD_FR) return fail(SRS_ERR_INVALID, "srs_pg_prove: the oracle must be over bn256::Fr");
    if (rowprog::shard_world(S->s) > 1)   // the polynomials of a sharded structure are PARTIAL sums: alpha / gamma need the exchanged ones
        return fail(SRS_ERR_INVALID, "srs_pg_prove: structure is row-sharded (srs_structure_set_shard); use the step-wise calls and add the ranks' polynomials");
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        HostTrace ht("srs_pg_prove");
        hipStream_t st = (hipStream_t)stream;
        rowprog::Structure *s = S->s;
        rowprog::PgSizes z;
        if (!rowprog::pg_sizes(s, n_instances - 1, z)) return fail(SRS_ERR_INVALID, "structure has no gates");
        if (n_betas < z.betas_count) return fail(SRS_ERR_INVALID, "srs_pg_prove: not enough betas");
        // K's domain "log" (quirk Q2) above F::S: the reference gets as far as coset_ifft and panics there (src/fft.rs:13); refused here
        // before anything is computed or written
        if (z.log_domain_K > ntt::FR_S)
            return fail(SRS_ERR_K_TOO_LARGE, "srs_pg_prove (compute_K_from_G): k=" + std::to_string(z.log_domain_K) + " should no larger than F::S=28");
        const size_t wlen = rowprog::num_witness_columns(s) * rowprog::rows(s);
        std::vector<const fe_t *> dW(n_instances), ch(n_instances);
        for (size_t j = 0; j < n_instances; ++j) {
            dW[j] = reinterpret_cast<const fe_t *>(W[j]);
            ch[j] = n_challenges ? reinterpret_cast<const fe_t *>(challenges[j]) : nullptr;
        }
        std::string err;
        size_t n_out = 0;
        // poly_F = compute_F(betas, delta, accumulator)                                                   :417-422
        int erc = rowprog::pg_sum(s, 0, dW.data(), ch.data(), n_challenges, 1, reinterpret_cast<const fe_t *>(betas), n_betas,
                                  reinterpret_cast<const fe_t *>(delta), reference_compat, st, reinterpret_cast<fe_t *>(poly_F), &n_out, err);
        if (erc) return fail(erc, "srs_pg_prove (compute_F): " + err);
        ht.mark("compute_F");
        // alpha = ro.absorb(poly_F).squeeze(MAX_BITS)                                                     :424-427
        fe_t alpha, gamma;
        std::memcpy(&alpha, &alpha_gamma[0], 32);
        std::memcpy(&gamma, &alpha_gamma[1], 32);
        if (ro) {
            poseidon::absorb(*ro->h, reinterpret_cast<const fe_t *>(poly_F), z.points_F);
            if (!poseidon::squeeze(*ro->h, 255, SRS_FIELD_FR, alpha, err)) return fail(SRS_ERR_INVALID, "srs_pg_prove (alpha): " + err);
        }
        // betas_stroke (BetaStrokeIter :449-462), F(alpha)
        std::vector<fe_t> bs(z.betas_count);
        {
            fe_t d;
            std::memcpy(&d, delta, 32);
            for (size_t i = 0; i < z.betas_count; ++i) {
                fe_t b;
                std::memcpy(&b, &betas[i], 32);
                bs[i] = Fr::add(b, Fr::mul(alpha, d));
                d = Fr::sqr(d);
            }
        }
        const fe_t f_alpha = rowprog::poly_eval(reinterpret_cast<const fe_t *>(poly_F), z.points_F, alpha);
        ht.mark("alpha,betas'");
        // poly_K = compute_K(F(alpha), betas_stroke, accumulator, incoming)                               :437-443
        std::vector<fe_t> poly_G(z.points_G);
        // G(1) = F(alpha) by definition (rowprog.hip, pg_sum): one evaluation point less for the leaf kernel.
        // r06: with one incoming trace and a small K domain (every BASELINE config: 256 points) G's values stay on the device and K's points,
        // the coset ifft and the copy of K follow in the same chain of launches -- one synchronisation instead of two, no host Horner
        rowprog::PgGValues gv;
        const bool k_on_device = n_instances == 2 && z.log_domain_K <= 12;
        erc = rowprog::pg_sum(s, 1, dW.data(), ch.data(), n_challenges, n_instances, bs.data(), bs.size(), nullptr, reference_compat, st,
                              poly_G.data(), &n_out, err, &f_alpha, k_on_device ? &gv : nullptr);
        if (erc) return fail(erc, "srs_pg_prove (compute_G): " + err);
        if (gv.vals_dev && !rowprog::pg_K_device_ok(gv, z.log_domain_K)) return fail(SRS_ERR_DEVICE, "srs_pg_prove: internal (device K on an unsupported shape)");
        ht.mark("compute_G");
        if (gv.vals_dev)
            erc = rowprog::pg_K_from_G_device(s, gv, f_alpha, z.instances_to_fold, z.log_domain_K, st, reinterpret_cast<fe_t *>(poly_K), err);
        else
            erc = rowprog::pg_K_from_G(poly_G.data(), poly_G.size(), f_alpha, z.instances_to_fold, z.log_domain_K, st,
                                       reinterpret_cast<fe_t *>(poly_K), err);
        if (erc) return fail(erc, "srs_pg_prove (compute_K_from_G): " + err);
        ht.mark("compute_K");
        const size_t n_K = (size_t)1 << z.log_domain_K;
        // gamma = ro.absorb(poly_K).squeeze(MAX_BITS)                                                     :445-448
        if (ro) {
            poseidon::absorb(*ro->h, reinterpret_cast<const fe_t *>(poly_K), n_K);
            if (!poseidon::squeeze(*ro->h, 255, SRS_FIELD_FR, gamma, err)) return fail(SRS_ERR_INVALID, "srs_pg_prove (gamma): " + err);
        }
        ht.mark("gamma");
        // L_j(gamma), e = F(alpha) L_0(gamma) + Z(gamma) K(gamma)  (calculate_e :748-764), fold_witness :176-210
        const std::vector<fe_t> L = rowprog::lagrange_eval(gamma, (uint32_t)z.lagrange_domain);
        const fe_t zg = Fr::sub(Fr::pow_u64(gamma, (uint64_t)1 << z.lagrange_domain), Fr::one());
        const fe_t kg = rowprog::poly_eval(reinterpret_cast<const fe_t *>(poly_K), n_K, gamma);
        const fe_t ev = Fr::add(Fr::mul(f_alpha, L[0]), Fr::mul(zg, kg));
        if (W_folded) {      // NULL: the caller folds later (srs_fold_lincomb with `lagrange`), e.g. under the next witness upload
            erc = rowprog::lincomb(SRS_FIELD_FR, reinterpret_cast<fe_t *>(W_folded), dW.data(), L.data(), n_instances, wlen, st, err);
            if (erc) return fail(erc, "srs_pg_prove (fold_witness): " + err);
        }
        std::memcpy(&alpha_gamma[0], &alpha, 32);
        std::memcpy(&alpha_gamma[1], &gamma, 32);
        std::memcpy(betas_stroke, bs.data(), bs.size() * sizeof(fe_t));
        std::memcpy(e, &ev, 32);
        std::memcpy(lagrange, L.data(), n_instances * sizeof(fe_t));
        ht.mark("e,L,fold");
        return SRS_OK;
    });
}

// VanillaFS::prove (src/nifs/sangria/mod.rs:253-277) as ONE call on device-resident traces: cross terms + their commitments, the
// challenge, the witness / error folds (in place) and the instance fold (host workers, joined with srs_job_wait).
// VanillaFS::prove on device-resident traces.  `incoming`: the incoming trace is still on the host (W2_host -> W2, uploaded
// here), its commitment is not known yet and is computed IN THE SAME batched MSM as the cross terms' (one chain of MSM
// launches for d+1 vectors instead of two chains), written to W_commitments[1] and absorbed -- followed by `u2_tail` -- before
// the cross-term commitments.
static int sangria_prove_impl(srs_structure *S, srs_ck *ck, srs_poseidon *ro, const srs_fe *challenges, size_t n_challenges, srs_fe *W1, srs_fe *W2,
                              const srs_fe *W2_host, const srs_fe *u2_tail, size_t n_u2_tail, srs_fe *E, void *stream, srs_fe *r_io,
                              srs_fe *const *T_dev, srs_affine *cross_term_commits, srs_affine *W_commitments, const srs_affine *E_commitment,
                              srs_affine *folded_commitments, uint64_t *jobs, bool incoming) {
    if (!S || !ck || !W1 || !W2 || !E || !r_io || !T_dev || !cross_term_commits || !W_commitments || !E_commitment || !folded_commitments || !jobs ||
        (n_u2_tail && !u2_tail))
        return fail(SRS_ERR_INVALID, "srs_sangria_prove: bad argument");
    const int curve = ck->key.curve, sf = srs_scalar_field_of(curve);
    if (ro && ro->h->field != (curve == SRS_CURVE_BN256 ? SRS_FIELD_FQ : SRS_FIELD_FR))
        return fail(SRS_ERR_INVALID, "srs_sangria_prove: the oracle's field is not the curve's base field");
    if (rowprog::shard_world(S->s) > 1 || ck->key.world > 1)   // partial cross-term commitments: the challenge needs the exchanged ones
        return fail(SRS_ERR_INVALID, "srs_sangria_prove: structure / key is sharded over processes; use the step-wise calls and add the ranks' partial commitments");
    const size_t d = srs_structure_num_cross_terms(S), rows = rowprog::rows(S->s), wlen = rowprog::num_witness_columns(S->s) * rows;
    HostTrace ht("srs_sangria_prove");
    int rc;
    if (!incoming) {
        rc = srs_commit_cross_terms(S, ck, W1, W2, challenges, n_challenges, SRS_SPACE_DEVICE, stream, T_dev, cross_term_commits);
        if (rc) return rc;
    } else {
        if (n_challenges && !challenges) return fail(SRS_ERR_INVALID, "srs_sangria_prove_incoming: bad argument");
        // run_sps_protocol_1 (src/plonk/mod.rs:465-495) squeezes U2's challenges out of a transcript that has already absorbed
        // W_commitments: with one or more challenges the reference's order is commit, THEN challenge, THEN cross terms -- the
        // commitment cannot share the cross terms' MSM.  Only zero-challenge structures (one gate, no lookups) may take this entry.
        if (rowprog::num_challenges(S->s) != 0)
            return fail(SRS_ERR_INVALID, "srs_sangria_prove_incoming: the structure has challenges (they depend on the trace's commitment); "
                                         "commit the trace first (srs_commit_upload), then srs_sangria_prove");
        if (wlen > ck->key.global_len || rows > ck->key.global_len)
            return fail(SRS_ERR_TOO_LONG_INPUT, "Can't commit too long input: input len: " + std::to_string(std::max(wlen, rows)) +
                                                    ", but limit is " + std::to_string(ck->key.global_len));
        rc = ensure_device();
        if (rc) return rc;
        std::vector<affine_t> cm(d + 1);
        rc = guarded([&]() -> int {
            hipStream_t st = (hipStream_t)stream;
            if (W2_host) SRS_HIP_CHECK(hipMemcpyAsync(W2, W2_host, wlen * sizeof(fe_t), hipMemcpyHostToDevice, st));
            if (d) {
                std::vector<fe_t *> dT(d);
                for (size_t k = 0; k < d; ++k) dT[k] = reinterpret_cast<fe_t *>(T_dev[k]);
                std::string err;
                int erc = rowprog::evaluate(S->s, 0, reinterpret_cast<const fe_t *>(W1), reinterpret_cast<const fe_t *>(W2),
                                            reinterpret_cast<const fe_t *>(challenges), n_challenges, dT.data(), st, err, false);
                if (erc) return fail(erc, "srs_sangria_prove_incoming: " + err);
            }
            std::vector<const srs_fe *> v(d + 1);
            std::vector<size_t> nn(d + 1, rows);
            v[0] = W2;
            nn[0] = wlen;
            for (size_t k = 0; k < d; ++k) v[k + 1] = T_dev[k];
            return srs_commit_batch(ck, v.data(), nn.data(), d + 1, SRS_SPACE_DEVICE, SRS_REPR_MONT, stream, reinterpret_cast<srs_affine *>(cm.data()));
        });
        if (rc) return rc;
        std::memcpy(&W_commitments[1], &cm[0], sizeof(affine_t));
        if (d) std::memcpy(cross_term_commits, &cm[1], d * sizeof(affine_t));
        if (ro) {   // U2 enters the transcript here: its W commitment, then whatever else the caller's U2 carries
            rc = srs_poseidon_absorb_point(ro, curve, &W_commitments[1]);
            if (rc) return rc;
            if (n_u2_tail) {
                rc = srs_poseidon_absorb_field(ro, u2_tail, n_u2_tail);
                if (rc) return rc;
            }
        }
    }
    ht.mark("cross terms + commitments");
    fe_t r;
    std::memcpy(&r, r_io, 32);
    if (ro) {       // generate_challenge (:162-179): the caller has absorbed pp_digest, U1, U2; the commitments and the squeeze happen here
        for (size_t k = 0; k < d; ++k) {
            rc = srs_poseidon_absorb_point(ro, curve, &cross_term_commits[k]);
            if (rc) return rc;
        }
        rc = srs_poseidon_squeeze(ro, 128, sf, reinterpret_cast<srs_fe *>(&r));
        if (rc) return rc;
        std::memcpy(r_io, &r, 32);
    }
    ht.mark("challenge");
    // W' = W1 + r W2, E' = E + sum r^k T_k  (accumulator.rs:364-404): stream-ordered, in place
    rc = srs_fold_witness(sf, W1, W1, W2, reinterpret_cast<const srs_fe *>(&r), wlen, SRS_SPACE_DEVICE, stream);
    if (rc) return rc;
    rc = srs_fold_error(sf, E, E, reinterpret_cast<const srs_fe *const *>(T_dev), d, reinterpret_cast<const srs_fe *>(&r), rows, SRS_SPACE_DEVICE, stream);
    if (rc) return rc;
    // U' (accumulator.rs:201-264): W_commitment' = C1 + r C2, E_commitment' = E_c + sum r^k T_c_k -- off the caller's thread
    std::vector<srs_fe> rp(d ? d : 1);
    srs_fe_powers(sf, reinterpret_cast<const srs_fe *>(&r), d, rp.data());
    rc = srs_point_lincomb_async(curve, &W_commitments[0], &W_commitments[1], reinterpret_cast<const srs_fe *>(&r), 1, SRS_REPR_MONT,
                                 &folded_commitments[0], &jobs[0]);
    if (rc) return rc;
    rc = srs_point_lincomb_async(curve, E_commitment, cross_term_commits, rp.data(), d, SRS_REPR_MONT, &folded_commitments[1], &jobs[1]);
    if (rc) (void)srs_job_wait(jobs[0]);     // never leave job 0 writing into folded_commitments[0] after an error return
    ht.mark("folds queued");
    return rc;
}

int srs_sangria_prove(srs_structure *S, srs_ck *ck, srs_poseidon *ro, const srs_fe *challenges, size_t n_challenges, srs_fe *W1, const srs_fe *W2,
                      srs_fe *E, void *stream, srs_fe *r_io, srs_fe *const *T_dev, srs_affine *cross_term_commits,
                      const srs_affine *W_commitments /* [2]: U1, U2 */, const srs_affine *E_commitment, srs_affine *folded_commitments /* [2]: W, E */,
                      uint64_t *jobs /* [2] */) {
    return sangria_prove_impl(S, ck, ro, challenges, n_challenges, W1, const_cast<srs_fe *>(W2), nullptr, nullptr, 0, E, stream, r_io, T_dev,
                              cross_term_commits, const_cast<srs_affine *>(W_commitments), E_commitment, folded_commitments, jobs, false);
}

int srs_sangria_prove_incoming(srs_structure *S, srs_ck *ck, srs_poseidon *ro, const srs_fe *challenges, size_t n_challenges, srs_fe *W1, srs_fe *W2,
                               const srs_fe *W2_host, const srs_fe *u2_tail, size_t n_u2_tail, srs_fe *E, void *stream, srs_fe *r_io,
                               srs_fe *const *T_dev, srs_affine *cross_term_commits, srs_affine *W_commitments /* [2]: U1 in, U2 out */,
                               const srs_affine *E_commitment, srs_affine *folded_commitments /* [2]: W, E */, uint64_t *jobs /* [2] */) {
    return sangria_prove_impl(S, ck, ro, challenges, n_challenges, W1, W2, W2_host, u2_tail, n_u2_tail, E, stream, r_io, T_dev, cross_term_commits,
                              W_commitments, E_commitment, folded_commitments, jobs, true);
}

int srs_fold_lincomb(int field, srs_fe *out, const srs_fe *const *W, const srs_fe *coefs, size_t J, size_t n, int space, void *stream) {
    if (!valid_field(field) || !coefs || !W || (n && !out) || J == 0) return fail(SRS_ERR_INVALID, "srs_fold_lincomb: bad argument");
    if (n == 0) return SRS_OK;
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        hipStream_t st = (hipStream_t)stream;
        std::string err;
        if (space == SRS_SPACE_DEVICE) {
            int erc = rowprog::lincomb(field, reinterpret_cast<fe_t *>(out), reinterpret_cast<const fe_t *const *>(W),
                                       reinterpret_cast<const fe_t *>(coefs), J, n, st, err);
            if (erc) return fail(erc, "srs_fold_lincomb: " + err);   // stream-ordered, see srs_fold_witness
        } else {
            fe_t *buf = nullptr;
            SRS_HIP_CHECK(hipMalloc((void **)&buf, J * n * sizeof(fe_t)));
            try {
                std::vector<const fe_t *> wp(J);
                for (size_t j = 0; j < J; ++j) {
                    SRS_HIP_CHECK(hipMemcpyAsync(buf + j * n, W[j], n * sizeof(fe_t), hipMemcpyHostToDevice, st));
                    wp[j] = buf + j * n;
                }
                int erc = rowprog::lincomb(field, buf, wp.data(), reinterpret_cast<const fe_t *>(coefs), J, n, st, err);
                if (erc) { (void)hipFree(buf); return fail(erc, "srs_fold_lincomb: " + err); }
                SRS_HIP_CHECK(hipMemcpyAsync(out, buf, n * sizeof(fe_t), hipMemcpyDeviceToHost, st));
                SRS_HIP_CHECK(hipStreamSynchronize(st));
            } catch (...) { (void)hipFree(buf); throw; }
            (void)hipFree(buf);
        }
        SRS_HIP_CHECK(hipGetLastError());
        return SRS_OK;
    });
}

int srs_fold_lincomb_sharded(int field, srs_fe *out, const srs_fe *const *W, const srs_fe *coefs, size_t J, size_t n, uint32_t rank,
                             uint32_t world, void *stream) {
    if (!valid_field(field) || !coefs || !W || (n && !out) || J == 0 || world == 0 || rank >= world)
        return fail(SRS_ERR_INVALID, "srs_fold_lincomb_sharded: bad argument");
    if (n == 0) return SRS_OK;
    int rc = ensure_device();
    if (rc) return rc;
    return guarded([&]() -> int {
        std::string err;
        int erc = rowprog::lincomb(field, reinterpret_cast<fe_t *>(out), reinterpret_cast<const fe_t *const *>(W),
                                   reinterpret_cast<const fe_t *>(coefs), J, n, (hipStream_t)stream, err, rank, world);
        if (erc) return fail(erc, "srs_fold_lincomb_sharded: " + err);
        SRS_HIP_CHECK(hipGetLastError());
        return SRS_OK;
    });
}

}  // extern "C"
