// capi.hip -- the extern "C" surface declared in include/sirius_amd.h.
#include "../../include/sirius_amd.h"

#include <cstring>
#include <string>
#include <vector>

#include "curve.cuh"
#include "devrt.h"
#include "msm.h"

namespace srs {
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
const char *get_error() { return g_err.c_str(); }
}  // namespace srs

using namespace srs;

static_assert(sizeof(srs_fe) == sizeof(fe_t) && sizeof(srs_affine) == sizeof(affine_t), "ABI layout");

struct srs_ck {
    msm::Key key;
    Arena staging;      // H2D staging of host scalars
};

namespace {

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

template <class C>
affine_t xyzz_to_affine(const xyzz_t &p) { return Ec<C>::to_affine(p); }

affine_t to_affine_curve(int curve, const xyzz_t &p) {
    return curve == SRS_CURVE_BN256 ? xyzz_to_affine<Bn256>(p) : xyzz_to_affine<Grumpkin>(p);
}

#if !defined(SRS_EMU)
bool g_device_ok = false;
#endif

int ensure_device() {
#if defined(SRS_EMU)
    return SRS_OK;
#else
    if (g_device_ok) return SRS_OK;
    return srs_init(-1);
#endif
}

}  // namespace

extern "C" {

const char *srs_last_error(void) { return get_error(); }
const char *srs_version(void) { return "sirius_amd 0.1.0 (gfx950)"; }

int srs_init(int device_ordinal) {
#if defined(SRS_EMU)
    (void)device_ordinal;
    return SRS_OK;
#else
    return guarded([&]() -> int {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
            return fail(SRS_ERR_DEVICE, "no HIP device visible: libsirius_amd has no CPU path");
        int dev = device_ordinal;
        if (dev < 0) SRS_HIP_CHECK(hipGetDevice(&dev));
        if (dev >= count) return fail(SRS_ERR_INVALID, "device ordinal out of range");
        SRS_HIP_CHECK(hipSetDevice(dev));
        hipDeviceProp_t prop;
        SRS_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(SRS_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
        g_device_ok = true;
        return SRS_OK;
    });
#endif
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

int srs_ck_create(int curve, const srs_affine *bases, size_t len, int space, srs_ck **out) {
    return srs_ck_create_sharded(curve, bases, len, space, 0, 1, out);
}

void srs_ck_free(srs_ck *ck) {
    if (!ck) return;
    if (ck->key.table) (void)hipFree(ck->key.table);
    ck->key.arena.release();
    ck->staging.release();
    delete ck;
}

size_t srs_ck_len(const srs_ck *ck) { return ck ? ck->key.global_len : 0; }

int srs_commit_batch(srs_ck *ck, const srs_fe *const *scalars, const size_t *n, size_t batch, int space,
                     int repr, void *stream, srs_affine *out) {
    if (!ck || !out || (batch && (!scalars || !n))) return fail(SRS_ERR_INVALID, "srs_commit: bad argument");
    if (batch == 0) return SRS_OK;
    for (size_t m = 0; m < batch; ++m) {
        if (n[m] > ck->key.global_len)
            return fail(SRS_ERR_TOO_LONG_INPUT, "Can't commit too long input: input len: " + std::to_string(n[m]) +
                                                    ", but limit is " + std::to_string(ck->key.global_len));
        if (n[m] && !scalars[m]) return fail(SRS_ERR_INVALID, "srs_commit: null scalar vector");
    }
    int rc = ensure_device();
    if (rc) return rc;
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
        for (size_t m = 0; m < batch; ++m) {
            affine_t a = to_affine_curve(ck->key.curve, res[m]);
            std::memcpy(&out[m], &a, sizeof(a));
        }
        return SRS_OK;
    });
}

int srs_commit(srs_ck *ck, const srs_fe *scalars, size_t n, int space, int repr, void *stream, srs_affine *out) {
    const srs_fe *v[1] = {scalars};
    size_t nn[1] = {n};
    return srs_commit_batch(ck, v, nn, 1, space, repr, stream, out);
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

}  // extern "C"
