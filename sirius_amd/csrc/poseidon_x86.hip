// poseidon_x86.hip -- the host permutation of the off-circuit random oracle (poseidon.hip) on AVX-512 IFMA.  Host code only.
//
// Why: with the challenges derived the way the reference derives them (ProtoGalaxy::prove absorbs ~313 field elements per prove,
// src/nifs/protogalaxy/mod.rs:80-133,400-481) the sponge sits on the critical path of every fold step: ~80 chained permutations of
// 15-20 us each on scalar 64-bit code.  A permutation is a chain, but inside a round the T state elements are independent: they map
// onto the 8 lanes of a 512-bit register (T <= 8), with 52-bit limbs and vpmadd52{lo,hi}uq as the multiplier:
//   s-box   : 3 lane-wise Montgomery products (radix 2^52, R = 2^260) for all T elements at once
//   MDS step: out[i] = sum_j M[i][j] s[j] with lanes = i: T broadcast-multiply-accumulates into ONE double-width sum, one reduction
// Same field elements as the scalar code (exact arithmetic; tests/test_poseidon.py runs both paths against the pinned oracle).
// Selected at run time (cpuid); SRS_POSEIDON_SCALAR=1 forces the scalar path.
#include "poseidon.h"

#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(_M_X64))
#include <immintrin.h>

#include <cstdlib>
#include <cstring>

namespace srs {
namespace poseidon {

namespace {
#define SRS_IFMA __attribute__((target("avx512f,avx512ifma,avx512vl,avx512dq")))

constexpr uint64_t M52 = (1ull << 52) - 1;
typedef unsigned __int128 u128;

struct V5 {
    __m512i l[5];
};

// ---- scalar helpers on 4 x u64 (set-up only)
inline void dbl_mod(uint64_t x[4], const uint64_t p[4]) {      // x < p -> 2x mod p
    uint64_t c = 0, r[4];
    for (int i = 0; i < 4; ++i) {
        const uint64_t n = (x[i] << 1) | c;
        c = x[i] >> 63;
        r[i] = n;
    }
    uint64_t d[4];
    u128 bw = 0;
    for (int i = 0; i < 4; ++i) {
        const u128 t = (u128)r[i] - p[i] - (uint64_t)bw;
        d[i] = (uint64_t)t;
        bw = (t >> 64) ? 1 : 0;
    }
    const bool take = c || !bw;
    for (int i = 0; i < 4; ++i) x[i] = take ? d[i] : r[i];
}
inline void split52(const uint64_t x[4], uint64_t o[5]) {
    o[0] = x[0] & M52;
    o[1] = ((x[0] >> 52) | (x[1] << 12)) & M52;
    o[2] = ((x[1] >> 40) | (x[2] << 24)) & M52;
    o[3] = ((x[2] >> 28) | (x[3] << 36)) & M52;
    o[4] = x[3] >> 16;
}
inline void join52(const uint64_t l[5], uint64_t o[4]) {       // limbs < 2^52, value < 2^256
    o[0] = l[0] | (l[1] << 52);
    o[1] = (l[1] >> 12) | (l[2] << 40);
    o[2] = (l[2] >> 24) | (l[3] << 28);
    o[3] = (l[3] >> 36) | (l[4] << 16);
}
}  // namespace

struct IfmaConsts {
    size_t t = 0, r_f = 0, r_p = 0;
    uint64_t p4[4];
    alignas(64) uint64_t p52[5][8];          // modulus limbs, broadcast
    alignas(64) uint64_t pinv[8];            // -p^-1 mod 2^52, broadcast
    alignas(64) uint64_t to260[5][8];        // 2^264 mod p  (product with it: 2^256-Montgomery -> 2^260-Montgomery), broadcast
    alignas(64) uint64_t to256[5][8];        // 2^256 mod p  (2^260-Montgomery -> 2^256-Montgomery), broadcast
    std::vector<uint64_t> rc;                // [(r_f + r_p)][5][8]: round constants, lanes = state index, 2^260-Montgomery
    std::vector<uint64_t> mds;               // [t (column j)][5][8]: lanes = row i, 2^260-Montgomery
};

bool ifma_available() {
    static const bool ok = [] {
        if (const char *e = std::getenv("SRS_POSEIDON_SCALAR"))
            if (e[0] == '1') return false;
        __builtin_cpu_init();
        return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512vl") &&
               __builtin_cpu_supports("avx512dq");
    }();
    return ok;
}

namespace {
// lane-wise Montgomery product a * b / 2^260 mod p; operands: limbs < 2^52, a * b < 2^260 * p * (small); result < 2 p, limbs < 2^52
SRS_IFMA inline V5 mm(const V5 &a, const V5 &b, const IfmaConsts &K) {
    const __m512i zero = _mm512_setzero_si512(), mask = _mm512_set1_epi64((long long)M52);
    const __m512i pinv = _mm512_loadu_si512((const void *)K.pinv);
    __m512i p[5];
    for (int j = 0; j < 5; ++j) p[j] = _mm512_loadu_si512((const void *)K.p52[j]);
    __m512i t[6] = {zero, zero, zero, zero, zero, zero};
    for (int i = 0; i < 5; ++i) {
        const __m512i bi = b.l[i];
        for (int j = 0; j < 5; ++j) t[j] = _mm512_madd52lo_epu64(t[j], a.l[j], bi);
        for (int j = 0; j < 5; ++j) t[j + 1] = _mm512_madd52hi_epu64(t[j + 1], a.l[j], bi);
        const __m512i m = _mm512_and_si512(_mm512_madd52lo_epu64(zero, t[0], pinv), mask);
        for (int j = 0; j < 5; ++j) t[j] = _mm512_madd52lo_epu64(t[j], m, p[j]);
        for (int j = 0; j < 5; ++j) t[j + 1] = _mm512_madd52hi_epu64(t[j + 1], m, p[j]);
        const __m512i c = _mm512_srli_epi64(t[0], 52);           // the low 52 bits of t[0] are zero now
        t[0] = _mm512_add_epi64(t[1], c);
        t[1] = t[2];
        t[2] = t[3];
        t[3] = t[4];
        t[4] = t[5];
        t[5] = zero;
    }
    V5 o;
    __m512i c = zero;
    for (int j = 0; j < 5; ++j) {
        const __m512i x = _mm512_add_epi64(t[j], c);
        o.l[j] = j < 4 ? _mm512_and_si512(x, mask) : x;
        c = _mm512_srli_epi64(x, 52);
    }
    return o;
}
SRS_IFMA inline V5 add_norm(const V5 &a, const V5 &b) {          // limb-wise sum, carries propagated (value a + b, limbs < 2^52)
    const __m512i mask = _mm512_set1_epi64((long long)M52);
    V5 o;
    __m512i c = _mm512_setzero_si512();
    for (int j = 0; j < 5; ++j) {
        const __m512i x = _mm512_add_epi64(_mm512_add_epi64(a.l[j], b.l[j]), c);
        o.l[j] = j < 4 ? _mm512_and_si512(x, mask) : x;
        c = _mm512_srli_epi64(x, 52);
    }
    return o;
}
SRS_IFMA inline V5 load_v5(const uint64_t (*src)[8]) {
    V5 o;
    for (int j = 0; j < 5; ++j) o.l[j] = _mm512_loadu_si512((const void *)src[j]);
    return o;
}

// out[i] = sum_j M[i][j] s[j]: lanes = i; one double-width accumulation, one Montgomery reduction
SRS_IFMA inline V5 mds_step(const V5 &s, const IfmaConsts &K) {
    const __m512i zero = _mm512_setzero_si512(), mask = _mm512_set1_epi64((long long)M52);
    const __m512i pinv = _mm512_loadu_si512((const void *)K.pinv);
    __m512i acc[11];
    for (int i = 0; i < 11; ++i) acc[i] = zero;
    for (size_t j = 0; j < K.t; ++j) {
        const __m512i idx = _mm512_set1_epi64((long long)j);
        const uint64_t(*col)[8] = reinterpret_cast<const uint64_t(*)[8]>(K.mds.data() + j * 40);
        __m512i sj[5];
        for (int b = 0; b < 5; ++b) sj[b] = _mm512_permutexvar_epi64(idx, s.l[b]);          // broadcast lane j
        for (int a = 0; a < 5; ++a) {
            const __m512i m = _mm512_loadu_si512((const void *)col[a]);
            for (int b = 0; b < 5; ++b) {
                acc[a + b] = _mm512_madd52lo_epu64(acc[a + b], m, sj[b]);
                acc[a + b + 1] = _mm512_madd52hi_epu64(acc[a + b + 1], m, sj[b]);
            }
        }
    }
    __m512i p[5];
    for (int j = 0; j < 5; ++j) p[j] = _mm512_loadu_si512((const void *)K.p52[j]);
    for (int i = 0; i < 5; ++i) {
        const __m512i m = _mm512_and_si512(_mm512_madd52lo_epu64(zero, acc[i], pinv), mask);
        for (int j = 0; j < 5; ++j) {
            acc[i + j] = _mm512_madd52lo_epu64(acc[i + j], m, p[j]);
            acc[i + j + 1] = _mm512_madd52hi_epu64(acc[i + j + 1], m, p[j]);
        }
        acc[i + 1] = _mm512_add_epi64(acc[i + 1], _mm512_srli_epi64(acc[i], 52));
    }
    V5 o;
    __m512i c = zero;
    for (int j = 0; j < 5; ++j) {
        const __m512i x = _mm512_add_epi64(acc[5 + j], c);
        o.l[j] = j < 4 ? _mm512_and_si512(x, mask) : x;
        c = _mm512_srli_epi64(x, 52);
    }
    return o;
}

SRS_IFMA void permute_impl(const IfmaConsts &K, uint64_t *state) {
    const size_t t = K.t, half = K.r_f / 2;
    alignas(64) uint64_t lanes[5][8];
    std::memset(lanes, 0, sizeof lanes);
    for (size_t i = 0; i < t; ++i) {
        uint64_t l[5];
        split52(state + 4 * i, l);
        for (int j = 0; j < 5; ++j) lanes[j][i] = l[j];
    }
    V5 s = mm(load_v5(lanes), load_v5(K.to260), K);                                   // -> 2^260-Montgomery, < 2 p
    const __mmask8 lane0 = 0x01;
    for (size_t r = 0; r < K.r_f + K.r_p; ++r) {
        s = add_norm(s, load_v5(reinterpret_cast<const uint64_t(*)[8]>(K.rc.data() + r * 40)));      // < 3 p
        const V5 s2 = mm(s, s, K);
        const V5 s5 = mm(mm(s2, s2, K), s, K);
        if (r < half || r >= half + K.r_p) {
            s = s5;
        } else {
            for (int j = 0; j < 5; ++j) s.l[j] = _mm512_mask_blend_epi64(lane0, s.l[j], s5.l[j]);     // partial round: state[0] only
        }
        s = mds_step(s, K);                                                                           // < 2 p
    }
    s = mm(s, load_v5(K.to256), K);                                                    // back to 2^256-Montgomery, < 2 p
    for (int j = 0; j < 5; ++j) _mm512_storeu_si512((void *)lanes[j], s.l[j]);
    for (size_t i = 0; i < t; ++i) {
        uint64_t l[5], x[4];
        for (int j = 0; j < 5; ++j) l[j] = lanes[j][i];
        join52(l, x);                                                                   // < 2 p < 2^255
        uint64_t d[4];
        u128 bw = 0;
        for (int k = 0; k < 4; ++k) {
            const u128 v = (u128)x[k] - K.p4[k] - (uint64_t)bw;
            d[k] = (uint64_t)v;
            bw = (v >> 64) ? 1 : 0;
        }
        for (int k = 0; k < 4; ++k) state[4 * i + k] = bw ? x[k] : d[k];
    }
}
}  // namespace

IfmaConsts *ifma_prepare(const uint64_t p4[4], uint64_t inv64, const uint64_t *rc64, const uint64_t *mds64, size_t t, size_t r_f, size_t r_p) {
    if (t > 8) return nullptr;
    IfmaConsts *K = new IfmaConsts();
    K->t = t;
    K->r_f = r_f;
    K->r_p = r_p;
    std::memcpy(K->p4, p4, 32);
    uint64_t pl[5];
    split52(p4, pl);
    for (int j = 0; j < 5; ++j)
        for (int l = 0; l < 8; ++l) K->p52[j][l] = pl[j];
    for (int l = 0; l < 8; ++l) K->pinv[l] = inv64 & M52;
    // 2^256 mod p and 2^264 mod p by modular doublings of 1
    uint64_t one[4] = {1, 0, 0, 0}, r256[4], r264[4];
    std::memcpy(r256, one, 32);
    for (int d = 0; d < 256; ++d) dbl_mod(r256, p4);
    std::memcpy(r264, r256, 32);
    for (int d = 0; d < 8; ++d) dbl_mod(r264, p4);
    uint64_t l[5];
    split52(r264, l);
    for (int j = 0; j < 5; ++j)
        for (int q = 0; q < 8; ++q) K->to260[j][q] = l[j];
    split52(r256, l);
    for (int j = 0; j < 5; ++j)
        for (int q = 0; q < 8; ++q) K->to256[j][q] = l[j];
    auto to260 = [&](const uint64_t *x256, uint64_t out[5]) {      // x 2^256 -> x 2^260: four modular doublings
        uint64_t x[4];
        std::memcpy(x, x256, 32);
        for (int d = 0; d < 4; ++d) dbl_mod(x, p4);
        split52(x, out);
    };
    const size_t rounds = r_f + r_p;
    K->rc.assign(rounds * 40, 0);
    K->mds.assign(t * 40, 0);
    for (size_t r = 0; r < rounds; ++r)
        for (size_t i = 0; i < t; ++i) {
            uint64_t o[5];
            to260(rc64 + (r * t + i) * 4, o);
            for (int j = 0; j < 5; ++j) K->rc[r * 40 + j * 8 + i] = o[j];
        }
    for (size_t j = 0; j < t; ++j)                     // column j, lanes = rows i
        for (size_t i = 0; i < t; ++i) {
            uint64_t o[5];
            to260(mds64 + (i * t + j) * 4, o);
            for (int a = 0; a < 5; ++a) K->mds[j * 40 + a * 8 + i] = o[a];
        }
    return K;
}
void ifma_release(IfmaConsts *K) { delete K; }
void ifma_permute(const IfmaConsts *K, uint64_t *state) { permute_impl(*K, state); }

}  // namespace poseidon
}  // namespace srs

#else   // device pass of hipcc / non-x86 hosts: the scalar permutation is the only one

namespace srs {
namespace poseidon {
#if !defined(__HIP_DEVICE_COMPILE__)
struct IfmaConsts {};
bool ifma_available() { return false; }
IfmaConsts *ifma_prepare(const uint64_t *, uint64_t, const uint64_t *, const uint64_t *, size_t, size_t, size_t) { return nullptr; }
void ifma_release(IfmaConsts *) {}
void ifma_permute(const IfmaConsts *, uint64_t *) {}
#endif
}  // namespace poseidon
}  // namespace srs

#endif
