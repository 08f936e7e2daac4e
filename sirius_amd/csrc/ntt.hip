// ntt.hip -- radix-2 NTT over bn256::Fr on gfx950: fft / ifft / coset_fft / coset_ifft.
//
// Replaces the bodies of reference src/fft.rs:160-198 (`fft`, `ifft`, `coset_fft`, `coset_ifft`
// over `best_fft`, :61-115).  Same contract: in place, natural order in AND out,
//   fft : a_i <- sum_j a_j w^(ij),  w = ROOT_OF_UNITY^(2^(S-k))            (:12-23)
//   ifft: same with w^-1, then * TWO_INV^k                                  (:168-182)
//   coset_fft : a_i *= ZETA^(i mod 3) first; coset_ifft: a_i *= ZETA^-(i mod 3) last (:186-228)
// Field arithmetic is exact, so any butterfly schedule reproduces the reference bit for bit.
//
// MI355X-first schedule (not the reference's bit-reverse + recursive rayon::join):
//   * n <= 2^10: one workgroup, whole vector in LDS, bit-reversed load + DIT stages.
//   * larger n: Cooley-Tukey over p = ceil(k/8) digits of <= 8 bits.  Pass j transforms digit j
//     for a tile of 8 neighbouring columns (256 B contiguous per row -> coalesced HBM traffic),
//     all 2^r x 8 elements staged in LDS (64 KiB: two workgroups per CU overlap compute and memory), butterflies from an
//     LDS-resident w_(2^r) table, then ONE multiply by a precomputed inter-digit twiddle that is
//     read coalesced (same index pattern as the data).  The last pass writes through the digit
//     reversal so the result lands in natural order; ifft's n^-1 is folded into the first table.
//     Physical traffic = P x 64 B/element (+32 B for the first-pass table), P = number of passes.
#include "ntt.h"
#define SRS_F29_CHAIN 1      // chained 9 x 29 products (field29_chain.inc): 2^24 fft 2.38 -> 2.31 ms, profiles/r04_ab_chain_blocks.txt
#include "field29.cuh"
#include "prof.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <tuple>
#include <mutex>
#include <vector>

namespace srs {
namespace ntt {

constexpr uint32_t COLS_LOG = 3;
constexpr uint32_t COLS = 1u << COLS_LOG;   // neighbouring columns per tile: 256 B rows; a 2^8-row tile is 64 KiB, so TWO
                                            // workgroups share a CU and one computes while the other loads / stores
constexpr uint32_t SMALL_LOG = 10; // single-workgroup path up to 2^10 points

__device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// T[i] = scale * base^(e(i)),  e(i) = (i >> lo_bits) * (i & (2^lo_bits - 1)) mod 2^m   (m = log2 of table)
// lo_bits == 0 : plain powers base^i.
__global__ void k_fill_table(fe_t *__restrict__ T, uint32_t log_entries, uint32_t lo_bits, fe_t base, fe_t scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << log_entries)) return;
    uint64_t e;
    if (lo_bits == 0) {
        e = i;
    } else {
        uint64_t k = i >> lo_bits, rest = i & (((uint64_t)1 << lo_bits) - 1);
        e = (k * rest) & (((uint64_t)1 << log_entries) - 1);
    }
    fe_t acc = scale;
    fe_t b = base;
    while (e) {
        if (e & 1) acc = Fr::mul(acc, b);
        b = Fr::sqr(b);
        e >>= 1;
    }
    T[i] = acc;
}

struct Scale3 {   // multiply element with global index i by z[i % 3] (z[0] unused = 1)
    fe_t z1, z2;
    int on;
};
__device__ __forceinline__ fe_t apply_scale3(const Scale3 &s, fe_t v, size_t idx) {
    uint32_t r = (uint32_t)(idx % 3);
    if (r == 1) return Fr::mul(v, s.z1);
    if (r == 2) return Fr::mul(v, s.z2);
    return v;
}

// one DIT butterfly stage over an LDS tile laid out [row][col], rows = 2^rbits
template <uint32_t NCOLS>
__device__ __forceinline__ void lds_stage(fe_t *tile, const fe_t *W, uint32_t rbits, uint32_t s) {
    const uint32_t half = 1u << s;
    const uint32_t pairs = (1u << (rbits - 1)) * NCOLS;
    for (uint32_t p = threadIdx.x; p < pairs; p += blockDim.x) {
        uint32_t c = p % NCOLS, q = p / NCOLS;
        uint32_t j = q & (half - 1);
        uint32_t lo = ((q >> s) << (s + 1)) | j;
        uint32_t hi = lo + half;
        fe_t a = tile[lo * NCOLS + c];
        fe_t b = tile[hi * NCOLS + c];
        if (j) b = Fr::mul(b, W[j << (rbits - 1 - s)]);
        tile[lo * NCOLS + c] = Fr::add(a, b);
        tile[hi * NCOLS + c] = Fr::sub(a, b);
    }
}

// two DIT stages (s, s + 1) in one LDS round trip: a thread owns the 4 elements base + {0, 1, 2, 3} * 2^s of a radix-4
// group -- the same 4 twiddle multiplications as two radix-2 stages, half the LDS traffic and half the barriers
template <uint32_t NCOLS>
__device__ __forceinline__ void lds_stage2(fe_t *tile, const fe_t *W, uint32_t rbits, uint32_t s) {
    const uint32_t h = 1u << s;
    const uint32_t groups = (1u << (rbits - 2)) * NCOLS;
    for (uint32_t p = threadIdx.x; p < groups; p += blockDim.x) {
        const uint32_t c = p % NCOLS, g = p / NCOLS;
        const uint32_t j = g & (h - 1);
        const uint32_t base = ((g >> s) << (s + 2)) | j;
        fe_t e0 = tile[base * NCOLS + c], e1 = tile[(base + h) * NCOLS + c];
        fe_t e2 = tile[(base + 2 * h) * NCOLS + c], e3 = tile[(base + 3 * h) * NCOLS + c];
        if (j) {                                                   // stage s: pairs (0,1) and (2,3), twiddle w^(j 2^(r-1-s))
            const fe_t t1 = W[j << (rbits - 1 - s)];
            e1 = Fr::mul(e1, t1);
            e3 = Fr::mul(e3, t1);
        }
        fe_t a0 = Fr::add(e0, e1), a1 = Fr::sub(e0, e1), a2 = Fr::add(e2, e3), a3 = Fr::sub(e2, e3);
        if (j) a2 = Fr::mul(a2, W[j << (rbits - 2 - s)]);      // stage s + 1: pairs (0,2) and (1,3)
        a3 = Fr::mul(a3, W[(j + h) << (rbits - 2 - s)]);
        tile[base * NCOLS + c] = Fr::add(a0, a2);
        tile[(base + 2 * h) * NCOLS + c] = Fr::sub(a0, a2);
        tile[(base + h) * NCOLS + c] = Fr::add(a1, a3);
        tile[(base + 3 * h) * NCOLS + c] = Fr::sub(a1, a3);
    }
}

// all `rbits` stages of a tile
template <uint32_t NCOLS>
__device__ __forceinline__ void lds_stages(fe_t *tile, const fe_t *W, uint32_t rbits) {
    uint32_t s = 0;
    for (; s + 1 < rbits; s += 2) {
        lds_stage2<NCOLS>(tile, W, rbits, s);
        __syncthreads();
    }
    if (s < rbits) {
        lds_stage<NCOLS>(tile, W, rbits, s);
        __syncthreads();
    }
}

// ---- single-workgroup transform, n = 2^k <= 2^SMALL_LOG; grid.x = batch of independent vectors
__global__ void SRS_KERNEL_BOUNDS(512, 1)
    k_ntt_small(fe_t *__restrict__ a, size_t stride, uint32_t k, const fe_t *__restrict__ Wg, fe_t post, int has_post,
                Scale3 pre, Scale3 fin) {
    __shared__ fe_t tile[1u << SMALL_LOG];
    __shared__ fe_t W[1u << (SMALL_LOG - 1)];
    const uint32_t n = 1u << k;
    fe_t *v = a + (size_t)blockIdx.x * stride;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        fe_t x = v[i];
        if (pre.on) x = apply_scale3(pre, x, i);
        tile[bitrev(i, k)] = x;
    }
    for (uint32_t i = threadIdx.x; i < (n >> 1); i += blockDim.x) W[i] = Wg[i];
    __syncthreads();
    lds_stages<1>(tile, W, k);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        fe_t x = tile[i];
        if (has_post) x = Fr::mul(x, post);
        if (fin.on) x = apply_scale3(fin, x, i);
        v[i] = x;
    }
}

// ---- one digit of the multi-pass transform
struct PassArgs {
    uint32_t log_n;
    uint32_t rbits;      // this digit's width
    uint32_t lbits;      // bits below this digit (0 for the last pass)
    uint32_t npass, pass;
    uint32_t radix_bits[4];   // widths of digit 1..p (digit 1 = most significant input bits)
};

// ---- the multi-pass kernels with the TILE ITSELF in the lazy 9 x 29-bit form (r04) --------------------------------------------------------
// r03 measured the 29-bit multiplier with canonical 8 x 32 data in LDS: every product paid an unpack of both operands and a pack +
// canonicalisation of the result (26 + 26 + 75 instructions for the 180 it saves) -- no gain.  Here an element is unpacked ONCE when the
// tile is loaded, lives in LDS as 9 words, goes through all stages with LAZY additions (no modular reduction: limbs are carry-normalised
// once per radix-4 group, values grow as bounded multiples of p) and is packed once, after its closing product -- the inter-digit
// twiddle, or a product with the radix' one in the last pass (which also carries ifft's / the coset's scaling when there is one).
// Twiddle tables are stored times 2^5 (plan.mul29), so x 2^256 * w 2^261 / 2^261 = x w 2^256: the ABI form, bit for bit.
//
// Bounds (P = modulus).  What Fr29::mul REQUIRES (field29.cuh): operand a with limbs < 2^31 (8 low limbs) -- any VALUE A * P it then represents,
// here A <= 469 --, operand b normalised (limbs < 2^29, value < P: the twiddles), so that every 64-bit column sum of the 9 + 9 products stays
// below 2^64 (9 * 2^31 * 2^29 + 9 * 2^29 * 2^29 + carry < 2^64); what it RETURNS: a normalised value < A B P / 2^261 + P, i.e. < (A / 169 + 1) P
// (2^261 / P > 169).  It does NOT need A * B < 169 P: only the output bound grows with A.  V_k = bound of a tile value, in units of P,
// entering radix-4 group k:
//   V_0 = 1 (canonical input), V_1 = 7, V_{k+1} = 4 V_k + 1  ->  29, 117, 469 after the 4th group;  a single trailing stage: 2 V + 1.
//   Inside a group every subtraction a - b adds (bound of b, + 1) * P, so nothing goes negative; products see A <= 2 V + 1 <= 235 and
//   return < 2.4 P.  The closing product sees A <= 469: result < 469 / 169 P + P < 3.8 P, which to_canonical_fe (< 4 P) takes.
//   Limbs: normalised (< 2^29, top limb < 2^31) at group boundaries; inside a group <= 2^29 + 2^30 + 2^30 < 2^32.
struct lazy {
    static constexpr uint32_t bound(uint32_t k) { return k == 0 ? 1u : (k == 1 ? 7u : 4u * bound(k - 1) + 1u); }
    // the closing product of a tile of up to 4 radix-4 groups (+ a trailing stage) must come back below the 4 P that to_canonical_fe takes
    static constexpr uint32_t closing_bound_x169(uint32_t a) { return a + 169u; }                  // (A / 169 + 1) P, times 169
    // Element e lives at word 9 e + (e >> 6): ONE padding word per 64 elements.  Without it the bit-reversed tile rows of the load phases sit
    // 288 x 2^j words apart (= 0 or 32 mod the 64 LDS banks): k_ntt_last_lazy's load put 32 lanes on one bank (SQ_LDS_BANK_CONFLICT 4x
    // k_ntt_pass_lazy's, profiles/r05_pmc_ntt.json), the pass kernel's 8, the butterfly groups up to 4; with the pad every access pattern of
    // these kernels is at most 2-way (tools/ntt_lds_banks.py enumerates them).
    SRS_HD static constexpr uint32_t words(uint32_t elems) { return elems * 9 + (elems >> 6) + 1; }
    SRS_D static f29_t get(const uint32_t *tile, uint32_t e) {
        f29_t x;
        const uint32_t at = e * 9 + (e >> 6);
#pragma unroll
        for (int l = 0; l < 9; ++l) x.v[l] = tile[at + l];
        return x;
    }
    SRS_D static void put(uint32_t *tile, uint32_t e, const f29_t &x) {
        const uint32_t at = e * 9 + (e >> 6);
#pragma unroll
        for (int l = 0; l < 9; ++l) tile[at + l] = x.v[l];
    }
};

// radix-4 group number K of a tile (stages 2 K and 2 K + 1), one group per thread and round
template <uint32_t NCOLS, uint32_t K>
__device__ __forceinline__ void lazy_stage2(uint32_t *tile, const uint32_t *W, uint32_t rbits) {
    using F = Fr29;
    constexpr uint32_t V = lazy::bound(K);
    const uint32_t s = 2 * K, h = 1u << s;
    const uint32_t groups = (1u << (rbits - 2)) * NCOLS;
    for (uint32_t p = threadIdx.x; p < groups; p += blockDim.x) {
        const uint32_t c = p % NCOLS, g = p / NCOLS;
        const uint32_t j = g & (h - 1);
        const uint32_t base = ((g >> s) << (s + 2)) | j;
        const uint32_t i0 = base * NCOLS + c, i1 = (base + h) * NCOLS + c, i2 = (base + 2 * h) * NCOLS + c, i3 = (base + 3 * h) * NCOLS + c;
        f29_t e0 = lazy::get(tile, i0), e1 = lazy::get(tile, i1), e2 = lazy::get(tile, i2), e3 = lazy::get(tile, i3);     // < V, normalised
        if (K > 0 && j) {                                          // stage s: pairs (0,1) and (2,3), twiddle w^(j 2^(r-1-s))
            const f29_t t1 = lazy::get(W, j << (rbits - 1 - s));
            e1 = F::mul(e1, t1);                                   // < 2.4
            e3 = F::mul(e3, t1);
        }
        // subtrahends < max(V, 2.4): + (V + 1) P for K > 0, + 2 P for K = 0 (V = 1, no product)
        constexpr uint32_t CA = K == 0 ? 2u : V + 1u;
        const f29_t a0 = F::add_lazy(e0, e1), a1 = F::template sub_lazy<CA, 0>(e0, e1);          // < 2 V, < 2 V + 1; limbs < 2^30, < 2^30.6
        f29_t a2 = F::add_lazy(e2, e3), a3 = F::template sub_lazy<CA, 0>(e2, e3);
        // stage s + 1: pairs (0,2) and (1,3)
        a3 = F::mul(a3, lazy::get(W, (j + h) << (rbits - 2 - s)));                               // < 2.4, normalised
        f29_t o0, o1, o2, o3;
        if (K > 0 && j) {
            a2 = F::mul(a2, lazy::get(W, j << (rbits - 2 - s)));                                 // < 2.4
            o0 = F::add_lazy(a0, a2);
            o2 = F::template sub_lazy<3, 0>(a0, a2);
        } else {                                                   // a2 < 2 V, limbs < 2^30
            o0 = F::add_lazy(a0, a2);
            o2 = F::template sub_lazy<2 * V + 1, 1>(a0, a2);       // < 4 V + 1
        }
        o1 = F::add_lazy(a1, a3);
        o3 = F::template sub_lazy<3, 0>(a1, a3);                   // < 2 V + 4
        lazy::put(tile, i0, F::normalize(o0));
        lazy::put(tile, i2, F::normalize(o2));
        lazy::put(tile, i1, F::normalize(o1));
        lazy::put(tile, i3, F::normalize(o3));
    }
}
// a single trailing stage s (odd digit widths), input bound V = lazy::bound(K)
template <uint32_t NCOLS, uint32_t K>
__device__ __forceinline__ void lazy_stage1(uint32_t *tile, const uint32_t *W, uint32_t rbits) {
    using F = Fr29;
    constexpr uint32_t V = lazy::bound(K);
    const uint32_t s = 2 * K, half = 1u << s;
    const uint32_t pairs = (1u << (rbits - 1)) * NCOLS;
    for (uint32_t p = threadIdx.x; p < pairs; p += blockDim.x) {
        const uint32_t c = p % NCOLS, q = p / NCOLS;
        const uint32_t j = q & (half - 1);
        const uint32_t lo = ((q >> s) << (s + 1)) | j, hi = lo + half;
        const f29_t a = lazy::get(tile, lo * NCOLS + c);
        f29_t b = lazy::get(tile, hi * NCOLS + c);
        if (j) b = F::mul(b, lazy::get(W, j << (rbits - 1 - s)));
        lazy::put(tile, lo * NCOLS + c, F::normalize(F::add_lazy(a, b)));
        lazy::put(tile, hi * NCOLS + c, F::normalize(F::template sub_lazy<V + 1, 0>(a, b)));
    }
}
template <uint32_t NCOLS, uint32_t RBITS>
__device__ __forceinline__ void lazy_stages(uint32_t *tile, const uint32_t *W) {
    static_assert(RBITS >= 4 && RBITS <= 8, "digit widths");
    // value bound entering the closing product: after RBITS / 2 groups (and a trailing stage for odd widths: 2 V + 1)
    constexpr uint32_t A_CLOSE = (RBITS & 1) ? 2u * lazy::bound(RBITS / 2) + 1u : lazy::bound(RBITS / 2);
    static_assert(lazy::closing_bound_x169(A_CLOSE) < 4u * 169u, "closing product must return < 4 P (Fr29::to_canonical_fe)");
    static_assert(2u * lazy::bound(RBITS / 2) + 1u <= 2u * 469u + 1u, "tile values stay within the limb headroom stated above");
    lazy_stage2<NCOLS, 0>(tile, W, RBITS);
    __syncthreads();
    lazy_stage2<NCOLS, 1>(tile, W, RBITS);
    __syncthreads();
    if constexpr (RBITS >= 6) {
        lazy_stage2<NCOLS, 2>(tile, W, RBITS);
        __syncthreads();
    }
    if constexpr (RBITS >= 8) {
        lazy_stage2<NCOLS, 3>(tile, W, RBITS);
        __syncthreads();
    }
    if constexpr (RBITS & 1) {
        lazy_stage1<NCOLS, RBITS / 2>(tile, W, RBITS);
        __syncthreads();
    }
}

template <uint32_t RBITS>
__global__ void SRS_KERNEL_BOUNDS(1024, 1)
    k_ntt_pass_lazy(const fe_t *__restrict__ src, fe_t *__restrict__ dst, PassArgs pa, const fe_t *__restrict__ Wg,
                    const fe_t *__restrict__ T, Scale3 pre) {
    __shared__ uint32_t tile[lazy::words((1u << RBITS) * COLS)];
    __shared__ uint32_t W[lazy::words(1u << (RBITS - 1))];
    const uint32_t rows = 1u << RBITS;
    const uint32_t tiles_per_hi = 1u << (pa.lbits - COLS_LOG);
    const size_t hi = blockIdx.x / tiles_per_hi;
    const uint32_t rest0 = (blockIdx.x % tiles_per_hi) * COLS;
    const size_t base = (hi << (RBITS + pa.lbits)) + rest0;
    for (uint32_t e = threadIdx.x; e < rows * COLS; e += blockDim.x) {
        uint32_t d = e / COLS, c = e % COLS;
        size_t idx = base + ((size_t)d << pa.lbits) + c;
        fe_t x = src[idx];
        if (pre.on) x = apply_scale3(pre, x, idx);
        lazy::put(tile, bitrev(d, RBITS) * COLS + c, Fr29::unpack(x));
    }
    for (uint32_t i = threadIdx.x; i < (rows >> 1); i += blockDim.x) lazy::put(W, i, Fr29::unpack(Wg[i]));
    __syncthreads();
    lazy_stages<COLS, RBITS>(tile, W);
    for (uint32_t e = threadIdx.x; e < rows * COLS; e += blockDim.x) {
        uint32_t kd = e / COLS, c = e % COLS;
        size_t tidx = ((size_t)kd << pa.lbits) + rest0 + c;      // T[k_j][rest]
        dst[base + ((size_t)kd << pa.lbits) + c] = Fr29::to_canonical_fe(Fr29::mul(lazy::get(tile, e), Fr29::unpack(T[tidx])));
    }
}

template <uint32_t RBITS>
__global__ void SRS_KERNEL_BOUNDS(1024, 1)
    k_ntt_last_lazy(const fe_t *__restrict__ src, fe_t *__restrict__ dst, PassArgs pa, const fe_t *__restrict__ Wg, Scale3 fin, fe_t one29) {
    __shared__ uint32_t tile[lazy::words((1u << RBITS) * COLS)];
    __shared__ uint32_t W[lazy::words(1u << (RBITS - 1))];
    const uint32_t rows = 1u << RBITS;
    const uint32_t r1 = pa.radix_bits[0];
    const uint32_t mid_bits = pa.log_n - r1 - RBITS;              // digits 2..p-1
    const uint32_t mid = blockIdx.x & ((1u << mid_bits) - 1);
    const uint32_t k1_0 = (blockIdx.x >> mid_bits) * COLS;
    uint32_t out_mid = 0;                                         // digit-reverse `mid` (see k_ntt_last)
    {
        uint32_t m = mid;
        uint32_t widths[2], nd = 0, off_of[2];
        for (uint32_t j = pa.npass - 1; j >= 2; --j) widths[nd++] = pa.radix_bits[j - 1];
        for (uint32_t t = 0; t < nd; ++t) {
            uint32_t j = pa.npass - 1 - t, off = 0;
            for (uint32_t q = 2; q < j; ++q) off += pa.radix_bits[q - 1];
            off_of[t] = off;
        }
        for (uint32_t t = 0; t < nd; ++t) {
            uint32_t dg = m & ((1u << widths[t]) - 1);
            m >>= widths[t];
            out_mid |= dg << off_of[t];
        }
    }
    for (uint32_t e = threadIdx.x; e < rows * COLS; e += blockDim.x) {
        uint32_t c = e / rows, i = e % rows;
        size_t idx = ((size_t)(k1_0 + c) << (pa.log_n - r1)) + ((size_t)mid << RBITS) + i;
        lazy::put(tile, bitrev(i, RBITS) * COLS + c, Fr29::unpack(src[idx]));
    }
    for (uint32_t i = threadIdx.x; i < (rows >> 1); i += blockDim.x) lazy::put(W, i, Fr29::unpack(Wg[i]));
    __syncthreads();
    lazy_stages<COLS, RBITS>(tile, W);
    // the closing product: with the radix' one (2^261 mod p: x 2^256 * 2^261 / 2^261 = x 2^256), or -- coset_ifft -- with zeta^-(i mod 3) * 2^5
    const f29_t one = Fr29::unpack(one29), z1 = Fr29::unpack(fin.z1), z2 = Fr29::unpack(fin.z2);
    for (uint32_t e = threadIdx.x; e < rows * COLS; e += blockDim.x) {
        uint32_t kp = e / COLS, c = e % COLS;
        size_t oidx = (size_t)(k1_0 + c) + ((size_t)out_mid << r1) + ((size_t)kp << (pa.log_n - RBITS));
        if (!fin.on) {                 // r05: nothing to scale -- the lazy value is reduced directly (Fr29::reduce_lazy) instead of multiplied by the radix' one
            dst[oidx] = Fr29::to_canonical_fe(Fr29::reduce_lazy(lazy::get(tile, e)));
            continue;
        }
        const uint32_t r3 = (uint32_t)(oidx % 3);
        const f29_t m = r3 == 1 ? z1 : (r3 == 2 ? z2 : one);
        dst[oidx] = Fr29::to_canonical_fe(Fr29::mul(lazy::get(tile, e), m));
    }
}

// ---------------------------------------------------------------------------------------------
// host side: constants, plans
// ---------------------------------------------------------------------------------------------
struct Consts {
    fe_t root, root_inv, two_inv, zeta, zeta2;
};
static const Consts &consts() {
    static Consts c = [] {
        Consts k;
        // ROOT_OF_UNITY = 7^((r-1) / 2^28)  [3P halo2curves bn256::Fr, pinned by src/fft.rs:241-260]
        uint32_t e[8];
        for (int i = 0; i < 8; ++i) e[i] = FrP::p(i);
        e[0] -= 1;
        for (int s = 0; s < 28; ++s) {
            for (int i = 0; i < 8; ++i) e[i] = (e[i] >> 1) | (i < 7 ? (e[i + 1] << 31) : 0);
        }
        k.root = Fr::pow(Fr::from_u64(7), e);
        k.root_inv = Fr::inv(k.root);
        k.two_inv = Fr::inv(Fr::from_u64(2));
        // ZETA (WithSmallOrderMulGroup<3>) [3P halo2curves], canonical value:
        fe_t z;
        const uint32_t zc[8] = {0x36636f23u, 0xb8ca0b2du, 0xec2bc5e9u, 0xcc37a73fu, 0x3fd84104u, 0x048b6e19u, 0xe131a029u, 0x30644e72u};
        for (int i = 0; i < 8; ++i) z.v[i] = zc[i];
        k.zeta = Fr::to_mont(z);
        k.zeta2 = Fr::sqr(k.zeta);
        return k;
    }();
    return c;
}
static fe_t omega_for(uint32_t k, bool inverse) {      // src/fft.rs:12-23
    fe_t w = inverse ? consts().root_inv : consts().root;
    for (uint32_t i = k; i < FR_S; ++i) w = Fr::sqr(w);
    return w;
}

fe_t omega(uint32_t k, bool inverse) { return omega_for(k, inverse); }
fe_t zeta() { return consts().zeta; }

struct Plan {
    uint32_t log_n = 0;
    bool inverse = false;
    uint32_t npass = 0;
    uint32_t radix_bits[4] = {0, 0, 0, 0};
    fe_t *W[4] = {nullptr, nullptr, nullptr, nullptr};   // w_(2^r)^e tables per pass (small path: W[0] = w_n^e)
    fe_t *T[4] = {nullptr, nullptr, nullptr, nullptr};   // inter-digit twiddles after pass j (j < npass-1)
    bool mul29 = false;                                  // multi-pass plans: butterflies on the 9 x 29-bit multiplier, tables times 2^5
    fe_t scale;                                          // n^-1 for ifft (small path only)
    fe_t *scratch = nullptr;                             // n elements (multi-pass ping buffer)
    hipEvent_t done = nullptr;                           // recorded behind the last transform that used `scratch` (see run)
    std::mutex *mu = nullptr;                            // orders the transforms that share `scratch` (one per plan: other lengths do not wait)
    fe_t unit;                                           // the tables' common factor in Montgomery form: 1, or 2^5 (mul29)
};

static std::mutex g_mu;
static uint32_t g_max_radix = 8;   // tuning knob (4..8): digits per pass; never changes results
static std::map<std::tuple<int, uint32_t, bool>, Plan> g_plans;      // per (device, length, direction): the tables live in that device's HBM
// r04: multi-pass transforms run on the lazy 9 x 29-bit tile (k_ntt_pass_lazy / k_ntt_last_lazy) -- 2^24: fft 2.33 vs 2.65 ms, ifft 2.21 vs
// 2.47, coset_ifft 2.12 vs 2.62; 2^20: 0.167 vs 0.201 (profiles/r04_ab_ntt_lazy_tile.txt); the canonical 8 x 32 tile kernels are gone (r06).

static void fill(fe_t *T, uint32_t log_entries, uint32_t lo_bits, const fe_t &base, const fe_t &scale, hipStream_t st) {
    size_t n = (size_t)1 << log_entries;
    SRS_LAUNCH(k_fill_table, ((uint32_t)((n + 255) / 256)), (256), 0, st, T, log_entries, lo_bits, base, scale);
}

static Plan &get_plan(uint32_t log_n, bool inverse, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    int device = 0;
    SRS_HIP_CHECK(hipGetDevice(&device));
    auto key = std::make_tuple(device, log_n, inverse);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    Plan p;
    p.log_n = log_n;
    p.inverse = inverse;
    fe_t ninv = Fr::one();
    if (inverse)
        for (uint32_t i = 0; i < log_n; ++i) ninv = Fr::mul(ninv, consts().two_inv);   // TWO_INV^k, src/fft.rs:25-27
    p.scale = ninv;
    if (log_n <= SMALL_LOG) {
        p.npass = 1;
        p.radix_bits[0] = log_n;
        if (log_n >= 1) {
            SRS_HIP_CHECK(hipMalloc((void **)&p.W[0], sizeof(fe_t) << (log_n - 1)));
            fill(p.W[0], log_n - 1, 0, omega_for(log_n, inverse), Fr::one(), st);
        }
    } else {
        p.npass = (log_n + g_max_radix - 1) / g_max_radix;
        if (p.npass > 4) { set_error("ntt: max radix too small for this length"); throw DeviceError{4}; }
        uint32_t left = log_n;
        for (uint32_t j = 0; j < p.npass; ++j) {
            uint32_t r = (left + (p.npass - j) - 1) / (p.npass - j);
            p.radix_bits[j] = r;
            left -= r;
            if (r < 4) { set_error("ntt: digit narrower than 4 bits (raise max radix)"); throw DeviceError{4}; }
        }
        p.mul29 = true;
        fe_t unit = Fr::one();              // the tables' common factor: 1, or 2^5 for the 2^261-radix multiplier (lazy tile)
        if (p.mul29)
            for (int d = 0; d < 5; ++d) unit = Fr::add(unit, unit);
        p.unit = unit;
        uint32_t below = log_n;
        for (uint32_t j = 0; j < p.npass; ++j) {
            uint32_t r = p.radix_bits[j];
            below -= r;
            SRS_HIP_CHECK(hipMalloc((void **)&p.W[j], sizeof(fe_t) << (r - 1)));
            fill(p.W[j], r - 1, 0, omega_for(r, inverse), unit, st);
            if (j + 1 < p.npass) {
                uint32_t m = r + below;     // M_j = 2^m entries: T[k][rest] = w_M^(k*rest), k < 2^r, rest < 2^below
                SRS_HIP_CHECK(hipMalloc((void **)&p.T[j], sizeof(fe_t) << m));
                fill(p.T[j], m, below, omega_for(m, inverse), j == 0 ? Fr::mul(ninv, unit) : unit, st);
            }
        }
        SRS_HIP_CHECK(hipMalloc((void **)&p.scratch, sizeof(fe_t) << log_n));
        SRS_HIP_CHECK(hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
        SRS_HIP_CHECK(hipEventRecord(p.done, st));
        p.mu = new std::mutex();
    }
    SRS_HIP_CHECK(hipStreamSynchronize(st));
    return g_plans.emplace(key, p).first->second;
}

void set_max_radix_bits(uint32_t bits) {
    release_plans();
    std::lock_guard<std::mutex> lk(g_mu);
    g_max_radix = bits < 4 ? 4 : (bits > 8 ? 8 : bits);
}

void release_plans() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_plans) {
        for (int j = 0; j < 4; ++j) {
            if (kv.second.W[j]) (void)hipFree(kv.second.W[j]);
            if (kv.second.T[j]) (void)hipFree(kv.second.T[j]);
        }
        if (kv.second.scratch) (void)hipFree(kv.second.scratch);
        if (kv.second.done) (void)hipEventDestroy(kv.second.done);
        delete kv.second.mu;
    }
    g_plans.clear();
}

// Threads per workgroup of a pass.  The kernels hold ~124 VGPRs, i.e. 4 waves per SIMD: a 1024-thread workgroup is then ALONE on its CU
// and its load, butterfly and store phases run one after the other with nothing to overlap them.  512 threads (each thread two
// butterfly groups per stage) let two workgroups share a CU (2 x 68 KiB of LDS) and one computes while the other moves data.
// (A/B: profiles/r03_ab_ntt_threads.txt)
static uint32_t pass_threads(uint32_t tile_elems) { return tile_elems >= 1024 ? 512u : 256u; }

template <uint32_t R>
static void launch_pass(const fe_t *src, fe_t *dst, const PassArgs &pa, const Plan &p, uint32_t j, const Scale3 &pre,
                        hipStream_t st) {
    uint32_t blocks = 1u << (pa.log_n - R - COLS_LOG);
    uint32_t threads = pass_threads((1u << R) * COLS);
    SRS_LAUNCH((k_ntt_pass_lazy<R>), (blocks), (threads), 0, st, src, dst, pa, (const fe_t *)p.W[j], (const fe_t *)p.T[j], pre);
}
template <uint32_t R>
static void launch_last(const fe_t *src, fe_t *dst, const PassArgs &pa, const Plan &p, uint32_t j, const Scale3 &fin,
                        hipStream_t st) {
    uint32_t blocks = 1u << (pa.log_n - R - COLS_LOG);
    uint32_t threads = pass_threads((1u << R) * COLS);
    // the closing product's constants in the multiplier's radix: one = 2^261 mod p, the coset factors times 2^5
    Scale3 f29 = fin;
    if (fin.on) {
        f29.z1 = Fr::mul(fin.z1, p.unit);
        f29.z2 = Fr::mul(fin.z2, p.unit);
    }
    SRS_LAUNCH((k_ntt_last_lazy<R>), (blocks), (threads), 0, st, src, dst, pa, (const fe_t *)p.W[j], f29, p.unit);
}
#define DISPATCH_R(fn, r, ...)                          \
    switch (r) {                                        \
    case 4: fn<4>(__VA_ARGS__); break;                  \
    case 5: fn<5>(__VA_ARGS__); break;                  \
    case 6: fn<6>(__VA_ARGS__); break;                  \
    case 7: fn<7>(__VA_ARGS__); break;                  \
    default: fn<8>(__VA_ARGS__); break;                 \
    }

// a: device pointer to `batch` vectors of 2^log_n elements, `stride` apart.  In place.
void run(fe_t *a, uint32_t log_n, size_t stride, uint32_t batch, bool inverse, bool coset, hipStream_t st) {
    Plan &p = get_plan(log_n, inverse, st);
    prof::Scope ps("ntt_transform", st, (uint64_t)batch << log_n);
    Scale3 none;
    none.z1 = none.z2 = Fr::one();
    none.on = 0;
    Scale3 pre = none, fin = none;
    if (coset && !inverse) {        // distribute_powers_zeta(into_coset = true): [zeta, zeta^2]
        pre.z1 = consts().zeta;
        pre.z2 = consts().zeta2;
        pre.on = 1;
    }
    if (coset && inverse) {         // moving out of the coset: [zeta^2, zeta]
        fin.z1 = consts().zeta2;
        fin.z2 = consts().zeta;
        fin.on = 1;
    }
    if (p.npass == 1) {
        uint32_t n = 1u << log_n;
        uint32_t threads = n >= 1024 ? 512 : (n >= 128 ? n / 2 : 64);
        SRS_LAUNCH(k_ntt_small, (batch), (threads), 0, st, a, stride, log_n, (const fe_t *)p.W[0], p.scale,
                   (int)(inverse ? 1 : 0), pre, fin);
        return;
    }
    // A plan owns ONE ping buffer, and plans are shared by every caller of this length: the transforms that use it are ordered here -- the
    // launches of one transform are issued under the PLAN's lock (two host threads on one stream cannot interleave their passes) and a
    // transform on another stream waits for the event recorded behind the previous one of the same length (r04: tests/test_commit_gpu.py::
    // test_two_host_threads_distinct_handles caught two threads' 2^12 transforms sharing the buffer).  r05: the lock is per plan, so
    // transforms of different lengths (the primary's and the secondary's circuits on two streams) no longer serialise.
    // release_plans() / srs_ntt_set_max_radix_bits must not run concurrently with transforms (they destroy the plans).
    std::lock_guard<std::mutex> scratch_lock(*p.mu);
    SRS_HIP_CHECK(hipStreamWaitEvent(st, p.done, 0));
    for (uint32_t b = 0; b < batch; ++b) {
        fe_t *v = a + (size_t)b * stride;
        PassArgs pa;
        pa.log_n = log_n;
        pa.npass = p.npass;
        for (int j = 0; j < 4; ++j) pa.radix_bits[j] = p.radix_bits[j];
        // ping-pong so that the last (out-of-place) pass lands in v:
        //   npass odd : v -> scratch (pass 1), scratch in place ..., scratch -> v (last)
        //   the first pass may be out of place because it keeps positions; middle passes are in place.
        uint32_t below = log_n;
        const fe_t *src = v;
        for (uint32_t j = 0; j + 1 < p.npass; ++j) {
            uint32_t r = p.radix_bits[j];
            below -= r;
            pa.rbits = r;
            pa.lbits = below;
            pa.pass = j;
            DISPATCH_R(launch_pass, r, src, p.scratch, pa, p, j, (j == 0 ? pre : none), st);
            src = p.scratch;
        }
        uint32_t r = p.radix_bits[p.npass - 1];
        pa.rbits = r;
        pa.lbits = 0;
        pa.pass = p.npass - 1;
        DISPATCH_R(launch_last, r, (const fe_t *)p.scratch, v, pa, p, p.npass - 1, fin, st);
    }
    SRS_HIP_CHECK(hipEventRecord(p.done, st));
}

}  // namespace ntt
}  // namespace srs
