// msm.hip -- Pedersen vector commitment  C = sum_i v[i] * ck[i]  on gfx950.
//
// Replaces the body of `CommitmentKey::commit` = `best_multiexp(v, &ck[..v.len()]).to_affine()`
// (reference src/commitment.rs:81-90; call sites src/plonk/mod.rs:441-447,
// src/nifs/sangria/mod.rs:151-154).  Group arithmetic is exact, so any evaluation order gives the
// reference's affine point bit for bit.
//
// MI355X-first design (not halo2's chunk-per-thread serial Pippenger):
//   * The commitment key is static across fold steps, and HBM is 288 GB: at key creation every
//     base is expanded into the 16 window multiples  T[w][i] = 2^(16 w) * P_i  (affine, 16x the
//     key size).  All 16 windows of a scalar then land in ONE shared set of 2^15 signed-digit
//     buckets: no per-window bucket sets, no Horner doublings, one bucket reduction per MSM.
//   * scalars -> 16 signed 16-bit digits (k_digits); zero digits are skipped like halo2 does.
//   * counting sort of the (bucket, table-index) pairs with a 128 KiB LDS histogram per
//     workgroup (k_hist / k_scatter): hot buckets (0/1/small witnesses) are
//     aggregated in LDS and cost one global atomic per tile, not one per entry.
//   * load-balanced bucket accumulation (k_accum0): one thread per (bucket, part) with at most
//     L0 gathered mixed adds, so a bucket holding 20 % of all entries is spread over thousands
//     of threads; partial sums are combined by further levels (k_accum1) and a wavefront-level
//     strided sum + 64-lane shuffle tree (k_accum_final).
//   * bucket reduction  sum_b (b+1) B_b  via the 2-D row/column split (k_rowcol) and log-depth
//     suffix scans in LDS (k_reduce_final); the XYZZ result is normalised on the host.
//   * everything after k_accum0 is a chain of ~30 DEPENDENT additions on a nearly idle chip: those kernels give every
//     addition to a QUAD of lanes (Ec::add_quad, curve.cuh: 4 levels of independent products exchanged with DPP
//     quad_perm moves), 3.9 us instead of 9.8 us per addition.
// Batched entry: the d-1 cross-term commitments share one base prefix and run
// as one set of launches (grid.y / grid.z = batch index).
// r04: the sets of a STREAMED commit (the witness comes up from host memory in chunks, capi.hip: commit_streamed) run in SLOT MODE:
// every bucket owns 2^6 persistent partial sums in HBM, a level-0 part adds its entries into "its" slot (k_accum0s), so the accumulation
// levels / wave pass / bucket fold that every chunk used to run are ONE reduction of the slots per commit (k_slot_reduce); the plan of a
// set (k_plan_s) picks the part length on the device; parts beyond the slots (hot buckets) go through the level kernels into the
// bucket's last slot, launched only when the key expects them (msm.h: overflow_missed / note_commit).  Whole MSMs of >= 2^23 scalars
// take the 13 x 20-bit "wide" windows over a second table (16 virtual MSMs of 2^15 buckets each).
// The 29-bit products of THIS translation unit chain every column's multiply-accumulates from the previous column's carry (inline
// v_mad_u64_u32, field29.cuh): no 64-bit join per column, 127 instead of 132 VGPRs in k_accum0 / k_accum0s (4 waves per SIMD instead of
// 3): 13.85 -> 14.4 G mixed additions/s, profiles/r04_ab_slots_cuts.txt; one asm statement per RUN of multiply-accumulates, not per
// instruction (hipcc pads each statement with an s_nop, ~1 cycle each): + 2.7 % more, profiles/r04_ab_chain_blocks.txt.  Device code only;
// host and emulator builds keep the C form.
#define SRS_F29_CHAIN 1
#include "msm.h"
#include "tuning.h"
#include "curve29.cuh"
#include "prof.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace srs {
namespace msm {

// ---------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t upper_bucket(const uint32_t *__restrict__ tp, uint32_t t) {
    // largest b in [0, NBUCKET) with tp[b] <= t   (tp is non-decreasing, tp[0] = 0, t < tp[NBUCKET])
    uint32_t lo = 0, hi = NBUCKET;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (tp[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// global index of local scalar i under block-cyclic sharding (stripe = 2^STRIPE_LOG entries):
// rank r of `world` owns stripes s with s % world == r.
__device__ __forceinline__ size_t shard_global_index(size_t i, uint32_t rank, uint32_t world) {
    if (world == 1) return i;
    size_t s = i >> STRIPE_LOG, o = i & ((1u << STRIPE_LOG) - 1);
    return ((s * world + rank) << STRIPE_LOG) + o;
}

// ---------------------------------------------------------------------------------------------
// key expansion: T[w][i] = 2^(16 w) P_i
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void k_table_step(const affine_t *__restrict__ prev, xyzz_t *__restrict__ tmp, uint32_t n, int nbits) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    affine_t p = prev[i];
    xyzz_t a;
    if (Ec<C>::is_identity(p)) {
        a = Ec<C>::identity();
    } else {
        a = Ec<C>::dbl_affine(p);
        for (int k = 1; k < nbits; ++k) a = Ec<C>::dbl(a);
    }
    tmp[i] = a;
}

// batch normalisation XYZZ -> affine, NORM_G points per thread share one inversion
template <class C>
__global__ void k_normalize(const xyzz_t *__restrict__ in, affine_t *__restrict__ out, uint32_t n) {
    using F = typename C::F;
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    size_t lo = (size_t)g * NORM_G;
    if (lo >= n) return;
    uint32_t cnt = (uint32_t)((n - lo < NORM_G) ? (n - lo) : NORM_G);
    fe_t pre[NORM_G];
    fe_t acc = F::one();
    for (uint32_t j = 0; j < cnt; ++j) {
        pre[j] = acc;
        xyzz_t p = in[lo + j];
        if (!Ec<C>::is_identity(p)) acc = F::mul(acc, F::mul(p.zz, p.zzz));
    }
    fe_t inv = F::inv(acc);
    for (uint32_t j = cnt; j-- > 0;) {
        xyzz_t p = in[lo + j];
        affine_t o;
        if (Ec<C>::is_identity(p)) {
            o = Ec<C>::affine_identity();
        } else {
            fe_t i = F::mul(inv, pre[j]);                 // 1 / (zz * zzz)
            inv = F::mul(inv, F::mul(p.zz, p.zzz));
            o.x = F::mul(p.x, F::mul(i, p.zzz));
            o.y = F::mul(p.y, F::mul(i, p.zz));
        }
        out[lo + j] = o;
    }
}

// The finished table is kept in the R' = 2^261 Montgomery form of the 9 x 29-bit multiplier that k_accum0 runs on
// (field29.cuh): one pass over all NWIN * len entries when the key is created.  k_table_unform is the inverse, used by
// the few readers of the plain bases (srs_ck_get_bases / save_file, the on-curve count).
template <class C>
__global__ void k_table_form(affine_t *__restrict__ t, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    t[i] = Ec29<C>::table_form(t[i]);
}
template <class C>
__device__ __forceinline__ affine_t table_unform(const affine_t &q) {
    using G = typename C::F;
    fe_t c = G::one();
    for (int d = 0; d < 5; ++d) c = G::halve(c);          // 2^256 / 32 = 2^251 mod p as an integer: x 2^261 * 2^251 / 2^256 = x 2^256
    affine_t o;
    o.x = G::mul(q.x, c);
    o.y = G::mul(q.y, c);
    return o;
}
template <class C>
__global__ void k_table_unform(const affine_t *__restrict__ t, affine_t *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = table_unform<C>(t[i]);
}

// synthetic key: P_i = [h(seed, g)] G with g the GLOBAL index of local base i  (NOT the reference's
// SHAKE256 + hash_to_curve key, src/commitment.rs:55-79 -- hash_to_curve is [3P]; any valid points
// with no known small relation serve parity and timing)
__device__ __forceinline__ uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
template <class C>
__global__ void k_gen_bases(xyzz_t *__restrict__ tmp, uint32_t n, uint64_t seed, uint32_t rank, uint32_t world) {
    using F = typename C::F;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t g = shard_global_index(i, rank, world);
    uint64_t st = seed ^ (g * 0x100000001b3ull + 0xcbf29ce484222325ull);
    uint32_t k[8];
    for (int j = 0; j < 4; ++j) {
        uint64_t v = splitmix64(st);
        k[2 * j] = (uint32_t)v;
        k[2 * j + 1] = (uint32_t)(v >> 32);
    }
    k[7] &= 0x0FFFFFFFu;   // < 2^252 < group order
    k[0] |= 1u;
    affine_t gen;
    gen.x = F::one();
    if (C::ID == 0) {
        gen.y = F::dbl(F::one());                                  // bn256 G1 generator (1, 2)
    } else {
        // grumpkin generator (1, sqrt(-16)) = 17631683881184975370165255887551781615748388533673675138860
        fe_t y;
        const uint32_t yc[8] = {0x823f272cu, 0x833fc48du, 0xf1181294u, 0x2d270d45u, 0x06a45d63u, 0xcf135e75u, 0x00000002u, 0u};
        for (int j = 0; j < 8; ++j) y.v[j] = yc[j];
        gen.y = F::to_mont(y);
    }
    tmp[i] = Ec<C>::mul_canon(k, gen);
}

// is_on_curve over a key (reference: load_or_setup_cache validates every point, src/commitment.rs:148-160)
template <class C>
__global__ void k_on_curve(const affine_t *__restrict__ pts, uint32_t n, uint32_t *__restrict__ bad) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!Ec<C>::is_on_curve(table_unform<C>(pts[i]))) atomicAdd(bad, 1u);
}

// ---------------------------------------------------------------------------------------------
// 1. scalars -> signed 16-bit digits
// dig[w * n + i] = 0xFFFF (zero digit) | (|d|-1) | sign << 15
// ---------------------------------------------------------------------------------------------
// the batch descriptor travels as a kernel argument (no host-to-device copies of pointer / length arrays per call)
struct BatchDesc {
    const fe_t *ptr[BATCH_ARGS];
    uint32_t n[BATCH_ARGS];
    uint32_t base[BATCH_ARGS];      // first base of MSM m inside the key (0 for the usual prefix commit; chunked commits slide it)
};

template <class C>
__global__ void k_digits(BatchDesc bd, uint16_t *__restrict__ dig, size_t dig_stride, int is_mont, uint32_t rank, uint32_t world,
                         uint32_t *__restrict__ count_zero, uint32_t n_zero) {
    using S = typename C::S;
    uint32_t m = blockIdx.y;
    uint32_t n = bd.n[m];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // the bucket counters k_hist adds into start at zero: cleared here (k_digits precedes k_hist on the stream) instead of by a
    // memset launch of its own per MSM
    for (uint32_t j = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; j < n_zero; j += gridDim.x * gridDim.y * blockDim.x)
        count_zero[j] = 0;
    if (i >= n) return;
    fe_t s = bd.ptr[m][shard_global_index(i, rank, world)];
    if (is_mont) s = S::from_mont(s);
    uint16_t *d = dig + (size_t)m * dig_stride;
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < NWIN; ++w) {
        uint32_t raw = (s.v[w >> 1] >> ((w & 1) * 16)) & 0xFFFFu;
        uint32_t v = raw + carry;
        uint32_t code;
        if (v > 0x8000u) {
            code = ((0x10000u - v) - 1u) | 0x8000u;   // negative digit, magnitude 1..0x7FFF
            carry = 1;
        } else {
            code = v ? (v - 1u) : 0xFFFFu;            // positive digit 1..0x8000, or zero
            carry = 0;
        }
        d[(size_t)w * n + i] = (uint16_t)code;
    }
}

// ---------------------------------------------------------------------------------------------
// 2. counting sort by bucket, LDS histogram per workgroup
// grid = (tiles, NWIN, batch), block = SORT_THREADS
// ---------------------------------------------------------------------------------------------
// fn(i, code) for every digit slot i in [lo, hi): 8 digits per 16-byte load once the pointer is aligned (a 2-byte load per
// lane uses 1/8 of the load width: the digit streams are read by k_hist, k_scatter and k_group)
template <class Fn>
__device__ __forceinline__ void for_each_digit(const uint16_t *__restrict__ d, uint32_t lo, uint32_t hi, Fn fn) {
    uint32_t head = lo;
    while (head < hi && (reinterpret_cast<uintptr_t>(d + head) & 15u)) ++head;      // at most 7 unaligned slots
    for (uint32_t i = lo + threadIdx.x; i < head; i += blockDim.x) fn(i, (uint32_t)d[i]);
    const uint32_t nvec = (hi - head) / 8;
    const uint4 *v = reinterpret_cast<const uint4 *>(d + head);
    auto eight = [&](uint32_t q, const uint4 &x) {
        const uint32_t i0 = head + q * 8;
        fn(i0 + 0, x.x & 0xFFFFu); fn(i0 + 1, x.x >> 16);
        fn(i0 + 2, x.y & 0xFFFFu); fn(i0 + 3, x.y >> 16);
        fn(i0 + 4, x.z & 0xFFFFu); fn(i0 + 5, x.z >> 16);
        fn(i0 + 6, x.w & 0xFFFFu); fn(i0 + 7, x.w >> 16);
    };
    // four 16-byte loads in flight per thread (r04: one load per loop round left a workgroup with 16 loads in flight -- the digit streams
    // of a large chunk took 25 memory latencies per thread)
    uint32_t q = threadIdx.x;
    for (; q + 3 * blockDim.x < nvec; q += 4 * blockDim.x) {
        const uint4 x0 = v[q], x1 = v[q + blockDim.x], x2 = v[q + 2 * blockDim.x], x3 = v[q + 3 * blockDim.x];
        eight(q, x0); eight(q + blockDim.x, x1); eight(q + 2 * blockDim.x, x2); eight(q + 3 * blockDim.x, x3);
    }
    for (; q < nvec; q += blockDim.x) eight(q, v[q]);
    for (uint32_t i = head + nvec * 8 + threadIdx.x; i < hi; i += blockDim.x) fn(i, (uint32_t)d[i]);
}

__device__ __forceinline__ void tile_histogram(uint32_t *h, const uint16_t *__restrict__ d, uint32_t lo, uint32_t hi) {
    for (uint32_t b = threadIdx.x; b < NBUCKET; b += blockDim.x) h[b] = 0;
    __syncthreads();
    for_each_digit(d, lo, hi, [&](uint32_t, uint32_t code) {
        if (code != 0xFFFFu) atomicAdd(&h[code & 0x7FFFu], 1u);
    });
    __syncthreads();
}

__global__ void SRS_KERNEL_BOUNDS(SORT_THREADS, 1)
    k_hist(const uint16_t *__restrict__ dig, size_t dig_stride, BatchDesc bd,
           uint32_t *__restrict__ count /* [batch][NBUCKET] */, uint32_t SORT_TILE,
           uint32_t *__restrict__ tile_hist /* two-pass sort: [batch][SEG][tiles * NWIN], or nullptr */) {
    __shared__ uint32_t h[NBUCKET];
    uint32_t m = blockIdx.z, w = blockIdx.y;
    uint32_t n = bd.n[m];
    uint32_t lo = blockIdx.x * SORT_TILE;
    const uint32_t T1 = gridDim.x * NWIN, t1 = w * gridDim.x + blockIdx.x;
    if (lo >= n) {
        if (tile_hist)
            for (uint32_t sgm = threadIdx.x; sgm < SEG; sgm += blockDim.x) tile_hist[((size_t)m * SEG + sgm) * T1 + t1] = 0;
        return;
    }
    uint32_t hi = lo + SORT_TILE < n ? lo + SORT_TILE : n;
    const uint16_t *d = dig + (size_t)m * dig_stride + (size_t)w * n;
    tile_histogram(h, d, lo, hi);
    uint32_t *cnt = count + (size_t)m * NBUCKET;
    // every workgroup flushes from its own offset: 256 of them walking the counters in the same order met at the same addresses
    // (k_hist 36.8 -> 33.1 us on average, profiles/r04_ab_chain_blocks.txt section 5)
    const uint32_t rot = ((blockIdx.y * gridDim.x + blockIdx.x) * 128u) & (NBUCKET - 1);
    for (uint32_t i = threadIdx.x; i < NBUCKET; i += blockDim.x) {      // (r05: two counters per 64-bit atomic change nothing: 35.6 us either way, profiles/r05_ab_misc.txt)
        const uint32_t b = (i + rot) & (NBUCKET - 1);
        uint32_t c = h[b];
        if (c) atomicAdd(&cnt[b], c);
    }
    if (tile_hist) {                       // entries of this tile per segment (= SEG_BUCKETS consecutive buckets)
        for (uint32_t sgm = threadIdx.x; sgm < SEG; sgm += blockDim.x) {
            uint32_t c = 0;
            for (uint32_t b = 0; b < SEG_BUCKETS; ++b) c += h[sgm * SEG_BUCKETS + ((b + sgm) & (SEG_BUCKETS - 1))];   // skewed: no bank conflicts
            tile_hist[((size_t)m * SEG + sgm) * T1 + t1] = c;
        }
    }
}

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds, uint32_t *total);   // defined with k_plan

// ---- two-pass (MSD) variant of the scatter, for large MSMs -----------------------------------------------------------
// The single-pass k_scatter writes every 4-byte entry to a random place of the whole sorted array (one bucket out of
// 2^15 per entry: a tile holds < 1 entry per bucket, so nothing coalesces and every (tile, bucket) pair costs a global
// atomic) -- 15 % of a 12 * 2^20 witness commit.  Two passes instead:
//   A  k_group   : entries -> SEG = 128 segments of 256 consecutive buckets.  A tile contributes thousands of entries to
//                  each segment, at offsets known from the per-tile segment counts (k_hist) scanned by k_scan_seg: the
//                  writes are long coalesced runs, no global atomics.
//   B  k_scatter2: tiles of the GROUPED array do the bucket-level counting sort.  A tile now touches 1-2 segments, i.e.
//                  a few hundred buckets inside a ~MB region that lives in L2: the random 4-byte writes merge there and
//                  the number of global reservations drops ~40x.
// Within a bucket the order of entries is irrelevant (exact group arithmetic), so neither pass needs to be stable.
__global__ void SRS_KERNEL_BOUNDS(1024, 1)
    k_scan_seg(uint32_t *__restrict__ tile_hist, uint32_t T1, const uint32_t *__restrict__ plan, size_t plan_stride) {
    // grid = (SEG, batch): exclusive scan of the T1 per-tile counts of one segment, offset by the segment's start off[seg * 256]
    __shared__ uint32_t lds[64];
    const uint32_t sgm = blockIdx.x, m = blockIdx.y;
    uint32_t *row = tile_hist + ((size_t)m * SEG + sgm) * T1;
    const uint32_t base = plan[(size_t)m * plan_stride + (size_t)sgm * SEG_BUCKETS];        // plan[0][b] = entry offset of bucket b
    uint32_t carry = 0;
    for (uint32_t at = 0; at < T1; at += blockDim.x) {
        const uint32_t i = at + threadIdx.x;
        const uint32_t v = i < T1 ? row[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, lds, &total);
        if (i < T1) row[i] = base + carry + ex;
        carry += total;
    }
}

// r03: both passes go through LDS.  A lane-per-entry global store of the first version met ~50 different cache lines per
// wavefront instruction -- the address path of a CU takes about one line per cycle, so k_group / k_scatter2 ran at ~1/4 of what
// their traffic costs.  Now a workgroup sorts a sub-tile of entries by segment (bucket) inside LDS and copies it out in index
// order: consecutive lanes store consecutive addresses of a run.
constexpr uint32_t GRP_PER = 16, GRP_SUB = SORT_THREADS * GRP_PER;      // digit slots per thread / per sub-tile of k_group
__global__ void SRS_KERNEL_BOUNDS(SORT_THREADS, 1)
    k_group(const uint16_t *__restrict__ dig, size_t dig_stride, BatchDesc bd, const uint32_t *__restrict__ tile_hist,
            uint16_t *__restrict__ gkey, uint32_t *__restrict__ gpay, size_t g_stride, uint32_t table_stride, uint32_t SORT_TILE,
            const uint32_t *__restrict__ plan, size_t plan_stride, uint16_t *__restrict__ tb, size_t tb_stride, uint32_t arr) {
    __shared__ uint32_t cur[SEG], cnt[SEG], lb[SEG], sc[64];
    __shared__ uint32_t spay[GRP_SUB];
    __shared__ uint16_t skey[GRP_SUB];
    const uint32_t m = blockIdx.z, w = blockIdx.y, tid = threadIdx.x;
    if (tb) {
        // slot mode: k_expand's work (the thread -> bucket map of level 0, from the plan array `arr`) rides on this launch, a bucket per
        // wavefront and round -- one dependent launch less per set of a chunked commit (r04)
        const uint32_t *tp = plan + (size_t)m * plan_stride + (size_t)arr * (NBUCKET + 1);
        uint16_t *map = tb + (size_t)m * tb_stride;
        const uint32_t waves = gridDim.x * gridDim.y * (blockDim.x >> 6);
        const uint32_t wv = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (tid >> 6), lane = tid & 63u;
        for (uint32_t b = wv; b < NBUCKET; b += waves) {
            const uint32_t s = tp[b], e = tp[b + 1];
            for (uint32_t t = s + lane; t < e; t += 64) map[t] = (uint16_t)b;
        }
    }
    const uint32_t n = bd.n[m];
    const uint32_t lo = blockIdx.x * SORT_TILE;
    if (lo >= n) return;
    const uint32_t hi = lo + SORT_TILE < n ? lo + SORT_TILE : n;
    const uint32_t T1 = gridDim.x * NWIN, t1 = w * gridDim.x + blockIdx.x;
    for (uint32_t sgm = tid; sgm < SEG; sgm += blockDim.x) cur[sgm] = tile_hist[((size_t)m * SEG + sgm) * T1 + t1];
    const uint16_t *d = dig + (size_t)m * dig_stride + (size_t)w * n;
    uint16_t *ok = gkey + (size_t)m * g_stride;
    uint32_t *op = gpay + (size_t)m * g_stride;
    const uint32_t pay0 = w * table_stride + bd.base[m];
    for (uint32_t sub = lo; sub < hi; sub += GRP_SUB) {                 // workgroup-uniform
        for (uint32_t sgm = tid; sgm < SEG; sgm += blockDim.x) cnt[sgm] = 0;
        __syncthreads();
        // the thread's GRP_PER consecutive digit slots (two 16-byte loads when they are aligned and inside the tile)
        const uint32_t i0 = sub + tid * GRP_PER;
        uint32_t code[GRP_PER], rank[GRP_PER];
        if (i0 + GRP_PER <= hi && (reinterpret_cast<uintptr_t>(d + i0) & 15u) == 0) {
            const uint4 *v = reinterpret_cast<const uint4 *>(d + i0);
#pragma unroll
            for (uint32_t q = 0; q < GRP_PER / 8; ++q) {
                const uint4 x = v[q];
                code[8 * q + 0] = x.x & 0xFFFFu; code[8 * q + 1] = x.x >> 16;
                code[8 * q + 2] = x.y & 0xFFFFu; code[8 * q + 3] = x.y >> 16;
                code[8 * q + 4] = x.z & 0xFFFFu; code[8 * q + 5] = x.z >> 16;
                code[8 * q + 6] = x.w & 0xFFFFu; code[8 * q + 7] = x.w >> 16;
            }
        } else {
#pragma unroll
            for (uint32_t k = 0; k < GRP_PER; ++k) code[k] = i0 + k < hi ? (uint32_t)d[i0 + k] : 0xFFFFu;
        }
#pragma unroll
        for (uint32_t k = 0; k < GRP_PER; ++k)
            rank[k] = code[k] != 0xFFFFu ? atomicAdd(&cnt[(code[k] & 0x7FFFu) / SEG_BUCKETS], 1u) : 0u;
        __syncthreads();
        uint32_t n_sub;
        const uint32_t ex = block_exclusive_scan(tid < SEG ? cnt[tid] : 0u, sc, &n_sub);
        if (tid < SEG) lb[tid] = ex;
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < GRP_PER; ++k) {
            if (code[k] != 0xFFFFu) {
                const uint32_t bkt = code[k] & 0x7FFFu, pos = lb[bkt / SEG_BUCKETS] + rank[k];
                skey[pos] = (uint16_t)bkt;
                spay[pos] = (pay0 + i0 + k) | ((code[k] & 0x8000u) << 16);
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < n_sub; i += blockDim.x) {             // index order: a segment's run goes out as one contiguous piece
            const uint32_t key = skey[i], sgm = key / SEG_BUCKETS, g = cur[sgm] + (i - lb[sgm]);
            ok[g] = (uint16_t)key;
            op[g] = spay[i];
        }
        __syncthreads();
        if (tid < SEG) cur[tid] += cnt[tid];
    }
}

constexpr uint32_t S2_PER = 8, S2_RANGE = 2 * SEG_BUCKETS;               // entries per thread; buckets a tile may span on the LDS path
static_assert(SORT_TILE2 == SORT_THREADS * S2_PER, "k_scatter2: one tile = S2_PER entries per thread");
__global__ void SRS_KERNEL_BOUNDS(SORT_THREADS, 2)
    k_scatter2(const uint16_t *__restrict__ gkey, const uint32_t *__restrict__ gpay, size_t g_stride,
               const uint32_t *__restrict__ plan, size_t plan_stride, uint32_t *__restrict__ cursor /* [batch][NBUCKET] */,
               uint32_t *__restrict__ sorted, size_t sorted_stride, uint32_t TILE2) {
    __shared__ uint32_t cnt[S2_RANGE], lb[S2_RANGE], gb[S2_RANGE], sc[64];
    __shared__ uint32_t spay[SORT_TILE2];
    __shared__ uint16_t skey[SORT_TILE2];
    const uint32_t m = blockIdx.y, tid = threadIdx.x;
    const uint32_t total = plan[(size_t)m * plan_stride + NBUCKET];        // number of non-zero digits of this MSM
    // XCD-aware tile mapping: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so workgroup b
    // takes tile (b % 8) * ceil(tiles / 8) + b / 8 -- every XCD sorts one contiguous eighth of the grouped array, i.e.
    // one eighth of the buckets, and the stores to a cache line all come from the same L2 and merge there
    const uint32_t n_tiles = (total + TILE2 - 1) / TILE2, per_xcd = (n_tiles + 7) / 8;
    if (blockIdx.x / 8 >= per_xcd) return;
    const uint32_t tile_id = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (tile_id >= n_tiles) return;
    const uint32_t lo = tile_id * TILE2;
    if (lo >= total) return;
    const uint32_t hi = lo + TILE2 < total ? lo + TILE2 : total;
    const uint16_t *key = gkey + (size_t)m * g_stride;
    const uint32_t *pay = gpay + (size_t)m * g_stride;
    uint32_t *cur = cursor + (size_t)m * NBUCKET;
    uint32_t *out = sorted + (size_t)m * sorted_stride;
    // the grouped array is ordered by segment: this tile only holds buckets of segments seg(first) .. seg(last)
    const uint32_t b_lo = ((uint32_t)key[lo] / SEG_BUCKETS) * SEG_BUCKETS;
    const uint32_t b_hi = ((uint32_t)key[hi - 1] / SEG_BUCKETS + 1) * SEG_BUCKETS;
    if (b_hi - b_lo > S2_RANGE) {            // a tile over more than two segments (few entries for the bucket range): entry by entry
        for (uint32_t i = lo + tid; i < hi; i += blockDim.x) out[atomicAdd(&cur[key[i]], 1u)] = pay[i];
        return;
    }
    for (uint32_t b = tid; b < S2_RANGE; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    const uint32_t i0 = lo + tid * S2_PER;
    uint32_t kk[S2_PER], pp[S2_PER], rank[S2_PER];
    if (i0 + S2_PER <= hi && (reinterpret_cast<uintptr_t>(key + i0) & 15u) == 0 && (reinterpret_cast<uintptr_t>(pay + i0) & 15u) == 0) {
        const uint4 x = *reinterpret_cast<const uint4 *>(key + i0);
        kk[0] = x.x & 0xFFFFu; kk[1] = x.x >> 16; kk[2] = x.y & 0xFFFFu; kk[3] = x.y >> 16;
        kk[4] = x.z & 0xFFFFu; kk[5] = x.z >> 16; kk[6] = x.w & 0xFFFFu; kk[7] = x.w >> 16;
        const uint4 p0 = *reinterpret_cast<const uint4 *>(pay + i0), p1 = *reinterpret_cast<const uint4 *>(pay + i0 + 4);
        pp[0] = p0.x; pp[1] = p0.y; pp[2] = p0.z; pp[3] = p0.w; pp[4] = p1.x; pp[5] = p1.y; pp[6] = p1.z; pp[7] = p1.w;
    } else {
#pragma unroll
        for (uint32_t k = 0; k < S2_PER; ++k) {
            kk[k] = i0 + k < hi ? (uint32_t)key[i0 + k] : 0xFFFFu;
            pp[k] = i0 + k < hi ? pay[i0 + k] : 0u;
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < S2_PER; ++k) rank[k] = kk[k] != 0xFFFFu ? atomicAdd(&cnt[kk[k] - b_lo], 1u) : 0u;
    __syncthreads();
    uint32_t n_tile;
    const uint32_t c = tid < S2_RANGE ? cnt[tid] : 0u;
    const uint32_t ex = block_exclusive_scan(c, sc, &n_tile);
    if (tid < S2_RANGE) {
        lb[tid] = ex;
        gb[tid] = c ? atomicAdd(&cur[b_lo + tid], c) : 0u;                 // reserve [base, base + c) of the bucket
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < S2_PER; ++k) {
        if (kk[k] != 0xFFFFu) {
            const uint32_t pos = lb[kk[k] - b_lo] + rank[k];
            skey[pos] = (uint16_t)(kk[k] - b_lo);
            spay[pos] = pp[k];
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n_tile; i += blockDim.x) {
        const uint32_t b = skey[i];
        out[gb[b] + (i - lb[b])] = spay[i];
    }
}

#ifndef SCATTER_RES_INFLIGHT
#define SCATTER_RES_INFLIGHT 4
#endif
__global__ void SRS_KERNEL_BOUNDS(SORT_THREADS, 1)
    k_scatter(const uint16_t *__restrict__ dig, size_t dig_stride, BatchDesc bd,
              uint32_t *__restrict__ cursor /* [batch][NBUCKET] */, uint32_t *__restrict__ sorted,
              size_t sorted_stride, uint32_t table_stride, uint32_t SORT_TILE) {
    __shared__ uint32_t h[NBUCKET];
    uint32_t m = blockIdx.z, w = blockIdx.y;
    uint32_t n = bd.n[m];
    uint32_t lo = blockIdx.x * SORT_TILE;
    if (lo >= n) return;
    uint32_t hi = lo + SORT_TILE < n ? lo + SORT_TILE : n;
    const uint16_t *d = dig + (size_t)m * dig_stride + (size_t)w * n;
    tile_histogram(h, d, lo, hi);
    uint32_t *cur = cursor + (size_t)m * NBUCKET;
    // the reservations are returning device-scope atomics: RES in flight per thread (32 = all of a thread's buckets in one round
    // measured no faster, r03: the kernel waits for its scattered 4-byte stores, profiles/r03_sq_counters_msm.txt)
    constexpr int RES = SCATTER_RES_INFLIGHT;
    for (uint32_t b0 = threadIdx.x; b0 < NBUCKET; b0 += RES * blockDim.x) {
        uint32_t c[RES], r[RES];
#pragma unroll
        for (int u = 0; u < RES; ++u) c[u] = b0 + u * blockDim.x < NBUCKET ? h[b0 + u * blockDim.x] : 0u;
#pragma unroll
        for (int u = 0; u < RES; ++u) r[u] = c[u] ? atomicAdd(&cur[b0 + u * blockDim.x], c[u]) : 0u;   // reserve [base, base+c)
#pragma unroll
        for (int u = 0; u < RES; ++u) if (c[u]) h[b0 + u * blockDim.x] = r[u];
    }
    __syncthreads();
    uint32_t *out = sorted + (size_t)m * sorted_stride;
    for_each_digit(d, lo, hi, [&](uint32_t i, uint32_t code) {
        if (code != 0xFFFFu) {
            uint32_t pos = atomicAdd(&h[code & 0x7FFFu], 1u);
            out[pos] = (w * table_stride + bd.base[m] + i) | ((code & 0x8000u) << 16);
        }
    });
}

// Wide windows: the NSEG_W virtual MSMs share ONE parts / map array per level parity instead of a worst-case stride each
// (a witness of small values puts every entry into segment 0).  base[l][v] = first level-l part of segment v in the flat
// thread space of level l (base[l][NSEG_W] = their number); in memory the parts of level l of segment v start at
// base[l & 1][v] of ping (even l) / pong (odd l): a level never has more parts than the level two below it.
struct Link {
    uint32_t base[MAX_LEVELS + 1][NSEG_W + 1];
};
__device__ __forceinline__ uint32_t link_segment(const uint32_t *__restrict__ base, uint32_t t) {   // base[v] <= t < base[v + 1]
    uint32_t v = 0;
#pragma unroll
    for (uint32_t u = 1; u < NSEG_W; ++u) v += (t >= base[u]) ? 1u : 0u;
    return v;
}

// ---------------------------------------------------------------------------------------------
// 2b. wide windows: scalars -> 13 signed 20-bit digits -> entries grouped by segment (the top 4 bits of the bucket)
// code = 0xFFFFFFFF (zero digit) | (|d| - 1) [19 bits] | sign << 31
// ---------------------------------------------------------------------------------------------
struct WideDesc {
    const fe_t *ptr;
    uint32_t n, base;               // scalars, first base inside the key
    uint32_t rank, world;           // as k_digits
    int is_mont;
};

template <class C>
__device__ __forceinline__ void wide_codes(const WideDesc &wd, uint32_t i, uint32_t (&code)[NWIN_W]) {
    using S = typename C::S;
    if (i >= wd.n) {
#pragma unroll
        for (int w = 0; w < NWIN_W; ++w) code[w] = 0xFFFFFFFFu;
        return;
    }
    fe_t s = wd.ptr[shard_global_index(i, wd.rank, wd.world)];
    if (wd.is_mont) s = S::from_mont(s);
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < NWIN_W; ++w) {
        const int bit = WBITS_W * w, limb = bit >> 5, sh = bit & 31;
        uint64_t two = s.v[limb];
        if (limb + 1 < 8) two |= (uint64_t)s.v[limb + 1] << 32;
        const uint32_t v = ((uint32_t)(two >> sh) & 0xFFFFFu) + carry;
        if (v > 0x80000u) {
            code[w] = ((0x100000u - v) - 1u) | 0x80000000u;   // negative digit, magnitude 1..0x7FFFF (v = 2^20: zero digit, carry)
            carry = 1;
        } else {
            code[w] = v ? (v - 1u) : 0xFFFFFFFFu;             // positive digit 1..0x80000, or zero
            carry = 0;
        }
    }
}

// The two segment passes over the scalars (the digits are recomputed rather than stored: 32 B read per scalar instead of
// 52 B written and read).  A workgroup takes WIDE_TILE scalars = up to 6656 entries; the 16 segment counters are kept per
// wavefront in LDS (64 words = 64 banks), so an atomic only meets the lanes of its own wavefront.
//   GROUP = false: tile_cnt[v][tile] = entries of the tile in segment v; seg_total[v] += the same
//   GROUP = true : tile_cnt holds the scanned offsets (k_seg_scan): the tile's entries are ordered by segment in LDS (the
//                  returning atomic is the entry's rank) and leave as 16 coalesced runs of
//                  (key = low 15 bits of the bucket, payload = table index | sign << 31)
template <class C, bool GROUP>
__global__ void SRS_KERNEL_BOUNDS(WIDE_THREADS, 1)
    k_seg_pass(WideDesc wd, uint32_t *__restrict__ tile_cnt, uint32_t T, uint32_t *__restrict__ seg_total,
               uint16_t *__restrict__ gkey, uint32_t *__restrict__ gpay, uint32_t table_stride) {
    constexpr uint32_t NW = WIDE_THREADS / 64, STAGE = GROUP ? WIDE_TILE * NWIN_W : 1;
    __shared__ uint32_t wc[NW][NSEG_W];          // counts, then (GROUP) the next free staging slot of (wavefront, segment)
    __shared__ uint32_t ss[NSEG_W + 1], goff[NSEG_W];
    __shared__ uint16_t skey[STAGE];
    __shared__ uint32_t spay[STAGE];
    const uint32_t tile = blockIdx.x, wave = threadIdx.x >> 6;
    if (threadIdx.x < NW * NSEG_W) (&wc[0][0])[threadIdx.x] = 0;
    uint32_t code[WIDE_PER][NWIN_W];
#pragma unroll
    for (uint32_t k = 0; k < WIDE_PER; ++k) wide_codes<C>(wd, tile * WIDE_TILE + k * WIDE_THREADS + threadIdx.x, code[k]);
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < WIDE_PER; ++k) {
#pragma unroll
        for (int w = 0; w < NWIN_W; ++w) {
            const uint32_t c = code[k][w];
            if (c != 0xFFFFFFFFu) atomicAdd(&wc[wave][(c >> 15) & (NSEG_W - 1)], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < NSEG_W) {
        const uint32_t v = threadIdx.x;
        uint32_t tot = 0;
        for (uint32_t w = 0; w < NW; ++w) tot += wc[w][v];
        if (!GROUP) {
            tile_cnt[(size_t)v * T + tile] = tot;
            if (tot) atomicAdd(&seg_total[v], tot);
        } else {
            ss[v + 1] = tot;
            goff[v] = tile_cnt[(size_t)v * T + tile];
        }
    }
    if (!GROUP) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t v = 0; v < NSEG_W; ++v) {
            const uint32_t c = ss[v + 1];
            ss[v] = run;
            run += c;
        }
        ss[NSEG_W] = run;
    }
    __syncthreads();
    if (threadIdx.x < NSEG_W) {                  // counts -> first staging slot of (wavefront, segment)
        const uint32_t v = threadIdx.x;
        uint32_t run = ss[v];
        for (uint32_t w = 0; w < NW; ++w) {
            const uint32_t c = wc[w][v];
            wc[w][v] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < WIDE_PER; ++k) {
        const uint32_t i = tile * WIDE_TILE + k * WIDE_THREADS + threadIdx.x;
#pragma unroll
        for (int w = 0; w < NWIN_W; ++w) {
            const uint32_t c = code[k][w];
            if (c != 0xFFFFFFFFu) {
                const uint32_t pos = atomicAdd(&wc[wave][(c >> 15) & (NSEG_W - 1)], 1u);
                skey[pos] = (uint16_t)(c & 0x7FFFu);
                spay[pos] = ((uint32_t)w * table_stride + wd.base + i) | (c & 0x80000000u);
            }
        }
    }
    __syncthreads();
    for (uint32_t v = 0; v < NSEG_W; ++v) {
        const uint32_t lo = ss[v], hi = ss[v + 1], g0 = goff[v];
        for (uint32_t e = lo + threadIdx.x; e < hi; e += WIDE_THREADS) {
            gkey[g0 + (e - lo)] = skey[e];
            gpay[g0 + (e - lo)] = spay[e];
        }
    }
}

// grid = NSEG_W: block v turns row v of tile_cnt into exclusive offsets into the grouped arrays (segment v starts where the
// totals of segments < v end).  Block 0 also leaves seg_off[0..NSEG_W] and the tile space of the per-segment counting sort:
// tile_base[v] = first workgroup of segment v when every segment is cut into tiles of `tile_g` entries.
__global__ void SRS_KERNEL_BOUNDS(1024, 1)
    k_seg_scan(uint32_t *__restrict__ tile_cnt, uint32_t T, const uint32_t *__restrict__ seg_total, uint32_t *__restrict__ seg_off,
               uint32_t *__restrict__ tile_base /* [NSEG_W + 1], then the tile size */,
               uint32_t *__restrict__ tile_base2 /* [NSEG_W + 1]: the same for the tiles of k_scatter2_g, then their size */,
               uint32_t small_tiles /* the launch of k_scatter2_g is sized for tiles of SORT_TILE2 / 4 (sets that may be sparse) */) {
    __shared__ uint32_t lds[64];
    const uint32_t v = blockIdx.x;
    uint32_t base = 0;
    for (uint32_t u = 0; u < v; ++u) base += seg_total[u];
    if (v == 0 && threadIdx.x == 0) {
        // tile size from the ACTUAL number of entries (zero digits are gone): <= SORT_TARGET_BLOCKS workgroups of 128 KiB LDS,
        // i.e. one round over the CUs, whatever the share of zeros
        uint32_t all = 0;
        for (uint32_t u = 0; u < NSEG_W; ++u) all += seg_total[u];
        uint32_t tile_g = (all + (SORT_TARGET_BLOCKS - NSEG_W) - 1) / (SORT_TARGET_BLOCKS - NSEG_W);
        tile_g = (tile_g + 1023u) & ~1023u;
        if (tile_g < SORT_TILE_MIN) tile_g = SORT_TILE_MIN;
        tile_base[NSEG_W + 1] = tile_g;
        uint32_t run = 0, tr = 0;
        for (uint32_t u = 0; u < NSEG_W; ++u) {
            seg_off[u] = run;
            tile_base[u] = tr;
            run += seg_total[u];
            tr += (seg_total[u] + tile_g - 1) / tile_g;
        }
        seg_off[NSEG_W] = run;
        tile_base[NSEG_W] = tr;
        // k_scatter2_g sorts a tile inside LDS only when it spans <= 2 sub-segments (S2_RANGE buckets); with < ~40 entries per bucket a tile
        // of SORT_TILE2 entries spans more and would go entry by entry through global atomics (the sets of a chunked commit, r05): smaller tiles
        const uint32_t tile2 = (small_tiles && all / (NSEG_W * NBUCKET) < 40u) ? SORT_TILE2 / 4 : SORT_TILE2;
        uint32_t t2 = 0;
        for (uint32_t u = 0; u < NSEG_W; ++u) {
            tile_base2[u] = t2;
            t2 += (seg_total[u] + tile2 - 1) / tile2;
        }
        tile_base2[NSEG_W] = t2;
        tile_base2[NSEG_W + 1] = tile2;
    }
    uint32_t *row = tile_cnt + (size_t)v * T;
    uint32_t carry = 0;
    for (uint32_t at = 0; at < T; at += blockDim.x) {
        const uint32_t i = at + threadIdx.x;
        const uint32_t c = i < T ? row[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(c, lds, &total);
        if (i < T) row[i] = base + carry + ex;
        carry += total;
    }
}

// counting sort inside the segments, from the grouped (key, payload) arrays: same LDS histogram as k_hist / k_scatter,
// one workgroup per tile of the flat tile space (tile_base)
__device__ __forceinline__ bool wide_tile(const uint32_t *__restrict__ seg_off, const uint32_t *__restrict__ tile_base,
                                          uint32_t &v, uint32_t &lo, uint32_t &hi) {
    if (blockIdx.x >= tile_base[NSEG_W]) return false;
    const uint32_t tile_g = tile_base[NSEG_W + 1];
    v = link_segment(tile_base, blockIdx.x);
    lo = seg_off[v] + (blockIdx.x - tile_base[v]) * tile_g;
    hi = seg_off[v + 1];
    if (hi > lo + tile_g) hi = lo + tile_g;
    return true;
}

__global__ void SRS_KERNEL_BOUNDS(SORT_THREADS, 1)
    k_hist_g(const uint16_t *__restrict__ gkey, const uint32_t *__restrict__ seg_off, const uint32_t *__restrict__ tile_base,
             uint32_t *__restrict__ count /* [NSEG_W][NBUCKET] */, uint32_t *__restrict__ tile_hist /* [SEG][gridDim.x] or nullptr */) {
    __shared__ uint32_t h[NBUCKET];
    uint32_t v, lo, hi;
    if (!wide_tile(seg_off, tile_base, v, lo, hi)) return;
    tile_histogram(h, gkey, lo, hi);
    uint32_t *cnt = count + (size_t)v * NBUCKET;
    for (uint32_t b = threadIdx.x; b < NBUCKET; b += blockDim.x) {
        uint32_t c = h[b];
        if (c) atomicAdd(&cnt[b], c);
    }
    if (tile_hist) {                       // entries of this tile per sub-segment (= SEG_BUCKETS consecutive buckets of its segment)
        for (uint32_t sgm = threadIdx.x; sgm < SEG; sgm += blockDim.x) {
            uint32_t c = 0;
            for (uint32_t b = 0; b < SEG_BUCKETS; ++b) c += h[sgm * SEG_BUCKETS + ((b + sgm) & (SEG_BUCKETS - 1))];
            tile_hist[(size_t)sgm * gridDim.x + blockIdx.x] = c;
        }
    }
}

// ---- the wide pipeline's counting sort in two passes through LDS (r03; the narrow pipeline's k_group / k_scatter2 on the grouped pairs of a
// segment): k_scan_seg_g turns the per-tile sub-segment counts into offsets, k_group_g sorts sub-tiles of a tile's pairs by sub-segment
// inside LDS and copies the runs out, k_scatter2_g sorts 8192-entry tiles of a segment by bucket inside LDS.
__global__ void SRS_KERNEL_BOUNDS(1024, 1)
    k_scan_seg_g(uint32_t *__restrict__ tile_hist, uint32_t TG, const uint32_t *__restrict__ tile_base, const uint32_t *__restrict__ plan,
                 size_t plan_stride) {
    // grid = (SEG, NSEG_W): the tiles of segment v are the workgroups tile_base[v] .. tile_base[v + 1] of k_hist_g
    __shared__ uint32_t lds[64];
    const uint32_t sgm = blockIdx.x, v = blockIdx.y;
    const uint32_t t0 = tile_base[v], t1 = tile_base[v + 1];
    uint32_t *row = tile_hist + (size_t)sgm * TG;
    const uint32_t base = plan[(size_t)v * plan_stride + (size_t)sgm * SEG_BUCKETS];        // absolute entry offset of the sub-segment's first bucket
    uint32_t carry = 0;
    for (uint32_t at = t0; at < t1; at += blockDim.x) {                                     // workgroup-uniform
        const uint32_t i = at + threadIdx.x;
        const uint32_t c = i < t1 ? row[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(c, lds, &total);
        if (i < t1) row[i] = base + carry + ex;
        carry += total;
    }
}

__global__ void SRS_KERNEL_BOUNDS(SORT_THREADS, 1)
    k_group_g(const uint16_t *__restrict__ gkey, const uint32_t *__restrict__ gpay, const uint32_t *__restrict__ seg_off,
              const uint32_t *__restrict__ tile_base, const uint32_t *__restrict__ tile_hist, uint16_t *__restrict__ gkey2,
              uint32_t *__restrict__ gpay2) {
    __shared__ uint32_t cur[SEG], cnt[SEG], lb[SEG], sc[64];
    __shared__ uint32_t spay[GRP_SUB];
    __shared__ uint16_t skey[GRP_SUB];
    const uint32_t tid = threadIdx.x;
    uint32_t v, lo, hi;
    if (!wide_tile(seg_off, tile_base, v, lo, hi)) return;
    for (uint32_t sgm = tid; sgm < SEG; sgm += blockDim.x) cur[sgm] = tile_hist[(size_t)sgm * gridDim.x + blockIdx.x];
    for (uint32_t sub = lo; sub < hi; sub += GRP_SUB) {                 // workgroup-uniform
        for (uint32_t sgm = tid; sgm < SEG; sgm += blockDim.x) cnt[sgm] = 0;
        __syncthreads();
        uint32_t key[GRP_PER], pay[GRP_PER], rank[GRP_PER];
#pragma unroll
        for (uint32_t k = 0; k < GRP_PER; ++k) {                        // lane-consecutive entries: coalesced 2- and 4-byte loads
            const uint32_t i = sub + k * SORT_THREADS + tid;
            key[k] = i < hi ? (uint32_t)gkey[i] : 0xFFFFu;
            pay[k] = i < hi ? gpay[i] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < GRP_PER; ++k) rank[k] = key[k] != 0xFFFFu ? atomicAdd(&cnt[key[k] / SEG_BUCKETS], 1u) : 0u;
        __syncthreads();
        uint32_t n_sub;
        const uint32_t ex = block_exclusive_scan(tid < SEG ? cnt[tid] : 0u, sc, &n_sub);
        if (tid < SEG) lb[tid] = ex;
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < GRP_PER; ++k) {
            if (key[k] != 0xFFFFu) {
                const uint32_t pos = lb[key[k] / SEG_BUCKETS] + rank[k];
                skey[pos] = (uint16_t)key[k];
                spay[pos] = pay[k];
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < n_sub; i += blockDim.x) {
            const uint32_t kk = skey[i], sgm = kk / SEG_BUCKETS, g = cur[sgm] + (i - lb[sgm]);
            gkey2[g] = (uint16_t)kk;
            gpay2[g] = spay[i];
        }
        __syncthreads();
        if (tid < SEG) cur[tid] += cnt[tid];
    }
}

__global__ void SRS_KERNEL_BOUNDS(SORT_THREADS, 2)
    k_scatter2_g(const uint16_t *__restrict__ gkey2, const uint32_t *__restrict__ gpay2, const uint32_t *__restrict__ seg_off,
                 const uint32_t *__restrict__ tile_base2, uint32_t *__restrict__ cursor /* [NSEG_W][NBUCKET], absolute */,
                 uint32_t *__restrict__ sorted) {
    __shared__ uint32_t cnt[S2_RANGE], lb[S2_RANGE], gb[S2_RANGE], sc[64];
    __shared__ uint32_t spay[SORT_TILE2];
    __shared__ uint16_t skey[SORT_TILE2];
    const uint32_t tid = threadIdx.x;
    // XCD-aware tile mapping as in k_scatter2: every XCD sorts one contiguous eighth of the flat tile space
    const uint32_t n_tiles = tile_base2[NSEG_W], per_xcd = (n_tiles + 7) / 8;
    if (blockIdx.x / 8 >= per_xcd) return;
    const uint32_t tile_id = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (tile_id >= n_tiles) return;
    const uint32_t v = link_segment(tile_base2, tile_id), tile2 = tile_base2[NSEG_W + 1];
    const uint32_t lo = seg_off[v] + (tile_id - tile_base2[v]) * tile2;
    const uint32_t hi = lo + tile2 < seg_off[v + 1] ? lo + tile2 : seg_off[v + 1];
    if (lo >= hi) return;
    uint32_t *cur = cursor + (size_t)v * NBUCKET;
    // the grouped array of a segment is ordered by sub-segment: this tile only holds buckets of sub-segments seg(first) .. seg(last)
    const uint32_t b_lo = ((uint32_t)gkey2[lo] / SEG_BUCKETS) * SEG_BUCKETS;
    const uint32_t b_hi = ((uint32_t)gkey2[hi - 1] / SEG_BUCKETS + 1) * SEG_BUCKETS;
    if (b_hi - b_lo > S2_RANGE) {            // a tile over more than two sub-segments (few entries for the bucket range): entry by entry
        for (uint32_t i = lo + tid; i < hi; i += blockDim.x) sorted[atomicAdd(&cur[gkey2[i]], 1u)] = gpay2[i];
        return;
    }
    for (uint32_t b = tid; b < S2_RANGE; b += blockDim.x) cnt[b] = 0;
    __syncthreads();
    uint32_t kk[S2_PER], pp[S2_PER], rank[S2_PER];
#pragma unroll
    for (uint32_t k = 0; k < S2_PER; ++k) {
        const uint32_t i = lo + k * SORT_THREADS + tid;
        kk[k] = i < hi ? (uint32_t)gkey2[i] : 0xFFFFu;
        pp[k] = i < hi ? gpay2[i] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < S2_PER; ++k) rank[k] = kk[k] != 0xFFFFu ? atomicAdd(&cnt[kk[k] - b_lo], 1u) : 0u;
    __syncthreads();
    uint32_t n_tile;
    const uint32_t c = tid < S2_RANGE ? cnt[tid] : 0u;
    const uint32_t ex = block_exclusive_scan(c, sc, &n_tile);
    if (tid < S2_RANGE) {
        lb[tid] = ex;
        gb[tid] = c ? atomicAdd(&cur[b_lo + tid], c) : 0u;                 // reserve [base, base + c) of the bucket (absolute)
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < S2_PER; ++k) {
        if (kk[k] != 0xFFFFu) {
            const uint32_t pos = lb[kk[k] - b_lo] + rank[k];
            skey[pos] = (uint16_t)(kk[k] - b_lo);
            spay[pos] = pp[k];
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n_tile; i += blockDim.x) {
        const uint32_t b = skey[i];
        sorted[gb[b] + (i - lb[b])] = spay[i];
    }
}

// ---------------------------------------------------------------------------------------------
// 3. plan: bucket offsets + (bucket, part) -> thread maps for every accumulation level
// grid = batch, block = PLAN_THREADS.   plan layout per MSM: (1 + nlevels) arrays of NBUCKET+1:
//   [0] off   : entry offsets of the sorted list            (off[NBUCKET] = #non-zero digits)
//   [1] tp0   : thread prefix of level 0, parts = max(1, ceil(count / L0))
//   [l] tp(l-1): thread prefix of level l-1 over the previous level's parts
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds, uint32_t *total) {
    // wave-level inclusive scan by shuffles, then one wave scans the per-wave totals (<= 16 waves)
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, nwaves = blockDim.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < nwaves ? lds[lane] : 0;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            uint32_t y = __shfl_up(w, d, 64);
            if (lane >= d) w += y;
        }
        if (lane < nwaves) lds[lane] = w;      // inclusive prefix of wave totals
    }
    __syncthreads();
    uint32_t base = wave ? lds[wave - 1] : 0;
    *total = lds[nwaves - 1];
    __syncthreads();
    return base + x - v;
}

__device__ __forceinline__ uint32_t block_max(uint32_t v, uint32_t *lds) {
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, nwaves = blockDim.x >> 6;
#pragma unroll
    for (uint32_t d = 32; d >= 1; d >>= 1) {
        uint32_t y = __shfl_xor(v, d, 64);
        v = y > v ? y : v;
    }
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    uint32_t m = 0;
    for (uint32_t w = 0; w < nwaves; ++w) m = lds[w] > m ? lds[w] : m;
    __syncthreads();
    return m;
}

// plan[level] arrays as described above; the last 4 words of the plan hold
//   [0] levels_needed : number of thread-sequential levels (>= 1) before the wavefront-level pass --
//       decided HERE from the actual heaviest bucket, so the common (balanced) case runs level 0 only
//       and later k_accum1 launches exit immediately.
__global__ void SRS_KERNEL_BOUNDS(PLAN_THREADS, 1)
    k_plan(const uint32_t *__restrict__ count, uint32_t *__restrict__ cursor, uint32_t *__restrict__ plan,
           size_t plan_stride, int nlevels, uint32_t l0_log, uint32_t l1_log,
           const uint32_t *__restrict__ seg_off /* wide windows: entry offsets are absolute, segment m starts at seg_off[m] */) {
    // grid = (batch, nlevels + 1): workgroup (m, y) scans ONE array of MSM m -- y = 0 the raw counts (entry offsets), y = l + 1 the
    // part counts of level l -- over all 2^15 buckets.  The arrays only depend on the counts (parts of level l = ceil(parts of level
    // l - 1 / 2^l1_log)), so the levels do not wait for each other (r03: one workgroup did them in turn, 45-55 us of pure latency
    // per MSM launch; now ~one round).  Every workgroup derives the number of levels actually needed the same way.
    // Thread t owns PER consecutive buckets.  r05: they stay in LDS (index i at i + i / PER: conflict-free for the owner's walk AND for the
    // coalesced loads / stores) and are walked -- sum, then running prefix written in place -- instead of being held in 32 registers: the
    // register form spilled 392 bytes per lane to scratch (31 us per launch; k_plan_s, which got this form in r04, takes 14).
    constexpr uint32_t PER = NBUCKET / PLAN_THREADS;             // 32
    __shared__ uint32_t tile[NBUCKET + NBUCKET / PER];
    __shared__ uint32_t lds[64];
    const uint32_t t = threadIdx.x;
    const uint32_t m = blockIdx.x;
    const int my_level = (int)blockIdx.y - 1;
    const uint32_t *cnt = count + (size_t)m * NBUCKET;
    uint32_t *cur = cursor + (size_t)m * NBUCKET;
    uint32_t *pl = plan + (size_t)m * plan_stride;
    {   // all 32 loads of a thread in flight at once (the counters were just written by another XCD's atomics: every load misses this L2)
        uint32_t c[PER];
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) c[j] = cnt[t + j * PLAN_THREADS];
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            const uint32_t i = t + j * PLAN_THREADS;
            tile[i + i / PER] = c[j];
        }
    }
    __syncthreads();
    const uint32_t base = t * PER + t;
    // parts of a bucket with c entries on level `level` (>= 1: an empty bucket still owns one, identity, part)
    auto value = [&](uint32_t c, int level) -> uint32_t {
        uint32_t p = (c + (1u << l0_log) - 1) >> l0_log;
        p = p ? p : 1u;
        for (int l = 0; l < level; ++l) p = (p + (1u << l1_log) - 1) >> l1_log;
        return p;
    };
    uint32_t local = 0;
#pragma unroll 8
    for (uint32_t j = 0; j < PER; ++j) local += tile[base + j];
    uint32_t total_entries;
    uint32_t run = block_exclusive_scan(local, lds, &total_entries);
    uint32_t total = total_entries;
    int needed = nlevels;
    if (my_level >= 0) {
        // (a) the heaviest bucket must be down to <= FINAL_FANIN parts for the wave-level pass;
        // (b) the TYPICAL bucket (2x the mean load) must be down to one part: the wave-level pass
        //     spends a whole wavefront per bucket, so it must find real work only in outliers.
        uint32_t mx = 0;
#pragma unroll 8
        for (uint32_t j = 0; j < PER; ++j) {
            const uint32_t v = value(tile[base + j], 0);
            mx = v > mx ? v : mx;
        }
        uint32_t p = block_max(mx, lds);
        needed = 1;
        while (p > FINAL_FANIN && needed < nlevels) {
            p = (p + (1u << l1_log) - 1) >> l1_log;
            ++needed;
        }
        const uint32_t typical = (2u * total_entries + NBUCKET - 1) / NBUCKET;
        uint32_t x = (typical + (1u << l0_log) - 1) >> l0_log;
        int by_mean = 1;
        while (x > 1 && by_mean < nlevels) {
            x = (x + (1u << l1_log) - 1) >> l1_log;
            ++by_mean;
        }
        needed = by_mean > needed ? by_mean : needed;
        if (t == 0 && my_level == 0) pl[plan_stride - 4] = (uint32_t)needed;
        if (my_level >= needed) return;                  // a level nobody runs (workgroup-uniform)
        local = 0;
#pragma unroll 8
        for (uint32_t j = 0; j < PER; ++j) local += value(tile[base + j], my_level);
        run = block_exclusive_scan(local, lds, &total);
    }
    uint32_t mx_own = 0;
#pragma unroll 8
    for (uint32_t j = 0; j < PER; ++j) {
        const uint32_t c = tile[base + j], v = my_level < 0 ? c : value(c, my_level);
        tile[base + j] = run;
        run += v;
        mx_own = v > mx_own ? v : mx_own;
    }
    __syncthreads();
    uint32_t *o = pl + (size_t)(my_level + 1) * (NBUCKET + 1);
    const uint32_t shift = (my_level < 0 && seg_off) ? seg_off[m] : 0u;
    for (uint32_t i = t; i < NBUCKET; i += PLAN_THREADS) {
        const uint32_t v = tile[i + i / PER] + shift;
        o[i] = v;
        if (my_level < 0) cur[i] = v;
    }
    if (t == PLAN_THREADS - 1) o[NBUCKET] = total + shift;
    // every bucket already a single part after the last level?  Then the wave-level pass is a pure copy:
    // k_accum_final exits and k_rowcol reads the last level's parts directly (part index == bucket index).
    if (my_level == needed - 1) {
        const uint32_t p = block_max(mx_own, lds);
        if (t == 0) pl[plan_stride - 3] = (p <= 1u) ? 1u : 0u;
    }
}

// ---- slot mode (msm.h: SLOT_LOG) ---------------------------------------------------------------------------------------------
// k_plan_s: the plan of one set in slot mode.   grid = (batch, nlevels + 2), block = PLAN_THREADS.   Arrays of NBUCKET + 1 per MSM:
//   [0]            off  : entry offsets of the sorted list (as k_plan)
//   [1]            tpo  : prefix of the OVERFLOW parts  max(0, parts(b) - (S - 1))   (level 0 of the overflow levels: these parts start
//                         from the identity and leave a partial sum in `ping`, which k_accum1 / k_ovf_final combine into slot S - 1)
//   [2 .. nlevels]      : prefix of the overflow parts after each further level (ceil(. / 2^l1_log))
//   [nlevels + 1]  tp0  : prefix of ALL level-0 parts  parts(b) = ceil(count(b) / 2^l0)  -- the dense thread space of k_accum0s
// The part length 2^l0 is chosen HERE from the mean bucket load, so that a typical bucket (mean + 6 sigma of a Poisson load) fits its
// S - 1 regular slots whatever the share of zero digits, and small sets get short parts (more threads): every workgroup derives the
// same value.  Header (last 4 words): [0] overflow levels needed, [1] 0, [2] l0, [3] overflow parts (also written to *h_ovf, which
// the host reads after the stream has been synchronised: msm::overflow_missed).
__device__ __forceinline__ uint32_t slot_l0_log(uint32_t total_entries, uint32_t S, uint32_t want_cap) {
    const uint32_t mean = total_entries / NBUCKET;
    uint32_t r = 0;
    while ((r + 1) * (r + 1) <= mean) ++r;                       // isqrt: <= 256 rounds, workgroup-uniform
    const uint32_t thresh = mean + 6 * r + 1;
    uint32_t lg = SLOT_L0_MIN_LOG;
    while (((thresh + (1u << lg) - 1) >> lg) > S - 1 && lg < SLOT_L0_MAX_LOG) ++lg;
    // not shorter than needed to fill the chip ~2.5 times over (2^19 threads), up to the 16 entries that amortise a part's load / store
    uint32_t want = 0;
    while (want < want_cap && (total_entries >> (19 + want + 1)) != 0) ++want;
    return lg > want ? lg : want;
}

__global__ void SRS_KERNEL_BOUNDS(PLAN_THREADS, 1)
    k_plan_s(const uint32_t *__restrict__ count, uint32_t *__restrict__ cursor, uint32_t *__restrict__ plan, size_t plan_stride, int nlevels,
             uint32_t S, uint32_t l1_log, const uint8_t *__restrict__ used_prev, uint8_t *__restrict__ used_next, int first,
             uint32_t *__restrict__ h_ovf, uint32_t want_cap) {
    // Thread t owns PER consecutive buckets; they stay in LDS (index i at i + i / PER: conflict-free for the owner's walk AND for the
    // coalesced loads / stores) and are walked twice -- sum, then running prefix written in place -- instead of being held in 32 registers
    // (k_plan's form spilled 100+ bytes per lane to scratch: 34 us per launch, a third of a small chunk's fixed cost).
    constexpr uint32_t PER = NBUCKET / PLAN_THREADS;             // 32
    __shared__ uint32_t tile[NBUCKET + NBUCKET / PER];
    __shared__ uint32_t lds[64];
    const uint32_t t = threadIdx.x, m = blockIdx.x, role = blockIdx.y;
    const uint32_t *cnt = count + (size_t)m * NBUCKET;
    uint32_t *pl = plan + (size_t)m * plan_stride;
    {   // all 32 loads of a thread in flight at once: issued one per loop round they cost a memory latency EACH (the counters were just
        // written by another XCD's atomics: every load misses this XCD's L2) -- that, not arithmetic, was the 30+ us of this kernel
        uint32_t c[PER];
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) c[j] = cnt[t + j * PLAN_THREADS];
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            const uint32_t i = t + j * PLAN_THREADS;
            tile[i + i / PER] = c[j];
        }
    }
    __syncthreads();
    const uint32_t base = t * PER + t;
    uint32_t local = 0;
#pragma unroll 8
    for (uint32_t j = 0; j < PER; ++j) local += tile[base + j];
    uint32_t total_entries;
    uint32_t run = block_exclusive_scan(local, lds, &total_entries);
    const uint32_t l0_log = slot_l0_log(total_entries, S, want_cap);
    uint32_t total = total_entries;
    const bool dense = role == (uint32_t)nlevels + 1;
    const uint32_t lv = role >= 2 && !dense ? role - 1 : 0;      // further levels an overflow role applies
    // the value of a bucket with c entries in this role's array
    auto value = [&](uint32_t c) -> uint32_t {
        if (role == 0) return c;
        uint32_t p = (c + (1u << l0_log) - 1) >> l0_log;         // level-0 parts
        if (dense) return p;
        p = p > S - 1 ? p - (S - 1) : 0u;                        // ... beyond the regular slots
        for (uint32_t l = 0; l < lv; ++l) p = (p + (1u << l1_log) - 1) >> l1_log;
        return p;
    };
    if (role != 0) {
        local = 0;
        uint32_t mx = 0;
#pragma unroll 4
        for (uint32_t j = 0; j < PER; ++j) {
            const uint32_t c = tile[base + j];
            local += value(c);
            if (!dense) {
                const uint32_t p = (c + (1u << l0_log) - 1) >> l0_log, o = p > S - 1 ? p - (S - 1) : 0u;
                mx = o > mx ? o : mx;
            }
        }
        if (!dense) {
            uint32_t p = block_max(mx, lds);
            int needed = 1;
            while (p > FINAL_FANIN && needed < nlevels) {
                p = (p + (1u << l1_log) - 1) >> l1_log;
                ++needed;
            }
            if (role == 1 && t == 0) {
                pl[plan_stride - 4] = (uint32_t)needed;
                pl[plan_stride - 3] = 0;
                pl[plan_stride - 2] = l0_log;
            }
            if ((int)role - 1 >= needed) return;                 // a level nobody runs (workgroup-uniform)
        }
        run = block_exclusive_scan(local, lds, &total);
    }
    // slots in use after this set: the thread's 32 bytes travel as two 16-byte words (byte loads in the loop cost a memory latency EACH:
    // that was 25 of this kernel's 33 us)
    static_assert(PER == 32, "k_plan_s: used[] travels as 2 x uint4 per thread");
    uint32_t was_w[8] = {0, 0, 0, 0, 0, 0, 0, 0}, now_w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (dense && !first) {
        const uint4 *up = reinterpret_cast<const uint4 *>(used_prev + (size_t)m * NBUCKET + (size_t)t * PER);
        const uint4 a = up[0], b = up[1];
        was_w[0] = a.x; was_w[1] = a.y; was_w[2] = a.z; was_w[3] = a.w; was_w[4] = b.x; was_w[5] = b.y; was_w[6] = b.z; was_w[7] = b.w;
    }
#pragma unroll
    for (uint32_t j = 0; j < PER; ++j) {
        const uint32_t c = tile[base + j], v = value(c);
        tile[base + j] = run;
        run += v;
        if (dense) {
            const uint32_t now = v < S - 1 ? v : S - 1, was = (was_w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            now_w[j >> 2] |= (now > was ? now : was) << (8 * (j & 3));
        }
    }
    if (dense) {
        uint4 *un = reinterpret_cast<uint4 *>(used_next + (size_t)m * NBUCKET + (size_t)t * PER);
        uint4 lo, hi;
        lo.x = now_w[0]; lo.y = now_w[1]; lo.z = now_w[2]; lo.w = now_w[3];
        hi.x = now_w[4]; hi.y = now_w[5]; hi.z = now_w[6]; hi.w = now_w[7];
        un[0] = lo;
        un[1] = hi;
    }
    __syncthreads();
    uint32_t *o = pl + (size_t)role * (NBUCKET + 1);
    uint32_t *cur = cursor + (size_t)m * NBUCKET;
    for (uint32_t i = t; i < NBUCKET; i += PLAN_THREADS) {
        const uint32_t v = tile[i + i / PER];
        o[i] = v;
        if (role == 0) cur[i] = v;
    }
    if (t == PLAN_THREADS - 1) {
        o[NBUCKET] = total;
        if (role == 1) {
            pl[plan_stride - 1] = total;
            h_ovf[m] = total;
        }
        if (role == 0) h_ovf[(size_t)LANDING_SLOTS * BATCH_ARGS + m] = total;      // the set's non-zero digits (msm::note_commit: the witness's density)
    }
}

__global__ void k_link(const uint32_t *__restrict__ plan, size_t plan_stride, int nlevels, Link *__restrict__ link) {
    const uint32_t t = threadIdx.x;
    for (int l = 0; l < nlevels; ++l) {
        uint32_t c = 0;
        if (t < NSEG_W) {
            const uint32_t *pl = plan + (size_t)t * plan_stride;
            if ((uint32_t)l < pl[plan_stride - 4]) c = pl[(size_t)(l + 1) * (NBUCKET + 1) + NBUCKET];
        }
        uint32_t x = c;
#pragma unroll
        for (uint32_t d = 1; d < NSEG_W; d <<= 1) {
            uint32_t y = __shfl_up(x, d, 64);
            if (t >= d) x += y;
        }
        if (t < NSEG_W) link->base[l][t] = x - c;
        if (t == NSEG_W - 1) link->base[l][NSEG_W] = x;
    }
}

// ---------------------------------------------------------------------------------------------
// 4. bucket accumulation
// level 0: thread t = (bucket, part): <= L0 gathered mixed adds  ->  parts0[t]  (XYZZ)
// grid = (blocks, batch)
// ---------------------------------------------------------------------------------------------
// thread -> bucket map of level 0: tb[t] = b for tp0[b] <= t < tp0[b + 1].  k_accum0 used to find its bucket by a binary search
// over the 2^15 prefix entries -- 15 DEPENDENT global loads (~15 us) in front of ~130 us of additions, in every thread.
// One wavefront per bucket writes the (contiguous) range instead; the map is 2 bytes per thread.   grid = (NBUCKET / 4, batch)
__global__ void SRS_KERNEL_BOUNDS(256, 1)
    k_expand(const uint32_t *__restrict__ plan, size_t plan_stride, uint16_t *__restrict__ tb, size_t tb_stride,
             const Link *__restrict__ link, uint32_t arr /* plan array holding the thread prefix: 1, or nlevels + 1 in slot mode */) {
    const uint32_t m = blockIdx.y, lane = threadIdx.x & 63u;
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t *tp = plan + (size_t)m * plan_stride + (size_t)arr * (NBUCKET + 1);
    uint16_t *out = tb + (link ? (size_t)link->base[0][m] : (size_t)m * tb_stride);
    const uint32_t s = tp[b], e = tp[b + 1];
    for (uint32_t t = s + lane; t < e; t += 64) out[t] = (uint16_t)b;
}

// The chain of gathered mixed additions of one level-0 part: init (nullptr: the identity) + sum of the entries src[s .. e), s < e.
// Behind every addition travel the NEXT gathered point (its index arrived an addition ago) and the index after it (r04: with the index
// loaded in the same round as the gather it addresses, every round parked the wave for a memory latency).
// r06: the loop runs the addition WITHOUT its exceptional cases (Ec29::madd_signed_fast: no identity tests, no branch, no doubling
// path in the loop body); a chain that met one -- an identity entry of a degenerate key, a partial sum that cancelled, an entry equal to
// the running sum -- is detected by a sticky flag and recomputed from its start with the complete formulas.
template <class C>
__device__ __forceinline__ xyzz29_t accumulate_part(const uint32_t *__restrict__ src, uint32_t s, uint32_t e, const affine_t *__restrict__ table,
                                                    const xyzz_t *__restrict__ init) {
    using E29 = Ec29<C>;
    using F = typename E29::F;
    bool exc = false;
    xyzz29_t acc;
    {
        uint32_t v = src[s];
        uint32_t vn = s + 1 < e ? src[s + 1] : 0u;
        affine_t p = table[v & 0x7FFFFFFFu];
        uint32_t j = s;
        if (init) {
            acc = E29::unpack(*init);
        } else {                              // a fresh part: the first entry IS the sum so far
            affine_t pn = p;
            uint32_t vnn = 0;
            if (s + 1 < e) {
                pn = table[vn & 0x7FFFFFFFu];
                if (s + 2 < e) vnn = src[s + 2];
            }
            const aff29_t q = E29::load_raw(p);
            acc.x = q.x;
            acc.y = (v >> 31) ? F::normalize(F::template neg_lazy<1, 0>(q.y)) : q.y;      // P - y  (y != 0 unless Q = O: flagged)
            acc.zz = E29::one();
            acc.zzz = acc.zz;
            exc = F::is_zero_exact(q.y);
            v = vn;
            vn = vnn;
            p = pn;
            j = s + 1;
        }
        for (; j < e; ++j) {
            uint32_t vnn = 0;
            affine_t pn = p;
            if (j + 1 < e) {
                pn = table[vn & 0x7FFFFFFFu];
                if (j + 2 < e) vnn = src[j + 2];
            }
            acc = E29::madd_signed_fast(acc, E29::load_raw(p), (v >> 31) != 0, exc);
            v = vn;
            vn = vnn;
            p = pn;
        }
    }
    if (exc) {                                // rare: once more, complete
        acc = init ? E29::unpack(*init) : E29::identity();
#pragma unroll 1
        for (uint32_t j = s; j < e; ++j) {
            const uint32_t v = src[j];
            acc = E29::madd_signed(acc, E29::load_raw(table[v & 0x7FFFFFFFu]), (v >> 31) != 0);
        }
    }
    return acc;
}

template <class C>
__global__ void SRS_KERNEL_BOUNDS(ACC_THREADS, 1)
    k_accum0(const uint32_t *__restrict__ sorted, size_t sorted_stride, const uint32_t *__restrict__ plan,
             size_t plan_stride, const uint16_t *__restrict__ tb, size_t tb_stride, const affine_t *__restrict__ table,
             xyzz_t *__restrict__ parts, size_t parts_stride, uint32_t l0, const Link *__restrict__ link) {
    uint32_t m = blockIdx.y;
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    size_t slot;                                   // index of this thread's part (and map entry)
    if (link) {                                    // wide windows: flat thread space over the segments, offsets are absolute
        if (t >= link->base[0][NSEG_W]) return;
        m = link_segment(link->base[0], t);
        slot = t;
        t -= link->base[0][m];
    } else {
        slot = (size_t)m * parts_stride + t;
    }
    const uint32_t *off = plan + (size_t)m * plan_stride;
    const uint32_t *tp = off + (NBUCKET + 1);
    if (t >= tp[NBUCKET]) return;
    uint32_t b = tb[link ? slot : (size_t)m * tb_stride + t];
    uint32_t part = t - tp[b];
    uint32_t s = off[b] + part * l0;
    uint32_t e = off[b + 1];
    if (e > s + l0) e = s + l0;
    const uint32_t *src = link ? sorted : sorted + (size_t)m * sorted_stride;
    // the additions run on the 9 x 29-bit limb form (curve29.cuh); the table is stored in its Montgomery form
    using E29 = Ec29<C>;
    const xyzz29_t acc = s < e ? accumulate_part<C>(src, s, e, table, nullptr) : E29::identity();
    parts[slot] = E29::pack(acc);                                // canonical R'-form: the later levels stay on the 29-bit multiplier
}

// 64-lane exchange of an XYZZ point
__device__ __forceinline__ xyzz_t shfl_down_point(const xyzz_t &p, unsigned delta) {
    xyzz_t o;
    shfl_down_words<32>(reinterpret_cast<const uint32_t *>(&p), reinterpret_cast<uint32_t *>(&o), delta, 64);
    return o;
}

// 64-lane exchange of a point in 9-limb coordinates (36 words), inside groups of `width` lanes
__device__ __forceinline__ xyzz29_t shfl_down_point29(const xyzz29_t &p, unsigned delta, int width) {
    xyzz29_t o;
    shfl_down_words<36>(reinterpret_cast<const uint32_t *>(&p), reinterpret_cast<uint32_t *>(&o), delta, width);
    return o;
}

// level >= 1: (bucket, part) over the previous level's parts, <= L1 full adds each.
// Two modes, chosen on the device from the number of outputs of this level:
//   many outputs  -> one lane per output (throughput-bound, every lane runs its own additions)
//   few outputs   -> one QUAD per output (latency-bound tail: Ec::add_quad cuts the dependent chain 3.3x)
// The grid is sized for the quad mode; in lane mode the surplus workgroups exit.
template <class C>
__global__ void SRS_KERNEL_BOUNDS(ACC_THREADS, 1)
    k_accum1(const xyzz_t *__restrict__ in, size_t in_stride, const uint32_t *__restrict__ plan,
             size_t plan_stride, int level, xyzz_t *__restrict__ out, size_t out_stride, uint32_t l1,
             const Link *__restrict__ link, uint32_t quad_max) {
    uint32_t m = blockIdx.y;
    const uint32_t lin = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t t;
    bool quad;
    size_t in_off, out_off;
    if (link) {                                // wide windows: flat thread space of this level over the segments
        const uint32_t n_all = link->base[level][NSEG_W];
        quad = n_all <= quad_max;
        t = quad ? (lin >> 2) : lin;
        if (t >= n_all) return;
        m = link_segment(link->base[level], t);
        t -= link->base[level][m];
        in_off = link->base[(level - 1) & 1][m];
        out_off = link->base[level & 1][m];
    } else {
        in_off = (size_t)m * in_stride;
        out_off = (size_t)m * out_stride;
    }
    if ((uint32_t)level >= plan[(size_t)m * plan_stride + plan_stride - 4]) return;   // level not needed (k_plan)
    const uint32_t *tp_prev = plan + (size_t)m * plan_stride + (size_t)level * (NBUCKET + 1);
    const uint32_t *tp = tp_prev + (NBUCKET + 1);
    const uint32_t n_out = tp[NBUCKET];
    if (!link) {
        quad = (uint64_t)n_out * gridDim.y <= quad_max;   // whole batch: latency-bound only while the chip is not full
        t = quad ? (lin >> 2) : lin;
    }
    if (t >= n_out) return;                    // in quad mode the 4 lanes of a quad leave together
    uint32_t b = upper_bucket(tp, t);
    uint32_t part = t - tp[b];
    uint32_t s = tp_prev[b] + part * l1;
    uint32_t e = tp_prev[b + 1];
    if (e > s + l1) e = s + l1;
    using E29 = Ec29<C>;
    const xyzz_t *src = in + in_off;
    xyzz29_t acc = E29::unpack(src[s]);
    if (quad) {
        const uint32_t q = threadIdx.x & 3u;
        for (uint32_t j = s + 1; j < e; ++j) acc = E29::add_quad(acc, E29::unpack(src[j]), q);
        if (q == 0) out[out_off + t] = E29::pack(acc);
    } else {
        for (uint32_t j = s + 1; j < e; ++j) acc = E29::add(acc, E29::unpack(src[j]));
        out[out_off + t] = E29::pack(acc);
    }
}

// final level: one wavefront per bucket; lanes stride over the remaining parts, then a 6-step
// shuffle tree; lane 0 stores the bucket.   grid = (NBUCKET / waves_per_block, batch)
template <class C>
__global__ void SRS_KERNEL_BOUNDS(FINAL_THREADS, 1)
    k_accum_final(const xyzz_t *__restrict__ ping, size_t ping_stride, const xyzz_t *__restrict__ pong,
                  size_t pong_stride, const uint32_t *__restrict__ plan, size_t plan_stride,
                  xyzz_t *__restrict__ buckets, const Link *__restrict__ link) {
    uint32_t m = blockIdx.y;
    if (plan[(size_t)m * plan_stride + plan_stride - 3]) return;              // all buckets single: nothing to combine
    const uint32_t level = plan[(size_t)m * plan_stride + plan_stride - 4];   // levels actually run
    const xyzz_t *in = (level & 1u) ? ping : pong;     // level 0 -> ping, level 1 -> pong, ...
    const size_t in_stride = (level & 1u) ? ping_stride : pong_stride;
    const uint32_t *tp = plan + (size_t)m * plan_stride + (size_t)level * (NBUCKET + 1);
    uint32_t lane = threadIdx.x & 63u;
    uint32_t b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t s = tp[b], e = tp[b + 1];
    const xyzz_t *src = in + (link ? (size_t)link->base[(level - 1u) & 1u][m] : (size_t)m * in_stride);
    xyzz_t *dst = buckets + (size_t)m * NBUCKET;
    if (e - s == 1) {                       // common case: the bucket is already one part
        if (lane == 0) dst[b] = src[s];
        return;
    }
    using E29 = Ec29<C>;
    xyzz29_t acc = E29::identity();
    for (uint32_t j = s + lane; j < e; j += 64) acc = E29::add(acc, E29::unpack(src[j]));
    for (unsigned d = 32; d >= 1; d >>= 1) {
        xyzz29_t other = shfl_down_point29(acc, d, 64);
        if (lane < d) acc = E29::add(acc, other);
    }
    if (lane == 0) dst[b] = E29::pack(acc);
}

// ---- slot mode: bucket accumulation INTO the persistent slots ------------------------------------------------------------------
// thread t = (bucket b, part p) of the dense thread space tp0 (plan array `arr`): <= 2^l0 gathered mixed additions, like k_accum0, but
//   p <  S - 1 : the running sum of slot (b, p) is the start value (identity when the slot holds nothing yet) and the result goes back
//                -- nothing is left for later levels to combine, whatever the number of sets a commit is cut into;
//   p >= S - 1 : a hot bucket (more entries than S - 1 parts hold): a fresh partial sum into `ovf`, combined by k_accum1 / k_ovf_final.
// grid = (ceil(cap / ACC_THREADS), batch)
template <class C>
__global__ void SRS_KERNEL_BOUNDS(ACC_THREADS, 1)
    k_accum0s(const uint32_t *__restrict__ sorted, size_t sorted_stride, const uint32_t *__restrict__ plan, size_t plan_stride, uint32_t arr,
              const uint16_t *__restrict__ tb, size_t tb_stride, const affine_t *__restrict__ table, xyzz_t *__restrict__ slots, uint32_t S,
              const uint8_t *__restrict__ used_prev, int first, xyzz_t *__restrict__ ovf, size_t ovf_stride) {
    const uint32_t m = blockIdx.y;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t *off = plan + (size_t)m * plan_stride;
    const uint32_t *tp = off + (size_t)arr * (NBUCKET + 1);
    if (t >= tp[NBUCKET]) return;
    const uint32_t *tpo = off + (NBUCKET + 1);
    const uint32_t l0 = 1u << off[plan_stride - 2];
    const uint32_t b = tb[(size_t)m * tb_stride + t];
    const uint32_t part = t - tp[b];
    uint32_t s = off[b] + part * l0;
    uint32_t e = off[b + 1];
    if (e > s + l0) e = s + l0;
    const uint32_t *src = sorted + (size_t)m * sorted_stride;
    using E29 = Ec29<C>;
    const bool regular = part < S - 1;
    xyzz_t *slot = regular ? slots + ((size_t)m * NBUCKET + b) * S + part : ovf + (size_t)m * ovf_stride + tpo[b] + (part - (S - 1));
    // a part is never empty (s < e); the slot's running sum is the start value when the slot holds one
    const bool resume = regular && !first && part < (uint32_t)used_prev[(size_t)m * NBUCKET + b];
    *slot = E29::pack(accumulate_part<C>(src, s, e, table, resume ? slot : nullptr));
}

// the wave-level pass of the overflow parts: one wavefront per bucket sums what the overflow levels left of it and adds the sum into the
// bucket's LAST slot (first set of a commit: every bucket's last slot is initialised here).   grid = (NBUCKET / waves_per_block, batch)
template <class C>
__global__ void SRS_KERNEL_BOUNDS(FINAL_THREADS, 1)
    k_ovf_final(const xyzz_t *__restrict__ ping, size_t ping_stride, const xyzz_t *__restrict__ pong, size_t pong_stride,
                const uint32_t *__restrict__ plan, size_t plan_stride, xyzz_t *__restrict__ slots, uint32_t S, int first) {
    const uint32_t m = blockIdx.y;
    const uint32_t level = plan[(size_t)m * plan_stride + plan_stride - 4];   // overflow levels actually run
    const xyzz_t *in = (level & 1u) ? ping + (size_t)m * ping_stride : pong + (size_t)m * pong_stride;     // level 0 -> ping, level 1 -> pong, ...
    const uint32_t *tp = plan + (size_t)m * plan_stride + (size_t)level * (NBUCKET + 1);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t s = tp[b], e = tp[b + 1];
    xyzz_t *dst = slots + ((size_t)m * NBUCKET + b) * S + (S - 1);
    using E29 = Ec29<C>;
    if (e == s) {                            // the usual bucket: nothing beyond its regular slots
        if (first && lane == 0) *dst = E29::pack(E29::identity());
        return;
    }
    xyzz29_t acc = E29::identity();
    for (uint32_t j = s + lane; j < e; j += 64) acc = E29::add(acc, E29::unpack(in[j]));
    for (unsigned d = 32; d >= 1; d >>= 1) {
        xyzz29_t other = shfl_down_point29(acc, d, 64);
        if (lane < d) acc = E29::add(acc, other);
    }
    if (lane == 0) {
        if (!first) acc = E29::add(acc, E29::unpack(*dst));
        *dst = E29::pack(acc);
    }
}

// the reduction of the slots at the end of a commit, <= 8 inputs per output: out[B * out_pb + g] = sum_j in[B * in_pb + g + j * out_pb] over
// the inputs that hold a sum -- index < used[B] (first level; nullptr: all of them) or == extra (the overflow slot, when its kernels ran).
// B = m * NBUCKET + b.  The inputs of an output are STRIDED (a bucket with u slots in use gives every one of its outputs u / out_pb of
// them, whatever u) and the thread space is g-major (a wavefront = 64 buckets at the same g: the wavefronts of a level carry equal
// chains, and a small MSM with one or two slots per bucket in use runs a few full wavefronts instead of one lane in eight of all of them).
// QUAD: one quad per output (the last level: 2^15 outputs per MSM).   grid = (ceil(n_out [* 4] / 256), batch)
template <class C, bool QUAD>
__global__ void SRS_KERNEL_BOUNDS(256, 1)
    k_slot_reduce(const xyzz_t *__restrict__ in, uint32_t in_pb, const uint8_t *__restrict__ used, size_t used_stride, int extra,
                  xyzz_t *__restrict__ out, uint32_t out_pb) {
    using E29 = Ec29<C>;
    const uint32_t m = blockIdx.y;
    const uint32_t lin = blockIdx.x * blockDim.x + threadIdx.x, t = QUAD ? (lin >> 2) : lin, q = lin & 3u;
    if (t >= NBUCKET * out_pb) return;
    const uint32_t g = t / NBUCKET, b = t % NBUCKET;
    const uint32_t lim = used ? (uint32_t)used[(size_t)m * used_stride + b] : in_pb;
    const xyzz_t *src = in + ((size_t)m * NBUCKET + b) * in_pb;
    xyzz29_t acc = E29::identity();
    for (uint32_t idx = g; idx < in_pb; idx += out_pb) {
        if (!(idx < lim || (int)idx == extra)) continue;                          // uniform inside a quad
        const xyzz29_t x = E29::unpack(src[idx]);
        if (QUAD) acc = E29::add_quad(acc, x, q); else acc = E29::add(acc, x);
    }
    if (!QUAD || q == 0) out[((size_t)m * NBUCKET + b) * out_pb + g] = E29::pack(acc);
}

// chunked commits: the finished buckets of this set (where k_rowcol would read them) go into the key's running buckets;
// one thread per bucket.   grid = NBUCKET / 64
template <class C>
__global__ void SRS_KERNEL_BOUNDS(64, 1)
    k_bucket_fold(const xyzz_t *__restrict__ buckets, const xyzz_t *__restrict__ ping, const xyzz_t *__restrict__ pong,
                  const uint32_t *__restrict__ plan, size_t plan_stride, xyzz_t *__restrict__ total, int first) {
    using E29 = Ec29<C>;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t *hdr = plan + plan_stride - 4;                              // [0] levels run, [1] all buckets single
    const xyzz_t *B = hdr[1] ? ((hdr[0] & 1u) ? ping : pong) : buckets;
    if (first) {
        total[b] = B[b];
    } else {
        total[b] = E29::pack(E29::add(E29::unpack(total[b]), E29::unpack(B[b])));
    }
}

// ---------------------------------------------------------------------------------------------
// 5. bucket reduction  S = sum_b (b+1) B_b,  b = hi * RED_COLS + lo
//    S = RED_COLS * sum_hi hi * R_hi  +  sum_lo (lo+1) * C_lo
// k_rowcol: block < RED_ROWS -> row sum R_hi ; else column sum C_lo.   grid = (ROWS + COLS, batch)
// ---------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void lds_tree_sum(xyzz_t *v, uint32_t n_pow2) {
    for (uint32_t s = n_pow2 >> 1; s >= 1; s >>= 1) {
        if (threadIdx.x < s) v[threadIdx.x] = Ec<C>::add(v[threadIdx.x], v[threadIdx.x + s]);
        __syncthreads();
    }
}

// 64-lane exchange of an XYZZ point inside groups of `width` lanes
__device__ __forceinline__ xyzz_t shfl_down_point_w(const xyzz_t &p, unsigned delta, int width) {
    xyzz_t o;
    shfl_down_words<32>(reinterpret_cast<const uint32_t *>(&p), reinterpret_cast<uint32_t *>(&o), delta, width);
    return o;
}

// Row / column sums, one QUAD per lane-task (Ec::add_quad): a quad first adds RC_SER consecutive elements, then the
// 16 quads of a wavefront combine by a shuffle tree (partner quad = 4 d lanes further).  A row (128 elements) is one
// wavefront; a column (256 elements) is the two wavefronts of a workgroup, joined through LDS.
// grid = (RED_ROWS / 2 + RED_COLS, batch), block = 128: workgroups < RED_ROWS / 2 do two rows, the others one column.
template <class C>
__global__ void SRS_KERNEL_BOUNDS(128, 1)
    k_rowcol(const xyzz_t *__restrict__ buckets, const xyzz_t *__restrict__ ping, size_t ping_stride,
             const xyzz_t *__restrict__ pong, size_t pong_stride, const uint32_t *__restrict__ plan, size_t plan_stride,
             xyzz_t *__restrict__ rc /* [batch][ROWS + COLS] */, const Link *__restrict__ link, int from_buckets) {
    constexpr uint32_t SER = 8;
    __shared__ xyzz_t half[1];
    const uint32_t m = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t q = lane & 3u, vl = lane >> 2;                                // quad role, quad index 0..15
    const uint32_t *hdr = plan + (size_t)m * plan_stride + plan_stride - 4;   // [0] levels run, [1] all buckets single
    const size_t ping_off = link ? (size_t)link->base[0][m] : (size_t)m * ping_stride;
    const size_t pong_off = link ? (size_t)link->base[1][m] : (size_t)m * pong_stride;
    const xyzz_t *B = (hdr[1] && !from_buckets) ? ((hdr[0] & 1u) ? ping + ping_off : pong + pong_off) : buckets + (size_t)m * NBUCKET;
    using E29 = Ec29<C>;
    xyzz_t *out = rc + (size_t)m * (RED_ROWS + RED_COLS);
    const bool is_row = blockIdx.x < RED_ROWS / 2;
    xyzz29_t acc;
    if (is_row) {
        const uint32_t hi = blockIdx.x * 2 + wave;                               // 16 quads x 8 = 128 = RED_COLS
        const xyzz_t *src = B + (size_t)hi * RED_COLS + vl * SER;
        acc = E29::unpack(src[0]);
        xyzz_t nx = src[1];
        for (uint32_t j = 1; j < SER; ++j) {             // next element is fetched behind the addition
            xyzz_t cur = nx;
            if (j + 1 < SER) nx = src[j + 1];
            acc = E29::add_quad(acc, E29::unpack(cur), q);
        }
    } else {
        const uint32_t lo = blockIdx.x - RED_ROWS / 2, sub = wave * 16 + vl;     // 32 quads x 8 = 256 = RED_ROWS
        const xyzz_t *src = B + (size_t)(sub * SER) * RED_COLS + lo;
        acc = E29::unpack(src[0]);
        xyzz_t nx = src[RED_COLS];
        for (uint32_t j = 1; j < SER; ++j) {
            xyzz_t cur = nx;
            if (j + 1 < SER) nx = src[(size_t)(j + 1) * RED_COLS];
            acc = E29::add_quad(acc, E29::unpack(cur), q);
        }
    }
    for (unsigned d = 8; d >= 1; d >>= 1) {
        xyzz29_t o = shfl_down_point29(acc, 4 * d, 64);
        if (vl < d) acc = E29::add_quad(acc, o, q);
    }
    if (is_row) {
        if (lane == 0) out[blockIdx.x * 2 + wave] = E29::pack(acc);
        return;
    }
    if (wave == 1 && lane == 0) half[0] = E29::pack(acc);
    __syncthreads();
    if (wave == 0 && vl == 0) {
        acc = E29::add_quad(acc, E29::unpack(half[0]), q);
        if (q == 0) out[RED_ROWS + (blockIdx.x - RED_ROWS / 2)] = E29::pack(acc);
    }
}

// sum_j j * X_j = sum_{j>=1} Suffix_j with Suffix_j = sum_{i>=j} X_i: an inclusive suffix scan and a tree sum, both
// log-depth in LDS, over 128 elements with one QUAD per element (512 threads).  grid = (3, batch):
//   block 0: A' = sum_a a * (R_2a + R_2a+1)     (the 256 row sums taken in pairs; drop Suffix_0)
//   block 1: B  = sum_lo (lo+1) * C_lo          (keep Suffix_0)
//   block 2: Z  = sum_a R_2a+1                  (plain tree sum)
//   block 3: T  = sum_lo C_lo                   (plain tree sum; wide windows only: the segment's unweighted total)
// since sum_hi hi R_hi = 2 A' + Z, the host finishes  S = RED_COLS * (2 A' + Z) + B.   out[gridDim.x * m + which] = A', B, Z (, T)
template <class C>
__global__ void SRS_KERNEL_BOUNDS(RED_THREADS, 1)
    k_reduce_final(const xyzz_t *__restrict__ rc, xyzz_t *__restrict__ out) {
    constexpr uint32_t N = RED_COLS;                       // 128 elements per block
    using E29 = Ec29<C>;
    __shared__ xyzz29_t v[N];                              // lazy 9-limb coordinates: packing (two products) only for the final store
    const uint32_t m = blockIdx.y, t = threadIdx.x >> 2, q = threadIdx.x & 3u, which = blockIdx.x;
    const xyzz_t *rows = rc + (size_t)m * (RED_ROWS + RED_COLS), *cols = rows + RED_ROWS;
    xyzz29_t x;
    if (which == 0) x = E29::add_quad(E29::unpack(rows[2 * t]), E29::unpack(rows[2 * t + 1]), q);
    else if (which == 1 || which == 3) x = E29::unpack(cols[t]);
    else x = E29::unpack(rows[2 * t + 1]);
    if (which < 2) {
        if (q == 0) v[t] = x;
        __syncthreads();
        for (uint32_t s = 1; s < N; s <<= 1) {             // suffix scan
            const bool has = t + s < N;
            xyzz29_t o = has ? v[t + s] : E29::identity();
            __syncthreads();
            if (has) x = E29::add_quad(x, o, q);
            if (q == 0) v[t] = x;
            __syncthreads();
        }
        if (which == 0 && t == 0) x = E29::identity();
    }
    if (q == 0) v[t] = x;
    __syncthreads();
    for (uint32_t s = N >> 1; s >= 1; s >>= 1) {           // tree sum
        if (t < s) {                                       // x is kept in registers by all 4 lanes: the quad's own
            x = E29::add_quad(x, v[t + s], q);             // slot is never re-read, so lane 0's store cannot race with it
            if (q == 0) v[t] = x;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)m * gridDim.x + which] = E29::pack(v[0]);
}

// wide windows: the 4 x NSEG_W partial sums of the segments -> 4 points.  A', B, Z add up (the bucket reduction is linear);
// bucket (v, lo) has weight v * NBUCKET + lo + 1, so the segment totals enter as U = sum_v v T_v (suffix sums, as above),
// which the host multiplies by NBUCKET.   grid = 1, block = 256: wavefront `which`, quad v
template <class C>
__global__ void SRS_KERNEL_BOUNDS(256, 1)
    k_wide_combine(const xyzz_t *__restrict__ in /* [NSEG_W][4] */, xyzz_t *__restrict__ out /* [4] */) {
    using E29 = Ec29<C>;
    const uint32_t which = threadIdx.x >> 6, lane = threadIdx.x & 63u, q = lane & 3u, vl = lane >> 2;
    xyzz29_t x = E29::unpack(in[vl * 4 + which]);
    if (which == 3) {
        for (uint32_t s = 1; s < NSEG_W; s <<= 1) {        // suffix scan over the 16 quads
            xyzz29_t o = shfl_down_point29(x, 4 * s, 64);
            if (vl + s < NSEG_W) x = E29::add_quad(x, o, q);
        }
        if (vl == 0) x = E29::identity();
    }
    for (unsigned d = NSEG_W / 2; d >= 1; d >>= 1) {
        xyzz29_t o = shfl_down_point29(x, 4 * d, 64);
        if (vl < d) x = E29::add_quad(x, o, q);
    }
    if (lane == 0) out[which] = E29::pack(x);
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
static inline uint32_t ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// windows 1 .. nwin-1 of a table whose window 0 (the bases, ABI form) is in place; `nbits` doublings between windows
template <class C>
static void expand_windows(affine_t *table, uint32_t n, int nwin, int nbits, xyzz_t *tmp, hipStream_t stream) {
    for (int w = 1; w < nwin; ++w) {
        const affine_t *prev = table + (size_t)(w - 1) * n;
        affine_t *cur = table + (size_t)w * n;
        SRS_LAUNCH((k_table_step<C>), (ceil_div(n, 256)), (256), 0, stream, prev, tmp, n, nbits);
        SRS_LAUNCH((k_normalize<C>), (ceil_div(ceil_div(n, NORM_G), 128)), (128), 0, stream, (const xyzz_t *)tmp, cur, n);
    }
}

// Which keys carry the second, 13-window table.  r03 ended with the wide pipeline opt-in: ahead at >= 12 M scalars (2^24 uniform 20.5 vs
// 22.5 ms) but with ONE full bench in which the row-program kernels ran slower on that box.  r04 repeated the comparison (three full
// benches each way on one box, profiles/r04_ab_wide_vs_narrow.txt): the anomaly did not come back -- the k = 20 step is the same with
// and without the second table (11.52 / 11.52 vs 11.55 / 11.57 ms), the 2^24 MSM is 19.9-20.0 vs 22.3 ms uniform and 10.30 vs 10.71 ms
// trace-like.  So: keys of >= 2^WIDE_MIN_KEY_LOG bases get the table (+81 % key memory: 29 GiB instead of 16 at 2^24) and WHOLE,
// device-resident MSMs of >= 2^WIDE_MIN_N_LOG scalars take the 20-bit windows; the sets of a streamed commit stay on the 16-bit windows
// and the slots (their chunks are 0.3-3 M scalars, where the wide pipeline's fixed costs lose).  SRS_MSM_WIDE=0 / 1: never / every key.
static bool wants_wide_table(size_t len) {
    // tuning msm_wide; unset: the environment's SRS_MSM_WIDE (a deployer's memory switch: 0 = no second table), else by key size
    static const int env_forced = [] { const char *e = std::getenv("SRS_MSM_WIDE"); return e ? std::atoi(e) : -1; }();
    const int forced = (int)tuning::get_or(tuning::MSM_WIDE, env_forced);
    if (forced == 0 || len == 0) return false;
    return forced == 1 || len >= ((size_t)1 << WIDE_MIN_KEY_LOG);
}

template <class C>
static void build_table_t(Key &k, hipStream_t stream) {
    const uint32_t n = (uint32_t)k.len;
    if (k.table_w) {
        SRS_HIP_CHECK(hipFree(k.table_w));
        k.table_w = nullptr;
    }
    if (n == 0) return;
    xyzz_t *tmp = nullptr;
    SRS_HIP_CHECK(hipMalloc((void **)&tmp, (size_t)n * sizeof(xyzz_t)));
    // the second table is an optimisation (+81 % key memory): a device that cannot hold it keeps the 16-bit windows for everything
    // (use_wide() is false without table_w; srs_ck_has_wide_table tells)
    if (wants_wide_table(n) && hipMalloc((void **)&k.table_w, (size_t)n * NWIN_W * sizeof(affine_t)) != hipSuccess) {
        (void)hipGetLastError();
        k.table_w = nullptr;
    }
    if (k.table_w) {
        SRS_HIP_CHECK(hipMemcpyAsync(k.table_w, k.table, (size_t)n * sizeof(affine_t), hipMemcpyDeviceToDevice, stream));
        expand_windows<C>(k.table_w, n, NWIN_W, WBITS_W, tmp, stream);
        const size_t total_w = (size_t)n * NWIN_W;
        SRS_LAUNCH((k_table_form<C>), (ceil_div(total_w, 256)), (256), 0, stream, k.table_w, total_w);
    }
    expand_windows<C>(k.table, n, NWIN, WBITS, tmp, stream);
    const size_t total = (size_t)n * NWIN;
    SRS_LAUNCH((k_table_form<C>), (ceil_div(total, 256)), (256), 0, stream, k.table, total);
    SRS_HIP_CHECK(hipStreamSynchronize(stream));
    SRS_HIP_CHECK(hipFree(tmp));
}

template <class C>
static void generate_t(Key &k, uint64_t seed, hipStream_t stream) {
    const uint32_t n = (uint32_t)k.len;
    if (n == 0) return;
    xyzz_t *tmp = nullptr;
    SRS_HIP_CHECK(hipMalloc((void **)&tmp, (size_t)n * sizeof(xyzz_t)));
    SRS_LAUNCH((k_gen_bases<C>), (ceil_div(n, 128)), (128), 0, stream, tmp, n, seed, k.rank, k.world);
    SRS_LAUNCH((k_normalize<C>), (ceil_div(ceil_div(n, NORM_G), 128)), (128), 0, stream, (const xyzz_t *)tmp, k.table, n);
    SRS_HIP_CHECK(hipStreamSynchronize(stream));
    SRS_HIP_CHECK(hipFree(tmp));
}
void generate_bases(Key &k, uint64_t seed, hipStream_t stream) {
    if (k.curve == 0) generate_t<Bn256>(k, seed, stream); else generate_t<Grumpkin>(k, seed, stream);
}

size_t count_off_curve(const Key &k, hipStream_t stream) {
    if (k.len == 0) return 0;
    uint32_t *d_bad = nullptr, bad = 0;
    SRS_HIP_CHECK(hipMalloc((void **)&d_bad, sizeof(uint32_t)));
    SRS_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(uint32_t), stream));
    const uint32_t n = (uint32_t)k.len;
    if (k.curve == 0) SRS_LAUNCH((k_on_curve<Bn256>), (ceil_div(n, 256)), (256), 0, stream, (const affine_t *)k.table, n, d_bad);
    else SRS_LAUNCH((k_on_curve<Grumpkin>), (ceil_div(n, 256)), (256), 0, stream, (const affine_t *)k.table, n, d_bad);
    SRS_HIP_CHECK(hipMemcpyAsync(&bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    SRS_HIP_CHECK(hipStreamSynchronize(stream));
    (void)hipFree(d_bad);
    return bad;
}

void read_bases(const Key &k, affine_t *out_dev, hipStream_t stream) {
    if (k.len == 0) return;
    if (k.curve == 0) SRS_LAUNCH((k_table_unform<Bn256>), (ceil_div(k.len, 256)), (256), 0, stream, (const affine_t *)k.table, out_dev, k.len);
    else SRS_LAUNCH((k_table_unform<Grumpkin>), (ceil_div(k.len, 256)), (256), 0, stream, (const affine_t *)k.table, out_dev, k.len);
}

void build_table(Key &k, hipStream_t stream) {
    if (k.curve == 0) build_table_t<Bn256>(k, stream); else build_table_t<Grumpkin>(k, stream);
}

// number of thread-sequential levels after level 0 so that the wavefront-level final pass sees
// at most ~FINAL_FANIN parts per bucket even if every entry fell into one bucket
// gathered mixed additions per level-0 thread.  Two losses pull in opposite directions: every part leaves a partial sum the
// later levels combine with the dearer full additions (+ 1.4 / L0 of the level-0 work), and the launch ends with a
// partly filled last wave of workgroups (the chip holds 2^18 level-0 threads at a time: + ~0.5 / waves, waves = entries /
// (L0 * 2^18)).  16 is best up to ~10 M scalars (measured on the chunks of a 12 * 2^20 commit: 64 cost 0.9 ms more in
// k_accum0 than it saved in k_accum1), 32 above, 64 from ~40 M.  Tuning msm_l0 = <log2> forces a value (tests).
static uint32_t l0_log_for(uint64_t M) {
    const int forced = (int)tuning::get_or(tuning::MSM_L0, 0);
    if (forced >= 1 && forced <= 7) return (uint32_t)forced;
    // small MSMs (the support circuit's 3 * 2^15 and 2^15-row commits) cannot fill the chip with 16-entry parts: 2^21 digit slots give
    // < 2^17 level-0 threads, each a chain of 16 dependent additions on a chip that holds 2^17.6 -- shorter parts, more threads (r03)
    // r05: parts of 8 from 2^20 digit slots on (was 2^21): the support circuit's batch (3 * 2^15 + 2 x 2^15 scalars, 1.6 M slots) 589 -> 501 us per call,
    // profiles/r05_ab_small_msm.txt -- with parts of 4 its first accumulation level has 5 partial sums per bucket to combine, with 8 two or three
    if (M < (1ull << 20)) return 2;
    if (M < (1ull << 22)) return 3;
    uint32_t lg = ACC_L0_LOG;
    while (lg < 7 && (M >> (lg + 1)) >= (5ull << 20)) ++lg;
    return lg;
}

// k_accum1 gives every output to a quad of lanes while a level has at most this many outputs (x batch): a level that cannot
// fill the chip's 2^17.6 resident lanes is a chain of dependent additions, and a quad shortens each by 3.3x for 1.3x the
// lane-cycles.  Tuning msm_quad_max = <log2> overrides (tests).
static uint32_t acc1_quad_max() {
    const int64_t lg = tuning::get_or(tuning::MSM_QUAD_MAX, 0);
    return (lg >= 1 && lg <= 24) ? (1u << lg) : ACC1_QUAD_MAX;
}

static int levels_for(uint64_t max_entries) {
    const uint64_t ACC_L0 = 1ull << l0_log_for(max_entries);
    uint64_t parts = (max_entries + ACC_L0 - 1) / ACC_L0;
    int levels = 1;
    while (parts > FINAL_FANIN && levels < MAX_LEVELS) {
        parts = (parts + ACC_L1 - 1) / ACC_L1;
        ++levels;
    }
    return levels;
}

// large MSMs take the two-pass scatter (k_group + k_scatter2); tuning msm_sort = 1 / 2 forces the single- / two-pass path
static bool use_two_pass(uint64_t M, uint32_t batch) {
    const int forced = (int)tuning::get_or(tuning::MSM_SORT, 0);
    if (forced == 1) return false;
    if (forced == 2) return true;
    return M * batch >= TWO_PASS_MIN_SLOTS;
}

size_t workspace_bytes(uint32_t n_max, uint32_t batch) {
    uint64_t M = (uint64_t)n_max * NWIN;
    int levels = levels_for(M);
    size_t plan_stride = (size_t)(levels + 1) * (NBUCKET + 1) + 4;
    uint64_t parts0 = (M >> l0_log_for(M)) + NBUCKET + 1;
    uint64_t parts1 = parts0 / ACC_L1 + NBUCKET + 1;
    size_t per = 0;
    per += Arena::pad(M * sizeof(uint16_t));                 // digits
    per += Arena::pad(M * sizeof(uint32_t));                 // sorted
    per += 2 * Arena::pad(NBUCKET * sizeof(uint32_t));       // count, cursor
    per += Arena::pad(plan_stride * sizeof(uint32_t));
    per += Arena::pad(parts0 * sizeof(xyzz_t));              // ping
    per += Arena::pad(parts0 * sizeof(uint16_t));            // thread -> bucket map of level 0
    per += Arena::pad(parts1 * sizeof(xyzz_t));              // pong
    per += Arena::pad((size_t)NBUCKET * sizeof(xyzz_t));     // buckets
    per += Arena::pad((RED_ROWS + RED_COLS) * sizeof(xyzz_t));
    if (use_two_pass(M, batch)) {
        per += Arena::pad(M * sizeof(uint16_t)) + Arena::pad(M * sizeof(uint32_t));                     // grouped keys / payloads
        per += Arena::pad((size_t)SEG * (SORT_TARGET_BLOCKS + 2 * NWIN) * sizeof(uint32_t));            // per-tile segment counts
    }
    return per * batch + Arena::pad(3 * batch * sizeof(xyzz_t)) + 4096;
}

// launches one set of <= BATCH_ARGS MSMs on `stream`; the 3 partial sums of MSM m land in landing slot `slot`
// (page-locked host memory) in stream order.  Returns false when every MSM is empty (nothing was launched).
template <class C>
static bool enqueue_t(Key &k, const fe_t *const *scalars_dev, const uint32_t *n_host, const uint32_t *base_host, uint32_t batch,
                      int is_mont, hipStream_t stream, uint32_t slot, Fold fold) {
    uint32_t n_max = 0;
    for (uint32_t m = 0; m < batch; ++m) n_max = std::max(n_max, n_host[m]);
    if (n_max == 0) return false;
    const uint64_t M = (uint64_t)n_max * NWIN;
    const int levels = levels_for(M);
    const uint32_t l0_log = l0_log_for(M);
    const size_t plan_stride = (size_t)(levels + 1) * (NBUCKET + 1) + 4;
    const uint64_t parts0_cap = (M >> l0_log) + NBUCKET + 1;
    const uint64_t parts1_cap = parts0_cap / ACC_L1 + NBUCKET + 1;

    Arena &A = k.arena;
    A.reserve(workspace_bytes(n_max, batch));
    A.reset();
    xyzz_t *d_out = A.take<xyzz_t>(3 * (size_t)batch);
    uint16_t *dig = A.take<uint16_t>(M * batch);
    uint32_t *sorted = A.take<uint32_t>(M * batch);
    uint32_t *count = A.take<uint32_t>((size_t)NBUCKET * batch);
    uint32_t *cursor = A.take<uint32_t>((size_t)NBUCKET * batch);
    uint32_t *plan = A.take<uint32_t>(plan_stride * batch);
    xyzz_t *ping = A.take<xyzz_t>(parts0_cap * batch);
    uint16_t *tb = A.take<uint16_t>(parts0_cap * batch);
    xyzz_t *pong = A.take<xyzz_t>(parts1_cap * batch);
    xyzz_t *buckets = A.take<xyzz_t>((size_t)NBUCKET * batch);
    xyzz_t *rc = A.take<xyzz_t>((size_t)(RED_ROWS + RED_COLS) * batch);
    const bool two_pass = use_two_pass(M, batch);
    uint16_t *gkey = two_pass ? A.take<uint16_t>(M * batch) : nullptr;
    uint32_t *gpay = two_pass ? A.take<uint32_t>(M * batch) : nullptr;
    uint32_t *tile_hist = two_pass ? A.take<uint32_t>((size_t)SEG * (SORT_TARGET_BLOCKS + 2 * NWIN) * batch) : nullptr;

    BatchDesc bd;
    for (uint32_t m = 0; m < BATCH_ARGS; ++m) {
        bd.ptr[m] = m < batch ? scalars_dev[m] : nullptr;
        bd.n[m] = m < batch ? n_host[m] : 0;
        bd.base[m] = (m < batch && base_host) ? base_host[m] : 0;
    }
    const uint32_t s_rank = k.compact_scalars ? 0u : k.rank, s_world = k.compact_scalars ? 1u : k.world;
    // tile = digits per workgroup: large enough that the fixed 2^15-bin zero/scan of the LDS histogram is
    // amortised, small enough to give ~SORT_TARGET_BLOCKS workgroups (one per CU, 128 KiB LDS each)
    uint32_t tile = (uint32_t)(((uint64_t)n_max * NWIN * batch + SORT_TARGET_BLOCKS - 1) / SORT_TARGET_BLOCKS);
    tile = (tile + 1023u) & ~1023u;
    if (tile < SORT_TILE_MIN) tile = SORT_TILE_MIN;
    const uint32_t tiles = ceil_div(n_max, tile);
    const Link *no_link = nullptr;
    SRS_LAUNCH((k_digits<C>), (ceil_div(n_max, 256), batch), (256), 0, stream, bd, dig, (size_t)M, is_mont, s_rank, s_world, count,
               (uint32_t)(NBUCKET * batch));
    SRS_LAUNCH(k_hist, (tiles, NWIN, batch), (SORT_THREADS), 0, stream, (const uint16_t *)dig, (size_t)M,
               bd, count, tile, two_pass ? tile_hist : (uint32_t *)nullptr);
    SRS_LAUNCH(k_plan, (batch, levels + 1), (PLAN_THREADS), 0, stream, (const uint32_t *)count, cursor, plan, plan_stride,
               levels, l0_log, (uint32_t)ACC_L1_LOG, (const uint32_t *)nullptr);
    if (two_pass) {
        const uint32_t T1 = tiles * NWIN;
        SRS_LAUNCH(k_scan_seg, (SEG, batch), (1024), 0, stream, tile_hist, T1, (const uint32_t *)plan, plan_stride);
        SRS_LAUNCH(k_group, (tiles, NWIN, batch), (SORT_THREADS), 0, stream, (const uint16_t *)dig, (size_t)M, bd,
                   (const uint32_t *)tile_hist, gkey, gpay, (size_t)M, (uint32_t)k.len, tile, (const uint32_t *)nullptr, (size_t)0,
                   (uint16_t *)nullptr, (size_t)0, 0u);
        // 8 x ceil(tiles / 8) workgroups: the XCD-aware mapping of k_scatter2 needs every (XCD, slot) pair to exist
        SRS_LAUNCH(k_scatter2, (8 * ceil_div(ceil_div(M, SORT_TILE2), 8), batch), (SORT_THREADS), 0, stream, (const uint16_t *)gkey,
                   (const uint32_t *)gpay, (size_t)M, (const uint32_t *)plan, plan_stride, cursor, sorted, (size_t)M,
                   (uint32_t)SORT_TILE2);
    } else {
        SRS_LAUNCH(k_scatter, (tiles, NWIN, batch), (SORT_THREADS), 0, stream, (const uint16_t *)dig, (size_t)M,
                   bd, cursor, sorted, (size_t)M, (uint32_t)k.len, tile);
    }
    SRS_LAUNCH(k_expand, (NBUCKET / 4, batch), (256), 0, stream, (const uint32_t *)plan, plan_stride, tb, (size_t)parts0_cap, no_link, 1u);

    uint64_t units = 0;
    for (uint32_t m = 0; m < batch; ++m) units += n_host[m];
    SRS_LAUNCH_TIMED("msm_accum0", units, (k_accum0<C>), (ceil_div(parts0_cap, ACC_THREADS), batch), (ACC_THREADS), 0, stream,
                     (const uint32_t *)sorted, (size_t)M, (const uint32_t *)plan, plan_stride, (const uint16_t *)tb, (size_t)parts0_cap,
                     (const affine_t *)k.table, ping, (size_t)parts0_cap, 1u << l0_log, no_link);
    xyzz_t *cur = ping, *nxt = pong;
    size_t cur_stride = parts0_cap, nxt_stride = parts1_cap;
    uint64_t cap = parts0_cap;
    for (int level = 1; level < levels; ++level) {
        cap = cap / ACC_L1 + NBUCKET + 1;
        SRS_LAUNCH((k_accum1<C>), (ceil_div(std::max<uint64_t>(cap, 4 * std::min<uint64_t>(cap, acc1_quad_max())), ACC_THREADS), batch), (ACC_THREADS), 0, stream,
                   (const xyzz_t *)cur, cur_stride, (const uint32_t *)plan, plan_stride, level, nxt, nxt_stride,
                   (uint32_t)ACC_L1, no_link, acc1_quad_max());
        std::swap(cur, nxt);
        std::swap(cur_stride, nxt_stride);
        // both buffers can hold any later level: parts shrink monotonically and pong >= level-1 cap
    }
    (void)cur;
    (void)cur_stride;
    SRS_LAUNCH((k_accum_final<C>), (NBUCKET / (FINAL_THREADS / 64), batch), (FINAL_THREADS), 0, stream,
               (const xyzz_t *)ping, (size_t)parts0_cap, (const xyzz_t *)pong, (size_t)parts1_cap,
               (const uint32_t *)plan, plan_stride, buckets, no_link);
    const xyzz_t *red_from = buckets;
    int from_buckets = 0;
    if (fold != FOLD_NONE) {
        if (!k.fold_buckets) SRS_HIP_CHECK(hipMalloc((void **)&k.fold_buckets, (size_t)NBUCKET * sizeof(xyzz_t)));
        SRS_LAUNCH((k_bucket_fold<C>), (NBUCKET / 64), (64), 0, stream, (const xyzz_t *)buckets, (const xyzz_t *)ping, (const xyzz_t *)pong,
                   (const uint32_t *)plan, plan_stride, k.fold_buckets, fold == FOLD_FIRST ? 1 : 0);
        if (fold != FOLD_LAST) return true;                  // no reduction, no result for this set
        red_from = k.fold_buckets;
        from_buckets = 1;
    }
    SRS_LAUNCH((k_rowcol<C>), (RED_ROWS / 2 + RED_COLS, batch), (128), 0, stream, red_from,
               (const xyzz_t *)ping, (size_t)parts0_cap, (const xyzz_t *)pong, (size_t)parts1_cap, (const uint32_t *)plan,
               plan_stride, rc, no_link, from_buckets);
    SRS_LAUNCH((k_reduce_final<C>), (3, batch), (RED_THREADS), 0, stream, (const xyzz_t *)rc, d_out);
    if (!k.h_result) SRS_HIP_CHECK(hipHostMalloc(&k.h_result, 3 * (size_t)BATCH_ARGS * LANDING_SLOTS * sizeof(xyzz_t)));
    xyzz_t *two = static_cast<xyzz_t *>(k.h_result) + 3 * (size_t)BATCH_ARGS * slot;
    SRS_HIP_CHECK(hipMemcpyAsync(two, d_out, 3 * (size_t)batch * sizeof(xyzz_t), hipMemcpyDeviceToHost, stream));
    k.slot_wide[slot] = false;
    return true;
}

static bool use_wide(const Key &k, uint32_t n_max, uint32_t batch);
// ---- slot mode: host side -------------------------------------------------------------------------------------------------------
static uint32_t slot_log() {
    const int64_t lg = tuning::get_or(tuning::MSM_SLOT_LOG, 0);
    return (lg >= 2 && lg <= 8) ? (uint32_t)lg : SLOT_LOG;
}
// the part length a large set prefers when its buckets would fit shorter parts: 2^4
static uint32_t slot_want_cap() { return 4u; }
// Slot mode is for the sets of a CHUNKED commit (fold != FOLD_NONE): that is where every set used to pay its own accumulation levels, wave-level
// pass and bucket fold.  A set that is a whole MSM keeps the r03 flow, by measurement (profiles/r04_ab_slots_cuts.txt): the batched
// cross-term commitments would each pay a slot reduction (k = 17 Sangria step 6.02 vs 5.76 ms), a single 2^24 MSM is 2 % slower (22.3 vs
// 21.85 ms).  Tuning msm_slots = 0: never; 2: every 16-bit-window set (what the emulator tests force).
static bool use_slots(const Key &k, uint32_t n_max, uint32_t batch, Fold fold) {
    const int mode = (int)tuning::get_or(tuning::MSM_SLOTS, 1);
    if (mode == 0) return false;
    if (fold != FOLD_NONE) return batch == 1;
    return mode == 2 && !use_wide(k, n_max, batch);
}
struct SlotShape {
    uint32_t S, nred;          // slots per bucket; reduction levels at the end of a commit (8 inputs per output)
    uint64_t M, cap, cap1;     // digit slots; level-0 threads = overflow parts (worst case); parts after the first overflow level
    int levels;                // overflow levels incl. level 0
    size_t plan_stride;
};
static SlotShape slot_shape(uint32_t n_max) {
    SlotShape h;
    h.S = 1u << slot_log();
    h.nred = (slot_log() + 2) / 3;
    h.M = (uint64_t)n_max * NWIN;
    // parts = sum_b ceil(c_b / L0) <= E / L0 + NBUCKET, and k_plan_s picks L0 >= (mean load) / (S - 1): never more than NBUCKET * S parts;
    // L0 >= 2^SLOT_L0_MIN_LOG bounds them for small sets
    h.cap = std::min<uint64_t>((uint64_t)NBUCKET * h.S, (h.M >> SLOT_L0_MIN_LOG) + NBUCKET);
    h.cap1 = h.cap / ACC_L1 + NBUCKET + 1;
    uint64_t parts = h.cap;
    h.levels = 1;
    while (parts > FINAL_FANIN && h.levels < MAX_LEVELS) {
        parts = (parts + ACC_L1 - 1) / ACC_L1;
        ++h.levels;
    }
    h.plan_stride = (size_t)(h.levels + 2) * (NBUCKET + 1) + 4;
    return h;
}
static size_t workspace_bytes_slots(uint32_t n_max, uint32_t batch) {
    const SlotShape h = slot_shape(n_max);
    size_t per = 0;
    per += Arena::pad(h.M * sizeof(uint16_t)) + Arena::pad(h.M * sizeof(uint32_t));          // digits, sorted
    per += 2 * Arena::pad(NBUCKET * sizeof(uint32_t)) + Arena::pad(h.plan_stride * sizeof(uint32_t));
    per += Arena::pad(h.cap * sizeof(xyzz_t)) + Arena::pad(h.cap * sizeof(uint16_t)) + Arena::pad(h.cap1 * sizeof(xyzz_t));   // overflow parts, map, pong
    per += Arena::pad((size_t)NBUCKET * (h.S / 8 + 1) * sizeof(xyzz_t)) + Arena::pad((size_t)NBUCKET * (h.S / 64 + 1) * sizeof(xyzz_t));   // reduction ping / pong
    per += Arena::pad((RED_ROWS + RED_COLS) * sizeof(xyzz_t));
    if (use_two_pass(h.M, batch)) {
        per += Arena::pad(h.M * sizeof(uint16_t)) + Arena::pad(h.M * sizeof(uint32_t));
        per += Arena::pad((size_t)SEG * (SORT_TARGET_BLOCKS + 2 * NWIN) * sizeof(uint32_t));
    }
    return per * batch + Arena::pad(3 * batch * sizeof(xyzz_t)) + 8192;
}

// one set of <= BATCH_ARGS MSMs in slot mode.  fold: NONE = a whole commit (first and last set), FIRST / MIDDLE / LAST = the sets of a
// chunked commit (batch == 1): only the last one reduces the slots and lands the 3 partial sums.
template <class C>
static bool enqueue_slots_t(Key &k, const fe_t *const *scalars_dev, const uint32_t *n_host, const uint32_t *base_host, uint32_t batch,
                            int is_mont, hipStream_t stream, uint32_t slot, Fold fold) {
    uint32_t n_max = 0;
    for (uint32_t m = 0; m < batch; ++m) n_max = std::max(n_max, n_host[m]);
    if (n_max == 0) return false;
    const bool first = fold == FOLD_NONE || fold == FOLD_FIRST, last = fold == FOLD_NONE || fold == FOLD_LAST;
    const SlotShape h = slot_shape(n_max);
    const uint32_t S = h.S;
    const uint64_t M = h.M;
    if (first) {
        k.slot_s = S;
        k.seq = 0;
        k.commit_ovf = k.expect_ovf;
        const size_t need = (size_t)batch * NBUCKET * S;
        if (k.slots_pts < need) {
            if (k.slots) SRS_HIP_CHECK(hipFree(k.slots));
            k.slots = nullptr;
            k.slots_pts = 0;
            SRS_HIP_CHECK(hipMalloc((void **)&k.slots, need * sizeof(xyzz_t)));
            k.slots_pts = need;
        }
        if (!k.used) SRS_HIP_CHECK(hipMalloc((void **)&k.used, 2 * (size_t)BATCH_ARGS * NBUCKET));
        if (!k.h_ovf) {
            SRS_HIP_CHECK(hipHostMalloc((void **)&k.h_ovf, 2 * (size_t)LANDING_SLOTS * BATCH_ARGS * sizeof(uint32_t)));
            for (size_t i = 0; i < 2 * (size_t)LANDING_SLOTS * BATCH_ARGS; ++i) k.h_ovf[i] = 0;
        }
    } else if (k.slot_s != S || batch != 1) {
        set_error("internal: msm slot mode: a commit's sets must share the slot layout");
        throw DeviceError{5};
    }
    const uint8_t *used_prev = k.used + (size_t)(k.seq & 1u) * BATCH_ARGS * NBUCKET;
    uint8_t *used_next = k.used + (size_t)((k.seq + 1) & 1u) * BATCH_ARGS * NBUCKET;
    ++k.seq;

    Arena &A = k.arena;
    A.reserve(workspace_bytes_slots(n_max, batch));
    A.reset();
    xyzz_t *d_out = A.take<xyzz_t>(3 * (size_t)batch);
    uint16_t *dig = A.take<uint16_t>(M * batch);
    uint32_t *sorted = A.take<uint32_t>(M * batch);
    uint32_t *count = A.take<uint32_t>((size_t)NBUCKET * batch);
    uint32_t *cursor = A.take<uint32_t>((size_t)NBUCKET * batch);
    uint32_t *plan = A.take<uint32_t>(h.plan_stride * batch);
    xyzz_t *ping = A.take<xyzz_t>(h.cap * batch);
    uint16_t *tb = A.take<uint16_t>(h.cap * batch);
    xyzz_t *pong = A.take<xyzz_t>(h.cap1 * batch);
    xyzz_t *red_a = A.take<xyzz_t>((size_t)NBUCKET * (S / 8 + 1) * batch);
    xyzz_t *red_b = A.take<xyzz_t>((size_t)NBUCKET * (S / 64 + 1) * batch);
    xyzz_t *rc = A.take<xyzz_t>((size_t)(RED_ROWS + RED_COLS) * batch);
    const bool two_pass = use_two_pass(M, batch);
    uint16_t *gkey = two_pass ? A.take<uint16_t>(M * batch) : nullptr;
    uint32_t *gpay = two_pass ? A.take<uint32_t>(M * batch) : nullptr;
    uint32_t *tile_hist = two_pass ? A.take<uint32_t>((size_t)SEG * (SORT_TARGET_BLOCKS + 2 * NWIN) * batch) : nullptr;

    BatchDesc bd;
    for (uint32_t m = 0; m < BATCH_ARGS; ++m) {
        bd.ptr[m] = m < batch ? scalars_dev[m] : nullptr;
        bd.n[m] = m < batch ? n_host[m] : 0;
        bd.base[m] = (m < batch && base_host) ? base_host[m] : 0;
    }
    const uint32_t s_rank = k.compact_scalars ? 0u : k.rank, s_world = k.compact_scalars ? 1u : k.world;
    uint32_t tile = (uint32_t)(((uint64_t)n_max * NWIN * batch + SORT_TARGET_BLOCKS - 1) / SORT_TARGET_BLOCKS);
    tile = (tile + 1023u) & ~1023u;
    if (tile < SORT_TILE_MIN) tile = SORT_TILE_MIN;
    const uint32_t tiles = ceil_div(n_max, tile);
    uint32_t *h_ovf = k.h_ovf + (size_t)slot * BATCH_ARGS;
    const uint32_t arr = (uint32_t)h.levels + 1;
    SRS_LAUNCH((k_digits<C>), (ceil_div(n_max, 256), batch), (256), 0, stream, bd, dig, (size_t)M, is_mont, s_rank, s_world, count,
               (uint32_t)(NBUCKET * batch));
    SRS_LAUNCH(k_hist, (tiles, NWIN, batch), (SORT_THREADS), 0, stream, (const uint16_t *)dig, (size_t)M,
               bd, count, tile, two_pass ? tile_hist : (uint32_t *)nullptr);
    SRS_LAUNCH(k_plan_s, (batch, h.levels + 2), (PLAN_THREADS), 0, stream, (const uint32_t *)count, cursor, plan, h.plan_stride, h.levels, S,
               (uint32_t)ACC_L1_LOG, used_prev, used_next, first ? 1 : 0, h_ovf, slot_want_cap());
    if (two_pass) {
        const uint32_t T1 = tiles * NWIN;
        SRS_LAUNCH(k_scan_seg, (SEG, batch), (1024), 0, stream, tile_hist, T1, (const uint32_t *)plan, h.plan_stride);
        SRS_LAUNCH(k_group, (tiles, NWIN, batch), (SORT_THREADS), 0, stream, (const uint16_t *)dig, (size_t)M, bd,
                   (const uint32_t *)tile_hist, gkey, gpay, (size_t)M, (uint32_t)k.len, tile, (const uint32_t *)plan, h.plan_stride, tb,
                   (size_t)h.cap, (uint32_t)h.levels + 1);                 // + the thread -> bucket map (k_expand's work)
        SRS_LAUNCH(k_scatter2, (8 * ceil_div(ceil_div(M, SORT_TILE2), 8), batch), (SORT_THREADS), 0, stream, (const uint16_t *)gkey,
                   (const uint32_t *)gpay, (size_t)M, (const uint32_t *)plan, h.plan_stride, cursor, sorted, (size_t)M,
                   (uint32_t)SORT_TILE2);
    } else {
        SRS_LAUNCH(k_scatter, (tiles, NWIN, batch), (SORT_THREADS), 0, stream, (const uint16_t *)dig, (size_t)M,
                   bd, cursor, sorted, (size_t)M, (uint32_t)k.len, tile);
    }
    uint64_t units = 0;
    for (uint32_t m = 0; m < batch; ++m) units += n_host[m];
    const Link *no_link = nullptr;
    if (!two_pass) SRS_LAUNCH(k_expand, (NBUCKET / 4, batch), (256), 0, stream, (const uint32_t *)plan, h.plan_stride, tb, (size_t)h.cap, no_link, arr);
    SRS_LAUNCH_TIMED("msm_accum0", units, (k_accum0s<C>), (ceil_div(h.cap, ACC_THREADS), batch), (ACC_THREADS), 0, stream,
                     (const uint32_t *)sorted, (size_t)M, (const uint32_t *)plan, h.plan_stride, arr, (const uint16_t *)tb, (size_t)h.cap,
                     (const affine_t *)k.table, k.slots, S, used_prev, first ? 1 : 0, ping, (size_t)h.cap);
    if (k.commit_ovf) {                      // hot buckets expected: the parts beyond the slots go through the level kernels into slot S - 1
        xyzz_t *cur = ping, *nxt = pong;
        size_t cur_stride = h.cap, nxt_stride = h.cap1;
        uint64_t cap = h.cap;
        for (int level = 1; level < h.levels; ++level) {
            cap = cap / ACC_L1 + NBUCKET + 1;
            SRS_LAUNCH((k_accum1<C>), (ceil_div(std::max<uint64_t>(cap, 4 * std::min<uint64_t>(cap, acc1_quad_max())), ACC_THREADS), batch), (ACC_THREADS), 0, stream,
                       (const xyzz_t *)cur, cur_stride, (const uint32_t *)plan, h.plan_stride, level, nxt, nxt_stride, (uint32_t)ACC_L1, no_link,
                       acc1_quad_max());
            std::swap(cur, nxt);
            std::swap(cur_stride, nxt_stride);
        }
        SRS_LAUNCH((k_ovf_final<C>), (NBUCKET / (FINAL_THREADS / 64), batch), (FINAL_THREADS), 0, stream, (const xyzz_t *)ping, (size_t)h.cap,
                   (const xyzz_t *)pong, (size_t)h.cap1, (const uint32_t *)plan, h.plan_stride, k.slots, S, first ? 1 : 0);
    }
    k.slot_mode[slot] = true;
    ++k.stat_slot_sets;
    k.slot_ovf_on[slot] = k.commit_ovf;
    k.slot_batch[slot] = batch;
    k.slot_wide[slot] = false;
    if (!last) return true;
    // the commit's one reduction: slots -> buckets (8 inputs per output and level), then the usual row / column sums
    const xyzz_t *in = k.slots;
    uint32_t in_pb = S;
    xyzz_t *outs[2] = {red_a, red_b};
    for (uint32_t l = 0; l < h.nred; ++l) {
        const uint32_t out_pb = (in_pb + 7) / 8, n_out = NBUCKET * out_pb;
        xyzz_t *out = outs[l & 1u];
        const uint8_t *used = l == 0 ? used_next : nullptr;
        const int extra = (l == 0 && k.commit_ovf) ? (int)S - 1 : -1;
        if (l + 1 == h.nred) {
            SRS_LAUNCH((k_slot_reduce<C, true>), (ceil_div((uint64_t)n_out * 4, 256), batch), (256), 0, stream, in, in_pb, used, (size_t)NBUCKET, extra,
                       out, out_pb);
        } else {
            SRS_LAUNCH((k_slot_reduce<C, false>), (ceil_div(n_out, 256), batch), (256), 0, stream, in, in_pb, used, (size_t)NBUCKET, extra, out,
                       out_pb);
        }
        in = out;
        in_pb = out_pb;
    }
    SRS_LAUNCH((k_rowcol<C>), (RED_ROWS / 2 + RED_COLS, batch), (128), 0, stream, in, (const xyzz_t *)ping, (size_t)h.cap,
               (const xyzz_t *)pong, (size_t)h.cap1, (const uint32_t *)plan, h.plan_stride, rc, no_link, 1);
    SRS_LAUNCH((k_reduce_final<C>), (3, batch), (RED_THREADS), 0, stream, (const xyzz_t *)rc, d_out);
    if (!k.h_result) SRS_HIP_CHECK(hipHostMalloc(&k.h_result, 3 * (size_t)BATCH_ARGS * LANDING_SLOTS * sizeof(xyzz_t)));
    xyzz_t *two = static_cast<xyzz_t *>(k.h_result) + 3 * (size_t)BATCH_ARGS * slot;
    SRS_HIP_CHECK(hipMemcpyAsync(two, d_out, 3 * (size_t)batch * sizeof(xyzz_t), hipMemcpyDeviceToHost, stream));
    return true;
}

bool overflow_missed(const Key &k, uint32_t slot) {
    if (slot >= LANDING_SLOTS || !k.slot_mode[slot] || k.slot_ovf_on[slot] || !k.h_ovf) return false;
    for (uint32_t m = 0; m < k.slot_batch[slot]; ++m)
        if (k.h_ovf[(size_t)slot * BATCH_ARGS + m]) return true;
    return false;
}
void note_commit(Key &k, const uint32_t *slots_used, uint32_t n_slots, uint64_t scalars) {
    bool any = false, slot_sets = false;
    uint64_t entries = 0;
    for (uint32_t i = 0; i < n_slots; ++i) {
        const uint32_t sl = slots_used[i];
        if (sl >= LANDING_SLOTS || !k.slot_mode[sl] || !k.h_ovf) continue;
        slot_sets = true;
        for (uint32_t m = 0; m < k.slot_batch[sl]; ++m) entries += k.h_ovf[(size_t)(LANDING_SLOTS + sl) * BATCH_ARGS + m];
        bool hot = false;
        for (uint32_t m = 0; m < k.slot_batch[sl]; ++m) hot = hot || k.h_ovf[(size_t)sl * BATCH_ARGS + m] != 0;
        if (hot) ++k.stat_hot_sets;
        any = any || hot;
    }
    // hot buckets are expected at once -- and from a key's FIRST commit (r05: Key::expect_ovf starts true; a witness of 0 / 1 / small values is
    // the common case and its first commit used to run twice) -- and un-expected only after COLD_COMMITS commits in a row without them:
    // launching the overflow kernels for nothing costs ~0.3 ms of empty launches per commit, missing them costs the commit run a second time
    constexpr uint32_t COLD_COMMITS = 3;
    if (!slot_sets) {                        // no slot-mode set: no entry count either -- the next commit's cuts fall back to the default schedule
        k.last_entries = k.last_scalars = 0;
        return;
    }
    k.last_entries = entries;
    k.last_scalars = scalars;
    if (any) {
        k.expect_ovf = true;
        k.cold_streak = 0;
    } else if (k.expect_ovf && ++k.cold_streak >= COLD_COMMITS) {
        k.expect_ovf = false;
        k.cold_streak = 0;
    }
}

// ---- the wide-window pipeline: ONE MSM of n >= 2^WIDE_MIN_N_LOG scalars over table_w -------------------------------------
struct WideShape {
    uint64_t M;                // entry capacity = n * NWIN_W
    int levels;
    uint32_t l0_log, T, tiles_g;
    size_t plan_stride;
    uint64_t parts0_cap, parts1_cap;
};
static WideShape wide_shape(uint32_t n) {
    WideShape w;
    w.M = (uint64_t)n * NWIN_W;
    w.levels = levels_for(w.M);                              // worst case: every entry in one bucket of one segment
    w.l0_log = l0_log_for(w.M);
    w.plan_stride = (size_t)(w.levels + 1) * (NBUCKET + 1) + 4;
    w.parts0_cap = (w.M >> w.l0_log) + (uint64_t)NSEG_W * (NBUCKET + 1);
    w.parts1_cap = w.parts0_cap / ACC_L1 + (uint64_t)NSEG_W * (NBUCKET + 1);
    w.T = ceil_div(n, WIDE_TILE);
    w.tiles_g = SORT_TARGET_BLOCKS;                          // the tile size is chosen on the device (k_seg_scan) so that this many suffice
    return w;
}
static size_t workspace_bytes_wide(uint32_t n) {
    const WideShape w = wide_shape(n);
    size_t b = 0;
    b += Arena::pad((4 + 4 * NSEG_W) * sizeof(xyzz_t));                          // results
    b += Arena::pad(w.M * sizeof(uint16_t)) + 2 * Arena::pad(w.M * sizeof(uint32_t));   // grouped keys / payloads, sorted
    b += Arena::pad(w.M * sizeof(uint16_t)) + Arena::pad(w.M * sizeof(uint32_t));       // ... grouped again by sub-segment (two-pass sort)
    b += Arena::pad((size_t)SEG * w.tiles_g * sizeof(uint32_t));                 // per-tile sub-segment counts
    b += Arena::pad((size_t)NSEG_W * w.T * sizeof(uint32_t));                    // per-tile segment counts
    b += Arena::pad(4 * (NSEG_W + 2) * sizeof(uint32_t)) + Arena::pad(sizeof(Link));
    b += 2 * Arena::pad((size_t)NSEG_W * NBUCKET * sizeof(uint32_t));            // count, cursor
    b += Arena::pad(w.plan_stride * NSEG_W * sizeof(uint32_t));
    b += Arena::pad(w.parts0_cap * sizeof(xyzz_t)) + Arena::pad(w.parts0_cap * sizeof(uint16_t));   // ping, map
    b += Arena::pad(w.parts1_cap * sizeof(xyzz_t));                              // pong
    b += Arena::pad((size_t)NSEG_W * NBUCKET * sizeof(xyzz_t));                  // buckets
    b += Arena::pad((size_t)NSEG_W * (RED_ROWS + RED_COLS) * sizeof(xyzz_t));
    return b + 4096;
}

static bool use_wide(const Key &k, uint32_t n_max, uint32_t batch) {
    const int min_log = (int)tuning::get_or(tuning::MSM_WIDE_MIN, (int64_t)WIDE_MIN_N_LOG);
    return k.table_w != nullptr && batch == 1 && n_max >= (1u << min_log);
}

template <class C>
static bool enqueue_wide_t(Key &k, const fe_t *scalars_dev, uint32_t n, uint32_t base, int is_mont, hipStream_t stream, uint32_t slot) {
    const WideShape w = wide_shape(n);
    Arena &A = k.arena;
    A.reserve(workspace_bytes_wide(n));
    A.reset();
    xyzz_t *d_out = A.take<xyzz_t>(4);
    xyzz_t *d_seg = A.take<xyzz_t>(4 * NSEG_W);
    uint16_t *gkey = A.take<uint16_t>(w.M);
    uint32_t *gpay = A.take<uint32_t>(w.M);
    uint32_t *sorted = A.take<uint32_t>(w.M);
    uint32_t *tile_cnt = A.take<uint32_t>((size_t)NSEG_W * w.T);
    // the counting sort inside the segments: two passes through LDS (k_group_g + k_scatter2_g)
    uint16_t *gkey2 = A.take<uint16_t>(w.M);
    uint32_t *gpay2 = A.take<uint32_t>(w.M);
    uint32_t *tile_hist = A.take<uint32_t>((size_t)SEG * w.tiles_g);
    uint32_t *seg3 = A.take<uint32_t>(4 * (NSEG_W + 2));
    uint32_t *seg_total = seg3, *seg_off = seg3 + (NSEG_W + 2), *tile_base = seg3 + 2 * (NSEG_W + 2), *tile_base2 = seg3 + 3 * (NSEG_W + 2);
    Link *link = A.take<Link>(1);
    uint32_t *count = A.take<uint32_t>((size_t)NSEG_W * NBUCKET);
    uint32_t *cursor = A.take<uint32_t>((size_t)NSEG_W * NBUCKET);
    uint32_t *plan = A.take<uint32_t>(w.plan_stride * NSEG_W);
    xyzz_t *ping = A.take<xyzz_t>(w.parts0_cap);
    uint16_t *tb = A.take<uint16_t>(w.parts0_cap);
    xyzz_t *pong = A.take<xyzz_t>(w.parts1_cap);
    xyzz_t *buckets = A.take<xyzz_t>((size_t)NSEG_W * NBUCKET);
    xyzz_t *rc = A.take<xyzz_t>((size_t)NSEG_W * (RED_ROWS + RED_COLS));

    WideDesc wd;
    wd.ptr = scalars_dev;
    wd.n = n;
    wd.base = base;
    wd.rank = k.compact_scalars ? 0u : k.rank;
    wd.world = k.compact_scalars ? 1u : k.world;
    wd.is_mont = is_mont;
    const uint32_t table_stride = (uint32_t)k.len;

    // sort: segment counts -> offsets -> grouping (MSD pass), then the counting sort inside the segments
    SRS_HIP_CHECK(hipMemsetAsync(seg_total, 0, (NSEG_W + 1) * sizeof(uint32_t), stream));
    SRS_HIP_CHECK(hipMemsetAsync(count, 0, (size_t)NSEG_W * NBUCKET * sizeof(uint32_t), stream));
    SRS_LAUNCH((k_seg_pass<C, false>), (w.T), (WIDE_THREADS), 0, stream, wd, tile_cnt, w.T, seg_total, (uint16_t *)nullptr,
               (uint32_t *)nullptr, table_stride);
    const uint32_t small_tiles = (w.M >> 19) < 64 ? 1u : 0u, tile2_host = small_tiles ? SORT_TILE2 / 4 : SORT_TILE2;
    SRS_LAUNCH(k_seg_scan, (NSEG_W), (1024), 0, stream, tile_cnt, w.T, (const uint32_t *)seg_total, seg_off, tile_base, tile_base2, small_tiles);
    SRS_LAUNCH((k_seg_pass<C, true>), (w.T), (WIDE_THREADS), 0, stream, wd, tile_cnt, w.T, seg_total, gkey, gpay, table_stride);
    SRS_LAUNCH(k_hist_g, (w.tiles_g), (SORT_THREADS), 0, stream, (const uint16_t *)gkey, (const uint32_t *)seg_off,
               (const uint32_t *)tile_base, count, tile_hist);
    SRS_LAUNCH(k_plan, (NSEG_W, w.levels + 1), (PLAN_THREADS), 0, stream, (const uint32_t *)count, cursor, plan, w.plan_stride, w.levels, w.l0_log,
               (uint32_t)ACC_L1_LOG, (const uint32_t *)seg_off);
    SRS_LAUNCH(k_scan_seg_g, (SEG, NSEG_W), (1024), 0, stream, tile_hist, w.tiles_g, (const uint32_t *)tile_base, (const uint32_t *)plan,
               w.plan_stride);
    SRS_LAUNCH(k_group_g, (w.tiles_g), (SORT_THREADS), 0, stream, (const uint16_t *)gkey, (const uint32_t *)gpay, (const uint32_t *)seg_off,
               (const uint32_t *)tile_base, (const uint32_t *)tile_hist, gkey2, gpay2);
    // tiles of SORT_TILE2 entries cut per segment: at most M / SORT_TILE2 + NSEG_W of them; 8 x ceil(. / 8) workgroups (XCD mapping)
    SRS_LAUNCH(k_scatter2_g, (8 * ceil_div(ceil_div(w.M, tile2_host) + NSEG_W, 8)), (SORT_THREADS), 0, stream, (const uint16_t *)gkey2,
               (const uint32_t *)gpay2, (const uint32_t *)seg_off, (const uint32_t *)tile_base2, cursor, sorted);
    SRS_LAUNCH(k_link, (1), (64), 0, stream, (const uint32_t *)plan, w.plan_stride, w.levels, link);
    const Link *lk = link;
    SRS_LAUNCH(k_expand, (NBUCKET / 4, NSEG_W), (256), 0, stream, (const uint32_t *)plan, w.plan_stride, tb, (size_t)0, lk, 1u);
    SRS_LAUNCH_TIMED("msm_accum0", n, (k_accum0<C>), (ceil_div(w.parts0_cap, ACC_THREADS)), (ACC_THREADS), 0, stream, (const uint32_t *)sorted,
                     (size_t)0, (const uint32_t *)plan, w.plan_stride, (const uint16_t *)tb, (size_t)0, (const affine_t *)k.table_w, ping, (size_t)0,
                     1u << w.l0_log, lk);
    xyzz_t *cur = ping, *nxt = pong;
    uint64_t cap = w.parts0_cap;
    for (int level = 1; level < w.levels; ++level) {
        cap = cap / ACC_L1 + (uint64_t)NSEG_W * (NBUCKET + 1);
        SRS_LAUNCH((k_accum1<C>), (ceil_div(std::max<uint64_t>(cap, 4 * std::min<uint64_t>(cap, acc1_quad_max())), ACC_THREADS)), (ACC_THREADS), 0,
                   stream, (const xyzz_t *)cur, (size_t)0, (const uint32_t *)plan, w.plan_stride, level, nxt, (size_t)0, (uint32_t)ACC_L1, lk,
                   acc1_quad_max());
        std::swap(cur, nxt);
    }
    SRS_LAUNCH((k_accum_final<C>), (NBUCKET / (FINAL_THREADS / 64), NSEG_W), (FINAL_THREADS), 0, stream, (const xyzz_t *)ping, (size_t)0,
               (const xyzz_t *)pong, (size_t)0, (const uint32_t *)plan, w.plan_stride, buckets, lk);
    SRS_LAUNCH((k_rowcol<C>), (RED_ROWS / 2 + RED_COLS, NSEG_W), (128), 0, stream, (const xyzz_t *)buckets, (const xyzz_t *)ping, (size_t)0,
               (const xyzz_t *)pong, (size_t)0, (const uint32_t *)plan, w.plan_stride, rc, lk, 0);
    SRS_LAUNCH((k_reduce_final<C>), (4, NSEG_W), (RED_THREADS), 0, stream, (const xyzz_t *)rc, d_seg);
    SRS_LAUNCH((k_wide_combine<C>), (1), (256), 0, stream, (const xyzz_t *)d_seg, d_out);
    if (!k.h_result) SRS_HIP_CHECK(hipHostMalloc(&k.h_result, 3 * (size_t)BATCH_ARGS * LANDING_SLOTS * sizeof(xyzz_t)));
    xyzz_t *land = static_cast<xyzz_t *>(k.h_result) + 3 * (size_t)BATCH_ARGS * slot;
    SRS_HIP_CHECK(hipMemcpyAsync(land, d_out, 4 * sizeof(xyzz_t), hipMemcpyDeviceToHost, stream));
    k.slot_wide[slot] = true;
    return true;
}

// host end of enqueue_t, after `stream` has been synchronised:  S = RED_COLS * (2 A' + Z) + B  (10 group operations per MSM)
template <class C>
static void finish_t(Key &k, uint32_t batch, uint32_t slot, bool launched, xyzz_t *result_host) {
    if (!launched) {
        for (uint32_t m = 0; m < batch; ++m) result_host[m] = Ec<C>::identity();
        return;
    }
    const xyzz_t *raw = static_cast<const xyzz_t *>(k.h_result) + 3 * (size_t)BATCH_ARGS * slot;
    // the device sums arrive in the R' = 2^261 form of the 29-bit multiplier: times 2^-5 (as a Montgomery product) -> ABI form
    using F = typename C::F;
    fe_t c = F::one();
    for (int d = 0; d < 5; ++d) c = F::halve(c);
    std::vector<xyzz_t> two(3 * (size_t)batch + (k.slot_wide[slot] ? 1 : 0));
    for (size_t i = 0; i < two.size(); ++i) {
        two[i].x = F::mul(raw[i].x, c);
        two[i].y = F::mul(raw[i].y, c);
        two[i].zz = F::mul(raw[i].zz, c);
        two[i].zzz = F::mul(raw[i].zzz, c);
    }
    for (uint32_t m = 0; m < batch; ++m) {
        xyzz_t a = Ec<C>::add(Ec<C>::dbl(two[3 * m]), two[3 * m + 2]);
        for (uint32_t j = 1; j < RED_COLS; j <<= 1) a = Ec<C>::dbl(a);
        result_host[m] = Ec<C>::add(a, two[3 * m + 1]);
    }
    if (k.slot_wide[slot]) {                                 // one MSM, 4 sums: + NBUCKET * U for the segment totals
        xyzz_t u = two[3];
        for (uint32_t j = 1; j < NBUCKET; j <<= 1) u = Ec<C>::dbl(u);
        result_host[0] = Ec<C>::add(result_host[0], u);
    }
}

bool may_fold(const Key &, uint32_t) { return true; }      // (r03: not for sets that took the wide pipeline; they no longer do)

bool enqueue(Key &k, const fe_t *const *scalars_dev, const uint32_t *n_host, const uint32_t *base_host, uint32_t batch, int is_mont,
             hipStream_t stream, uint32_t slot, Fold fold) {
    if (batch > BATCH_ARGS || slot >= LANDING_SLOTS) {
        set_error("internal: msm::enqueue batch / slot out of range");
        throw DeviceError{5};
    }
    if (fold != FOLD_NONE && batch != 1) {
        set_error("internal: msm::enqueue fold needs one MSM per set");
        throw DeviceError{5};
    }
    if (batch == 1 && fold == FOLD_NONE && use_wide(k, n_host[0], 1)) {      // the sets of a chunked commit stay on the 16-bit windows
        const uint32_t base = base_host ? base_host[0] : 0;
        k.slot_mode[slot] = false;
        ++k.stat_other_sets;
        return k.curve == 0 ? enqueue_wide_t<Bn256>(k, scalars_dev[0], n_host[0], base, is_mont, stream, slot)
                            : enqueue_wide_t<Grumpkin>(k, scalars_dev[0], n_host[0], base, is_mont, stream, slot);
    }
    uint32_t n_max = 0;
    for (uint32_t m = 0; m < batch; ++m) n_max = std::max(n_max, n_host[m]);
    // the sets of one commit take the same flow: decided by its first set
    const bool first = fold == FOLD_NONE || fold == FOLD_FIRST;
    const bool slots = first ? use_slots(k, n_max, batch, fold) : k.slot_s != 0;
    if (slots)
        return k.curve == 0 ? enqueue_slots_t<Bn256>(k, scalars_dev, n_host, base_host, batch, is_mont, stream, slot, fold)
                            : enqueue_slots_t<Grumpkin>(k, scalars_dev, n_host, base_host, batch, is_mont, stream, slot, fold);
    if (first) k.slot_s = 0;
    k.slot_mode[slot] = false;
    ++k.stat_other_sets;
    return k.curve == 0 ? enqueue_t<Bn256>(k, scalars_dev, n_host, base_host, batch, is_mont, stream, slot, fold)
                        : enqueue_t<Grumpkin>(k, scalars_dev, n_host, base_host, batch, is_mont, stream, slot, fold);
}
void finish(Key &k, uint32_t batch, uint32_t slot, bool launched, xyzz_t *result_host) {
    if (k.curve == 0) finish_t<Bn256>(k, batch, slot, launched, result_host); else finish_t<Grumpkin>(k, batch, slot, launched, result_host);
}
void reserve(Key &k, uint32_t n_max, uint32_t batch) {
    size_t b = use_wide(k, n_max, batch) ? std::max(workspace_bytes_wide(n_max), workspace_bytes(n_max, batch)) : workspace_bytes(n_max, batch);
    if (batch == 1) b = std::max(b, workspace_bytes_slots(n_max, batch));        // (a chunked commit's sets: slot mode)
    k.arena.reserve(b);
}

void release(Key &k) {
    if (k.table_w) (void)hipFree(k.table_w);
    k.table_w = nullptr;
    if (k.fold_buckets) (void)hipFree(k.fold_buckets);
    k.fold_buckets = nullptr;
    if (k.h_result) (void)hipHostFree(k.h_result);
    k.h_result = nullptr;
    if (k.slots) (void)hipFree(k.slots);
    k.slots = nullptr;
    k.slots_pts = 0;
    if (k.used) (void)hipFree(k.used);
    k.used = nullptr;
    if (k.h_ovf) (void)hipHostFree(k.h_ovf);
    k.h_ovf = nullptr;
    k.arena.release();
}

void run(Key &k, const fe_t *const *scalars_dev, const uint32_t *n_host, uint32_t batch, int is_mont,
         hipStream_t stream, xyzz_t *result_host) {
    for (uint32_t at = 0; at < batch;) {                           // the batch descriptor is a kernel argument of BATCH_ARGS slots
        uint32_t b = std::min<uint32_t>(BATCH_ARGS, batch - at);
        if (use_wide(k, n_host[at], 1)) b = 1;                     // large vectors go through the wide-window pipeline one by one
        bool launched = enqueue(k, scalars_dev + at, n_host + at, nullptr, b, is_mont, stream, 0);
        if (launched) {
            SRS_HIP_CHECK(hipStreamSynchronize(stream));
            SRS_HIP_CHECK(hipGetLastError());
            const uint32_t sl = 0;
            if (overflow_missed(k, sl)) {                          // hot buckets the prediction did not expect: once more, with the overflow kernels
                k.expect_ovf = true;
                ++k.stat_redo;
                launched = enqueue(k, scalars_dev + at, n_host + at, nullptr, b, is_mont, stream, 0);
                SRS_HIP_CHECK(hipStreamSynchronize(stream));
                SRS_HIP_CHECK(hipGetLastError());
            }
            uint64_t total = 0;
            for (uint32_t m = 0; m < b; ++m) total += n_host[at + m];
            note_commit(k, &sl, 1, total);
        }
        finish(k, b, 0, launched, result_host + at);
        prof::collect();
        at += b;
    }
}

}  // namespace msm
}  // namespace srs
