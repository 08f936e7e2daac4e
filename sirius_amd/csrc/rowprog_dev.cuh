// rowprog_dev.cuh -- device-side definitions shared by the ahead-of-time kernels in rowprog.hip and by the row programs
// compiled at run time with hiprtc (jit.hip embeds this file verbatim: keep it free of host-only includes).
#pragma once
#include "field.cuh"
#include "field29.cuh"
#include "lanes.cuh"      // SRS_SWEEP_ACC

namespace srs {
namespace rowprog {

struct Insn {   // 16 bytes, read wave-uniformly
    uint32_t op, dst, a, b;
};

enum : uint32_t { I_LD_SEL = 0, I_LD_FIX, I_LD_ADV, I_ADD, I_SUB, I_MUL, I_SQR, I_DBL, I_NEG };
constexpr uint32_t UNIFORM_BIT = 0x80000000u;
constexpr uint32_t RP_THREADS = 128;
constexpr uint32_t DMAX = 8;   // cross terms kept in VGPRs per pass; higher degrees take ceil(d / 8) passes over the points
constexpr uint32_t DEGREE_LIMIT = 255;   // evaluation points 0..d must stay < 2^8 (small_times)

constexpr uint32_t JMAX = 16;  // witnesses combined by one advice load (cross terms: 2; ProtoGalaxy G: L + 1 <= 16)

// everything a row needs besides the program
struct RowCtx {
    uint32_t rows;
    const uint8_t *const *sel;
    const fe_t *const *fix;
    const fe_t *W[JMAX];      // column-major [num_advice][rows] each
    uint32_t J;               // number of witnesses
    const fe_t *wcoef;        // [npts][J] combination coefficients, or nullptr:
                              //   J == 1: W[0];  J == 2: W[0] + pt * W[1]  (cross-term points X = pt)
    uint32_t shard_rank, shard_world, local_rows;   // multi-GPU: this rank evaluates only the rows of ITS block-cyclic stripes
                              //   (2^ROW_STRIPE_LOG rows each, the stripes of the sharded commitment key); world == 1: all rows
    uint32_t half;            // J == 2, wcoef == nullptr: (W[0] + W[1]) / 2 + pt * (W[0] - W[1]) / 2, i.e. the Lagrange fold
                              //   L_0(X) W[0] + L_1(X) W[1] over the domain {1, -1} at the integer point X = pt (compute_G, L = 1)
    uint32_t pt0;             // sweep form only: the first evaluation point is X = pt0 instead of X = 0 (compute_G skips X = 1 when the
                              //   caller knows G(1) = F(alpha) and X = 0 with it: the points are 2 .. d + 1)
};

struct DevArgs {
    const Insn *prog;
    uint32_t n_insn;
    uint32_t result;          // operand code of the expression value
    RowCtx ctx;
    const fe_t *utab;         // [npts][n_uniform]
    uint32_t n_uniform;
    uint32_t npts;            // d + 1 (interpolate) or 1 (plain evaluation)
    uint32_t d;               // number of outputs in interpolate mode
    const fe_t *vinv;         // [d][npts]: T_k = sum_j vinv[(k-1)*npts + j] * P(j)
    const fe_t *vinv29;       // the same matrix * 2^5: operands of the 2^261-radix multiplier (sweep form)
    fe_t *const *out;         // d (interpolate) or 1 (plain) output vectors of `rows`
};

template <class F>
__device__ __forceinline__ fe_t small_times(const fe_t &x, uint32_t j) {   // j * x for a tiny j
    fe_t acc = F::zero();
    bool any = false;
    for (int b = j < 32u ? 4 : 7; b >= 0; --b) {
        if (any) acc = F::dbl(acc);
        if ((j >> b) & 1u) {
            acc = any ? F::add(acc, x) : x;
            any = true;
        }
    }
    return acc;
}

// One multiplier body per kernel for the straight-line programs (emit_spec_source): a program with every 2 KB multiplier
// inlined is hundreds of KB of code streamed once per evaluation point; called, it is tens of KB.
template <class F>
__device__ __attribute__((noinline)) fe_t mul_ni(fe_t a, fe_t b) { return F::mul(a, b); }
template <class F>
__device__ __attribute__((noinline)) fe_t sqr_ni(fe_t a) { return F::sqr(a); }

template <class F>
__device__ __attribute__((noinline)) f29_t mul29_ni(f29_t a, f29_t b) { return Fp29<typename F::Params>::mul(a, b); }
template <class F>
__device__ __attribute__((noinline)) f29_t sqr29_ni(f29_t a) { return Fp29<typename F::Params>::sqr(a); }

// column loads (PlonkEvalDomain::eval_column_var / eval_advice_var, src/plonk/eval.rs:57-69,153-228)
template <class F>
__device__ __forceinline__ fe_t ld_sel(const RowCtx &C, uint32_t col, uint32_t rr) { return C.sel[col][rr] ? F::one() : F::zero(); }
template <class F>
__device__ __forceinline__ fe_t ld_fix(const RowCtx &C, uint32_t col, uint32_t rr) { return C.fix[col][rr]; }
template <class F>
__device__ __forceinline__ fe_t ld_adv(const RowCtx &C, uint32_t col, uint32_t rr, uint32_t pt) {
    size_t idx = (size_t)col * C.rows + rr;
    if (C.wcoef == nullptr) {
        fe_t r = C.W[0][idx];
        if (C.J == 2 && C.half) {
            const fe_t w1 = C.W[1][idx];
            const fe_t a = F::halve(F::add(r, w1)), b = F::halve(F::sub(r, w1));
            return pt ? F::add(a, small_times<F>(b, pt)) : a;
        }
        if (C.J == 2 && pt) r = F::add(r, small_times<F>(C.W[1][idx], pt));
        return r;
    }
    const fe_t *cf = C.wcoef + (size_t)pt * C.J;
    fe_t r = F::mul(cf[0], C.W[0][idx]);
    for (uint32_t j = 1; j < C.J; ++j) r = F::add(r, F::mul(cf[j], C.W[j][idx]));
    return r;
}

// sweep form (rowprog.hip, emit_sweep_source): an advice leaf as an affine function of the evaluation point,
// value(pt) = cur + pt * step.  Only for the point sets without a coefficient table (C.wcoef == nullptr):
//   J == 1: the witness itself;  J == 2: W[0] + pt W[1] (cross terms);  J == 2, half: the Lagrange fold at integer points.
template <class F>
__device__ __forceinline__ void adv_affine(const RowCtx &C, uint32_t col, uint32_t rr, fe_t &cur, fe_t &step) {
    const size_t idx = (size_t)col * C.rows + rr;
    cur = C.W[0][idx];
    step = F::zero();
    if (C.J == 2) {
        const fe_t w1 = C.W[1][idx];
        if (C.half) {
            step = F::halve(F::sub(cur, w1));
            cur = F::halve(F::add(cur, w1));
        } else {
            step = w1;
        }
    }
    for (uint32_t i = 0; i < C.pt0; ++i) cur = F::add(cur, step);
}

constexpr uint32_t ROW_STRIPE_LOG = 10;   // == msm::STRIPE_LOG: rows and key entries are sharded alike

// thread index -> (row to evaluate, whether it exists).  Under sharding thread t is the t-th row of this rank's stripes.
__device__ __forceinline__ uint32_t shard_row(const RowCtx &C, uint32_t t, bool &live) {
    if (C.shard_world <= 1) {
        live = t < C.rows;
        return live ? t : C.rows - 1;
    }
    live = t < C.local_rows;
    if (!live) t = C.local_rows ? C.local_rows - 1 : 0;
    const uint32_t s = t >> ROW_STRIPE_LOG, o = t & ((1u << ROW_STRIPE_LOG) - 1);
    const uint32_t row = ((s * C.shard_world + C.shard_rank) << ROW_STRIPE_LOG) + o;
    if (row >= C.rows) { live = false; return C.rows - 1; }
    return row;
}

// Body of a straight-line ("specialised") row-program kernel: `eval(ctx, row, pt, U)` is the compiled program.
// The point values P(0..d) are parked in LDS (thread-private column, no barrier) and the inverse Vandermonde is
// applied after the last point: keeping the d accumulators T_k live across the straight-line program cost 48 VGPRs
// and pushed the kernel into scratch spills at 2 waves/SIMD.
template <class F, class Eval>
__device__ __forceinline__ void spec_kernel_body(const DevArgs &A, Eval eval) {
    __shared__ fe_t Pv[(DMAX + 1) * RP_THREADS];
    bool live;
    const uint32_t row = shard_row(A.ctx, blockIdx.x * RP_THREADS + threadIdx.x, live);
    for (uint32_t pt = 0; pt < A.npts; ++pt) {
        fe_t P = eval(A.ctx, row, pt, A.utab + (size_t)pt * A.n_uniform);
        if (A.d == 0) {
            if (live) A.out[0][row] = P;
        } else {
            Pv[pt * RP_THREADS + threadIdx.x] = P;
        }
    }
    if (A.d && live) {
        for (uint32_t k = 0; k < A.d; ++k) {
            fe_t T = F::zero();
            for (uint32_t pt = 0; pt < A.npts; ++pt) T = F::add(T, F::mul(A.vinv[k * A.npts + pt], Pv[pt * RP_THREADS + threadIdx.x]));
            A.out[k][row] = T;
        }
    }
}

// ---- sweep form: per-point accumulators of a thread in LDS, limb-planar: word `limb` of point `pt` lives at
// acc[(pt * 9 + limb) * RP_THREADS] (acc already offset by the thread index): conflict-free 4-byte lanes.  The values are LAZY
// 9 x 29-bit sums (ABI form, i.e. value * 2^256, plus a few multiples of p); sw_fold brings one below 2p again (product with
// 2^261 mod p, the radix' one), sw_finish turns it into the canonical 8 x 32 element.
constexpr uint32_t SW_WORDS = 9;
// The accumulators are DYNAMIC shared memory, sized by the launch for the points actually evaluated (sweep_smem_bytes): at
// 9 points they are 41.5 KB and only three workgroups fit a CU; the cross terms of the benchmark circuits have 6 - 7.
SRS_HD constexpr uint32_t sweep_smem_bytes(uint32_t npts) { return npts * SW_WORDS * RP_THREADS * 4u; }
__device__ __forceinline__ f29_t sw_load(const uint32_t *acc, uint32_t pt) {
    f29_t o;
#pragma unroll
    for (uint32_t i = 0; i < 9; ++i) o.v[i] = acc[(pt * SW_WORDS + i) * RP_THREADS];
    return o;
}
__device__ __forceinline__ void sw_store(uint32_t *acc, uint32_t pt, const f29_t &x) {
#pragma unroll
    for (uint32_t i = 0; i < 9; ++i) acc[(pt * SW_WORDS + i) * RP_THREADS] = x.v[i];
}
template <class F>
__device__ __forceinline__ void sw_fold(uint32_t *acc, uint32_t pt, const fe_t &one261) {
    using G = Fp29<typename F::Params>;
    sw_store(acc, pt, G::mul(sw_load(acc, pt), G::unpack(one261)));
}
template <class F>
__device__ __forceinline__ fe_t sw_finish(const uint32_t *acc, uint32_t pt, const fe_t &one261) {
    using G = Fp29<typename F::Params>;
    return G::to_canonical_fe(G::mul(sw_load(acc, pt), G::unpack(one261)));
}

// The kernel body for a program in sweep form: `sweep(ctx, row, npts, utab, nu, acc, accumulate)`, `one_idx` = index of 2^261 mod p
// in the uniform table.  The inverse Vandermonde step runs on the 9 x 29-bit multiplier as well: vinv29 = vinv * 2^5, so the
// product of an ABI-form point value with it is ABI form again; the d sums are reduced once each.
template <class F, class Sweep>
__device__ __forceinline__ void sweep_kernel_body(const DevArgs &A, uint32_t one_idx, Sweep sweep) {
    using G = Fp29<typename F::Params>;
    SRS_SWEEP_ACC(acc_all);
    uint32_t *acc = acc_all + threadIdx.x;
    bool live;
    const uint32_t row = shard_row(A.ctx, blockIdx.x * RP_THREADS + threadIdx.x, live);
    sweep(A.ctx, row, A.npts, A.utab, A.n_uniform, acc, false);
    if (!live) return;
    const fe_t one261 = A.utab[one_idx];
    if (A.d == 0) {
        A.out[0][row] = sw_finish<F>(acc, 0, one261);
        return;
    }
    for (uint32_t pt = 0; pt < A.npts; ++pt) sw_fold<F>(acc, pt, one261);          // < 2p: d products each below
    for (uint32_t k = 0; k < A.d; ++k) {
        f29_t T = G::mul(sw_load(acc, 0), G::unpack(A.vinv29[k * A.npts]));
        for (uint32_t pt = 1; pt < A.npts; ++pt)
            T = G::normalize(G::add_lazy(T, G::mul(sw_load(acc, pt), G::unpack(A.vinv29[k * A.npts + pt]))));   // <= 2 (d + 1) p <= 18 p
        A.out[k][row] = G::to_canonical_fe(G::mul(T, G::unpack(one261)));
    }
}

}  // namespace rowprog
}  // namespace srs
