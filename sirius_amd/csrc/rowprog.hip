// rowprog.hip -- per-row polynomial work of the Sangria NIFS on gfx950:
//   * cross terms  T_k[row], k = 1..d   (reference VanillaFS::commit_cross_terms, src/nifs/sangria/mod.rs:102-158)
//   * plain gate evaluation per row       (deciders, src/nifs/sangria/mod.rs:334-383, src/plonk/mod.rs:304-361)
//   * witness / error-vector folds        (RelaxedPlonkWitness::fold, src/nifs/sangria/accumulator.rs:364-404)
//
// What the reference computes: with P the compressed + homogenised gate polynomial of the structure
// (src/plonk/util.rs:34-56, src/polynomial/expression.rs:356-429, src/plonk/mod.rs:68-121),
//   T_k[row] = coefficient of X^k in  P(fixed[row], W1[row] + X*W2[row], ch1 + X*ch2)
// obtained symbolically (GroupedPoly::new, src/polynomial/grouped_poly.rs:88-138) and evaluated by a
// per-row interpreter, one pass over all columns PER TERM (src/polynomial/graph_evaluator.rs:361-388).
//
// MI355X-first formulation: the coefficients are mathematically determined, and field arithmetic
// is exact, so they are recovered numerically instead -- evaluate P at the d+1 points X = 0..d and
// apply the (constant) inverse Vandermonde matrix.  One pass over the rows yields ALL d terms;
// the program is the small homogeneous expression (no symbolic blow-up), compiled on the host to an
// SSA register program with CSE; sub-expressions that do not depend on the row (constants,
// challenges, powers of u) are evaluated once per point on the host ("uniform table").
// Row registers live in LDS as [slot][thread] (conflict-free 32-byte lanes); the d accumulators
// T_k live in VGPRs.  Columns are read coalesced (column-major, consecutive rows per lane).
#include "rowprog.h"
#include "ntt.h"
#include "prof.h"
#include "jit.h"
#include "tuning.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>

namespace srs {
namespace rowprog {

// device-side definitions (RowCtx, DevArgs, column loads, specialised-kernel body): rowprog_dev.cuh
// interpret one row program at evaluation point `pt`; registers = LDS slots [slot][thread]
template <class F>
__device__ __forceinline__ fe_t interp(fe_t *slots, const Insn *__restrict__ prog, uint32_t n_insn, uint32_t result,
                                       const RowCtx &C, uint32_t row, uint32_t pt, const fe_t *__restrict__ U) {
    const uint32_t tid = threadIdx.x, nthr = blockDim.x, mask = C.rows - 1;
    for (uint32_t ip = 0; ip < n_insn; ++ip) {
        const Insn in = prog[ip];
        fe_t r;
        if (in.op <= I_LD_ADV) {
            uint32_t rr = (row + (uint32_t)(int32_t)in.b) & mask;     // (row + rot) rem_euclid 2^k
            if (in.op == I_LD_SEL) r = ld_sel<F>(C, in.a, rr);
            else if (in.op == I_LD_FIX) r = ld_fix<F>(C, in.a, rr);
            else r = ld_adv<F>(C, in.a, rr, pt);
        } else {
            fe_t a = (in.a & UNIFORM_BIT) ? U[in.a & ~UNIFORM_BIT] : slots[in.a * nthr + tid];
            if (in.op <= I_MUL) {
                fe_t b = (in.b & UNIFORM_BIT) ? U[in.b & ~UNIFORM_BIT] : slots[in.b * nthr + tid];
                r = in.op == I_ADD ? F::add(a, b) : (in.op == I_SUB ? F::sub(a, b) : F::mul(a, b));
            } else if (in.op == I_SQR) {
                r = F::sqr(a);
            } else if (in.op == I_DBL) {
                r = F::dbl(a);
            } else {
                r = F::neg(a);
            }
        }
        slots[in.dst * nthr + tid] = r;
    }
    return (result & UNIFORM_BIT) ? U[result & ~UNIFORM_BIT] : slots[result * nthr + tid];
}

template <class F, uint32_t NSLOT>
__global__ void SRS_KERNEL_BOUNDS(RP_THREADS, 1) k_rowprog(DevArgs A) {
    __shared__ fe_t slots[NSLOT * RP_THREADS];
    bool live;
    const uint32_t row = shard_row(A.ctx, blockIdx.x * RP_THREADS + threadIdx.x, live);
    fe_t T[DMAX];
#pragma unroll
    for (uint32_t k = 0; k < DMAX; ++k) T[k] = F::zero();
    for (uint32_t pt = 0; pt < A.npts; ++pt) {
        fe_t P = interp<F>(slots, A.prog, A.n_insn, A.result, A.ctx, row, pt, A.utab + (size_t)pt * A.n_uniform);
        if (A.d == 0) {
            if (live) A.out[0][row] = P;
        } else {
#pragma unroll
            for (uint32_t k = 0; k < DMAX; ++k) {
                if (k < A.d) T[k] = F::add(T[k], F::mul(A.vinv[k * A.npts + pt], P));
            }
        }
    }
    if (A.d && live) {
#pragma unroll
        for (uint32_t k = 0; k < DMAX; ++k)
            if (k < A.d) A.out[k][row] = T[k];
    }
}

// ---------------------------------------------------------------------------------------------
// Specialised row programs.  The interpreter above is the general path (any circuit).  For the gate
// sets of the reference's own configurations (MainGate<5> + MainGate<3>, MainGate<5>: SURVEY.md H5)
// the SSA program is additionally emitted as straight-line C++ (emit_spec_source below), compiled
// ahead of time (tools/gen_rowprog_spec.py -> rowprog_spec.inc) and selected by program
// fingerprint: registers instead of LDS slots, no decode, loads scheduled by the compiler.
// ---------------------------------------------------------------------------------------------
#define SRS_SPEC_PART 1
#include "rowprog_spec.inc"
#undef SRS_SPEC_PART

template <class F, int ID>
__global__ void SRS_KERNEL_BOUNDS(RP_THREADS, 2) k_rowprog_spec(DevArgs A) {
    // sweep form: every column is read once for all points, bodies on the 9 x 29-bit multiplier.  The callers of these kernels
    // (cross terms, plain gate evaluation) never pass a coefficient table, and the specialised programs have d <= DMAX.
    sweep_kernel_body<F>(A, SpecCall<F, ID>::one(), [](const RowCtx &C, uint32_t row, uint32_t npts, const fe_t *U, uint32_t nu, uint32_t *acc, bool accumulate) {
        SpecCall<F, ID>::sweep(C, row, npts, U, nu, acc, accumulate);
    });
}

// ---------------------------------------------------------------------------------------------
// ProtoGalaxy: pow-weighted sums of gate evaluations (reference src/nifs/protogalaxy/poly/mod.rs)
//   Out[p] = sum_{i < n} pow_i(c^(p)) * f_i^(p),   pow_i(c) = prod_{b in bits(i)} c_b
//   f_i = gates[i / 2^k] evaluated at row(i)  (get_evaluate_witness_fn, src/plonk/mod.rs:683-718)
// compute_F : one leaf value, P weight vectors c^(p) = beta + X_p * delta          (:68-203)
// compute_G : P leaf values (witness folded with L_j(X_p) on the fly -- FoldedWitness is never
//             materialised), one weight vector beta'                                (:308-425)
// evaluate_e: P = 1                                                   (protogalaxy/mod.rs:571-640)
// The reference reduces with a binary tree where the right child is scaled by c_height; the same
// tree is evaluated here level by level (LDS inside a workgroup, then across workgroups).
// ---------------------------------------------------------------------------------------------
struct GateProg {
    const Insn *prog;
    uint32_t n_insn, result, n_uniform, utab_off;   // utab_off: offset (in fe) of this gate's table in PgArgs.utab
};
struct PgArgs {
    const GateProg *gates;
    uint32_t n_gates, log_rows;
    RowCtx ctx;
    int compat;               // Q1: row_index = index & 2^k  ==> every leaf is evaluated at row 0
    uint32_t leaf_pts;        // 1 (F, e) or P (G)
    uint32_t P;
    const fe_t *utab;         // per gate: [leaf_pts][n_uniform]
    const fe_t *weights;      // [levels][wpts]
    uint32_t wpts;            // P (F) or 1 (G, e)
    uint32_t tile_log;        // leaves per workgroup = 2^tile_log (<= 7)
    fe_t *partial;            // [n_gates * tiles_per_gate][P]
    uint32_t shard_rank, shard_world;   // process-per-GPU runs (set_shard): a 1024-leaf tile IS a key stripe (ROW_STRIPE_LOG), this rank
                              //   evaluates the tiles t % world == rank and leaves zeros for the others: its result is a PARTIAL sum
    // reference leaf rows (compat): every leaf of a gate is that gate AT ROW 0 -- one value per (gate, point) for the whole launch.  r03 shared it
    // among the 8 leaves of a thread; r04 hoists it out of the leaf pass: a one-workgroup launch (hoist_mode 1) evaluates the gates and leaves
    // hoist[gate * P + p], the leaf launch (hoist_mode 2) reads it -- every leaf still enters the tree with its own weight.  0: evaluate per thread.
    fe_t *hoist;
    int hoist_mode;
};
__device__ __forceinline__ bool pg_tile_is_mine(const PgArgs &A, uint32_t tile) {
    return A.shard_world <= 1 || tile % A.shard_world == A.shard_rank;
}
// a tile of another rank: zero partial sums (block-uniform exit before any barrier)
__device__ __forceinline__ void pg_zero_partial(const PgArgs &A, uint32_t gate, uint32_t tile) {
    for (uint32_t p = threadIdx.x; p < A.P; p += blockDim.x) A.partial[((size_t)gate * gridDim.x + tile) * A.P + p] = Fr::zero();
}

// binary-tree combine of blockDim.x values: v[2t] + v[2t+1] * w[level]; result in red[0]
template <class F>
__device__ __forceinline__ void weighted_tree(fe_t *red, fe_t x, const fe_t *__restrict__ w, uint32_t wstride, uint32_t levels) {
    const uint32_t tid = threadIdx.x;
    red[tid] = x;
    __syncthreads();
    for (uint32_t l = 0; l < levels; ++l) {
        uint32_t stride = 1u << l;
        if ((tid & (2 * stride - 1)) == 0) red[tid] = F::add(red[tid], F::mul(red[tid + stride], w[(size_t)l * wstride]));
        __syncthreads();
    }
}

// grid = (tiles_per_gate, n_gates), block = T = 2^(tile_log - log2 LPT) threads.  Thread t owns the LPT leaves
// base + l * T + t (l < LPT): consecutive lanes read consecutive rows, so every column load is coalesced.  The sum
// sum_i pow_i(c) f_i does not care about the order of additions (exact arithmetic), only that leaf i meets the weights
// of ITS index bits: the in-register combine over l therefore uses the weights of bits log2(T).., the LDS tree over
// the threads those of bits 0 .. log2(T) - 1.
template <class F, uint32_t NSLOT, uint32_t LPT>
__global__ void SRS_KERNEL_BOUNDS(RP_THREADS, 1) k_pg_leaves(PgArgs A) {
    __shared__ fe_t slots[NSLOT * RP_THREADS];
    __shared__ fe_t red[RP_THREADS];
    constexpr uint32_t LPT_LOG = LPT == 8 ? 3 : (LPT == 4 ? 2 : (LPT == 2 ? 1 : 0));
    const uint32_t gate = blockIdx.y, tile = blockIdx.x;
    const GateProg G = A.gates[gate];
    if (A.compat && A.hoist_mode == 1) {      // r06, as the ahead-of-time kernels: the reference's leaf rows make every leaf of a gate that gate AT ROW 0 --
        // the hoisting launch, grid = (leaf_pts, gates), evaluates it once per (gate, point); every lane computes the same value, lane 0 stores it
        const fe_t x = interp<F>(slots, G.prog, G.n_insn, G.result, A.ctx, 0u, tile, A.utab + G.utab_off + (size_t)tile * G.n_uniform);
        if (threadIdx.x == 0) A.hoist[(size_t)gate * A.leaf_pts + tile] = x;
        return;
    }
    if (!pg_tile_is_mine(A, tile)) { pg_zero_partial(A, gate, tile); return; }
    const uint32_t TL = A.tile_log - LPT_LOG;                          // log2(blockDim.x)
    const uint32_t row0 = (tile << A.tile_log) + threadIdx.x;          // < rows (rows is a multiple of the tile)
    fe_t v[LPT];
    for (uint32_t p = 0; p < A.P; ++p) {
        if (A.compat && A.hoist_mode == 2) {                           // the leaf launch reads the hoisted values
            if (A.leaf_pts > 1 || p == 0) {
                const fe_t x = A.hoist[(size_t)gate * A.leaf_pts + (A.leaf_pts > 1 ? p : 0)];
#pragma unroll
                for (uint32_t l = 0; l < LPT; ++l) v[l] = x;
            }
        } else if (A.leaf_pts > 1 || p == 0) {
#pragma unroll
            for (uint32_t l = 0; l < LPT; ++l)
                v[l] = interp<F>(slots, G.prog, G.n_insn, G.result, A.ctx, A.compat ? 0u : row0 + (l << TL), p,
                                 A.utab + G.utab_off + (size_t)p * G.n_uniform);
        }
        const fe_t *w = A.weights + (A.wpts > 1 ? p : 0);
        fe_t t[LPT];
#pragma unroll
        for (uint32_t l = 0; l < LPT; ++l) t[l] = v[l];
#pragma unroll
        for (uint32_t lvl = 0; lvl < LPT_LOG; ++lvl) {
            fe_t c = w[(size_t)(TL + lvl) * A.wpts];
#pragma unroll
            for (uint32_t i = 0; i < (LPT >> (lvl + 1)); ++i) t[i] = F::add(t[2 * i], F::mul(t[2 * i + 1], c));
        }
        weighted_tree<F>(red, t[0], w, A.wpts, TL);
        if (threadIdx.x == 0) A.partial[((size_t)gate * gridDim.x + tile) * A.P + p] = red[0];
        __syncthreads();
    }
}

// k_pg_leaves with the gate programs of a known gate set compiled ahead of time (rowprog_spec.inc, PgSpecCall):
// no LDS register file, no decode -- same leaf mapping and tree.
template <class F, int ID, uint32_t LPT>
__global__ void SRS_KERNEL_BOUNDS(RP_THREADS, 2) k_pg_leaves_spec(PgArgs A) {
    __shared__ fe_t red[RP_THREADS];
    constexpr uint32_t LPT_LOG = LPT == 8 ? 3 : (LPT == 4 ? 2 : (LPT == 2 ? 1 : 0));
    const uint32_t gate = blockIdx.y, tile = blockIdx.x;
    if (!pg_tile_is_mine(A, tile)) { pg_zero_partial(A, gate, tile); return; }
    const GateProg G = A.gates[gate];
    const uint32_t TL = A.tile_log - LPT_LOG;
    const uint32_t row0 = (tile << A.tile_log) + threadIdx.x;
    fe_t v[LPT];
    for (uint32_t p = 0; p < A.P; ++p) {
        if (A.leaf_pts > 1 || p == 0) {
#pragma unroll
            for (uint32_t l = 0; l < LPT; ++l)
                v[l] = PgSpecCall<F, ID>::run(gate, A.ctx, A.compat ? 0u : row0 + (l << TL), p,
                                              A.utab + G.utab_off + (size_t)p * G.n_uniform);
        }
        const fe_t *w = A.weights + (A.wpts > 1 ? p : 0);
        fe_t t[LPT];
#pragma unroll
        for (uint32_t l = 0; l < LPT; ++l) t[l] = v[l];
#pragma unroll
        for (uint32_t lvl = 0; lvl < LPT_LOG; ++lvl) {
            fe_t c = w[(size_t)(TL + lvl) * A.wpts];
#pragma unroll
            for (uint32_t i = 0; i < (LPT >> (lvl + 1)); ++i) t[i] = F::add(t[2 * i], F::mul(t[2 * i + 1], c));
        }
        weighted_tree<F>(red, t[0], w, A.wpts, TL);
        if (threadIdx.x == 0) A.partial[((size_t)gate * gridDim.x + tile) * A.P + p] = red[0];
        __syncthreads();
    }
}

// k_pg_leaves_spec in sweep form (compute_G with integer points, evaluate_e): every leaf row is swept ONCE for all P points
// (its columns are read once, the Lagrange fold advances by one addition per point).  The thread's LPT leaves need the weights
// w_l = prod_{b in bits(l)} c_(TL + b) of the leaf-index bits it owns: the host multiplies the term coefficients of the
// uniform table by w_l (one table per l, A.utab = [gate][l][P][nu]), so leaf l simply ACCUMULATES onto the same lazy 9 x 29-bit
// sums in LDS; between leaves the sums are folded below 2p.  Needs wpts == 1, P <= DMAX + 1 and affine advice leaves.
template <class F, int ID, uint32_t LPT, bool COMPAT>
__global__ void SRS_KERNEL_BOUNDS(RP_THREADS, 2) k_pg_leaves_sweep(PgArgs A) {
    __shared__ fe_t red[RP_THREADS];
    SRS_SWEEP_ACC(acc_all);                                 // sweep_smem_bytes(P) of dynamic LDS
    constexpr uint32_t LPT_LOG = LPT == 8 ? 3 : (LPT == 4 ? 2 : (LPT == 2 ? 1 : 0));
    const uint32_t gate = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    if (!(COMPAT && A.hoist_mode == 1) && !pg_tile_is_mine(A, tile)) { pg_zero_partial(A, gate, tile); return; }
    const GateProg G = A.gates[gate];
    const uint32_t TL = A.tile_log - LPT_LOG;
    const uint32_t row0 = (tile << A.tile_log) + tid;
    uint32_t *acc = acc_all + tid;
    const fe_t *U0 = A.utab + G.utab_off;
    const fe_t one261 = U0[PgSpecCall<F, ID>::one(gate)];
    if constexpr (COMPAT) {
        // the reference's leaf rows (`index & 2^k`, src/plonk/mod.rs:714): the LPT leaves of a thread all sit at row 0 of this gate,
        // so they share ONE gate evaluation f(X_p); sum_l w_l f = (sum_l w_l) f exactly -- table LPT holds the term coefficients
        // times the sum of the thread's leaf weights (pg_sum).  The value is the same for every thread of every tile: hoisted (PgArgs::hoist)
        if (A.hoist_mode == 2) {                                // the leaf launch: the hoisted values, then the trees
            for (uint32_t p = 0; p < A.P; ++p) {
                weighted_tree<F>(red, A.hoist[(size_t)gate * A.P + p], A.weights, A.wpts, TL);
                if (tid == 0) A.partial[((size_t)gate * gridDim.x + tile) * A.P + p] = red[0];
                __syncthreads();
            }
            return;
        }
        if (A.hoist_mode == 1) {                                // the hoisting launch, grid = (P, gates): workgroup p evaluates point p alone
            RowCtx c = A.ctx;                                   // (the points are independent: P short chains side by side instead of one long one)
            c.pt0 += tile;
            PgSpecCall<F, ID>::sweep(gate, c, 0u, 1u, U0 + ((size_t)LPT * A.P + tile) * G.n_uniform, G.n_uniform, acc, false);
            if (tid == 0) A.hoist[(size_t)gate * A.P + tile] = sw_finish<F>(acc, 0, one261);
            return;
        }
        PgSpecCall<F, ID>::sweep(gate, A.ctx, 0u, A.P, U0 + (size_t)LPT * A.P * G.n_uniform, G.n_uniform, acc, false);
    } else {
        for (uint32_t l = 0; l < LPT; ++l) {
            PgSpecCall<F, ID>::sweep(gate, A.ctx, row0 + (l << TL), A.P, U0 + (size_t)l * A.P * G.n_uniform, G.n_uniform, acc, l != 0);
            if (l + 1 < LPT)
                for (uint32_t p = 0; p < A.P; ++p) sw_fold<F>(acc, p, one261);
        }
    }
    for (uint32_t p = 0; p < A.P; ++p) {
        weighted_tree<F>(red, sw_finish<F>(acc, p, one261), A.weights, A.wpts, TL);
        if (tid == 0) A.partial[((size_t)gate * gridDim.x + tile) * A.P + p] = red[0];
        __syncthreads();
    }
}

// compute_F as a POLYNOMIAL tree (no evaluation points, no ifft).  F(X) = sum_i f_i prod_{b in bits(i)} (beta_b + X delta_b)
// has degree t = log2(#leaves); the reference evaluates it at next_pow2(t + 1) points (one weighted tree per point, t'n
// multiplies) and interpolates.  The same coefficients come out of ONE tree whose nodes are polynomials:
//   node = left + right * (beta_h + X delta_h)   ->   2 (deg + 1) multiplies per node, ~4 n in total instead of 32 n
// (exact arithmetic: identical coefficients, the ones above degree t are exactly zero in the reference's ifft too).
// k_pg_F_leaves: every thread evaluates its 8 leaves (same coalesced mapping as k_pg_leaves) and folds them over the
// three leaf-index bits it owns into a cubic; nodes are stored coefficient-major nodes[m * n_nodes + node].
struct PgFLevels {
    fe_t beta[3], delta[3];      // weights of leaf-index bits TL, TL + 1, TL + 2
};
template <class F, uint32_t NSLOT, int ID>
__global__ void SRS_KERNEL_BOUNDS(RP_THREADS, 1) k_pg_F_leaves(PgArgs A, PgFLevels Lv, fe_t *__restrict__ nodes, uint32_t n_nodes) {
    __shared__ fe_t slots[NSLOT * RP_THREADS];
    constexpr uint32_t LPT = 8, LPT_LOG = 3;
    const uint32_t gate = blockIdx.y, tile = blockIdx.x;
    if (!(A.compat && A.hoist_mode == 1) && !pg_tile_is_mine(A, tile)) {          // another rank's tile: zero cubic
        const uint32_t node = (gate * gridDim.x + tile) * blockDim.x + threadIdx.x;
        for (uint32_t m = 0; m < 4; ++m) nodes[(size_t)m * n_nodes + node] = F::zero();
        return;
    }
    const GateProg G = A.gates[gate];
    const uint32_t TL = A.tile_log - LPT_LOG;
    const uint32_t row0 = (tile << A.tile_log) + threadIdx.x;
    fe_t v[LPT];
#pragma unroll
    for (uint32_t l = 0; l < LPT; ++l) {
        if (A.compat && l) {                               // reference leaf rows (src/plonk/mod.rs:714): all LPT leaves are the gate at row 0
            v[l] = v[0];
            continue;
        }
        if (A.compat && A.hoist_mode == 2) {               // ... evaluated once per launch (PgArgs::hoist)
            v[0] = A.hoist[gate];
            continue;
        }
        const uint32_t row = A.compat ? 0u : row0 + (l << TL);
        if constexpr (ID >= 0) {                           // sweep form with one point: the 9 x 29-bit multiplier
            using PC = PgSpecCall<F, (ID >= 0 ? ID : 0)>;
            uint32_t *acc = reinterpret_cast<uint32_t *>(slots) + threadIdx.x;      // NSLOT >= 2: 9 words per thread fit
            PC::sweep(gate, A.ctx, row, 1, A.utab + G.utab_off, G.n_uniform, acc, false);
            v[l] = sw_finish<F>(acc, 0, (A.utab + G.utab_off)[PC::one(gate)]);
        } else {
            v[l] = interp<F>(slots, G.prog, G.n_insn, G.result, A.ctx, row, 0, A.utab + G.utab_off);
        }
        if (A.compat && A.hoist_mode == 1) {               // the one-workgroup launch: the gate at row 0, nothing else
            if (threadIdx.x == 0) A.hoist[gate] = v[0];
            return;
        }
    }
    fe_t c0[4], c1[4];                                     // bit TL: (v0 + v1 (beta + X delta))
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        c0[j] = F::add(v[2 * j], F::mul(v[2 * j + 1], Lv.beta[0]));
        c1[j] = F::mul(v[2 * j + 1], Lv.delta[0]);
    }
    fe_t e0[2], e1[2], e2[2];                              // bit TL + 1
#pragma unroll
    for (uint32_t j = 0; j < 2; ++j) {
        e0[j] = F::add(c0[2 * j], F::mul(c0[2 * j + 1], Lv.beta[1]));
        e1[j] = F::add(F::add(c1[2 * j], F::mul(c1[2 * j + 1], Lv.beta[1])), F::mul(c0[2 * j + 1], Lv.delta[1]));
        e2[j] = F::mul(c1[2 * j + 1], Lv.delta[1]);
    }
    const uint32_t node = (gate * gridDim.x + tile) * blockDim.x + threadIdx.x;      // bit TL + 2
    nodes[(size_t)0 * n_nodes + node] = F::add(e0[0], F::mul(e0[1], Lv.beta[2]));
    nodes[(size_t)1 * n_nodes + node] = F::add(F::add(e1[0], F::mul(e1[1], Lv.beta[2])), F::mul(e0[1], Lv.delta[2]));
    nodes[(size_t)2 * n_nodes + node] = F::add(F::add(e2[0], F::mul(e2[1], Lv.beta[2])), F::mul(e1[1], Lv.delta[2]));
    nodes[(size_t)3 * n_nodes + node] = F::mul(e2[1], Lv.delta[2]);
}

// one level of the polynomial tree: out[i] = in[2i] + in[2i+1] * (beta + X delta); nodes >= m_valid are zero.
// coefficient-major arrays: in[m * n_in + node] (degree deg_in), out[m * n_out + node] (degree deg_in + 1).
// One thread per (coefficient m, node i) -- consecutive threads = consecutive nodes of one coefficient (coalesced): the upper levels
// have few nodes, and a thread per NODE walking its deg + 2 coefficients in turn left them at 16-48 us of pure latency each (r02:
// 18 launches = 0.5 ms of a step); r03.
template <class F>
__global__ void k_pg_F_level(const fe_t *__restrict__ in, uint32_t n_in, uint32_t m_valid, uint32_t deg_in, fe_t beta, fe_t delta,
                             fe_t *__restrict__ out, uint32_t n_out) {
    const uint32_t lin = blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= n_out * (deg_in + 2)) return;
    const uint32_t m = lin / n_out, i = lin - m * n_out;
    const bool hasL = 2 * i < m_valid, hasR = 2 * i + 1 < m_valid;
    fe_t acc = F::zero();
    if (m <= deg_in) {
        if (hasL) acc = in[(size_t)m * n_in + 2 * i];
        if (hasR) acc = F::add(acc, F::mul(in[(size_t)m * n_in + 2 * i + 1], beta));
    }
    if (m >= 1 && hasR) acc = F::add(acc, F::mul(in[(size_t)(m - 1) * n_in + 2 * i + 1], delta));
    out[(size_t)m * n_out + i] = acc;
}

// the top of the tree in ONE workgroup: the last `nlev` <= PG_F_TAIL_LEVELS levels (n_in = 2^nlev <= 32 nodes) through LDS, the same
// (coefficient, node) threads; writes the final polynomial (degree deg_in + nlev) to out[0 .. deg_in + nlev]
constexpr uint32_t PG_F_TAIL_LEVELS = 5, PG_F_TAIL_MAXDEG = 40;
struct PgFTail {
    fe_t beta[PG_F_TAIL_LEVELS], delta[PG_F_TAIL_LEVELS];
};
template <class F>
__global__ void SRS_KERNEL_BOUNDS(512, 1)
    k_pg_F_tail(const fe_t *__restrict__ in, uint32_t n_in, uint32_t m_valid, uint32_t deg_in, uint32_t nlev, PgFTail T, fe_t *__restrict__ out) {
    __shared__ fe_t buf[2][(PG_F_TAIL_MAXDEG + 1) * 16];        // level outputs: <= 16 nodes x <= 41 coefficients
    const uint32_t t = threadIdx.x;
    uint32_t cur_n = n_in, deg = deg_in, valid = m_valid;
    for (uint32_t lv = 0; lv < nlev; ++lv) {
        const uint32_t n_out = cur_n >> 1;
        const fe_t *src = lv == 0 ? in : buf[(lv - 1) & 1];
        fe_t *dst = buf[lv & 1];
        for (uint32_t lin = t; lin < n_out * (deg + 2); lin += blockDim.x) {
            const uint32_t m = lin / n_out, i = lin - m * n_out;
            const bool hasL = 2 * i < valid, hasR = 2 * i + 1 < valid;
            fe_t acc = F::zero();
            if (m <= deg) {
                if (hasL) acc = src[(size_t)m * cur_n + 2 * i];
                if (hasR) acc = F::add(acc, F::mul(src[(size_t)m * cur_n + 2 * i + 1], T.beta[lv]));
            }
            if (m >= 1 && hasR) acc = F::add(acc, F::mul(src[(size_t)(m - 1) * cur_n + 2 * i + 1], T.delta[lv]));
            dst[(size_t)m * n_out + i] = acc;
        }
        __syncthreads();
        cur_n = n_out;
        ++deg;
        valid = (valid + 1) >> 1;
    }
    for (uint32_t m = t; m <= deg; m += blockDim.x) out[m] = buf[(nlev - 1) & 1][m];
}

// r05: SEVERAL levels of the polynomial tree per launch, for ANY number of nodes: workgroup g takes the 2^nlev consecutive input nodes
// [g 2^nlev, (g + 1) 2^nlev) through nlev <= PG_F_MULTI_LEVELS levels in LDS (the same (coefficient, node) threads as k_pg_F_level) and writes
// ONE output node of degree deg_in + nlev.  compute_F's 18 levels above the leaf cubics (k = 20) are 3 launches instead of 13 + 1
// (each ~7 us of kernel + a launch gap on a chain nothing overlaps).   grid = n_in >> nlev, block = 256
constexpr uint32_t PG_F_MULTI_LEVELS = 6, PG_F_MULTI_NODES = 1u << (PG_F_MULTI_LEVELS - 1);
struct PgFMulti {
    fe_t beta[PG_F_MULTI_LEVELS], delta[PG_F_MULTI_LEVELS];
};
template <class F>
__global__ void SRS_KERNEL_BOUNDS(256, 1)
    k_pg_F_multi(const fe_t *__restrict__ in, uint32_t n_in, uint32_t m_valid, uint32_t deg_in, uint32_t nlev, PgFMulti T, fe_t *__restrict__ out,
                 uint32_t n_out_total) {
    // level outputs, two buffers of 32 x (deg_in + nlev + 1) elements each: DYNAMIC LDS sized by the launch -- the first launch of a k = 20
    // compute_F (4096 workgroups, degree 3 -> 9) needs 20 KB and shares a CU eight ways; with the worst-case 84 KB it ran one workgroup per
    // CU, 16 rounds: 219 us (profiles/r05_ab_misc.txt)
    SRS_DYN_LDS(fe_t, dyn, 2 * (PG_F_TAIL_MAXDEG + 1) * PG_F_MULTI_NODES);
    const uint32_t bstride = PG_F_MULTI_NODES * (deg_in + nlev + 1);
    fe_t *const buf[2] = {dyn, dyn + bstride};
    const uint32_t t = threadIdx.x, g = blockIdx.x;
    const uint32_t first = g << nlev;                                      // this workgroup's first input node
    uint32_t cur_n = 1u << nlev, deg = deg_in;
    uint32_t valid = m_valid > first ? m_valid - first : 0u;               // its input nodes that are not padding
    if (valid > cur_n) valid = cur_n;
    for (uint32_t lv = 0; lv < nlev; ++lv) {
        const uint32_t n_out = cur_n >> 1;
        const fe_t *src = lv == 0 ? in + first : buf[(lv - 1) & 1];
        const size_t src_stride = lv == 0 ? (size_t)n_in : (size_t)cur_n;  // coefficient-major: src[m * stride + node]
        fe_t *dst = buf[lv & 1];
        for (uint32_t lin = t; lin < n_out * (deg + 2); lin += blockDim.x) {
            const uint32_t m = lin / n_out, i = lin - m * n_out;
            const bool hasL = 2 * i < valid, hasR = 2 * i + 1 < valid;
            fe_t acc = F::zero();
            if (m <= deg) {
                if (hasL) acc = src[(size_t)m * src_stride + 2 * i];
                if (hasR) acc = F::add(acc, F::mul(src[(size_t)m * src_stride + 2 * i + 1], T.beta[lv]));
            }
            if (m >= 1 && hasR) acc = F::add(acc, F::mul(src[(size_t)(m - 1) * src_stride + 2 * i + 1], T.delta[lv]));
            dst[(size_t)m * n_out + i] = acc;
        }
        __syncthreads();
        cur_n = n_out;
        ++deg;
        valid = (valid + 1) >> 1;
    }
    for (uint32_t m = t; m <= deg; m += blockDim.x) out[(size_t)m * n_out_total + g] = buf[(nlev - 1) & 1][m];
}

#define SRS_SPEC_PART 2
#include "rowprog_spec.inc"
#undef SRS_SPEC_PART

// continues the tree over the per-tile partials: in[m_valid][P] (zero beyond m_valid) -> out[ceil(m / 2^lv)][P]
// grid = number of output nodes, block = 2^lv threads
template <class F>
__global__ void SRS_KERNEL_BOUNDS(RP_THREADS, 1)
    k_pg_reduce(const fe_t *__restrict__ in, uint32_t m_valid, uint32_t P, const fe_t *__restrict__ weights, uint32_t wpts,
                uint32_t level0, uint32_t lv, fe_t *__restrict__ out) {
    __shared__ fe_t red[RP_THREADS];
    const uint32_t node = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t p = 0; p < P; ++p) {
        fe_t x = node < m_valid ? in[(size_t)node * P + p] : F::zero();
        weighted_tree<F>(red, x, weights + (size_t)level0 * wpts + (wpts > 1 ? p : 0), wpts, lv);
        if (threadIdx.x == 0) out[(size_t)blockIdx.x * P + p] = red[0];
        __syncthreads();
    }
}

// r05: the same reduction with the P evaluation points SIDE BY SIDE: thread (p, t) of a workgroup of P * 2^lv <= 1024 threads runs point p's
// tree (k_pg_reduce walks the points one after the other: P x lv dependent multiply-add-barrier rounds on a chain nothing overlaps --
// 64 + 26 us per compute_G at P = 5).   grid = number of output nodes, block = (2^lv, P)
constexpr uint32_t PG_REDUCE_PAR_THREADS = 1024;
template <class F>
__global__ void SRS_KERNEL_BOUNDS(PG_REDUCE_PAR_THREADS, 1)
    k_pg_reduce_par(const fe_t *__restrict__ in, uint32_t m_valid, uint32_t P, const fe_t *__restrict__ weights, uint32_t wpts,
                    uint32_t level0, uint32_t lv, fe_t *__restrict__ out) {
    __shared__ fe_t red[PG_REDUCE_PAR_THREADS];
    const uint32_t t = threadIdx.x, p = threadIdx.y, width = blockDim.x;
    const uint32_t node = blockIdx.x * width + t;
    fe_t *r = red + (size_t)p * width;
    const fe_t *w = weights + (size_t)level0 * wpts + (wpts > 1 ? p : 0);
    r[t] = node < m_valid ? in[(size_t)node * P + p] : F::zero();
    __syncthreads();
    for (uint32_t l = 0; l < lv; ++l) {
        const uint32_t stride = 1u << l;
        if ((t & (2 * stride - 1)) == 0) r[t] = F::add(r[t], F::mul(r[t + stride], w[(size_t)l * wpts]));
        __syncthreads();
    }
    if (t == 0) out[(size_t)blockIdx.x * P + p] = r[0];
}

// K(X) on the coset (compute_K_from_G, poly/mod.rs:475-509): thread i: X = ZETA * w^i,
//   K(X) = (G(X) - F(alpha) * L0(X)) / Z(X),  L0(X) = (1/n)(X^n - 1)/(X - 1),  Z(X) = X^n - 1,  n = instances_to_fold
__global__ void k_pg_K_points(const fe_t *__restrict__ polyG, uint32_t nG, fe_t f_alpha, fe_t zeta, fe_t omega, fe_t inv_n,
                              uint32_t n_fold, uint32_t count, fe_t *__restrict__ out, int *__restrict__ err) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    fe_t X = Fr::mul(zeta, Fr::pow_u64(omega, i));
    fe_t g = Fr::zero(), xp = Fr::one();
    for (uint32_t k = 0; k < nG; ++k) {            // UnivariatePoly::eval (univariate.rs:67-75)
        g = Fr::add(g, Fr::mul(xp, polyG[k]));
        xp = Fr::mul(xp, X);
    }
    fe_t xn1 = Fr::sub(Fr::pow_u64(X, n_fold), Fr::one());
    fe_t xm1 = Fr::sub(X, Fr::one());
    fe_t l0;
    if (Fr::is_zero(xn1) && Fr::is_zero(xm1)) l0 = Fr::one();   // lagrange.rs:67-69
    else l0 = Fr::mul(inv_n, Fr::mul(xn1, Fr::inv(xm1)));
    if (Fr::is_zero(xn1)) { *err = 1; return; }                 // "Z(X) must be not equal to 0"
    out[i] = Fr::mul(Fr::sub(g, Fr::mul(f_alpha, l0)), Fr::inv(xn1));
}

constexpr uint32_t PG_K_SMALL_MAX_NODES = 32, PG_K_SMALL_MAX_LOG = 12;
// r06: the same K points for a SMALL domain straight from compute_G's values at integer nodes (one incoming trace): every workgroup
// interpolates G (coefficients c = M v, M = the constant inverse Vandermonde matrix of the nodes, v = the values, v[0] = g0 when
// `has_g0`: G(1) = F(alpha)), then thread i evaluates G at X_i by Horner and applies the domain's precomputed L_0(X_i) and 1 / Z(X_i)
// (tab = [xs | l0 | inv_z], count entries each).  Exact field arithmetic: the host route's values bit for bit.
__global__ void k_pg_K_small(const fe_t *__restrict__ vals, uint32_t n_vals, fe_t g0, int has_g0, const fe_t *__restrict__ M, uint32_t n_nodes,
                             fe_t f_alpha, const fe_t *__restrict__ tab, uint32_t count, fe_t *__restrict__ out) {
    __shared__ fe_t c[PG_K_SMALL_MAX_NODES];
    if (threadIdx.x < n_nodes) {
        fe_t acc = Fr::zero();
        for (uint32_t j = 0; j < n_nodes; ++j) {
            const fe_t v = has_g0 ? (j == 0 ? g0 : vals[j - 1]) : vals[j];
            acc = Fr::add(acc, Fr::mul(M[threadIdx.x * n_nodes + j], v));
        }
        c[threadIdx.x] = acc;
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const fe_t X = tab[i];
    fe_t g = Fr::zero();
    for (uint32_t k = n_nodes; k-- > 0;) g = Fr::add(Fr::mul(g, X), c[k]);
    out[i] = Fr::mul(Fr::sub(g, Fr::mul(f_alpha, tab[count + i])), tab[2 * count + i]);
}

// number of rows with a[i] != b[i] (b == nullptr: a[i] != 0): the deciders' mismatch count
// (PlonkStructure::is_sat src/plonk/mod.rs:329-346; is_sat_accumulation src/nifs/sangria/mod.rs:352-376)
__global__ void k_count_mismatch(const fe_t *__restrict__ a, const fe_t *__restrict__ b, size_t n, uint32_t *__restrict__ count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_t x = a[i];
    bool bad = b ? !Fr::eq(x, b[i]) : !Fr::is_zero(x);
    if (bad) atomicAdd(count, 1u);
}

// ---------------------------------------------------------------------------------------------
// lookup arguments (log-derivative), src/plonk/lookup.rs
// ---------------------------------------------------------------------------------------------
// evaluate_m (lookup.rs:275-303): m[i] = #{ j : l[j] == t[i] } for the FIRST row i holding a given table value,
// 0 for its repeats.  The reference builds a HashMap over l and walks t serially; here t is inserted into an
// open-addressing table (slot = row index of the smallest row with that value, resolved with atomicMin so the
// result does not depend on scheduling), l is counted into the slots, and every row of t reads its slot back.
// Values compare by their Montgomery limbs (a bijection of the canonical form `to_repr` the reference hashes).
constexpr uint32_t M_EMPTY = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t fe_hash(const fe_t &x) {
    uint32_t h = 0x9E3779B9u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h ^= x.v[i];
        h *= 0x85EBCA6Bu;
        h ^= h >> 15;
    }
    return h;
}
__global__ void k_m_insert(const fe_t *__restrict__ t, uint32_t n, uint32_t *__restrict__ slot_row, uint32_t mask) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe_t key = t[i];
    uint32_t s = fe_hash(key) & mask;
    for (;;) {
        uint32_t cur = atomicCAS(&slot_row[s], M_EMPTY, i);
        if (cur == M_EMPTY) return;
        if (Fr::eq(t[cur], key)) { atomicMin(&slot_row[s], i); return; }
        s = (s + 1) & mask;
    }
}
__global__ void k_m_count(const fe_t *__restrict__ l, uint32_t n, const fe_t *__restrict__ t, const uint32_t *__restrict__ slot_row,
                          uint32_t *__restrict__ slot_cnt, uint32_t mask) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const fe_t key = l[j];
    uint32_t s = fe_hash(key) & mask;
    for (;;) {
        uint32_t cur = slot_row[s];
        if (cur == M_EMPTY) return;                       // value not in the table
        if (Fr::eq(t[cur], key)) { atomicAdd(&slot_cnt[s], 1u); return; }
        s = (s + 1) & mask;
    }
}
template <class F>
__global__ void k_m_emit(const fe_t *__restrict__ t, uint32_t n, const uint32_t *__restrict__ slot_row,
                         const uint32_t *__restrict__ slot_cnt, uint32_t mask, fe_t *__restrict__ m) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe_t key = t[i];
    uint32_t s = fe_hash(key) & mask;
    for (;;) {
        uint32_t cur = slot_row[s];
        if (cur == i || Fr::eq(t[cur], key)) break;     // every row of t was inserted: the probe terminates
        s = (s + 1) & mask;
    }
    uint32_t c = slot_row[s] == i ? slot_cnt[s] : 0u;
    m[i] = c ? F::from_u64(c) : F::zero();               // F::from_u128(count), lookup.rs:295-300
}

// evaluate_h_g (lookup.rs:305-317): h = 1 / (l + r), g = m / (t + r), with 1/0 := 0.
// One field inversion per HG_CHUNK elements (Montgomery's trick); elements of a chunk are blockDim apart, so
// every load / store is coalesced.  blockIdx.y: 0 -> h, 1 -> g.
constexpr uint32_t HG_THREADS = 128, HG_CHUNK = 8;
template <class F>
__global__ void SRS_KERNEL_BOUNDS(HG_THREADS, 1)
k_lookup_hg(const fe_t *__restrict__ l, const fe_t *__restrict__ t, const fe_t *__restrict__ m, fe_t r, uint32_t n,
            fe_t *__restrict__ h, fe_t *__restrict__ g) {
    const bool is_g = blockIdx.y == 1;
    const fe_t *__restrict__ src = is_g ? t : l;
    fe_t *__restrict__ dst = is_g ? g : h;
    const uint32_t base = blockIdx.x * HG_THREADS * HG_CHUNK + threadIdx.x;
    fe_t v[HG_CHUNK], pref[HG_CHUNK];
    uint32_t zero_mask = 0;
    fe_t acc = F::one();
#pragma unroll
    for (uint32_t j = 0; j < HG_CHUNK; ++j) {
        uint32_t idx = base + j * HG_THREADS;
        fe_t x = idx < n ? F::add(src[idx], r) : F::one();
        if (F::is_zero(x)) { zero_mask |= 1u << j; x = F::one(); }
        v[j] = x;
        pref[j] = acc;
        acc = F::mul(acc, x);
    }
    fe_t inv = F::inv(acc);
#pragma unroll
    for (int j = (int)HG_CHUNK - 1; j >= 0; --j) {
        uint32_t idx = base + (uint32_t)j * HG_THREADS;
        fe_t o = F::mul(inv, pref[j]);
        inv = F::mul(inv, v[j]);
        if (idx < n) {
            if ((zero_mask >> j) & 1u) o = F::zero();
            else if (is_g) o = F::mul(o, m[idx]);
            dst[idx] = o;
        }
    }
}

// batch_invert_assigned (src/util/mod.rs:119-153): out[i] = num[i] * (has_den[i] ? den[i]^-1 : 1), with 0^-1 := 0
// (ff::BatchInvert leaves zero denominators at zero).  Same chunked Montgomery trick and access pattern as k_lookup_hg.
template <class F>
__global__ void SRS_KERNEL_BOUNDS(HG_THREADS, 1)
k_assigned_invert(const fe_t *__restrict__ num, const fe_t *__restrict__ den, const uint8_t *__restrict__ has_den, uint32_t n,
                  fe_t *__restrict__ out) {
    const uint32_t base = blockIdx.x * HG_THREADS * HG_CHUNK + threadIdx.x;
    fe_t v[HG_CHUNK], pref[HG_CHUNK];
    uint32_t zero_mask = 0, triv_mask = 0;
    fe_t acc = F::one();
#pragma unroll
    for (uint32_t j = 0; j < HG_CHUNK; ++j) {
        uint32_t idx = base + j * HG_THREADS;
        fe_t x = F::one();
        if (idx < n && (!has_den || has_den[idx])) {
            x = den[idx];
            if (F::is_zero(x)) { zero_mask |= 1u << j; x = F::one(); }
        } else {
            triv_mask |= 1u << j;
        }
        v[j] = x;
        pref[j] = acc;
        acc = F::mul(acc, x);
    }
    fe_t inv = F::inv(acc);
#pragma unroll
    for (int j = (int)HG_CHUNK - 1; j >= 0; --j) {
        uint32_t idx = base + (uint32_t)j * HG_THREADS;
        fe_t o = F::mul(inv, pref[j]);
        inv = F::mul(inv, v[j]);
        if (idx < n) {
            fe_t a = num[idx];
            if ((zero_mask >> j) & 1u) a = F::zero();
            else if (!((triv_mask >> j) & 1u)) a = F::mul(a, o);
            out[idx] = a;
        }
    }
}

// partial sums of a[i] - b[i] (is_sat_log_derivative, src/plonk/mod.rs:366-378): one value per workgroup
template <class F>
__global__ void k_sum_diff(const fe_t *__restrict__ a, const fe_t *__restrict__ b, uint32_t n, fe_t *__restrict__ partial) {
    __shared__ fe_t red[256];
    fe_t acc = F::zero();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc = F::add(acc, F::sub(a[i], b[i]));
    red[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = F::add(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// out[i] = sum_j coef[j] * W[j][i]   (ProtoGalaxy::fold_witness, protogalaxy/mod.rs:176-210)
struct LincombArgs {
    const fe_t *w[JMAX];
    fe_t coef[JMAX];
    uint32_t J;
};
// AFFINE: the coefficients sum to one (Lagrange values, the only caller on the hot path: ProtoGalaxy::fold_witness), so
// sum_j c_j w_j = w_0 + sum_{j>=1} c_j (w_j - w_0): one product less per element, the same field element.
// world > 1 (process-per-GPU ranks): n counts the elements of THIS rank's block-cyclic stripes; only those are folded.
template <class F, bool AFFINE>
__global__ void k_lincomb(fe_t *__restrict__ out, LincombArgs a, size_t n, uint32_t rank, uint32_t world) {
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; l < n; l += stride) {
        const size_t i = world <= 1 ? l : ((((l >> ROW_STRIPE_LOG) * world + rank) << ROW_STRIPE_LOG) + (l & ((1u << ROW_STRIPE_LOG) - 1)));
        fe_t acc;
        if (AFFINE) {
            const fe_t w0 = a.w[0][i];
            acc = w0;
            for (uint32_t j = 1; j < a.J; ++j) acc = F::add(acc, F::mul(a.coef[j], F::sub(a.w[j][i], w0)));
        } else {
            acc = F::mul(a.coef[0], a.w[0][i]);
            for (uint32_t j = 1; j < a.J; ++j) acc = F::add(acc, F::mul(a.coef[j], a.w[j][i]));
        }
        out[i] = acc;
    }
}

// the same sum on listed rows of every column (the halo rows of a sharded fold): element c * col_len + rows[i]
template <class F>
__global__ void k_lincomb_rows(fe_t *__restrict__ out, LincombArgs a, const uint32_t *__restrict__ rows, size_t n_rows, size_t cols, size_t col_len) {
    size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n_rows * cols) return;
    const size_t i = (l / n_rows) * col_len + rows[l % n_rows];
    fe_t acc = F::mul(a.coef[0], a.w[0][i]);
    for (uint32_t j = 1; j < a.J; ++j) acc = F::add(acc, F::mul(a.coef[j], a.w[j][i]));
    out[i] = acc;
}

// ---- folds ----
template <class F>
__global__ void k_fold_w(fe_t *__restrict__ out, const fe_t *__restrict__ w1, const fe_t *__restrict__ w2, fe_t r, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = F::add(w1[i], F::mul(r, w2[i]));
}
struct FoldEArgs {
    const fe_t *t[DMAX];
    fe_t rpow[DMAX];
    uint32_t n_terms;
};
template <class F>
__global__ void k_fold_e(fe_t *out, const fe_t *e, FoldEArgs fa, size_t n) {   // out may alias e
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        fe_t acc = e[i];
#pragma unroll
        for (uint32_t k = 0; k < DMAX; ++k)
            if (k < fa.n_terms) acc = F::add(acc, F::mul(fa.rpow[k], fa.t[k][i]));
        out[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// host: Expression AST (reference src/polynomial/expression.rs:112-120)
// ---------------------------------------------------------------------------------------------
enum NodeKind { N_CONST, N_POLY, N_CHAL, N_NEG, N_SUM, N_PROD, N_SCALED };
struct Node {
    int kind;
    fe_t c;           // N_CONST / N_SCALED
    int64_t index;    // N_POLY / N_CHAL
    int32_t rot;      // N_POLY
    int a, b;
};
struct Ast {
    std::vector<Node> n;
    int add(int kind, int a = -1, int b = -1, int64_t index = 0, int32_t rot = 0, const fe_t *c = nullptr) {
        Node x;
        x.kind = kind;
        x.a = a;
        x.b = b;
        x.index = index;
        x.rot = rot;
        std::memset(&x.c, 0, sizeof(x.c));
        if (c) x.c = *c;
        n.push_back(x);
        return (int)n.size() - 1;
    }
};

template <class F>
struct HostField {
    static fe_t add(const fe_t &a, const fe_t &b) { return F::add(a, b); }
};

struct FieldOps {   // runtime-dispatched host field arithmetic
    int field;
    fe_t zero() const { return Fr::zero(); }
    fe_t one() const { return field == 0 ? Fr::one() : Fq::one(); }
    fe_t add(const fe_t &a, const fe_t &b) const { return field == 0 ? Fr::add(a, b) : Fq::add(a, b); }
    fe_t sub(const fe_t &a, const fe_t &b) const { return field == 0 ? Fr::sub(a, b) : Fq::sub(a, b); }
    fe_t mul(const fe_t &a, const fe_t &b) const { return field == 0 ? Fr::mul(a, b) : Fq::mul(a, b); }
    fe_t neg(const fe_t &a) const { return field == 0 ? Fr::neg(a) : Fq::neg(a); }
    fe_t halve(const fe_t &a) const { return field == 0 ? Fr::halve(a) : Fq::halve(a); }
    fe_t inv(const fe_t &a) const { return field == 0 ? Fr::inv(a) : Fq::inv(a); }
    fe_t from_u64(uint64_t v) const { return field == 0 ? Fr::from_u64(v) : Fq::from_u64(v); }
    bool is_zero(const fe_t &a) const { return Fr::is_zero(a); }
    bool eq(const fe_t &a, const fe_t &b) const { return Fr::eq(a, b); }
};

// gate stream: postfix words, see include/sirius_amd.h (SRS_EX_*)
static bool parse_gates(const uint64_t *w, size_t nw, size_t num_gates, Ast &ast, std::vector<int> &roots, std::string &err) {
    std::vector<int> st;
    size_t i = 0;
    while (i < nw) {
        uint64_t op = w[i++];
        switch (op) {
        case 0: {   // CONST c[4]
            if (i + 4 > nw) { err = "truncated constant"; return false; }
            fe_t c;
            std::memcpy(&c, &w[i], 32);
            i += 4;
            st.push_back(ast.add(N_CONST, -1, -1, 0, 0, &c));
            break;
        }
        case 1: {   // POLY index rot
            if (i + 2 > nw) { err = "truncated query"; return false; }
            st.push_back(ast.add(N_POLY, -1, -1, (int64_t)w[i], (int32_t)(int64_t)w[i + 1]));
            i += 2;
            break;
        }
        case 2:
            if (i + 1 > nw) { err = "truncated challenge"; return false; }
            st.push_back(ast.add(N_CHAL, -1, -1, (int64_t)w[i]));
            i += 1;
            break;
        case 3:
            if (st.empty()) { err = "stack underflow"; return false; }
            st.back() = ast.add(N_NEG, st.back());
            break;
        case 4:
        case 5: {
            if (st.size() < 2) { err = "stack underflow"; return false; }
            int b = st.back();
            st.pop_back();
            int a = st.back();
            st.back() = ast.add(op == 4 ? N_SUM : N_PROD, a, b);
            break;
        }
        case 6: {
            if (st.empty() || i + 4 > nw) { err = "bad scaled"; return false; }
            fe_t c;
            std::memcpy(&c, &w[i], 32);
            i += 4;
            st.back() = ast.add(N_SCALED, st.back(), -1, 0, 0, &c);
            break;
        }
        case 7:
            if (st.size() != 1) { err = "gate expression does not reduce to one value"; return false; }
            roots.push_back(st.back());
            st.clear();
            break;
        default:
            err = "unknown expression opcode";
            return false;
        }
    }
    if (!st.empty() || roots.size() != num_gates) { err = "gate count mismatch"; return false; }
    return true;
}

static void collect_challenges(const Ast &ast, int r, std::vector<int64_t> &set) {
    const Node &x = ast.n[r];
    if (x.kind == N_CHAL) {
        if (std::find(set.begin(), set.end(), x.index) == set.end()) set.push_back(x.index);
    }
    if (x.a >= 0) collect_challenges(ast, x.a, set);
    if (x.b >= 0) collect_challenges(ast, x.b, set);
}
static size_t num_challenges(const Ast &ast, int r) {   // Expression::num_challenges, expression.rs:163-167
    std::vector<int64_t> set;
    collect_challenges(ast, r, set);
    return set.size();
}

// compress_expression (src/plonk/util.rs:34-56)
static int compress(Ast &ast, const std::vector<int> &gates, size_t challenge_index, const FieldOps &f) {
    fe_t z = f.zero();
    if (gates.size() > 1) {
        int acc = ast.add(N_CONST, -1, -1, 0, 0, &z);
        for (int g : gates) {
            int y = ast.add(N_CHAL, -1, -1, (int64_t)challenge_index);
            acc = ast.add(N_SUM, g, ast.add(N_PROD, acc, y));
        }
        return acc;
    }
    if (gates.size() == 1) return gates[0];
    return ast.add(N_CONST, -1, -1, 0, 0, &z);
}

static int challenge_in_degree(Ast &ast, size_t idx, size_t degree) {   // expression.rs:501-513
    int r = ast.add(N_CHAL, -1, -1, (int64_t)idx);
    for (size_t i = 2; i <= degree; ++i) r = ast.add(N_PROD, r, ast.add(N_CHAL, -1, -1, (int64_t)idx));
    return r;
}

// Expression::homogeneous (src/polynomial/expression.rs:356-429)
struct Ctx {
    size_t num_selectors, num_fixed, num_advice, num_challenges;
    size_t num_lookups = 0;     // each adds the 5 fold variables (l, t, m, h, g) after the advice columns
    size_t num_fold_vars() const { return num_advice + 5 * num_lookups; }   // expression.rs:61-63
    // Column of fold variable j inside the CONCATENATED witness W[0] || W[1] (|| W[2]).  This is index_map of
    // PlonkEvalDomain::eval_advice_var (src/plonk/eval.rs:169-201) composed with the round sizes of
    // ConstraintSystemMetainfo::build (constraint_system_metainfo.rs:58-79): in both the 2-round and the
    // 3-round layout (l,t,m) of lookup li sits at columns num_advice + 3 li + {0,1,2} and (h,g) at
    // num_advice + 3 L + 2 li + {0,1}.
    size_t witness_col(size_t j) const {
        if (j < num_advice) return j;
        size_t li = (j - num_advice) / 5, sub = (j - num_advice) % 5;
        return sub < 3 ? num_advice + li * 3 + sub : num_advice + 3 * num_lookups + li * 2 + (sub - 3);
    }
};
static bool homogeneous(Ast &ast, int r, const Ctx &ctx, int &out, size_t &degree, std::string &err) {
    const Node x = ast.n[r];
    switch (x.kind) {
    case N_CONST: out = r; degree = 0; return true;
    case N_POLY: {
        size_t i = (size_t)x.index;
        if (i < ctx.num_selectors + ctx.num_fixed) degree = 0;
        else if (i < ctx.num_selectors + ctx.num_fixed + ctx.num_fold_vars()) degree = 1;   // Advice | Lookup
        else { err = "unknown query index " + std::to_string(i); return false; }
        out = r;
        return true;
    }
    case N_CHAL: out = r; degree = 1; return true;
    case N_NEG: {
        int a; if (!homogeneous(ast, x.a, ctx, a, degree, err)) return false;
        out = ast.add(N_NEG, a);
        return true;
    }
    case N_SCALED: {
        int a; if (!homogeneous(ast, x.a, ctx, a, degree, err)) return false;
        out = ast.add(N_SCALED, a, -1, 0, 0, &x.c);
        return true;
    }
    case N_PROD: {
        int a, b; size_t da, db;
        if (!homogeneous(ast, x.a, ctx, a, da, err) || !homogeneous(ast, x.b, ctx, b, db, err)) return false;
        out = ast.add(N_PROD, a, b);
        degree = da + db;
        return true;
    }
    default: {   // N_SUM
        int a, b; size_t da, db;
        if (!homogeneous(ast, x.a, ctx, a, da, err) || !homogeneous(ast, x.b, ctx, b, db, err)) return false;
        if (da > db) {
            out = ast.add(N_SUM, a, ast.add(N_PROD, b, challenge_in_degree(ast, ctx.num_challenges, da - db)));
            degree = da;
        } else if (da < db) {
            out = ast.add(N_SUM, ast.add(N_PROD, a, challenge_in_degree(ast, ctx.num_challenges, db - da)), b);
            degree = db;
        } else {
            out = ast.add(N_SUM, a, b);
            degree = da;
        }
        return true;
    }
    }
}

// Expression::degree (src/polynomial/expression.rs:431-447)
static size_t expr_degree(const Ast &ast, int r, const Ctx &ctx) {
    const Node &x = ast.n[r];
    switch (x.kind) {
    case N_CONST: return 0;
    case N_POLY: {
        size_t i = (size_t)x.index;
        return (i >= ctx.num_selectors + ctx.num_fixed) ? 1 : 0;
    }
    case N_CHAL: return 1;
    case N_NEG:
    case N_SCALED: return expr_degree(ast, x.a, ctx);
    case N_SUM: return std::max(expr_degree(ast, x.a, ctx), expr_degree(ast, x.b, ctx));
    default: return expr_degree(ast, x.a, ctx) + expr_degree(ast, x.b, ctx);
    }
}

// ---------------------------------------------------------------------------------------------
// host: compile an expression into (uniform program, row program)
// ---------------------------------------------------------------------------------------------
// A value is either
//   KNOWN   : compile-time constant (folded)                 -> becomes a uniform-table entry
//   UNIFORM : depends on challenges only (evaluated per call, per point on the host)
//   ROW     : depends on the row (virtual register)
struct Val {
    int cls;      // 0 known, 1 uniform, 2 row
    int id;       // uniform index / virtual register
    fe_t k;       // known value
};
struct UOp {      // uniform program: u[dst] = op(u[a], u[b]);  leaves: constant / challenge
    int op;       // 0 const, 1 challenge(index), 2 add, 3 sub, 4 mul, 5 neg, 6 scaled copy: u[a] * 2^(5 b), b signed (sweep form)
    int a, b;
    fe_t c;
    int64_t chal;
};
struct VInsn {
    uint32_t op;
    int dst;      // virtual register
    int a, b;     // operands: >= 0 virtual reg; < 0: uniform index = -(x)-1 ; loads: a = column, b = rotation
};

struct Compiler {
    const Ast &ast;
    FieldOps f;
    Ctx ctx;
    bool fold_mode;                 // true: advice/challenges are W1 + X*W2 (cross terms); false: plain
    std::vector<UOp> uops;
    std::vector<VInsn> vins;
    int nvreg = 0;
    std::map<int, Val> memo;                                     // AST node -> value
    std::map<std::tuple<int, int, int>, int> u_cse;              // (op, a, b) -> uniform index
    std::map<std::tuple<uint32_t, int, int>, int> r_cse;         // (op, a, b) -> vreg
    std::map<std::tuple<int64_t, int>, int> chal_cse;
    std::string err;

    Compiler(const Ast &a, FieldOps fo, Ctx c, bool fm) : ast(a), f(fo), ctx(c), fold_mode(fm) {}

    int u_const(const fe_t &c) {
        for (size_t i = 0; i < uops.size(); ++i)
            if (uops[i].op == 0 && f.eq(uops[i].c, c)) return (int)i;
        UOp u{};
        u.op = 0;
        u.c = c;
        uops.push_back(u);
        return (int)uops.size() - 1;
    }
    int u_chal(int64_t idx) {
        auto key = std::make_tuple(idx, 0);
        auto it = chal_cse.find(key);
        if (it != chal_cse.end()) return it->second;
        UOp u{};
        u.op = 1;
        u.chal = idx;
        uops.push_back(u);
        return chal_cse[key] = (int)uops.size() - 1;
    }
    int u_op(int op, int a, int b) {
        if ((op == 2 || op == 4) && a > b) std::swap(a, b);
        auto key = std::make_tuple(op, a, b);
        auto it = u_cse.find(key);
        if (it != u_cse.end()) return it->second;
        UOp u{};
        u.op = op;
        u.a = a;
        u.b = b;
        uops.push_back(u);
        return u_cse[key] = (int)uops.size() - 1;
    }
    Val known(const fe_t &k) { Val v; v.cls = 0; v.id = -1; v.k = k; return v; }
    Val uniform(int id) { Val v{}; v.cls = 1; v.id = id; return v; }
    Val rowv(int id) { Val v{}; v.cls = 2; v.id = id; return v; }
    int as_uniform(const Val &v) { return v.cls == 0 ? u_const(v.k) : v.id; }
    int operand(const Val &v) { return v.cls == 2 ? v.id : -(as_uniform(v)) - 1; }
    Val r_op(uint32_t op, int a, int b) {
        if ((op == I_ADD || op == I_MUL) && a > b) std::swap(a, b);
        auto key = std::make_tuple(op, a, b);
        auto it = r_cse.find(key);
        if (it != r_cse.end()) return rowv(it->second);
        VInsn in{op, nvreg++, a, b};
        vins.push_back(in);
        r_cse[key] = in.dst;
        return rowv(in.dst);
    }

    Val v_add(const Val &a, const Val &b) {
        if (a.cls == 0 && b.cls == 0) return known(f.add(a.k, b.k));
        if (a.cls == 0 && f.is_zero(a.k)) return b;
        if (b.cls == 0 && f.is_zero(b.k)) return a;
        if (a.cls < 2 && b.cls < 2) return uniform(u_op(2, as_uniform(a), as_uniform(b)));
        return r_op(I_ADD, operand(a), operand(b));
    }
    Val v_neg(const Val &a) {
        if (a.cls == 0) return known(f.neg(a.k));
        if (a.cls == 1) return uniform(u_op(5, a.id, -1));
        return r_op(I_NEG, a.id, 0);
    }
    Val v_mul(const Val &a, const Val &b) {
        if (a.cls == 0 && b.cls == 0) return known(f.mul(a.k, b.k));
        if ((a.cls == 0 && f.is_zero(a.k)) || (b.cls == 0 && f.is_zero(b.k))) return known(f.zero());
        if (a.cls == 0 && f.eq(a.k, f.one())) return b;
        if (b.cls == 0 && f.eq(b.k, f.one())) return a;
        if (a.cls < 2 && b.cls < 2) return uniform(u_op(4, as_uniform(a), as_uniform(b)));
        if (a.cls == 2 && b.cls == 2 && a.id == b.id) return r_op(I_SQR, a.id, 0);
        return r_op(I_MUL, operand(a), operand(b));
    }

    Val walk(int r) {
        auto it = memo.find(r);
        if (it != memo.end()) return it->second;
        const Node &x = ast.n[r];
        Val v;
        switch (x.kind) {
        case N_CONST: v = known(x.c); break;
        case N_CHAL: v = uniform(u_chal(x.index)); break;
        case N_POLY: {
            size_t i = (size_t)x.index;
            uint32_t op;
            int col;
            if (i < ctx.num_selectors) { op = I_LD_SEL; col = (int)i; }
            else if (i < ctx.num_selectors + ctx.num_fixed) { op = I_LD_FIX; col = (int)(i - ctx.num_selectors); }
            else if (i < ctx.num_selectors + ctx.num_fixed + ctx.num_fold_vars()) { op = I_LD_ADV; col = (int)ctx.witness_col(i - ctx.num_selectors - ctx.num_fixed); }
            else { err = "column index " + std::to_string(i) + " out of range"; v = known(f.zero()); break; }
            auto key = std::make_tuple(op, col, (int)x.rot);
            auto c = r_cse.find(key);
            if (c != r_cse.end()) { v = rowv(c->second); break; }
            VInsn in{op, nvreg++, col, (int)x.rot};
            vins.push_back(in);
            r_cse[key] = in.dst;
            v = rowv(in.dst);
            break;
        }
        case N_NEG: v = v_neg(walk(x.a)); break;
        case N_SUM: { Val a = walk(x.a); Val b = walk(x.b); v = v_add(a, b); break; }
        case N_PROD: { Val a = walk(x.a); Val b = walk(x.b); v = v_mul(a, b); break; }
        default: { Val a = walk(x.a); v = v_mul(a, known(x.c)); break; }
        }
        memo[r] = v;
        return v;
    }
};

// linear-scan allocation of virtual registers to LDS slots
static bool allocate(const std::vector<VInsn> &vins, int nvreg, int result_vreg, std::vector<Insn> &out,
                     uint32_t &result_code, uint32_t &nslots) {
    std::vector<int> last(nvreg, -1);
    for (size_t i = 0; i < vins.size(); ++i) {
        const VInsn &in = vins[i];
        if (in.op > I_LD_ADV) {
            if (in.a >= 0) last[in.a] = (int)i;
            if (in.op <= I_MUL && in.b >= 0) last[in.b] = (int)i;
        }
    }
    if (result_vreg >= 0) last[result_vreg] = (int)vins.size();
    std::vector<int> slot(nvreg, -1);
    std::vector<int> free_list;
    uint32_t n = 0;
    auto enc = [&](int x) -> uint32_t { return x >= 0 ? (uint32_t)slot[x] : (UNIFORM_BIT | (uint32_t)(-x - 1)); };
    for (size_t i = 0; i < vins.size(); ++i) {
        const VInsn &in = vins[i];
        Insn o;
        o.op = in.op;
        if (in.op <= I_LD_ADV) {
            o.a = (uint32_t)in.a;
            o.b = (uint32_t)in.b;
        } else {
            o.a = enc(in.a);
            o.b = in.op <= I_MUL ? enc(in.b) : 0;
            // operands dying here free their slots before the destination is chosen
            if (in.a >= 0 && last[in.a] == (int)i) free_list.push_back(slot[in.a]);
            if (in.op <= I_MUL && in.b >= 0 && in.b != in.a && last[in.b] == (int)i) free_list.push_back(slot[in.b]);
        }
        int s;
        if (!free_list.empty()) { s = free_list.back(); free_list.pop_back(); } else s = (int)n++;
        slot[in.dst] = s;
        o.dst = (uint32_t)s;
        out.push_back(o);
        if (last[in.dst] < 0) free_list.push_back(s);   // dead value (cannot happen after CSE, but stay safe)
    }
    nslots = n ? n : 1;
    result_code = result_vreg >= 0 ? (uint32_t)slot[result_vreg] : 0;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Structure
// ---------------------------------------------------------------------------------------------
struct Program {
    std::vector<UOp> uops;
    std::vector<Insn> insns;
    uint32_t result = 0, nslots = 1;
    Insn *d_insns = nullptr;
    std::vector<VInsn> vins;        // SSA form (virtual registers), kept for emit_spec_source
    int result_vreg = -1;
    // "sweep" form (plan_sweep): the expression as a sum of terms coef * body, evaluated for ALL points per column load
    struct SweepTerm { int node; int coef; int sign; int level; };   // node: vreg, or -(u)-1 for a row-independent term; coef: uniform index (pre-scaled)
    struct SweepCluster { std::vector<int> terms, body, loads; bool linear = false; };     // body / loads: vregs in SSA order; linear: see plan_sweep
    std::vector<SweepTerm> sw_terms;
    std::vector<SweepCluster> sw_clusters;
    std::vector<int> sw_level;      // per vreg: power of 2^-5 its 9 x 29-bit value carries (field29.cuh: R' = 2^261 vs the ABI's 2^256)
    std::vector<int> sw_raise;      // [delta] -> uniform index of the raw constant 2^(261 - 5 delta): product with it adds delta levels
    int sw_one = -1;                // uniform index of 2^261 mod p (the radix' one): product with it folds a lazy value below 2p
    std::vector<int> sw_coef;       // the uniform entries that are term coefficients (dedicated entries: ProtoGalaxy scales them by the leaf weight)
    std::map<std::pair<int, int>, int> sw_uat;   // (uniform index, level) -> index of the copy scaled by 2^(-5 level) (operand of a body addition)
    bool sweep_ok = false;
    uint64_t fingerprint = 0;       // FNV-1a of the SSA program
    int spec_id = -1;               // index into the ahead-of-time specialised kernels, or -1
    jit::Kernel jit;                // straight-line kernel compiled at structure creation (jit.hip), or empty
};

static uint64_t fingerprint_of(const Program &p) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ull; } };
    for (auto &in : p.vins) { mix(in.op); mix((uint64_t)(int64_t)in.dst); mix((uint64_t)(int64_t)in.a); mix((uint64_t)(int64_t)in.b); }
    mix((uint64_t)(int64_t)p.result_vreg);
    mix(p.result);
    mix(p.uops.size());
    return h;
}

// straight-line C++ for the SSA program (one function template over the field)
std::string emit_spec_source(const Program &p, const std::string &name, bool shared_mul) {
    std::string o;
    auto opnd = [](int x) { return x >= 0 ? "v" + std::to_string(x) : "U[" + std::to_string(-x - 1) + "]"; };
    // shared_mul: the multiplications call ONE shared body (mul_ni / sqr_ni, rowprog_dev.cuh) instead of inlining 2 KB of
    // code each, so a program is tens of KB, not hundreds.  Run-time compiled kernels need that: with every multiplier
    // inlined they ran 30 % slower than the same ISA linked into the library (profiles/r01_jit_vs_aot.txt), called they
    // match it.  The ahead-of-time kernels keep the inlined form (3 % faster there).
    const std::string MUL = shared_mul ? "mul_ni<F>(" : "F::mul(", SQR = shared_mul ? "sqr_ni<F>(" : "F::sqr(";
    o += "template <class F>\n__device__ __forceinline__ fe_t " + name +
         "(const RowCtx &C, uint32_t row, uint32_t pt, const fe_t *__restrict__ U) {\n";
    o += "    const uint32_t mask = C.rows - 1; (void)mask; (void)pt; (void)U;\n";
    for (auto &in : p.vins) {
        std::string d = "    const fe_t v" + std::to_string(in.dst) + " = ";
        std::string rr = "(row + " + std::to_string((uint32_t)in.b) + "u) & mask";
        switch (in.op) {
        case I_LD_SEL: o += d + "ld_sel<F>(C, " + std::to_string(in.a) + ", " + rr + ");\n"; break;
        case I_LD_FIX: o += d + "ld_fix<F>(C, " + std::to_string(in.a) + ", " + rr + ");\n"; break;
        case I_LD_ADV: o += d + "ld_adv<F>(C, " + std::to_string(in.a) + ", " + rr + ", pt);\n"; break;
        case I_ADD: o += d + "F::add(" + opnd(in.a) + ", " + opnd(in.b) + ");\n"; break;
        case I_SUB: o += d + "F::sub(" + opnd(in.a) + ", " + opnd(in.b) + ");\n"; break;
        case I_MUL: o += d + MUL + opnd(in.a) + ", " + opnd(in.b) + ");\n"; break;
        case I_SQR: o += d + SQR + opnd(in.a) + ");\n"; break;
        case I_DBL: o += d + "F::dbl(" + opnd(in.a) + ");\n"; break;
        default: o += d + "F::neg(" + opnd(in.a) + ");\n"; break;
        }
    }
    if (p.result_vreg >= 0) o += "    return v" + std::to_string(p.result_vreg) + ";\n";
    else o += "    return U[" + std::to_string(p.result & ~UNIFORM_BIT) + "];\n";
    o += "}\n";
    return o;
}

// ---------------------------------------------------------------------------------------------
// Sweep form.  The straight-line program above evaluates ONE point per pass over the row's columns: d + 1 passes, each
// re-reading every column from L2 / HBM (profiles/r01_pmc_step_kernels.json: 6.3x the algorithmic bytes).  The sweep form
// turns the loops inside out: the expression is flattened into a sum of TERMS  coef(pt) * body(row, pt)  -- the linear
// skeleton (+, -, scaling by row-independent values) is distributed, the coefficients become new entries of the uniform
// table (host work per call) -- and terms sharing columns form CLUSTERS.  A cluster loads its columns once, then loops
// over the points: advice leaves are affine in the point (W1 + X W2, or the Lagrange fold (w0 + w1)/2 + X (w0 - w1)/2), so the
// next point costs one addition per leaf; the per-point accumulators live in LDS.  Bodies run on the 9 x 29-bit
// multiplier (field29.cuh), whose Montgomery radix 2^261 differs from the ABI's 2^256: a product of two ABI-form values
// comes out 2^-5 short ("level" + 1).  Levels are tracked statically; the term's final multiplication by its coefficient
// uses a coefficient pre-scaled by 2^(5 (level + 1)) on the host, which lands every term back in ABI form -- the sum is
// the same field element as the straight-line program's, canonical, bit for bit.
// ---------------------------------------------------------------------------------------------
struct SweepBuilder {
    Program &p;
    std::map<std::tuple<int, int, int>, int> cse;          // (op, a, b) -> uniform index, for the entries created here
    std::vector<int> def;                                  // vreg -> index into p.vins
    explicit SweepBuilder(Program &prog) : p(prog) {
        int nv = 0;
        for (auto &in : p.vins) nv = std::max(nv, in.dst + 1);
        def.assign(nv, -1);
        for (size_t i = 0; i < p.vins.size(); ++i) def[p.vins[i].dst] = (int)i;
    }
    int uop(int op, int a, int b, const fe_t *c = nullptr) {
        if (op == 4 && a > b) std::swap(a, b);
        auto key = std::make_tuple(op, a, b);
        if (!c) { auto it = cse.find(key); if (it != cse.end()) return it->second; }
        UOp u{};
        u.op = op;
        u.a = a;
        u.b = b;
        if (c) u.c = *c;
        p.uops.push_back(u);
        const int id = (int)p.uops.size() - 1;
        if (!c) cse[key] = id;
        return id;
    }
    int u_mul(int a, int b) { return a < 0 ? b : (b < 0 ? a : uop(4, a, b)); }      // -1 = coefficient one
    int u_scaled(int a, int e) { return e == 0 ? a : uop(6, a, e); }
    int one = -1;
    int u_one(const FieldOps &f) {
        if (one < 0) { fe_t o = f.one(); one = uop(0, 0, 0, &o); }
        return one;
    }
    int two = -1;
    int u_two(const FieldOps &f) {
        if (two < 0) { fe_t t = f.add(f.one(), f.one()); two = uop(0, 0, 0, &t); }
        return two;
    }
    bool is_linear_mul(const VInsn &in) const { return in.op == I_MUL && ((in.a < 0) != (in.b < 0)); }

    // distribute the linear skeleton below `x` (operand code) with coefficient `coef` (uniform index or -1) and sign
    void lin(int x, int coef, int sign, const FieldOps &f, int depth) {
        if (x < 0) {                                         // row-independent value
            p.sw_terms.push_back({x, coef, sign, 0});
            return;
        }
        const VInsn &in = p.vins[def[x]];
        if (depth < 64) {
            switch (in.op) {
            case I_ADD: lin(in.a, coef, sign, f, depth + 1); lin(in.b, coef, sign, f, depth + 1); return;
            case I_SUB: lin(in.a, coef, sign, f, depth + 1); lin(in.b, coef, -sign, f, depth + 1); return;
            case I_NEG: lin(in.a, coef, -sign, f, depth + 1); return;
            case I_DBL: lin(in.a, u_mul(coef, u_two(f)), sign, f, depth + 1); return;
            case I_MUL:
                if (is_linear_mul(in)) {
                    const int u = in.a < 0 ? -in.a - 1 : -in.b - 1, v = in.a < 0 ? in.b : in.a;
                    lin(v, u_mul(coef, u), sign, f, depth + 1);
                    return;
                }
                break;
            default: break;
            }
        }
        p.sw_terms.push_back({x, coef, sign, 0});
    }
    // level of a body value; uniform operands of body additions are re-scaled on the host, so they never raise a level
    int level_of(int v) {
        if (p.sw_level[v] >= 0) return p.sw_level[v];
        const VInsn &in = p.vins[def[v]];
        int l = 0;
        auto lv = [&](int x) { return x < 0 ? 0 : level_of(x); };
        switch (in.op) {
        case I_LD_SEL: case I_LD_FIX: case I_LD_ADV: l = 0; break;
        case I_MUL: l = lv(in.a) + lv(in.b) + 1; break;
        case I_SQR: l = 2 * lv(in.a) + 1; break;
        case I_ADD: case I_SUB: l = std::max(lv(in.a), lv(in.b)); break;
        default: l = lv(in.a); break;
        }
        return p.sw_level[v] = l;
    }
    void collect(int v, std::vector<char> &seen, std::vector<int> &body, std::vector<int> &loads) {
        if (v < 0 || seen[v]) return;
        seen[v] = 1;
        const VInsn &in = p.vins[def[v]];
        if (in.op <= I_LD_ADV) { loads.push_back(v); return; }
        collect(in.a, seen, body, loads);
        if (in.op <= I_MUL) collect(in.b, seen, body, loads);
        body.push_back(v);
    }
};

// registers a cluster may spend on hoisted column values: an advice leaf is (value, step) = 16 VGPRs, a fixed one 8
static constexpr int SWEEP_LOAD_BUDGET = 112;

static void plan_sweep(Program &p, const FieldOps &f) {
    p.sweep_ok = false;
    p.sw_terms.clear();
    p.sw_clusters.clear();
    if (p.result_vreg < 0 || p.vins.empty()) return;
    SweepBuilder B(p);
    B.lin(p.result_vreg, -1, +1, f, 0);
    if (p.sw_terms.empty() || p.sw_terms.size() > 4096) { p.sw_terms.clear(); return; }
    p.sw_level.assign(B.def.size(), -1);
    for (auto &t : p.sw_terms) {
        t.level = t.node < 0 ? 0 : B.level_of(t.node);
        if (t.level > 40) { p.sw_terms.clear(); return; }
        if (t.node < 0) {                                    // constant term: coefficient * value, a dedicated ABI-form entry
            t.coef = B.uop(6, B.u_mul(t.coef, -t.node - 1), 0);
        } else {                                             // coef * 2^(5 (level + 1)): the closing product returns to ABI form
            t.coef = B.u_scaled(t.coef < 0 ? B.u_one(f) : t.coef, t.level + 1);
        }
    }
    p.sw_coef.clear();
    for (auto &t : p.sw_terms) p.sw_coef.push_back(t.coef);
    std::sort(p.sw_coef.begin(), p.sw_coef.end());
    p.sw_coef.erase(std::unique(p.sw_coef.begin(), p.sw_coef.end()), p.sw_coef.end());
    // helper constants of the 2^261-radix bodies
    int max_level = 0;
    for (int l : p.sw_level) max_level = std::max(max_level, l);
    {                                                        // a constant of its own (never shared with a coefficient entry)
        fe_t c = f.one();
        for (int k = 0; k < 5; ++k) c = f.add(c, c);
        p.sw_one = B.uop(0, 0, 0, &c);
    }
    p.sw_raise.assign(max_level + 1, -1);
    for (int d = 1; d <= max_level; ++d) p.sw_raise[d] = B.u_scaled(B.u_one(f), 1 - d);
    p.sw_uat.clear();
    for (size_t i = 0; i < p.vins.size(); ++i) {            // uniform operands of body additions at level > 0
        const VInsn &in = p.vins[i];
        if ((in.op != I_ADD && in.op != I_SUB) || p.sw_level[in.dst] <= 0) continue;
        for (int x : {in.a, in.b})
            if (x < 0) p.sw_uat[{-x - 1, p.sw_level[in.dst]}] = B.u_scaled(-x - 1, -p.sw_level[in.dst]);
    }
    // Terms that are AFFINE in the evaluation point -- a product chain with at most one advice leaf and point-independent
    // coefficients (no challenge in the uniform program: the ProtoGalaxy gate polynomials) -- need no evaluation per point: their
    // clusters compute the value at the first point and the slope (the same chain with the leaf's step in place of its value)
    // once, and every further point is one lazy addition.  chain(v) = number of advice leaves of a pure product chain, -1 otherwise.
    bool const_u = true;
    for (const UOp &u : p.uops) const_u = const_u && u.op != 1;
    std::function<int(int)> chain = [&](int v) -> int {
        if (v < 0) return 0;
        const VInsn &in = p.vins[B.def[v]];
        if (in.op == I_LD_ADV) return 1;
        if (in.op == I_LD_SEL || in.op == I_LD_FIX) return 0;
        if (in.op != I_MUL) return -1;
        const int a = chain(in.a), b = chain(in.b);
        return (a < 0 || b < 0) ? -1 : a + b;
    };
    std::vector<char> term_linear(p.sw_terms.size(), 0);
    for (size_t ti = 0; ti < p.sw_terms.size(); ++ti) {
        if (!const_u) break;
        const auto &t = p.sw_terms[ti];
        const int c = t.node < 0 ? 0 : chain(t.node);
        term_linear[ti] = c == 0 || c == 1;
    }
    // clusters: terms in expression order; a term joins the open cluster while the hoisted columns fit the budget
    const size_t nv = B.def.size();
    std::vector<char> in_cluster(nv, 0);
    Program::SweepCluster cur;
    int cost = 0;
    auto load_cost = [&](int v) { return p.vins[B.def[v]].op == I_LD_ADV ? 16 : 8; };
    // a term q * (s * y) keeps the hoisted affine factor q s and its step in registers across the point loop (emit_sweep_source):
    // 18 VGPRs, while the fixed leaf q is only needed before the loop
    constexpr int hoist_regs = 10;
    auto hoist_cost = [&](int v) -> int {
        if (v < 0) return 0;
        const VInsn &in = p.vins[B.def[v]];
        if (in.op != I_MUL) return 0;
        for (int o1 = 0; o1 < 2; ++o1) {
            const int a = o1 ? in.b : in.a, u = o1 ? in.a : in.b;
            if (a < 0 || u < 0) continue;
            const int oa = p.vins[B.def[a]].op, ou = p.vins[B.def[u]].op;
            if ((oa != I_LD_FIX && oa != I_LD_SEL) || ou != I_MUL) continue;
            const VInsn &iu = p.vins[B.def[u]];
            for (int x : {iu.a, iu.b})
                if (x >= 0 && p.vins[B.def[x]].op == I_LD_ADV) return hoist_regs;
        }
        return 0;
    };
    auto flush = [&]() {
        if (cur.terms.empty()) return;
        std::sort(cur.body.begin(), cur.body.end());
        std::sort(cur.loads.begin(), cur.loads.end());
        p.sw_clusters.push_back(cur);
        cur = Program::SweepCluster();
        cost = 0;
        std::fill(in_cluster.begin(), in_cluster.end(), 0);
    };
    for (int pass = 0; pass < 2; ++pass) {                   // the per-point clusters first, then the affine ones
    flush();
    for (size_t ti = 0; ti < p.sw_terms.size(); ++ti) {
        if ((int)term_linear[ti] != pass) continue;
        cur.linear = pass == 1;
        const auto &t = p.sw_terms[ti];
        if (t.node < 0) { cur.terms.push_back((int)ti); continue; }
        std::vector<char> seen(nv, 0);
        std::vector<int> body, loads;
        B.collect(t.node, seen, body, loads);
        int extra = 0;
        for (int v : loads) if (!in_cluster[v]) extra += load_cost(v);
        extra += hoist_cost(t.node);
        if (!cur.terms.empty() && cost + extra > SWEEP_LOAD_BUDGET) flush();
        for (int v : loads) if (!in_cluster[v]) { in_cluster[v] = 1; cur.loads.push_back(v); cost += load_cost(v); }
        for (int v : body) if (!in_cluster[v]) { in_cluster[v] = 1; cur.body.push_back(v); }
        cost += hoist_cost(t.node);
        cur.linear = pass == 1;
        cur.terms.push_back((int)ti);
    }
    }
    flush();
    p.sweep_ok = !p.sw_clusters.empty();
}

// the sweep form as C++ (see the comment on the signature below)
std::string emit_sweep_source(const Program &p, const std::string &name, bool shared_mul) {
    std::string o;
    if (!p.sweep_ok) return o;
    // shared_mul: one multiplier body per kernel (mul29_ni / sqr29_ni, rowprog_dev.cuh) -- the run-time compiled form, see emit_spec_source
    const std::string MUL = shared_mul ? "mul29_ni<F>(" : "G::mul(", SQR = shared_mul ? "sqr29_ni<F>(" : "G::sqr(";
    std::vector<int> def;
    {
        int nv = 0;
        for (auto &in : p.vins) nv = std::max(nv, in.dst + 1);
        def.assign(nv, -1);
        for (size_t i = 0; i < p.vins.size(); ++i) def[p.vins[i].dst] = (int)i;
    }
    auto S = [](int x) { return std::to_string(x); };
    // uniform entries the bodies need in a re-scaled form are added by plan_sweep only for coefficients; an addition of a
    // row value of level l and a uniform value is emitted with a run-time multiplication-free trick: the uniform operand is
    // multiplied by the level-raising constant like any other lower-level operand (rare: constants inside products).
    // NAME(C, row, npts, Uall, nu, acc, accumulate): adds (accumulate) or stores the LAZY 9 x 29-bit sum P(pt) * 2^256 + small
    // multiples of p into the thread's limb-planar LDS accumulators (sw_load / sw_store, rowprog_dev.cuh); NAME_one<F>() = the
    // uniform index of 2^261 mod p, with which the caller folds (sw_fold) and finishes (sw_finish) them.
    o += "template <class F> __device__ constexpr uint32_t " + name + "_one() { return " + S(p.sw_one) + "u; }\n";
    o += "template <class F>\n__device__ __forceinline__ void " + name +
         "(const RowCtx &C, uint32_t row, uint32_t npts, const fe_t *__restrict__ Uall, uint32_t nu, uint32_t *__restrict__ acc, bool accumulate) {\n";
    o += "    using G = Fp29<typename F::Params>;\n    const uint32_t mask = C.rows - 1; (void)mask;\n";
    bool first = true;
    double acc_bound = 2.0;                                      // an accumulator handed in by the caller is folded: < 2p
    // number of advice leaves of a product chain (the affine clusters hold nothing else)
    std::function<int(int)> chain = [&](int v) -> int {
        if (v < 0) return 0;
        const VInsn &in = p.vins[def[v]];
        if (in.op == I_LD_ADV) return 1;
        if (in.op == I_LD_SEL || in.op == I_LD_FIX) return 0;
        return chain(in.a) + chain(in.b);
    };
    for (size_t ci = 0; ci < p.sw_clusters.size(); ++ci) {
        const auto &cl = p.sw_clusters[ci];
        o += std::string("    {   // cluster ") + S((int)ci) + (cl.linear ? " (affine in the point)\n" : "\n");
        for (int v : cl.loads) {
            const VInsn &in = p.vins[def[v]];
            const std::string rr = "(row + " + std::to_string((uint32_t)in.b) + "u) & mask";
            if (in.op == I_LD_SEL) o += "        const fe_t l" + S(v) + " = ld_sel<F>(C, " + S(in.a) + ", " + rr + ");\n";
            else if (in.op == I_LD_FIX) o += "        const fe_t l" + S(v) + " = ld_fix<F>(C, " + S(in.a) + ", " + rr + ");\n";
            else o += "        fe_t l" + S(v) + ", s" + S(v) + "; adv_affine<F>(C, " + S(in.a) + ", " + rr + ", l" + S(v) + ", s" + S(v) + ");\n";
        }
        std::vector<char> is_load(def.size(), 0);            // column values are unpacked where they are used: a 9-limb copy of
        for (int v : cl.loads) is_load[v] = 1;               // every hoisted column, live across the loop body, spills
        // Hoisted affine factors (per-point clusters): q * (s * y) with q a fixed / selector column and s an advice leaf is computed as
        // (q s) * y, and q s is affine in the point -- its value and step cost two products per row, every point one lazy addition
        // instead of a product.  (The MainGate's q_5 s^5 terms: 3 instead of 4 products per state and point.)
        struct Hoist { int v, a, b, c, u; };
        std::vector<Hoist> hoists;
        std::map<int, size_t> hoist_of;                       // v -> index
        std::vector<char> hoisted_inner(def.size(), 0);
        if (!cl.linear) {
            std::vector<int> uses(def.size(), 0);
            for (int v : cl.body) {
                const VInsn &in = p.vins[def[v]];
                if (in.a >= 0) ++uses[in.a];
                if (in.op <= I_MUL && in.b >= 0) ++uses[in.b];
            }
            for (int ti : cl.terms) if (p.sw_terms[ti].node >= 0) ++uses[p.sw_terms[ti].node];
            auto fixed_leaf = [&](int x) { return x >= 0 && is_load[x] && p.vins[def[x]].op != I_LD_ADV; };
            auto adv_leaf = [&](int x) { return x >= 0 && is_load[x] && p.vins[def[x]].op == I_LD_ADV; };
            for (int v : cl.body) {
                const VInsn &in = p.vins[def[v]];
                if (in.op != I_MUL) continue;
                for (int o1 = 0; o1 < 2 && !hoist_of.count(v); ++o1) {
                    const int a = o1 ? in.b : in.a, u = o1 ? in.a : in.b;
                    if (!fixed_leaf(a) || u < 0 || is_load[u] || p.vins[def[u]].op != I_MUL || uses[u] != 1 || hoisted_inner[u]) continue;
                    const VInsn &iu = p.vins[def[u]];
                    for (int o2 = 0; o2 < 2; ++o2) {
                        const int b = o2 ? iu.b : iu.a, c = o2 ? iu.a : iu.b;
                        if (!adv_leaf(b) || c == b) continue;
                        hoist_of[v] = hoists.size();
                        hoists.push_back({v, a, b, c, u});
                        hoisted_inner[u] = 1;
                        break;
                    }
                }
            }
        }
        // One evaluation of the cluster's terms.  IN: indentation; X: name prefix of the temporaries; slope: the advice leaves
        // enter with their STEP and the terms without an advice leaf are left out (affine clusters).  -> expression of the lazy
        // sum and its bound in units of p (empty: nothing to add).
        auto gen = [&](const std::string &IN, const std::string &X, bool slope, double &total_b) -> std::string {
        std::map<int, double> bound;                           // static bound of a value in units of p (normalised limbs)
        for (int v : cl.loads) bound[v] = 1.0;
        auto bnd = [&](int x) { return x >= 0 ? bound[x] : 1.0; };
        auto lvl = [&](int x) { return x >= 0 ? p.sw_level[x] : 0; };
        int tmp = 0;
        const std::string D = IN + "const f29_t ";
        // operand x at `level` with a bound <= max_bound: the expression text; b_out = its bound
        auto prep = [&](int x, int level, double max_bound, double &b_out) -> std::string {
            if (x < 0) {                                         // row-independent operand: the host keeps a copy at every level needed
                b_out = 1.0;
                const int u = -x - 1;
                if (level <= 0) return "G::unpack(U[" + S(u) + "])";
                auto it = p.sw_uat.find({u, level});
                return "G::unpack(U[" + S(it == p.sw_uat.end() ? u : it->second) + "])";
            }
            std::string e;
            if (is_load[x]) e = std::string("G::unpack(") + ((slope && p.vins[def[x]].op == I_LD_ADV) ? "s" : "l") + S(x) + ")";
            else e = X + "x" + S(x);
            double b = bnd(x);
            if (lvl(x) < level) {                                // product with the raw constant 2^(261 - 5 delta): + delta levels
                const std::string t = X + "r" + S(tmp++);
                o += D + t + " = " + MUL + e + ", G::unpack(U[" + S(p.sw_raise[level - lvl(x)]) + "]));\n";
                e = t;
                b = 2.0;
            }
            if (b > max_bound) {                                 // product with 2^261 mod p: the same value, below 2p again
                const std::string t = X + "r" + S(tmp++);
                o += D + t + " = " + MUL + e + ", G::unpack(U[" + S(p.sw_one) + "]));\n";
                e = t;
                b = 2.0;
            }
            b_out = b;
            return e;
        };
        // in slope mode only the values the degree-1 terms need are computed
        std::vector<char> want(def.size(), 1);
        if (slope) {
            std::fill(want.begin(), want.end(), 0);
            std::function<void(int)> mark = [&](int v) {
                if (v < 0 || want[v]) return;
                want[v] = 1;
                const VInsn &in = p.vins[def[v]];
                if (in.op > I_LD_ADV) { mark(in.a); if (in.op <= I_MUL) mark(in.b); }
            };
            for (int ti : cl.terms) {
                const auto &t = p.sw_terms[ti];
                if (t.node >= 0 && chain(t.node) == 1) mark(t.node);
            }
        }
        for (int v : cl.body) {
            if (!want[v] || hoisted_inner[v]) continue;
            const VInsn &in = p.vins[def[v]];
            const std::string d = D + X + "x" + S(v) + " = ";
            double ba, bb;
            if (hoist_of.count(v)) {                             // (q s) * y with the affine factor kept in `aff`
                const Hoist &h = hoists[hoist_of[v]];
                std::string c = prep(h.c, lvl(h.c), 12.0, bb);
                o += d + MUL + "aff" + S(v) + ", " + c + ");\n";
                bound[v] = 2.0;
                continue;
            }
            switch (in.op) {
            case I_MUL: {
                std::string a = prep(in.a, lvl(in.a), 12.0, ba), b = prep(in.b, lvl(in.b), 12.0, bb);      // a uniform operand enters at level 0
                o += d + MUL + a + ", " + b + ");\n";
                bound[v] = 2.0;
                break;
            }
            case I_SQR: {
                std::string a = prep(in.a, lvl(in.a), 12.0, ba);
                o += d + SQR + a + ");\n";
                bound[v] = 2.0;
                break;
            }
            case I_ADD: {
                const int L = p.sw_level[v];
                std::string a = prep(in.a, L, 30.0, ba), b = prep(in.b, L, 30.0, bb);
                o += d + "G::normalize(G::add_lazy(" + a + ", " + b + "));\n";
                bound[v] = ba + bb;
                break;
            }
            case I_SUB: {
                const int L = p.sw_level[v];
                std::string a = prep(in.a, L, 30.0, ba), b = prep(in.b, L, 30.0, bb);
                const int cp = (int)bb + 1;
                o += d + "G::normalize(G::template sub_lazy<" + S(cp) + ", 0>(" + a + ", " + b + "));\n";
                bound[v] = ba + cp;
                break;
            }
            case I_DBL: {
                std::string a = prep(in.a, lvl(in.a), 30.0, ba);
                o += d + "G::normalize(G::add_lazy(" + a + ", " + a + "));\n";
                bound[v] = 2 * ba;
                break;
            }
            default: {
                std::string a = prep(in.a, lvl(in.a), 30.0, ba);
                const int cp = (int)ba + 1;
                o += d + "G::normalize(G::template neg_lazy<" + S(cp) + ", 0>(" + a + "));\n";
                bound[v] = cp;
                break;
            }
            }
        }
        // the terms: terms sharing a coefficient are summed first, then ONE closing product with the pre-scaled coefficient per
        // group -> ABI form; everything stays lazy (sums of values < 2p, normalised limbs) down to the accumulator in LDS
        std::vector<int> order;                                  // coefficient groups in first-appearance order
        std::map<int, std::vector<int>> groups;
        for (int ti : cl.terms) {
            const auto &t = p.sw_terms[ti];
            if (slope && (t.node < 0 || chain(t.node) != 1)) continue;
            if (!groups.count(t.coef)) order.push_back(t.coef);
            groups[t.coef].push_back(ti);
        }
        std::string total;
        total_b = 0;
        auto lazy_add = [&](std::string &sum, double &b, const std::string &e, double be, bool minus) {
            const std::string r = X + "r" + S(tmp++);
            if (sum.empty()) {
                if (minus) {
                    const int cp = (int)be + 1;
                    o += D + r + " = G::normalize(G::template neg_lazy<" + S(cp) + ", 0>(" + e + "));\n";
                    b = cp;
                } else {
                    sum = e;
                    b = be;
                    return;
                }
            } else if (minus) {
                const int cp = (int)be + 1;
                o += D + r + " = G::normalize(G::template sub_lazy<" + S(cp) + ", 0>(" + sum + ", " + e + "));\n";
                b += cp;
            } else {
                o += D + r + " = G::normalize(G::add_lazy(" + sum + ", " + e + "));\n";
                b += be;
            }
            sum = r;
        };
        auto fold = [&](std::string &sum, double &b, double limit) {
            if (b <= limit) return;
            const std::string r = X + "r" + S(tmp++);
            o += D + r + " = " + MUL + sum + ", G::unpack(U[" + S(p.sw_one) + "]));\n";
            sum = r;
            b = 2.0;
        };
        for (int key : order) {
            const auto &g = groups[key];
            const auto &t0 = p.sw_terms[g[0]];
            std::string val;
            bool negate = t0.sign < 0;                           // the group enters the cluster sum as +-(first +- ...)
            double vb = 1.0;
            if (t0.node < 0) {
                val = "G::unpack(U[" + S(t0.coef) + "])";        // constant term (ABI form, canonical)
            } else {
                // all terms of a group share the coefficient, hence the level (the scale 2^(5 (level + 1)) is part of the entry)
                std::string sum;
                double b = 0;
                for (size_t j = 0; j < g.size(); ++j) {
                    const auto &t = p.sw_terms[g[j]];
                    double bj;
                    std::string e = prep(t.node, lvl(t.node), 12.0, bj);
                    lazy_add(sum, b, e, bj, (t.sign < 0) != negate);
                    fold(sum, b, 12.0);
                }
                const std::string r = X + "r" + S(tmp++);
                o += D + r + " = " + MUL + sum + ", G::unpack(U[" + S(t0.coef) + "]));\n";
                val = r;
                vb = 2.0;
            }
            lazy_add(total, total_b, val, vb, negate);
            fold(total, total_b, 24.0);
        }
        if (!total.empty() && slope) fold(total, total_b, 2.0);          // the step is added once per point: keep it below 2p
        return total;
        };   // gen

        if (!cl.linear) {
            for (const Hoist &h : hoists) {
                o += "        f29_t aff" + S(h.v) + " = " + MUL + "G::unpack(l" + S(h.a) + "), G::unpack(l" + S(h.b) + "));\n";
                o += "        const f29_t affs" + S(h.v) + " = npts > 1 ? " + MUL + "G::unpack(l" + S(h.a) + "), G::unpack(s" + S(h.b) + ")) : aff" + S(h.v) + ";\n";
            }
            o += "        for (uint32_t pt = 0; pt < npts; ++pt) {\n";
            o += "            const fe_t *__restrict__ U = Uall + (size_t)pt * nu; (void)U;\n";
            double total_b = 0;
            const std::string total = gen("            ", "", false, total_b);
            if (first) {
                o += "            sw_store(acc, pt, accumulate ? G::normalize(G::add_lazy(sw_load(acc, pt), " + total + ")) : " + total + ");\n";
            } else {
                o += "            sw_store(acc, pt, G::normalize(G::add_lazy(sw_load(acc, pt), " + total + ")));\n";
            }
            acc_bound += total_b;
            if (acc_bound > 100.0) {                                 // fold the accumulators before they outgrow the 261-bit limbs
                o += "            sw_store(acc, pt, " + MUL + "sw_load(acc, pt), G::unpack(U[" + S(p.sw_one) + "])));\n";
                acc_bound = 2.0;
            }
            for (int v : cl.loads)
                if (p.vins[def[v]].op == I_LD_ADV) o += "            l" + S(v) + " = F::add(l" + S(v) + ", s" + S(v) + ");\n";
            for (const Hoist &h : hoists)                         // < 2p (1 + points): fine as a multiplier operand (<= 20p * 2p < 165 p^2)
                o += "            aff" + S(h.v) + " = G::normalize(G::add_lazy(aff" + S(h.v) + ", affs" + S(h.v) + "));\n";
            o += "        }\n    }\n";
        } else {
            // value at the first point + slope, then one lazy addition per further point (coefficients do not depend on the point)
            o += "        const fe_t *__restrict__ U = Uall; (void)U;\n";
            double vb = 0, sb = 0;
            std::string val = gen("        ", "v", false, vb);
            const std::string step = gen("        ", "d", true, sb);
            if (val.empty()) { val = "G::unpack(F::zero())"; vb = 1.0; }
            if (vb > 2.0) {
                o += "        const f29_t vfold = " + MUL + val + ", G::unpack(U[" + S(p.sw_one) + "]));\n";
                val = "vfold";
                vb = 2.0;
            }
            o += "        f29_t cur = " + val + ";\n";
            o += "        for (uint32_t pt = 0; pt < npts; ++pt) {\n";
            if (first) o += "            sw_store(acc, pt, accumulate ? G::normalize(G::add_lazy(sw_load(acc, pt), cur)) : cur);\n";
            else o += "            sw_store(acc, pt, G::normalize(G::add_lazy(sw_load(acc, pt), cur)));\n";
            if (!step.empty()) o += "            cur = G::normalize(G::add_lazy(cur, " + step + "));\n";
            const double worst = vb + (step.empty() ? 0.0 : sb * (double)DMAX);      // cur at the last of <= DMAX + 1 points
            acc_bound += worst;
            if (acc_bound > 100.0) {
                o += "            sw_store(acc, pt, " + MUL + "sw_load(acc, pt), G::unpack(U[" + S(p.sw_one) + "])));\n";
                acc_bound = 2.0;
            }
            o += "        }\n    }\n";
        }
        first = false;
    }
    o += "}\n";
    return o;
}

// the translation unit hiprtc compiles: the emitted program `jit_fn` wrapped in the kernel body the ahead-of-time kernels use
static std::string jit_translation_unit(const std::string &fn_source, int field, bool has_sweep = false) {
    const std::string fname = field == 0 ? "Fr" : "Fq";
    std::string body;
    if (has_sweep)      // same choice as k_rowprog_spec: sweep form whenever the advice leaves are affine in the point
        body = "    if (A.ctx.wcoef == nullptr && A.npts <= DMAX + 1) {\n        sweep_kernel_body<" + fname +
               ">(A, jit_fn_sweep_one<" + fname + ">(), [](const RowCtx &C, uint32_t row, uint32_t npts, const fe_t *U, uint32_t nu, uint32_t *acc, "
               "bool accumulate) { jit_fn_sweep<" + fname + ">(C, row, npts, U, nu, acc, accumulate); });\n        return;\n    }\n";
    return "#include \"rowprog_dev.cuh\"\nnamespace srs {\nnamespace rowprog {\n" + fn_source +
           "extern \"C\" __global__ void __launch_bounds__(128, 2) srs_jit_rowprog(DevArgs A) {\n" + body + "    spec_kernel_body<" + fname +
           ">(A, [](const RowCtx &C, uint32_t row, uint32_t pt, const fe_t *U) { return jit_fn<" + fname + ">(C, row, pt, U); });\n}\n}\n}\n";
}

// Host-only check of the run-time compilation path (no device): a small program in the emitted form -- column loads,
// called and inlined multipliers, a uniform -- must compile with hiprtc against the headers embedded in the library.
bool jit_selfcheck(size_t *code_bytes, std::string &log) {
    const std::string fn =
        "template <class F>\n__device__ __forceinline__ fe_t jit_fn(const RowCtx &C, uint32_t row, uint32_t pt, const fe_t *__restrict__ U) {\n"
        "    const uint32_t mask = C.rows - 1;\n"
        "    const fe_t v0 = ld_fix<F>(C, 0, (row + 0u) & mask);\n"
        "    const fe_t v1 = ld_adv<F>(C, 0, (row + 1u) & mask, pt);\n"
        "    const fe_t v2 = ld_sel<F>(C, 0, (row + 0u) & mask);\n"
        "    const fe_t v3 = mul_ni<F>(v0, v1);\n"
        "    const fe_t v4 = sqr_ni<F>(v3);\n"
        "    const fe_t v5 = F::mul(U[0], v2);\n"
        "    return F::sub(F::add(v4, v5), F::dbl(F::neg(v1)));\n}\n"
        // and one in sweep form: affine advice leaf, called 9 x 29-bit multipliers, lazy add / sub, closing product
        "template <class F> __device__ constexpr uint32_t jit_fn_sweep_one() { return 1u; }\n"
        "template <class F>\n__device__ __forceinline__ void jit_fn_sweep(const RowCtx &C, uint32_t row, uint32_t npts, const fe_t *__restrict__ Uall, "
        "uint32_t nu, uint32_t *__restrict__ acc, bool accumulate) {\n"
        "    using G = Fp29<typename F::Params>;\n    const uint32_t mask = C.rows - 1;\n"
        "    const fe_t l0 = ld_fix<F>(C, 0, (row + 0u) & mask);\n"
        "    fe_t l1, s1; adv_affine<F>(C, 0, (row + 1u) & mask, l1, s1);\n"
        "    for (uint32_t pt = 0; pt < npts; ++pt) {\n"
        "        const fe_t *__restrict__ U = Uall + (size_t)pt * nu;\n"
        "        const f29_t x0 = G::unpack(l0), x1 = G::unpack(l1);\n"
        "        const f29_t x2 = mul29_ni<F>(x0, x1);\n"
        "        const f29_t x3 = sqr29_ni<F>(x2);\n"
        "        const f29_t x4 = G::normalize(G::template sub_lazy<3, 0>(G::normalize(G::add_lazy(x3, x2)), x3));\n"
        "        const f29_t x5 = G::mul(x4, G::unpack(U[0]));\n"
        "        sw_store(acc, pt, accumulate ? G::normalize(G::add_lazy(sw_load(acc, pt), x5)) : x5);\n"
        "        l1 = F::add(l1, s1);\n    }\n}\n";
    for (int field = 0; field < 2; ++field) {
        size_t bytes = 0;
        if (!jit::compile_only(jit_translation_unit(fn, field, true), &bytes, log)) return false;
        if (code_bytes) *code_bytes = bytes;
    }
    return true;
}

struct Structure {
    int field = 0;
    uint32_t k = 0;
    size_t rows = 0, num_selectors = 0, num_fixed = 0, num_advice = 0;
    size_t s_num_challenges = 0;   // PlonkStructure::num_challenges (compressed().num_challenges())
    size_t h_num_challenges = 0;   // homogeneous().num_challenges()  (challenge i folds with i + this)
    size_t degree = 0;             // homogeneous degree = number of cross terms
    Program cross;                 // homogeneous expression, fold mode
    Program plain_compressed;      // compressed expression, single witness (decider, plonk/mod.rs:328)
    Program plain_homogeneous;     // homogeneous expression, single witness (decider, sangria/mod.rs:351)
    size_t num_lookups = 0;        // lookup arguments (src/plonk/lookup.rs:72-82); 5 fold variables each
    bool has_vector_lookup = false;
    std::vector<Program> lookup_progs;   // lookup_polys L_i then table_polys T_i (LookupEvalDomain: advice columns, challenges = [r])
    std::vector<Program> gate_progs;   // S.gates one by one (ProtoGalaxy leaves, plonk/mod.rs:697-701)
    int pg_spec_id = -1;           // ahead-of-time specialised leaf kernel for this gate set, or -1
    size_t max_gate_degree = 0;    // max_i gates[i].degree()  (get_points_count, poly/mod.rs:535-545)
    std::vector<fe_t> vinv_g, vinv_g1;   // compute_G at integer points: inverse Vandermonde of the nodes 0..d_G (rows 1..d_G) / 1..d_G+1 (all rows)
    GateProg *d_gate_progs = nullptr;
    std::vector<fe_t> vinv;        // [degree][degree+1]
    // device data
    uint8_t **d_sel_ptrs = nullptr;
    fe_t **d_fix_ptrs = nullptr;
    std::vector<void *> owned;
    fe_t *d_vinv = nullptr, *d_vinv29 = nullptr;
    fe_t *d_kM[2] = {nullptr, nullptr};      // pg_K_from_G_device: the full inverse Vandermonde matrix of the nodes 0..d_G / 1..d_G+1 (made on first use)
    Arena arena;
    std::vector<uint8_t> host_stage;   // source of the per-call staging copy (must outlive the asynchronous copy)
    uint32_t shard_rank = 0, shard_world = 1;   // cross terms: evaluate only this rank's row stripes (set_shard)
    int32_t min_rot = 0, max_rot = 0;           // range of the rotations of every column query in the gates / lookup expressions
};

static bool build_program(const Ast &ast, int root, const FieldOps &f, const Ctx &ctx, bool fold_mode, Program &p,
                          std::string &err) {
    Compiler c(ast, f, ctx, fold_mode);
    Val v = c.walk(root);
    if (!c.err.empty()) { err = c.err; return false; }
    int result_vreg = -1;
    uint32_t result_uniform = 0;
    if (v.cls == 2) result_vreg = v.id; else result_uniform = UNIFORM_BIT | (uint32_t)c.as_uniform(v);
    p.uops = c.uops;
    allocate(c.vins, c.nvreg, result_vreg, p.insns, p.result, p.nslots);
    if (result_vreg < 0) p.result = result_uniform;
    p.vins = c.vins;
    p.result_vreg = result_vreg;
    p.fingerprint = fingerprint_of(p);
    p.spec_id = -1;
    for (size_t i = 0; i < sizeof(kSpecs) / sizeof(kSpecs[0]); ++i)
        if (kSpecs[i].fingerprint == p.fingerprint && kSpecs[i].id >= 0) p.spec_id = kSpecs[i].id;
    plan_sweep(p, f);               // appends coefficient entries to p.uops (the fingerprint above is that of the plain program)
    if (!p.sweep_ok) p.spec_id = -1;   // the ahead-of-time kernels ARE the sweep form
    return true;
}

// inverse Vandermonde for the points 0..d: vinv[(k-1)*(d+1) + j] = coefficient of X^k in L_j(X)
static std::vector<fe_t> inverse_vandermonde(const FieldOps &f, size_t d) {
    size_t m = d + 1;
    std::vector<fe_t> out(d * m);
    for (size_t j = 0; j < m; ++j) {
        std::vector<fe_t> poly(1, f.one());        // prod_{t != j} (X - t)
        fe_t denom = f.one();
        for (size_t t = 0; t < m; ++t) {
            if (t == j) continue;
            fe_t ft = f.from_u64(t);
            std::vector<fe_t> nx(poly.size() + 1, f.zero());
            for (size_t i = 0; i < poly.size(); ++i) {
                nx[i + 1] = f.add(nx[i + 1], poly[i]);
                nx[i] = f.sub(nx[i], f.mul(poly[i], ft));
            }
            poly.swap(nx);
            denom = f.mul(denom, f.sub(f.from_u64(j), ft));
        }
        fe_t di = f.inv(denom);
        for (size_t k = 1; k <= d; ++k) out[(k - 1) * m + j] = f.mul(poly[k], di);
    }
    return out;
}

// the same for the nodes first .. first + d, all rows: out[k * (d + 1) + j] = coefficient of X^k in L_j(X), k = 0..d
static std::vector<fe_t> inverse_vandermonde_at(const FieldOps &f, size_t d, uint64_t first) {
    const size_t m = d + 1;
    std::vector<fe_t> out(m * m);
    for (size_t j = 0; j < m; ++j) {
        std::vector<fe_t> poly(1, f.one());        // prod_{t != j} (X - x_t)
        fe_t denom = f.one();
        for (size_t t = 0; t < m; ++t) {
            if (t == j) continue;
            const fe_t ft = f.from_u64(first + t);
            std::vector<fe_t> nx(poly.size() + 1, f.zero());
            for (size_t i = 0; i < poly.size(); ++i) {
                nx[i + 1] = f.add(nx[i + 1], poly[i]);
                nx[i] = f.sub(nx[i], f.mul(poly[i], ft));
            }
            poly.swap(nx);
            denom = f.mul(denom, f.sub(f.from_u64(first + j), ft));
        }
        const fe_t di = f.inv(denom);
        for (size_t k = 0; k <= d; ++k) out[k * m + j] = f.mul(poly[k], di);
    }
    return out;
}

static void upload_program(Program &p, Structure &S) {
    if (p.insns.empty()) return;
    SRS_HIP_CHECK(hipMalloc((void **)&p.d_insns, p.insns.size() * sizeof(Insn)));
    S.owned.push_back(p.d_insns);
    SRS_HIP_CHECK(hipMemcpy(p.d_insns, p.insns.data(), p.insns.size() * sizeof(Insn), hipMemcpyHostToDevice));
}

Structure *create(int field, uint32_t k, size_t num_selectors, size_t num_fixed, size_t num_advice,
                  const uint8_t *const *selectors, const fe_t *const *fixed, int space_device,
                  const uint64_t *gates, size_t gates_words, size_t num_gates, size_t num_lookups, bool has_vector_lookup,
                  const uint64_t *lookup_exprs, size_t lookup_words, int &rc, std::string &err) {
    rc = 4;
    FieldOps f{field};
    Ast ast;
    std::vector<int> roots, lroots;
    if (!parse_gates(gates, gates_words, num_gates, ast, roots, err)) return nullptr;
    if (num_lookups && !parse_gates(lookup_exprs, lookup_words, 2 * num_lookups, ast, lroots, err)) return nullptr;
    if (!num_lookups && has_vector_lookup) { err = "has_vector_lookup without lookups"; return nullptr; }
    int32_t min_rot = 0, max_rot = 0;
    for (const Node &nd : ast.n)
        if (nd.kind == N_POLY) { min_rot = std::min(min_rot, nd.rot); max_rot = std::max(max_rot, nd.rot); }
    if (num_selectors + num_fixed == 0) { err = "Fixed & Selectors can't be empty in one time"; return nullptr; }   // eval.rs:47-54
    // destroy(), not delete: an exception half-way (e.g. hipMalloc of the fixed columns at a large k) must free the device
    // allocations and the run-time compiled module the structure already owns
    std::unique_ptr<Structure, void (*)(Structure *)> S(new Structure(), &destroy);
    S->field = field;
    S->k = k;
    S->rows = (size_t)1 << k;
    S->num_selectors = num_selectors;
    S->num_fixed = num_fixed;
    S->num_advice = num_advice;
    S->num_lookups = num_lookups;
    S->has_vector_lookup = has_vector_lookup;
    S->min_rot = min_rot;
    S->max_rot = max_rot;
    // ConstraintSystemMetainfo::build: the gate-compression challenge comes after the lookup challenges
    // (r1 [, r2]), i.e. ctx.num_challenges starts at 2 / 1 / 0
    // (src/table/constraint_system_metainfo.rs:81-97) -> CompressedGates::new (src/plonk/mod.rs:84-107)
    Ctx ctx{num_selectors, num_fixed, num_advice, has_vector_lookup ? (size_t)2 : (num_lookups ? (size_t)1 : (size_t)0)};
    ctx.num_lookups = num_lookups;
    int compressed = compress(ast, roots, ctx.num_challenges, f);
    ctx.num_challenges = num_challenges(ast, compressed);
    S->s_num_challenges = ctx.num_challenges;
    int homog;
    size_t degree;
    if (!homogeneous(ast, compressed, ctx, homog, degree, err)) { rc = 7; return nullptr; }
    S->h_num_challenges = num_challenges(ast, homog);
    S->degree = degree;
    if (degree > DEGREE_LIMIT) { err = "folding degree " + std::to_string(degree) + " exceeds the supported maximum 255"; return nullptr; }
    if (!build_program(ast, homog, f, ctx, true, S->cross, err) ||
        !build_program(ast, compressed, f, ctx, false, S->plain_compressed, err) ||
        !build_program(ast, homog, f, ctx, false, S->plain_homogeneous, err)) {
        rc = 7;
        return nullptr;
    }
    if (degree) S->vinv = inverse_vandermonde(f, degree);
    S->gate_progs.resize(roots.size());
    for (size_t g = 0; g < roots.size(); ++g) {
        if (!build_program(ast, roots[g], f, ctx, false, S->gate_progs[g], err)) { rc = 7; return nullptr; }
        S->max_gate_degree = std::max(S->max_gate_degree, expr_degree(ast, roots[g], ctx));
    }
    if (field == 0 && S->max_gate_degree >= 1 && S->max_gate_degree <= 64) {     // 6 inversions each: once per structure, not per prove
        S->vinv_g = inverse_vandermonde(f, S->max_gate_degree);
        S->vinv_g1 = inverse_vandermonde_at(f, S->max_gate_degree, 1);
    }
    for (size_t e = 0; e < sizeof(kPgSpecs) / sizeof(kPgSpecs[0]); ++e) {
        if ((size_t)kPgSpecs[e].n_gates != S->gate_progs.size()) continue;
        bool same = true;
        for (size_t g = 0; g < S->gate_progs.size(); ++g) same = same && kPgSpecs[e].fp[g] == S->gate_progs[g].fingerprint;
        for (size_t g = 0; g < S->gate_progs.size(); ++g) same = same && S->gate_progs[g].sweep_ok;      // the leaf kernels use the sweep form
        if (same) S->pg_spec_id = kPgSpecs[e].id;
    }
    // lookup / table polynomials see the advice COLUMNS only (LookupEvalDomain, src/plonk/eval.rs:106-134)
    {
        Ctx lctx{num_selectors, num_fixed, num_advice, 0};
        S->lookup_progs.resize(lroots.size());
        for (size_t i = 0; i < lroots.size(); ++i)
            if (!build_program(ast, lroots[i], f, lctx, false, S->lookup_progs[i], err)) { rc = 7; return nullptr; }
    }
    // ---- no ahead-of-time kernel for this gate set: compile the cross-term program now (jit.hip).  Worth it from 2^14
    //      rows on (one hiprtc compile ~ a second); single-pass degrees only (the kernel body parks d + 1 <= 9 points).
    if (S->cross.spec_id < 0 && S->degree >= 1 && S->degree <= DMAX && !S->cross.insns.empty() && jit::enabled() &&
        (k >= 14 || tuning::get_or(tuning::JIT_ALWAYS, 0) != 0)) {
        const bool shared = true;
        const std::string src = jit_translation_unit(emit_spec_source(S->cross, "jit_fn", shared) + emit_sweep_source(S->cross, "jit_fn_sweep", shared),
                                                     field, S->cross.sweep_ok);
        std::string log;
        (void)jit::compile(src, "srs_jit_rowprog", S->cross.jit, log);      // on failure the structure stays on the interpreter (srs_structure_jit_info reports which)
    }
    // ---- device residency: programs, fixed columns, selectors
    rc = 5;
    for (auto &lp : S->lookup_progs) upload_program(lp, *S);
    upload_program(S->cross, *S);
    upload_program(S->plain_compressed, *S);
    upload_program(S->plain_homogeneous, *S);
    for (auto &gp : S->gate_progs) upload_program(gp, *S);
    if (!S->vinv.empty()) {
        SRS_HIP_CHECK(hipMalloc((void **)&S->d_vinv, S->vinv.size() * sizeof(fe_t)));
        S->owned.push_back(S->d_vinv);
        SRS_HIP_CHECK(hipMemcpy(S->d_vinv, S->vinv.data(), S->vinv.size() * sizeof(fe_t), hipMemcpyHostToDevice));
        std::vector<fe_t> v29(S->vinv);                        // * 2^5: operands of the 2^261-radix multiplier (sweep_kernel_body)
        for (auto &x : v29)
            for (int k = 0; k < 5; ++k) x = f.add(x, x);
        SRS_HIP_CHECK(hipMalloc((void **)&S->d_vinv29, v29.size() * sizeof(fe_t)));
        S->owned.push_back(S->d_vinv29);
        SRS_HIP_CHECK(hipMemcpy(S->d_vinv29, v29.data(), v29.size() * sizeof(fe_t), hipMemcpyHostToDevice));
    }
    const hipMemcpyKind kind = space_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    std::vector<uint8_t *> selp(num_selectors);
    std::vector<fe_t *> fixp(num_fixed);
    for (size_t i = 0; i < num_selectors; ++i) {
        SRS_HIP_CHECK(hipMalloc((void **)&selp[i], S->rows));
        S->owned.push_back(selp[i]);
        SRS_HIP_CHECK(hipMemcpy(selp[i], selectors[i], S->rows, kind));
    }
    for (size_t i = 0; i < num_fixed; ++i) {
        SRS_HIP_CHECK(hipMalloc((void **)&fixp[i], S->rows * sizeof(fe_t)));
        S->owned.push_back(fixp[i]);
        SRS_HIP_CHECK(hipMemcpy(fixp[i], fixed[i], S->rows * sizeof(fe_t), kind));
    }
    SRS_HIP_CHECK(hipMalloc((void **)&S->d_sel_ptrs, (num_selectors + 1) * sizeof(void *)));
    S->owned.push_back(S->d_sel_ptrs);
    SRS_HIP_CHECK(hipMalloc((void **)&S->d_fix_ptrs, (num_fixed + 1) * sizeof(void *)));
    S->owned.push_back(S->d_fix_ptrs);
    if (num_selectors) SRS_HIP_CHECK(hipMemcpy(S->d_sel_ptrs, selp.data(), num_selectors * sizeof(void *), hipMemcpyHostToDevice));
    if (num_fixed) SRS_HIP_CHECK(hipMemcpy(S->d_fix_ptrs, fixp.data(), num_fixed * sizeof(void *), hipMemcpyHostToDevice));
    rc = 0;
    return S.release();
}

void destroy(Structure *S) {
    if (!S) return;
    jit::release(S->cross.jit);
    for (void *p : S->owned) (void)hipFree(p);
    S->arena.release();
    delete S;
}

static uint32_t shard_local_rows(size_t rows, uint32_t rank, uint32_t world) {     // rows of this rank's block-cyclic stripes
    if (world <= 1) return (uint32_t)rows;
    const size_t Sz = (size_t)1 << ROW_STRIPE_LOG, full = rows >> ROW_STRIPE_LOG, rem = rows & (Sz - 1);
    size_t cnt = (full / world) * Sz;
    if (rank < full % world) cnt += Sz;
    if (rank == full % world) cnt += rem;
    return (uint32_t)cnt;
}
void set_shard(Structure *S, uint32_t rank, uint32_t world) {
    S->shard_rank = rank;
    S->shard_world = world ? world : 1;
}
uint32_t shard_world(const Structure *S) { return S->shard_world; }
uint32_t shard_rank(const Structure *S) { return S->shard_rank; }
void rotation_range(const Structure *S, int32_t *lo, int32_t *hi) { *lo = S->min_rot; *hi = S->max_rot; }
uint32_t log_rows(const Structure *S) { return S->k; }
size_t degree(const Structure *S) { return S->degree; }
size_t num_challenges(const Structure *S) { return S->s_num_challenges; }
size_t num_advice(const Structure *S) { return S->num_advice; }
size_t num_witness_columns(const Structure *S) { return S->num_advice + 5 * S->num_lookups; }
size_t num_lookups(const Structure *S) { return S->num_lookups; }
size_t rows(const Structure *S) { return S->rows; }
int field(const Structure *S) { return S->field; }

// evaluate the uniform program for one point: challenge i -> ch[i] + pt * ch[i + fold_offset]
static bool eval_uniform(const Program &p, const FieldOps &f, const fe_t *ch, size_t n_ch, size_t fold_offset, bool fold,
                         uint32_t pt, fe_t *out, std::string &err) {
    fe_t fpt = f.from_u64(pt);
    for (size_t i = 0; i < p.uops.size(); ++i) {
        const UOp &u = p.uops[i];
        switch (u.op) {
        case 0: out[i] = u.c; break;
        case 1: {
            size_t a = (size_t)u.chal;
            if (a >= n_ch) { err = "challenge index " + std::to_string(a) + " out of boundary " + std::to_string(n_ch); return false; }
            out[i] = ch[a];
            if (fold) {
                size_t b = a + fold_offset;
                if (b >= n_ch) { err = "challenge index " + std::to_string(b) + " out of boundary " + std::to_string(n_ch); return false; }
                if (pt) out[i] = f.add(out[i], f.mul(fpt, ch[b]));
            }
            break;
        }
        case 2: out[i] = f.add(out[u.a], out[u.b]); break;
        case 3: out[i] = f.sub(out[u.a], out[u.b]); break;
        case 4: out[i] = f.mul(out[u.a], out[u.b]); break;
        case 6: {                                     // u[a] * 2^(5 b): operands of the 2^261-radix multiplier (sweep form)
            fe_t x = out[u.a];
            for (int k = 0; k < 5 * (u.b < 0 ? -u.b : u.b); ++k) x = u.b < 0 ? f.halve(x) : f.add(x, x);
            out[i] = x;
            break;
        }
        default: out[i] = f.neg(out[u.a]); break;
        }
    }
    return true;
}

template <class F>
static void launch_rowprog(const DevArgs &A, uint32_t nslots, hipStream_t st) {
    uint32_t blocks = (A.ctx.local_rows + RP_THREADS - 1) / RP_THREADS;
    if (nslots <= 8) SRS_LAUNCH((k_rowprog<F, 8>), (blocks), (RP_THREADS), 0, st, A);
    else if (nslots <= 10) SRS_LAUNCH((k_rowprog<F, 10>), (blocks), (RP_THREADS), 0, st, A);
    else if (nslots <= 12) SRS_LAUNCH((k_rowprog<F, 12>), (blocks), (RP_THREADS), 0, st, A);
    else if (nslots <= 16) SRS_LAUNCH((k_rowprog<F, 16>), (blocks), (RP_THREADS), 0, st, A);
    else if (nslots <= 24) SRS_LAUNCH((k_rowprog<F, 24>), (blocks), (RP_THREADS), 0, st, A);
    else SRS_LAUNCH((k_rowprog<F, 32>), (blocks), (RP_THREADS), 0, st, A);
}

// 0 interpreter, 1 ahead-of-time specialised kernel, 2 compiled at run time (include/sirius_amd.h SRS_KERNEL_*)
int kernel_kind(const Structure *S, int which) {
    if (which == 2) return S->pg_spec_id >= 0 ? 1 : 0;
    const Program &p = which == 0 ? S->cross : S->plain_compressed;
    if (p.spec_id >= 0) return 1;
    if (p.jit.function) return 2;
    return 0;
}

const char *spec_source(Structure *S, int which, uint64_t *fingerprint, int *spec_id, std::string &buf) {
    if (which >= 3 && (size_t)(which - 3) >= S->gate_progs.size()) { buf.clear(); return buf.c_str(); }
    Program &p = which >= 3 ? S->gate_progs[which - 3] : (which == 0 ? S->cross : (which == 1 ? S->plain_compressed : S->plain_homogeneous));
    buf = emit_spec_source(p, "spec_fn", false);   // ahead-of-time generation: inlined multipliers
    buf += emit_sweep_source(p, "spec_fn_sweep", false);  // empty when the program has no sweep form
    if (fingerprint) *fingerprint = p.fingerprint;
    if (spec_id) *spec_id = p.spec_id;
    if (spec_id && p.spec_id < 0 && p.jit.function) *spec_id = -2;      // compiled at run time
    return buf.c_str();
}

// mode 0: cross terms (needs W2), outputs `degree` vectors; mode 1/2: plain evaluation of the
// compressed / homogeneous expression on W1, one output vector.
static int evaluate_prog(Structure *S, Program &p, int mode, const fe_t *W1_dev, const fe_t *W2_dev, const fe_t *challenges_host,
                         size_t n_ch, fe_t *const *out_dev_ptrs_host, hipStream_t st, std::string &err, bool sync = true);
// sync = false: the kernels are only enqueued; the caller synchronises `st` before the next call on this structure
// (the uniform tables live in the structure's arena).  Used when an MSM over the outputs follows on the same stream.
int evaluate(Structure *S, int mode, const fe_t *W1_dev, const fe_t *W2_dev, const fe_t *challenges_host, size_t n_ch,
             fe_t *const *out_dev_ptrs_host, hipStream_t st, std::string &err, bool sync) {
    Program &p = mode == 0 ? S->cross : (mode == 1 ? S->plain_compressed : S->plain_homogeneous);
    return evaluate_prog(S, p, mode, W1_dev, W2_dev, challenges_host, n_ch, out_dev_ptrs_host, st, err, sync);
}
static int evaluate_prog(Structure *S, Program &p, int mode, const fe_t *W1_dev, const fe_t *W2_dev, const fe_t *challenges_host,
                         size_t n_ch, fe_t *const *out_dev_ptrs_host, hipStream_t st, std::string &err, bool sync) {
    FieldOps f{S->field};
    const uint32_t d = mode == 0 ? (uint32_t)S->degree : 0;
    const uint32_t npts = mode == 0 ? d + 1 : 1;
    const uint32_t nout = mode == 0 ? d : 1;
    if (mode == 0 && d == 0) return 0;
    if (p.nslots > 32) { err = "row program needs more than 32 live registers"; return 4; }
    const size_t nu = p.uops.size() ? p.uops.size() : 1;
    std::vector<fe_t> utab(nu * npts);
    for (uint32_t pt = 0; pt < npts; ++pt)
        if (!eval_uniform(p, f, challenges_host, n_ch, S->h_num_challenges, mode == 0, pt, utab.data() + (size_t)pt * nu, err)) return 7;
    // uniform tables and output pointers travel in ONE host-to-device copy (each small pageable copy costs ~50 us of
    // host time, during which the GPU idles)
    Arena &A = S->arena;
    const size_t utab_bytes = utab.size() * sizeof(fe_t), stage_bytes = utab_bytes + nout * sizeof(void *);
    A.reserve(Arena::pad(stage_bytes) + 1024);
    A.reset();
    uint8_t *d_stage = A.take<uint8_t>(stage_bytes);
    fe_t *d_utab = reinterpret_cast<fe_t *>(d_stage);
    fe_t **d_out = reinterpret_cast<fe_t **>(d_stage + utab_bytes);
    S->host_stage.resize(stage_bytes);
    std::memcpy(S->host_stage.data(), utab.data(), utab_bytes);
    std::memcpy(S->host_stage.data() + utab_bytes, out_dev_ptrs_host, nout * sizeof(void *));
    SRS_HIP_CHECK(hipMemcpyAsync(d_stage, S->host_stage.data(), stage_bytes, hipMemcpyHostToDevice, st));
    DevArgs a;
    a.prog = p.d_insns;
    a.n_insn = (uint32_t)p.insns.size();
    a.result = p.result;
    a.ctx.rows = (uint32_t)S->rows;
    a.ctx.sel = S->d_sel_ptrs;
    a.ctx.fix = S->d_fix_ptrs;
    for (uint32_t j = 0; j < JMAX; ++j) a.ctx.W[j] = nullptr;
    a.ctx.W[0] = W1_dev;
    a.ctx.W[1] = W2_dev;
    a.ctx.J = mode == 0 ? 2 : 1;
    a.ctx.wcoef = nullptr;
    a.ctx.half = 0;
    a.ctx.pt0 = 0;
    // only the cross terms are sharded (their consumers, the sharded MSM and the error fold, touch this rank's stripes
    // only); the deciders' plain evaluations always cover every row
    a.ctx.shard_rank = mode == 0 ? S->shard_rank : 0;
    a.ctx.shard_world = mode == 0 ? S->shard_world : 1;
    a.ctx.local_rows = shard_local_rows(S->rows, a.ctx.shard_rank, a.ctx.shard_world);
    a.utab = d_utab;
    a.n_uniform = (uint32_t)nu;
    a.npts = npts;
    // d <= 8: one pass.  Higher folding degrees (many gates compressed with y^(n-1)): every pass evaluates all
    // d + 1 points again and accumulates the next 8 coefficients (a.d = outputs of THIS pass).
    for (uint32_t k0 = 0; k0 < (d ? d : 1); k0 += DMAX) {
        a.d = d ? std::min<uint32_t>(DMAX, d - k0) : 0;
        a.vinv = S->d_vinv ? S->d_vinv + (size_t)k0 * npts : nullptr;
        a.vinv29 = S->d_vinv29 ? S->d_vinv29 + (size_t)k0 * npts : nullptr;
        a.out = d_out + k0;
        prof::Scope ps(mode == 0 ? "rowprog_cross_terms" : "rowprog_eval", st, S->rows);
        if (p.spec_id >= 0) launch_spec(p.spec_id, S->field, a, st);
        else if (p.jit.function) {
            if (!jit::launch(p.jit, (a.ctx.local_rows + RP_THREADS - 1) / RP_THREADS, RP_THREADS, p.sweep_ok ? sweep_smem_bytes(a.npts) : 0u, &a, st)) {
                err = "launch of the run-time compiled row program failed";
                return 5;
            }
        }
        else if (S->field == 0) launch_rowprog<Fr>(a, p.nslots, st); else launch_rowprog<Fq>(a, p.nslots, st);
    }
    if (!sync) return 0;
    SRS_HIP_CHECK(hipStreamSynchronize(st));   // utab / pointer staging lives in the arena
    SRS_HIP_CHECK(hipGetLastError());
    prof::collect();
    return 0;
}

void fold_w(int field, fe_t *out, const fe_t *w1, const fe_t *w2, const fe_t &r, size_t n, hipStream_t st) {
    if (!n) return;
    uint32_t blocks = (uint32_t)std::min<size_t>((n + 255) / 256, 256 * 16);
    if (field == 0) SRS_LAUNCH((k_fold_w<Fr>), (blocks), (256), 0, st, out, w1, w2, r, n);
    else SRS_LAUNCH((k_fold_w<Fq>), (blocks), (256), 0, st, out, w1, w2, r, n);
}

int fold_e(int field, fe_t *out, const fe_t *e, const fe_t *const *t_dev_ptrs_host, size_t n_terms, const fe_t &r, size_t n,
           hipStream_t st, std::string &err) {
    (void)err;
    FieldOps f{field};
    if (!n) return 0;
    uint32_t blocks = (uint32_t)std::min<size_t>((n + 255) / 256, 256 * 16);
    fe_t acc = r;   // r^1, r^2, ...  (accumulator.rs:380-383)
    const fe_t *src = e;
    for (size_t k0 = 0; k0 == 0 || k0 < n_terms; k0 += DMAX) {      // 8 terms per launch; later launches accumulate onto out
        FoldEArgs fa;
        fa.n_terms = (uint32_t)std::min<size_t>(DMAX, n_terms - k0);
        for (uint32_t k = 0; k < DMAX; ++k) {
            fa.t[k] = k < fa.n_terms ? t_dev_ptrs_host[k0 + k] : nullptr;
            fa.rpow[k] = acc;
            if (k < fa.n_terms) acc = f.mul(acc, r);
        }
        if (field == 0) SRS_LAUNCH((k_fold_e<Fr>), (blocks), (256), 0, st, out, src, fa, n);
        else SRS_LAUNCH((k_fold_e<Fq>), (blocks), (256), 0, st, out, src, fa, n);
        src = out;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// ProtoGalaxy host side
// ---------------------------------------------------------------------------------------------
static size_t next_pow2(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }
static uint32_t ilog2(size_t v) { uint32_t l = 0; while (((size_t)1 << (l + 1)) <= v) ++l; return l; }

// PolyContext (src/nifs/protogalaxy/poly/mod.rs:205-269)
bool pg_sizes(const Structure *S, size_t traces_len, PgSizes &o) {
    size_t count = S->rows * S->gate_progs.size();           // get_count_of_valuation :511-516
    if (count == 0) return false;
    o.count_with_padding = next_pow2(count);                   // :518-533
    o.betas_count = ilog2(o.count_with_padding);               // :237-239
    o.points_F = next_pow2(o.betas_count + 1);                 // :241-243
    o.instances_to_fold = traces_len + 1;
    o.points_G = next_pow2(traces_len * S->max_gate_degree + 1);   // get_points_count :535-545
    o.lagrange_domain = ilog2(o.instances_to_fold);            // :253-255
    // Q2: fft_log_domain_size_K returns a COUNT that is then used as a LOG (:263-268, :481)
    size_t c = o.points_G + 1;
    c = c > o.instances_to_fold ? c - o.instances_to_fold : 0;
    o.log_domain_K = (uint32_t)next_pow2(c);
    return true;
}

// iter_eval_lagrange_poly_for_cyclic_group (src/polynomial/lagrange.rs:50-75), Fr, host
std::vector<fe_t> lagrange_eval(const fe_t &X, uint32_t log_n) {
    const size_t n = (size_t)1 << log_n;
    fe_t w = ntt::omega(log_n, false);
    fe_t xn1 = Fr::sub(Fr::pow_u64(X, n), Fr::one());
    std::vector<fe_t> out(n), val(n), den(n), pre(n + 1);
    // one inversion for n and all the X - w^i (Montgomery's trick; a zero difference -- X on the domain -- is left out of the product)
    fe_t value = Fr::one(), acc = Fr::from_u64(n);
    pre[0] = acc;
    for (size_t i = 0; i < n; ++i) {
        val[i] = value;
        den[i] = Fr::sub(X, value);
        if (!Fr::is_zero(den[i])) acc = Fr::mul(acc, den[i]);
        pre[i + 1] = acc;
        value = Fr::mul(value, w);
    }
    fe_t inv = Fr::inv(acc);
    std::vector<fe_t> dinv(n);
    for (size_t i = n; i-- > 0;) {
        if (Fr::is_zero(den[i])) continue;
        dinv[i] = Fr::mul(inv, pre[i]);
        inv = Fr::mul(inv, den[i]);
    }
    const fe_t inv_n = inv;                                   // what is left: 1 / n
    for (size_t i = 0; i < n; ++i) {
        if (Fr::is_zero(den[i])) out[i] = Fr::is_zero(xn1) ? Fr::one() : Fr::zero();     // xn1 == 0 exactly when X is on the domain
        else out[i] = Fr::mul(Fr::mul(val[i], inv_n), Fr::mul(xn1, dinv[i]));
    }
    return out;
}

fe_t poly_eval(const fe_t *c, size_t n, const fe_t &x) {       // UnivariatePoly::eval (univariate.rs:67-75)
    fe_t acc = Fr::zero(), xp = Fr::one();
    for (size_t i = 0; i < n; ++i) {
        acc = Fr::add(acc, Fr::mul(xp, c[i]));
        xp = Fr::mul(xp, x);
    }
    return acc;
}

template <uint32_t NS>
static void launch_pg_leaves(const PgArgs &A, uint32_t tiles, uint32_t gates, uint32_t threads, uint32_t lpt, hipStream_t st) {
    if (lpt == 8) SRS_LAUNCH((k_pg_leaves<Fr, NS, 8>), (tiles, gates), (threads), 0, st, A);
    else SRS_LAUNCH((k_pg_leaves<Fr, NS, 1>), (tiles, gates), (threads), 0, st, A);
}

// mode 0: compute_F, 1: compute_G, 2: evaluate_e.  W_dev: J device witness pointers (J = 1 for F / e).
// challenges_host: J arrays of n_ch challenges.  weights_in: betas (F, e) / betas_stroke (G), betas_count values.
// out_host: points_F / points_G coefficients (after ifft) or the single value e.

int pg_sum(Structure *S, int mode, const fe_t *const *W_dev, const fe_t *const *challenges_host, size_t n_ch, size_t J,
           const fe_t *weights_in, size_t n_weights, const fe_t *delta, int compat, hipStream_t st, fe_t *out_host,
           size_t *n_out, std::string &err, const fe_t *g_at_one, PgGValues *keep_on_device) {
    if (S->field != 0) { err = "ProtoGalaxy polynomials need the 2-adic field bn256::Fr"; return 4; }
    if (J == 0 || J > JMAX) { err = "unsupported number of traces"; return 4; }
    PgSizes sz;
    if (!pg_sizes(S, mode == 1 ? J - 1 : 1, sz)) { *n_out = 0; return 0; }
    if (n_weights < sz.betas_count) { err = "not enough betas"; return 4; }
    FieldOps f{0};
    const uint32_t P_out = mode == 0 ? (uint32_t)sz.points_F : (mode == 1 ? (uint32_t)sz.points_G : 1u);
    // compute_G with one incoming trace (L = 1, Lagrange domain {1, -1}): the folded witness L_0(X) w_0 + L_1(X) w_1 =
    // (w_0 + w_1)/2 + X (w_0 - w_1)/2 is LINEAR in X, so G has degree <= max gate degree d_G.  Instead of the reference's
    // next_pow2(d_G + 1) roots of unity (2 multiplies per advice load to fold the witness, then an ifft) G is evaluated at
    // the integers 0..d_G (fold = halvings + additions, d_G + 1 points) and interpolated with the constant inverse
    // Vandermonde matrix: the same polynomial, hence the same coefficients (exact arithmetic), ~35 % less work.
    const bool g_int = mode == 1 && J == 2 && tuning::get_or(tuning::PG_G_FFT, 0) == 0;      // (tuning pg_g_fft: the L >= 2 route on L = 1, for the equality test)
    const uint32_t dG = (uint32_t)S->max_gate_degree;
    // ... and one of the d_G + 1 values is free for a caller that has F: at X = 1 the fold IS the accumulator (L_0(1) = 1), the
    // weights are betas_stroke = beta + alpha delta^(2^b), so G(1) = sum_i pow_i(betas_stroke) f_i(acc) = F(alpha) -- an
    // identity of the definitions, whatever the traces hold.  ProtoGalaxy::prove has F(alpha) (it is compute_K's first
    // argument): with `g_at_one` the leaves are evaluated at X = 2 .. d_G + 1 only (d_G points), and the interpolation
    // nodes are 1 .. d_G + 1.  Needs the sweep-form leaf kernel (below); otherwise all d_G + 1 points are evaluated.
    const uint32_t lpt_probe = S->k >= 10 ? 8u : 1u;
    const bool skip_one = g_int && g_at_one != nullptr && dG >= 2 && S->pg_spec_id >= 0 && lpt_probe == 8 && dG + 1 <= DMAX + 1 &&
                          true;
    const uint32_t pt0 = skip_one ? 2u : 0u;
    const uint32_t P = g_int ? (skip_one ? dG : dG + 1) : P_out;   // evaluation points actually used
    const uint32_t leaf_pts = mode == 1 ? P : 1u;
    const uint32_t wpts = mode == 0 ? P : 1u;
    const uint32_t levels = (uint32_t)sz.betas_count;
    const uint32_t n_gates = (uint32_t)S->gate_progs.size();
    // ---- weights [levels][wpts]
    std::vector<fe_t> weights((size_t)levels * wpts);
    std::vector<fe_t> pts;                                     // evaluation points X_p (F: w_t'^p ; G: w_G^p)
    if (g_int) {
        for (uint32_t p = 0; p < P; ++p) pts.push_back(Fr::from_u64(pt0 + p));
    } else if (mode != 2) {
        fe_t w = ntt::omega(ilog2(P), false), x = Fr::one();
        for (uint32_t p = 0; p < P; ++p) { pts.push_back(x); x = Fr::mul(x, w); }     // iter_cyclic_subgroup
    }
    if (mode == 0) {
        fe_t d = *delta;                                       // deltas: delta^(2^b)  (:97-99)
        for (uint32_t b = 0; b < levels; ++b) {
            for (uint32_t p = 0; p < P; ++p) weights[(size_t)b * P + p] = Fr::add(weights_in[b], Fr::mul(pts[p], d));   // :102-111
            d = Fr::sqr(d);
        }
    } else {
        for (uint32_t b = 0; b < levels; ++b) weights[b] = weights_in[b];
    }
    // ---- witness combination coefficients and per-point challenges
    std::vector<fe_t> wcoef;                                   // [P][J] = L_j(X_p)   (FoldedWitness::new, folded_witness.rs:20-47)
    std::vector<std::vector<fe_t>> ch_pt(leaf_pts, std::vector<fe_t>(n_ch ? n_ch : 1, Fr::zero()));
    if (mode == 1) {
        wcoef.resize((size_t)P * J);
        // integer-point G folds the witnesses by halvings in the kernel: L_j(X_p) is only needed to fold challenges
        for (uint32_t p = 0; p < P && (!g_int || n_ch); ++p) {
            std::vector<fe_t> L = lagrange_eval(pts[p], (uint32_t)sz.lagrange_domain);
            for (size_t j = 0; j < J; ++j) {
                wcoef[(size_t)p * J + j] = L[j];
                for (size_t c = 0; c < n_ch; ++c)             // fold_plonk_challenges :142-180
                    ch_pt[p][c] = Fr::add(ch_pt[p][c], Fr::mul(challenges_host[j][c], L[j]));
            }
        }
    } else {
        for (size_t c = 0; c < n_ch; ++c) ch_pt[0][c] = challenges_host[0][c];
    }
    // 8 leaves per thread once a gate has >= 1024 rows: 1024-leaf tiles, the first three tree levels in registers
    const uint32_t lpt = S->k >= 10 ? 8u : 1u;
    const uint32_t tile_log = lpt == 8 ? 10u : std::min<uint32_t>(7, S->k);
    // the specialised leaf kernel in sweep form (k_pg_leaves_sweep): integer-point G and evaluate_e of a known gate set
    const bool sweep_leaves = S->pg_spec_id >= 0 && lpt == 8 && mode != 0 && (mode != 1 || g_int) && P <= DMAX + 1;
    // ---- uniform tables per gate / leaf point
    uint32_t max_slots = 1;
    std::vector<GateProg> gp(n_gates);
    std::vector<fe_t> utab;
    for (uint32_t g = 0; g < n_gates; ++g) {
        Program &p = S->gate_progs[g];
        max_slots = std::max(max_slots, p.nslots);
        const size_t nu = p.uops.size() ? p.uops.size() : 1;
        gp[g].prog = p.d_insns;
        gp[g].n_insn = (uint32_t)p.insns.size();
        gp[g].result = p.result;
        gp[g].n_uniform = (uint32_t)nu;
        gp[g].utab_off = (uint32_t)utab.size();
        utab.resize(utab.size() + nu * leaf_pts * (sweep_leaves ? lpt + 1 : 1u));
        for (uint32_t lp = 0; lp < leaf_pts; ++lp)
            if (!eval_uniform(p, f, ch_pt[lp].data(), n_ch, 0, false, 0, utab.data() + gp[g].utab_off + (size_t)lp * nu, err)) return 7;
        if (sweep_leaves) {
            // k_pg_leaves_sweep: one table per leaf slot l of a thread, the term coefficients times w_l = prod_{b in bits(l)} c_(TL + b)
            // table lpt: the SUM of the w_l (reference_compat: the lpt leaves of a thread share one evaluation at row 0)
            const uint32_t TL = tile_log - 3;
            fe_t wsum = Fr::one();
            for (uint32_t l = 1; l <= lpt; ++l) {
                fe_t wl = Fr::one();
                for (uint32_t b = 0; b < 3; ++b)
                    if ((l >> b) & 1u) wl = Fr::mul(wl, weights[(size_t)(TL + b) * wpts]);
                if (l == lpt) wl = wsum; else wsum = Fr::add(wsum, wl);
                fe_t *dst = utab.data() + gp[g].utab_off + (size_t)l * leaf_pts * nu;
                std::memcpy(dst, utab.data() + gp[g].utab_off, (size_t)leaf_pts * nu * sizeof(fe_t));
                for (uint32_t lp = 0; lp < leaf_pts; ++lp)
                    for (int ci : p.sw_coef) dst[(size_t)lp * nu + ci] = Fr::mul(dst[(size_t)lp * nu + ci], wl);
            }
        }
    }
    if (max_slots > 32) { err = "row program needs more than 32 live registers"; return 4; }
    const uint32_t tile = (1u << tile_log) / lpt, tiles_per_gate = (uint32_t)(S->rows >> tile_log);
    const size_t n_tiles_valid = (size_t)n_gates * tiles_per_gate;
    const size_t n_tiles_padded = sz.count_with_padding >> tile_log;
    // ---- compute_F with >= 1024 rows per gate: polynomial tree (see k_pg_F_leaves); the evaluate-and-interpolate route
    //      below stays for small tables and as the cross-check (tuning pg_f_eval = 1)
    if (mode == 0 && lpt == 8 && tuning::get_or(tuning::PG_F_EVAL, 0) == 0) {
        const uint32_t TL = tile_log - 3, T = tile;
        const size_t n0 = n_tiles_padded * T;
        Arena &A = S->arena;
        A.reserve(2 * Arena::pad((4 * n0 + 64) * sizeof(fe_t)) + Arena::pad((utab.size() + 1) * sizeof(fe_t)) +
                  Arena::pad(gp.size() * sizeof(GateProg)) + Arena::pad((n_gates + 1) * sizeof(fe_t)) + 4096);
        A.reset();
        fe_t *d_utab = A.take<fe_t>(utab.size() + 1);
        GateProg *d_gp = A.take<GateProg>(gp.size());
        fe_t *cur = A.take<fe_t>(4 * n0 + 64), *nxt = A.take<fe_t>(4 * n0 + 64);
        fe_t *d_hoist = A.take<fe_t>(n_gates + 1);
        SRS_HIP_CHECK(hipMemcpyAsync(d_utab, utab.data(), utab.size() * sizeof(fe_t), hipMemcpyHostToDevice, st));
        SRS_HIP_CHECK(hipMemcpyAsync(d_gp, gp.data(), gp.size() * sizeof(GateProg), hipMemcpyHostToDevice, st));
        std::vector<fe_t> deltas(levels);
        {
            fe_t d = *delta;                                   // deltas: delta^(2^b)  (:97-99)
            for (uint32_t b = 0; b < levels; ++b) { deltas[b] = d; d = Fr::sqr(d); }
        }
        PgArgs a;
        a.gates = d_gp;
        a.n_gates = n_gates;
        a.log_rows = S->k;
        a.ctx.rows = (uint32_t)S->rows;
        a.ctx.sel = S->d_sel_ptrs;
        a.ctx.fix = S->d_fix_ptrs;
        for (uint32_t j = 0; j < JMAX; ++j) a.ctx.W[j] = j < 1 ? W_dev[j] : nullptr;
        a.ctx.J = 1;
        a.ctx.wcoef = nullptr;
        a.ctx.half = 0;
        a.ctx.pt0 = 0;
        a.ctx.shard_rank = 0;
        a.ctx.shard_world = 1;
        a.ctx.local_rows = a.ctx.rows;
        a.compat = compat;
        a.leaf_pts = 1;
        a.P = 1;
        a.utab = d_utab;
        a.weights = nullptr;
        a.wpts = 1;
        a.tile_log = tile_log;
        a.partial = nullptr;
        a.shard_rank = S->shard_rank;                      // lpt == 8 here: tile_log == ROW_STRIPE_LOG
        a.shard_world = S->shard_world;
        a.hoist = d_hoist;
        a.hoist_mode = 0;
        PgFLevels lv;
        for (uint32_t j = 0; j < 3; ++j) { lv.beta[j] = weights_in[TL + j]; lv.delta[j] = deltas[TL + j]; }
        {
            prof::Scope ps("pg_F_leaves", st, S->rows * n_gates);
            auto leaves = [&](uint32_t tiles) {
                if (S->pg_spec_id == 0) SRS_LAUNCH((k_pg_F_leaves<Fr, 2, 0>), (tiles, n_gates), (T), 0, st, a, lv, cur, (uint32_t)n0);
                else if (max_slots <= 8) SRS_LAUNCH((k_pg_F_leaves<Fr, 8, -1>), (tiles, n_gates), (T), 0, st, a, lv, cur, (uint32_t)n0);
                else if (max_slots <= 16) SRS_LAUNCH((k_pg_F_leaves<Fr, 16, -1>), (tiles, n_gates), (T), 0, st, a, lv, cur, (uint32_t)n0);
                else SRS_LAUNCH((k_pg_F_leaves<Fr, 32, -1>), (tiles, n_gates), (T), 0, st, a, lv, cur, (uint32_t)n0);
            };
            if (compat) {                                  // the gates at row 0 once, then the leaf pass reads them
                a.hoist_mode = 1;
                leaves(1);
                a.hoist_mode = 2;
            }
            leaves(tiles_per_gate);
        }
        // remaining leaf-index bits, in the order adjacent nodes differ: thread bits 0 .. TL-1, then tile / gate bits TL+3 ..
        std::vector<uint32_t> order;
        for (uint32_t b = 0; b < TL; ++b) order.push_back(b);
        for (uint32_t b = TL + 3; b < levels; ++b) order.push_back(b);
        size_t n_in = n0, m_valid = n_tiles_valid * T;
        uint32_t deg = 3;
        size_t at = 0;
        // (dynamic LDS of a launch stays below 48 KiB: degree <= 23 after the launch -- every table size an NTT of <= 2^28 allows; beyond, the r03 flow)
        while (at < order.size() && deg + std::min<size_t>(PG_F_MULTI_LEVELS, order.size() - at) <= 23) {
            // up to six levels per launch (k_pg_F_multi); what is left beyond degree 23: one launch per level + the one-workgroup tail
            const uint32_t nlev = (uint32_t)std::min<size_t>(PG_F_MULTI_LEVELS, order.size() - at);
            PgFMulti tm;
            for (uint32_t l = 0; l < PG_F_MULTI_LEVELS; ++l) {
                tm.beta[l] = l < nlev ? weights_in[order[at + l]] : Fr::zero();
                tm.delta[l] = l < nlev ? deltas[order[at + l]] : Fr::zero();
            }
            const size_t n_out = n_in >> nlev;
            const size_t lds = 2 * (size_t)PG_F_MULTI_NODES * (deg + nlev + 1) * sizeof(fe_t);
            SRS_LAUNCH((k_pg_F_multi<Fr>), ((uint32_t)n_out), (256), lds, st, (const fe_t *)cur, (uint32_t)n_in, (uint32_t)m_valid, deg, nlev, tm, nxt,
                       (uint32_t)n_out);
            n_in = n_out;
            m_valid = (m_valid + (((size_t)1 << nlev) - 1)) >> nlev;
            deg += nlev;
            at += nlev;
            std::swap(cur, nxt);
        }
        for (; at < order.size(); ++at) {
            // the last levels (<= 32 nodes left) run in one workgroup (k_pg_F_tail)
            const size_t left = order.size() - at;
            if (n_in <= 32 && left <= PG_F_TAIL_LEVELS && deg + left <= PG_F_TAIL_MAXDEG) break;
            const uint32_t b = order[at];
            const size_t n_out = n_in / 2;
            SRS_LAUNCH((k_pg_F_level<Fr>), ((uint32_t)((n_out * (deg + 2) + 127) / 128)), (128), 0, st, (const fe_t *)cur, (uint32_t)n_in,
                       (uint32_t)m_valid, deg, weights_in[b], deltas[b], nxt, (uint32_t)n_out);
            n_in = n_out;
            m_valid = (m_valid + 1) / 2;
            ++deg;
            std::swap(cur, nxt);
        }
        if (at < order.size()) {
            const uint32_t nlev = (uint32_t)(order.size() - at);
            PgFTail tl;
            for (uint32_t l = 0; l < PG_F_TAIL_LEVELS; ++l) {
                tl.beta[l] = l < nlev ? weights_in[order[at + l]] : Fr::zero();
                tl.delta[l] = l < nlev ? deltas[order[at + l]] : Fr::zero();
            }
            SRS_LAUNCH((k_pg_F_tail<Fr>), (1), (512), 0, st, (const fe_t *)cur, (uint32_t)n_in, (uint32_t)m_valid, deg, nlev, tl, nxt);
            deg += nlev;
            std::swap(cur, nxt);
        }
        // n_in == 1: cur[m] = coefficient m, m <= deg = levels; the reference's vector has fft_points_count_F entries
        std::vector<fe_t> coef(deg + 1);
        SRS_HIP_CHECK(hipMemcpyAsync(coef.data(), cur, coef.size() * sizeof(fe_t), hipMemcpyDeviceToHost, st));
        SRS_HIP_CHECK(hipStreamSynchronize(st));
        SRS_HIP_CHECK(hipGetLastError());
        prof::collect();
        for (uint32_t m = 0; m < P; ++m) out_host[m] = m < coef.size() ? coef[m] : Fr::zero();
        *n_out = P;
        return 0;
    }
    // ---- device staging
    Arena &A = S->arena;
    size_t need = Arena::pad(weights.size() * sizeof(fe_t)) + Arena::pad((wcoef.size() + 1) * sizeof(fe_t)) +
                  Arena::pad((utab.size() + 1) * sizeof(fe_t)) + Arena::pad(gp.size() * sizeof(GateProg)) +
                  2 * Arena::pad((n_tiles_padded + 1) * P * sizeof(fe_t)) + Arena::pad(sizeof(fe_t) * P) +
                  Arena::pad(((size_t)n_gates * P + 1) * sizeof(fe_t)) + 4096;
    A.reserve(need);
    A.reset();
    fe_t *d_w = A.take<fe_t>(weights.size());
    fe_t *d_coef = A.take<fe_t>(wcoef.size() + 1);
    fe_t *d_utab = A.take<fe_t>(utab.size() + 1);
    GateProg *d_gp = A.take<GateProg>(gp.size());
    fe_t *buf0 = A.take<fe_t>((n_tiles_padded + 1) * P);
    fe_t *buf1 = A.take<fe_t>((n_tiles_padded + 1) * P);
    fe_t *d_hoist = A.take<fe_t>((size_t)n_gates * P + 1);
    SRS_HIP_CHECK(hipMemcpyAsync(d_w, weights.data(), weights.size() * sizeof(fe_t), hipMemcpyHostToDevice, st));
    if (!wcoef.empty()) SRS_HIP_CHECK(hipMemcpyAsync(d_coef, wcoef.data(), wcoef.size() * sizeof(fe_t), hipMemcpyHostToDevice, st));
    SRS_HIP_CHECK(hipMemcpyAsync(d_utab, utab.data(), utab.size() * sizeof(fe_t), hipMemcpyHostToDevice, st));
    SRS_HIP_CHECK(hipMemcpyAsync(d_gp, gp.data(), gp.size() * sizeof(GateProg), hipMemcpyHostToDevice, st));
    PgArgs a;
    a.gates = d_gp;
    a.n_gates = n_gates;
    a.log_rows = S->k;
    a.ctx.rows = (uint32_t)S->rows;
    a.ctx.sel = S->d_sel_ptrs;
    a.ctx.fix = S->d_fix_ptrs;
    for (uint32_t j = 0; j < JMAX; ++j) a.ctx.W[j] = j < J ? W_dev[j] : nullptr;
    a.ctx.J = (uint32_t)(mode == 1 ? J : 1);
    a.ctx.wcoef = (mode == 1 && !g_int) ? d_coef : nullptr;
    a.ctx.half = g_int ? 1u : 0u;
    a.ctx.shard_rank = 0;
    a.ctx.shard_world = 1;
    a.ctx.local_rows = a.ctx.rows;
    a.ctx.pt0 = pt0;
    a.compat = compat;
    a.leaf_pts = leaf_pts;
    a.P = P;
    a.utab = d_utab;
    a.weights = d_w;
    a.wpts = wpts;
    a.tile_log = tile_log;
    a.partial = buf0;
    // partial sums of a sharded structure: by tiles when a tile is a stripe (k >= 10), else rank 0 evaluates everything
    if (tile_log == ROW_STRIPE_LOG) { a.shard_rank = S->shard_rank; a.shard_world = S->shard_world; }
    else if (S->shard_world > 1 && S->shard_rank != 0) { a.shard_rank = 0xFFFFFFFFu; a.shard_world = 2; }     // no tile is mine
    else { a.shard_rank = 0; a.shard_world = 1; }
    {
        prof::Scope ps(mode == 0 ? "pg_F_leaves" : (mode == 1 ? "pg_G_leaves" : "pg_e_leaves"), st, S->rows * n_gates);
        a.hoist = d_hoist;
        a.hoist_mode = 0;
        if (S->pg_spec_id >= 0 && lpt == 8) {
            if (sweep_leaves && compat) {     // k_pg_leaves_sweep<.., COMPAT>: the gates at row 0 once, then the leaf pass
                a.hoist_mode = 1;
                launch_pg_spec(S->pg_spec_id, a, P, n_gates, tile, st, sweep_leaves);      // one workgroup per (point, gate)
                a.hoist_mode = 2;
            }
            launch_pg_spec(S->pg_spec_id, a, tiles_per_gate, n_gates, tile, st, sweep_leaves);
        }
        else {
            auto leaves = [&](uint32_t tiles) {
                if (max_slots <= 8) launch_pg_leaves<8>(a, tiles, n_gates, tile, lpt, st);
                else if (max_slots <= 12) launch_pg_leaves<12>(a, tiles, n_gates, tile, lpt, st);
                else if (max_slots <= 16) launch_pg_leaves<16>(a, tiles, n_gates, tile, lpt, st);
                else launch_pg_leaves<32>(a, tiles, n_gates, tile, lpt, st);
            };
            if (compat) {                      // the interpreter too evaluates the gates at row 0 once per (gate, point), then runs the weighted trees
                a.hoist_mode = 1;
                leaves(leaf_pts);
                a.hoist_mode = 2;
            }
            leaves(tiles_per_gate);
        }
    }
    // ---- upper levels of the tree over the tile partials (zero padding beyond the real gates)
    size_t m_valid = n_tiles_valid, m = n_tiles_padded;
    uint32_t level0 = tile_log;
    fe_t *cur = buf0, *nxt = buf1;
    while (m > 1) {
        uint32_t lv = std::min<uint32_t>(7, ilog2(m));
        uint32_t outs = (uint32_t)(m >> lv);
        if (((size_t)P << lv) <= PG_REDUCE_PAR_THREADS && (((size_t)P << lv) % 64 == 0 || outs == 1))
            SRS_LAUNCH((k_pg_reduce_par<Fr>), (outs), (1u << lv, P), 0, st, (const fe_t *)cur, (uint32_t)m_valid, P, (const fe_t *)d_w, wpts,
                       level0, lv, nxt);
        else
        SRS_LAUNCH((k_pg_reduce<Fr>), (outs), (1u << lv), 0, st, (const fe_t *)cur, (uint32_t)m_valid, P, (const fe_t *)d_w, wpts,
                   level0, lv, nxt);
        level0 += lv;
        m = outs;
        m_valid = outs;
        std::swap(cur, nxt);
    }
    if (g_int && keep_on_device && dG + 1 <= PG_K_SMALL_MAX_NODES) {      // the caller goes on to K on the device (pg_K_from_G_device): no round trip through the host here
        keep_on_device->vals_dev = cur;
        keep_on_device->n_dev = P;
        keep_on_device->degree = dG;
        keep_on_device->skip_one = skip_one;
        *n_out = 0;
        return 0;
    }
    if (g_int) {
        std::vector<fe_t> val(P);
        SRS_HIP_CHECK(hipMemcpyAsync(val.data(), cur, (size_t)P * sizeof(fe_t), hipMemcpyDeviceToHost, st));
        SRS_HIP_CHECK(hipStreamSynchronize(st));
        SRS_HIP_CHECK(hipGetLastError());
        prof::collect();
        for (uint32_t m = 0; m < P_out; ++m) out_host[m] = Fr::zero();
        if (skip_one) {                                          // values at the nodes 1 .. dG + 1: G(1) = F(alpha), then the evaluated ones
            std::vector<fe_t> at(dG + 1);
            at[0] = *g_at_one;
            for (uint32_t j = 0; j < P; ++j) at[j + 1] = val[j];
            const std::vector<fe_t> vall = !S->vinv_g1.empty() ? S->vinv_g1 : inverse_vandermonde_at(f, dG, 1);   // [dG + 1][dG + 1], rows k = 0..dG
            for (uint32_t k = 0; k <= dG && k < P_out; ++k) {
                fe_t acc = Fr::zero();
                for (uint32_t j = 0; j <= dG; ++j) acc = Fr::add(acc, Fr::mul(vall[(size_t)k * (dG + 1) + j], at[j]));
                out_host[k] = acc;
            }
            *n_out = P_out;
            return 0;
        }
        const std::vector<fe_t> vinv = !S->vinv_g.empty() ? S->vinv_g : (dG ? inverse_vandermonde(f, dG) : std::vector<fe_t>());    // [dG][dG + 1], rows k = 1..dG
        out_host[0] = val[0];                                                                       // G(0)
        for (uint32_t k = 1; k <= dG && k < P_out; ++k) {
            fe_t acc = Fr::zero();
            for (uint32_t j = 0; j < P; ++j) acc = Fr::add(acc, Fr::mul(vinv[(size_t)(k - 1) * P + j], val[j]));
            out_host[k] = acc;
        }
        *n_out = P_out;
        return 0;
    }
    if (mode != 2 && P > 1) ntt::run(cur, ilog2(P), P, 1, true, false, st);      // fft::ifft(&mut points)  (:197, :419)
    SRS_HIP_CHECK(hipMemcpyAsync(out_host, cur, (size_t)P * sizeof(fe_t), hipMemcpyDeviceToHost, st));
    SRS_HIP_CHECK(hipStreamSynchronize(st));
    SRS_HIP_CHECK(hipGetLastError());
    prof::collect();
    *n_out = P;
    return 0;
}

// compute_K_from_G (src/nifs/protogalaxy/poly/mod.rs:475-509)
int pg_K_from_G(const fe_t *polyG_host, size_t nG, const fe_t &f_alpha, size_t instances_to_fold, uint32_t log_domain_K,
                hipStream_t st, fe_t *out_host, std::string &err) {
    if (log_domain_K > ntt::FR_S) { err = "k=" + std::to_string(log_domain_K) + " should no larger than F::S=28"; return 3; }   // fft.rs:13
    const size_t count = (size_t)1 << log_domain_K;
    static thread_local ThreadArena scratch;                         // grow-only: no hipMalloc / hipFree (implicit sync) per call
    scratch.reserve(Arena::pad((nG + 1) * sizeof(fe_t)) + Arena::pad(count * sizeof(fe_t)) + Arena::pad(sizeof(int)) + 256);
    scratch.reset();
    fe_t *d_g = scratch.take<fe_t>(nG + 1), *d_out = scratch.take<fe_t>(count);
    int *d_err = scratch.take<int>(1);
    int herr = 0;
    if (count <= 4096) {
        // The K domain is tiny (256 points in every configuration, quirk Q2): the points are computed on the host with ONE
        // inversion for all denominators (Montgomery's trick).  A GPU thread per point spends two Fermat inversions =
        // 760 dependent multiplications = 0.6 ms of pure latency on it.
        // Everything that depends on the domain only -- the points X_i = zeta omega^i, 1 / Z(X_i) and L_0(X_i) = Z(X_i) / (n (X_i - 1))
        // (lagrange.rs:50-75) -- is computed once per (domain, n) and kept: two inversions and 2 x 256 products less per prove.
        struct KDomain { std::vector<fe_t> xs, l0, inv_z; bool zero_z = false; };
        static std::mutex mu;
        static std::map<std::pair<uint32_t, size_t>, KDomain> cache;
        const KDomain *dom;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = cache.find({log_domain_K, instances_to_fold});
            if (it == cache.end()) {
                KDomain d;
                const fe_t zeta = ntt::zeta(), omega = ntt::omega(log_domain_K, false), one = Fr::one();
                const fe_t inv_n = Fr::inv(Fr::from_u64(instances_to_fold));
                d.xs.resize(count); d.l0.resize(count); d.inv_z.resize(count);
                std::vector<fe_t> xn1(count), xm1(count), pref(2 * count);
                fe_t X = zeta, acc = one;
                for (size_t i = 0; i < count; ++i) { d.xs[i] = X; X = Fr::mul(X, omega); }
                for (size_t i = 0; i < count; ++i) {
                    xn1[i] = Fr::sub(Fr::pow_u64(d.xs[i], instances_to_fold), one);
                    xm1[i] = Fr::sub(d.xs[i], one);
                    if (Fr::is_zero(xn1[i])) d.zero_z = true;                                  // X = 1 included (then X - 1 = 0 too)
                }
                if (!d.zero_z) {
                    for (size_t i = 0; i < count; ++i) {
                        pref[2 * i] = acc;
                        acc = Fr::mul(acc, xn1[i]);
                        pref[2 * i + 1] = acc;
                        acc = Fr::mul(acc, xm1[i]);
                    }
                    fe_t inv = Fr::inv(acc);
                    for (size_t i = count; i-- > 0;) {
                        const fe_t inv_xm1 = Fr::mul(inv, pref[2 * i + 1]);
                        inv = Fr::mul(inv, xm1[i]);
                        d.inv_z[i] = Fr::mul(inv, pref[2 * i]);
                        inv = Fr::mul(inv, xn1[i]);
                        d.l0[i] = Fr::mul(inv_n, Fr::mul(xn1[i], inv_xm1));
                    }
                }
                it = cache.emplace(std::make_pair(log_domain_K, instances_to_fold), std::move(d)).first;
            }
            dom = &it->second;
        }
        if (dom->zero_z) { err = "Z(X) must be not equal to 0"; return 4; }
        std::vector<fe_t> kp(count);
        // 256 x (nG Horner steps + 2 products) = ~3 k field products = ~0.1 ms on one core: less than starting helper threads
        // costs (the r01 / early r02 version spawned 8 std::threads per call: 0.2 ms of the gap after compute_G)
        for (size_t i = 0; i < count; ++i) {
            fe_t gi = Fr::zero();
            for (size_t k = nG; k-- > 0;) gi = Fr::add(Fr::mul(gi, dom->xs[i]), polyG_host[k]);   // UnivariatePoly::eval (univariate.rs:67-75), Horner: same value
            kp[i] = Fr::mul(Fr::sub(gi, Fr::mul(f_alpha, dom->l0[i])), dom->inv_z[i]);
        }
        SRS_HIP_CHECK(hipMemcpyAsync(d_out, kp.data(), count * sizeof(fe_t), hipMemcpyHostToDevice, st));
        ntt::run(d_out, log_domain_K, count, 1, true, true, st);     // UnivariatePoly::coset_ifft
        SRS_HIP_CHECK(hipMemcpyAsync(out_host, d_out, count * sizeof(fe_t), hipMemcpyDeviceToHost, st));
        SRS_HIP_CHECK(hipStreamSynchronize(st));
        return 0;
    }
    SRS_HIP_CHECK(hipMemcpyAsync(d_g, polyG_host, nG * sizeof(fe_t), hipMemcpyHostToDevice, st));
    SRS_HIP_CHECK(hipMemsetAsync(d_err, 0, sizeof(int), st));
    fe_t inv_n = Fr::inv(Fr::from_u64(instances_to_fold));
    SRS_LAUNCH(k_pg_K_points, ((uint32_t)((count + 63) / 64)), (64), 0, st, (const fe_t *)d_g, (uint32_t)nG, f_alpha,
               ntt::zeta(), ntt::omega(log_domain_K, false), inv_n, (uint32_t)instances_to_fold, (uint32_t)count, d_out, d_err);
    ntt::run(d_out, log_domain_K, count, 1, true, true, st);     // UnivariatePoly::coset_ifft
    SRS_HIP_CHECK(hipMemcpyAsync(out_host, d_out, count * sizeof(fe_t), hipMemcpyDeviceToHost, st));
    SRS_HIP_CHECK(hipMemcpyAsync(&herr, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    SRS_HIP_CHECK(hipStreamSynchronize(st));
    if (herr) { err = "Z(X) must be not equal to 0"; return 4; }
    return 0;
}

// the small K domain's points X_i = zeta omega^i, L_0(X_i) and 1 / Z(X_i) on the device ([xs | l0 | inv_z]), per (domain, n, device)
static const fe_t *k_domain_table_dev(uint32_t log_domain_K, size_t instances_to_fold, bool &zero_z) {
    struct Tab { fe_t *dev = nullptr; bool zero_z = false; };
    static std::mutex mu;
    static std::map<std::tuple<int, uint32_t, size_t>, Tab> cache;
    int device = 0;
    SRS_HIP_CHECK(hipGetDevice(&device));
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_tuple(device, log_domain_K, instances_to_fold);
    auto it = cache.find(key);
    if (it == cache.end()) {
        const size_t count = (size_t)1 << log_domain_K;
        std::vector<fe_t> h(3 * count), xn1(count), xm1(count), pref(2 * count);
        const fe_t zeta = ntt::zeta(), omega = ntt::omega(log_domain_K, false), one = Fr::one();
        const fe_t inv_n = Fr::inv(Fr::from_u64(instances_to_fold));
        Tab t;
        fe_t X = zeta, acc = one;
        for (size_t i = 0; i < count; ++i) { h[i] = X; X = Fr::mul(X, omega); }
        for (size_t i = 0; i < count; ++i) {
            xn1[i] = Fr::sub(Fr::pow_u64(h[i], instances_to_fold), one);
            xm1[i] = Fr::sub(h[i], one);
            if (Fr::is_zero(xn1[i])) t.zero_z = true;
        }
        if (!t.zero_z) {                                   // one inversion for all denominators (Montgomery's trick), as the host route
            for (size_t i = 0; i < count; ++i) {
                pref[2 * i] = acc;
                acc = Fr::mul(acc, xn1[i]);
                pref[2 * i + 1] = acc;
                acc = Fr::mul(acc, xm1[i]);
            }
            fe_t inv = Fr::inv(acc);
            for (size_t i = count; i-- > 0;) {
                const fe_t inv_xm1 = Fr::mul(inv, pref[2 * i + 1]);
                inv = Fr::mul(inv, xm1[i]);
                h[2 * count + i] = Fr::mul(inv, pref[2 * i]);
                inv = Fr::mul(inv, xn1[i]);
                h[count + i] = Fr::mul(inv_n, Fr::mul(xn1[i], inv_xm1));
            }
            SRS_HIP_CHECK(hipMalloc((void **)&t.dev, 3 * count * sizeof(fe_t)));        // kept for the life of the process (24 KiB per domain)
            SRS_HIP_CHECK(hipMemcpy(t.dev, h.data(), 3 * count * sizeof(fe_t), hipMemcpyHostToDevice));
        }
        it = cache.emplace(key, t).first;
    }
    zero_z = it->second.zero_z;
    return it->second.dev;
}

bool pg_K_device_ok(const PgGValues &g, uint32_t log_domain_K) {
    return g.vals_dev != nullptr && g.degree + 1 <= PG_K_SMALL_MAX_NODES && log_domain_K <= PG_K_SMALL_MAX_LOG;
}

int pg_K_from_G_device(Structure *S, const PgGValues &g, const fe_t &f_alpha, size_t instances_to_fold, uint32_t log_domain_K, hipStream_t st,
                       fe_t *out_host, std::string &err) {
    if (!pg_K_device_ok(g, log_domain_K)) { err = "internal: pg_K_from_G_device on an unsupported shape"; return 5; }
    const uint32_t n_nodes = g.degree + 1, count = 1u << log_domain_K;
    bool zero_z = false;
    const fe_t *tab = k_domain_table_dev(log_domain_K, instances_to_fold, zero_z);
    if (zero_z) { err = "Z(X) must be not equal to 0"; return 4; }
    fe_t *&M = S->d_kM[g.skip_one ? 1 : 0];
    if (!M) {
        FieldOps f{0};
        std::vector<fe_t> full((size_t)n_nodes * n_nodes);
        if (g.skip_one) {
            full = !S->vinv_g1.empty() ? S->vinv_g1 : inverse_vandermonde_at(f, g.degree, 1);                   // [d + 1][d + 1], nodes 1 .. d + 1
        } else {
            const std::vector<fe_t> rows = !S->vinv_g.empty() ? S->vinv_g : (g.degree ? inverse_vandermonde(f, g.degree) : std::vector<fe_t>());
            for (uint32_t j = 0; j < n_nodes; ++j) full[j] = j == 0 ? Fr::one() : Fr::zero();                       // coefficient 0 = G(0)
            for (size_t i = 0; i < rows.size(); ++i) full[n_nodes + i] = rows[i];                                    // rows k = 1 .. d, nodes 0 .. d
        }
        SRS_HIP_CHECK(hipMalloc((void **)&M, full.size() * sizeof(fe_t)));
        S->owned.push_back(M);
        SRS_HIP_CHECK(hipMemcpy(M, full.data(), full.size() * sizeof(fe_t), hipMemcpyHostToDevice));
    }
    static thread_local ThreadArena scratch;
    scratch.reserve(Arena::pad((size_t)count * sizeof(fe_t)) + 256);
    scratch.reset();
    fe_t *d_out = scratch.take<fe_t>(count);
    const uint32_t threads = std::max<uint32_t>(64u, std::min<uint32_t>(256u, count));
    SRS_LAUNCH(k_pg_K_small, ((count + threads - 1) / threads), (threads), 0, st, g.vals_dev, g.n_dev, f_alpha, g.skip_one ? 1 : 0, (const fe_t *)M,
               n_nodes, f_alpha, tab, count, d_out);
    ntt::run(d_out, log_domain_K, count, 1, true, true, st);     // UnivariatePoly::coset_ifft
    SRS_HIP_CHECK(hipMemcpyAsync(out_host, d_out, (size_t)count * sizeof(fe_t), hipMemcpyDeviceToHost, st));
    SRS_HIP_CHECK(hipStreamSynchronize(st));
    SRS_HIP_CHECK(hipGetLastError());
    prof::collect();
    return 0;
}

size_t count_mismatch(const fe_t *a_dev, const fe_t *b_dev, size_t n, hipStream_t st) {
    if (!n) return 0;
    uint32_t *d = nullptr, h = 0;
    SRS_HIP_CHECK(hipMalloc((void **)&d, sizeof(uint32_t)));
    SRS_HIP_CHECK(hipMemsetAsync(d, 0, sizeof(uint32_t), st));
    SRS_LAUNCH(k_count_mismatch, ((uint32_t)((n + 255) / 256)), (256), 0, st, a_dev, b_dev, n, d);
    SRS_HIP_CHECK(hipMemcpyAsync(&h, d, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SRS_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(d);
    return h;
}

// ---------------------------------------------------------------------------------------------
// lookup arguments: host side
// ---------------------------------------------------------------------------------------------
// Arguments::evaluate_coefficient_1 (src/plonk/lookup.rs:319-341): ls[i] = L_i(row), ts[i] = T_i(row), ms[i].
// advice_dev: the advice columns, column-major num_advice * rows; outputs: HOST arrays of L DEVICE vectors.
int lookup_coeff_1(Structure *S, const fe_t *advice_dev, const fe_t &r, fe_t *const *ls, fe_t *const *ts, fe_t *const *ms,
                   hipStream_t st, std::string &err) {
    const size_t L = S->num_lookups;
    if (!L) { err = "structure has no lookup arguments"; return 4; }    // SpsError::LackOfLookupArguments
    const uint32_t n = (uint32_t)S->rows;
    for (size_t i = 0; i < 2 * L; ++i) {
        fe_t *outs[1] = {i < L ? ls[i] : ts[i - L]};
        int rc = evaluate_prog(S, S->lookup_progs[i], 1, advice_dev, nullptr, &r, 1, outs, st, err);
        if (rc) return rc;
    }
    uint32_t cap = 2;
    while (cap < 2 * n) cap <<= 1;
    uint32_t *d_row = nullptr, *d_cnt = nullptr;
    SRS_HIP_CHECK(hipMalloc((void **)&d_row, (size_t)cap * 2 * sizeof(uint32_t)));
    d_cnt = d_row + cap;
    try {
        const uint32_t blocks = (n + 255) / 256;
        for (size_t i = 0; i < L; ++i) {
            prof::Scope ps("lookup_m", st, n);
            SRS_HIP_CHECK(hipMemsetAsync(d_row, 0xFF, (size_t)cap * sizeof(uint32_t), st));
            SRS_HIP_CHECK(hipMemsetAsync(d_cnt, 0, (size_t)cap * sizeof(uint32_t), st));
            SRS_LAUNCH(k_m_insert, (blocks), (256), 0, st, (const fe_t *)ts[i], n, d_row, cap - 1);
            SRS_LAUNCH(k_m_count, (blocks), (256), 0, st, (const fe_t *)ls[i], n, (const fe_t *)ts[i], (const uint32_t *)d_row, d_cnt, cap - 1);
            if (S->field == 0) SRS_LAUNCH((k_m_emit<Fr>), (blocks), (256), 0, st, (const fe_t *)ts[i], n, (const uint32_t *)d_row, (const uint32_t *)d_cnt, cap - 1, ms[i]);
            else SRS_LAUNCH((k_m_emit<Fq>), (blocks), (256), 0, st, (const fe_t *)ts[i], n, (const uint32_t *)d_row, (const uint32_t *)d_cnt, cap - 1, ms[i]);
        }
        SRS_HIP_CHECK(hipStreamSynchronize(st));
        SRS_HIP_CHECK(hipGetLastError());
    } catch (...) { (void)hipFree(d_row); throw; }
    (void)hipFree(d_row);
    prof::collect();
    return 0;
}

// Arguments::evaluate_h_g (lookup.rs:305-317) for one lookup: h = 1/(l + r), g = m/(t + r)
void lookup_coeff_2(int field, const fe_t *l, const fe_t *t, const fe_t *m, const fe_t &r, size_t n, fe_t *h, fe_t *g, hipStream_t st) {
    if (!n) return;
    prof::Scope ps("lookup_hg", st, n);
    const uint32_t gx = (uint32_t)((n + HG_THREADS * HG_CHUNK - 1) / (HG_THREADS * HG_CHUNK));
    if (field == 0) SRS_LAUNCH((k_lookup_hg<Fr>), (gx, 2), (HG_THREADS), 0, st, l, t, m, r, (uint32_t)n, h, g);
    else SRS_LAUNCH((k_lookup_hg<Fq>), (gx, 2), (HG_THREADS), 0, st, l, t, m, r, (uint32_t)n, h, g);
}

void assigned_invert(int field, const fe_t *num, const fe_t *den, const uint8_t *has_den, size_t n, fe_t *out, hipStream_t st) {
    if (!n) return;
    const uint32_t gx = (uint32_t)((n + HG_THREADS * HG_CHUNK - 1) / (HG_THREADS * HG_CHUNK));
    if (field == 0) SRS_LAUNCH((k_assigned_invert<Fr>), (gx), (HG_THREADS), 0, st, num, den, has_den, (uint32_t)n, out);
    else SRS_LAUNCH((k_assigned_invert<Fq>), (gx), (HG_THREADS), 0, st, num, den, has_den, (uint32_t)n, out);
}

// PlonkStructure::is_sat_log_derivative (src/plonk/mod.rs:366-398) on the concatenated witness:
// number of lookups i whose sum_row (h_i - g_i) != 0; h_i / g_i are columns 2 i / 2 i + 1 of the LAST round.
size_t log_derivative_mismatches(Structure *S, const fe_t *W_dev, hipStream_t st) {
    const size_t L = S->num_lookups;
    if (!L) return 0;
    FieldOps f{S->field};
    const uint32_t n = (uint32_t)S->rows, blocks = std::min<uint32_t>((n + 255) / 256, 256);
    const fe_t *last = W_dev + (S->num_advice + 3 * L) * S->rows;
    fe_t *d_part = nullptr;
    SRS_HIP_CHECK(hipMalloc((void **)&d_part, L * blocks * sizeof(fe_t)));
    std::vector<fe_t> part(L * blocks);
    try {
        for (size_t i = 0; i < L; ++i) {
            const fe_t *h = last + (2 * i) * S->rows, *g = last + (2 * i + 1) * S->rows;
            if (S->field == 0) SRS_LAUNCH((k_sum_diff<Fr>), (blocks), (256), 0, st, h, g, n, d_part + i * blocks);
            else SRS_LAUNCH((k_sum_diff<Fq>), (blocks), (256), 0, st, h, g, n, d_part + i * blocks);
        }
        SRS_HIP_CHECK(hipMemcpyAsync(part.data(), d_part, part.size() * sizeof(fe_t), hipMemcpyDeviceToHost, st));
        SRS_HIP_CHECK(hipStreamSynchronize(st));
    } catch (...) { (void)hipFree(d_part); throw; }
    (void)hipFree(d_part);
    size_t bad = 0;
    for (size_t i = 0; i < L; ++i) {
        fe_t acc = f.zero();
        for (uint32_t b = 0; b < blocks; ++b) acc = f.add(acc, part[i * blocks + b]);
        if (!f.is_zero(acc)) ++bad;
    }
    return bad;
}

int lincomb(int field, fe_t *out, const fe_t *const *w_dev, const fe_t *coefs, size_t J, size_t n, hipStream_t st, std::string &err,
            uint32_t rank, uint32_t world) {
    if (J == 0 || J > JMAX) { err = "unsupported number of witnesses"; return 4; }
    if (world > 1) {                           // elements of this rank's stripes (block-cyclic, 2^ROW_STRIPE_LOG each)
        const size_t SL = (size_t)1 << ROW_STRIPE_LOG, full = n / SL;
        size_t mine = (full > rank ? (full - rank + world - 1) / world : 0) * SL;
        if (n % SL && full % world == rank) mine += n % SL;
        n = mine;
    } else {
        rank = 0;
        world = 1;
    }
    if (!n) return 0;
    LincombArgs a;
    a.J = (uint32_t)J;
    for (uint32_t j = 0; j < JMAX; ++j) {
        a.w[j] = j < J ? w_dev[j] : nullptr;
        a.coef[j] = j < J ? coefs[j] : Fr::zero();
    }
    uint32_t blocks = (uint32_t)std::min<size_t>((n + 255) / 256, 256 * 16);
    FieldOps f{field};
    fe_t sum = f.zero();
    for (size_t j = 0; j < J; ++j) sum = f.add(sum, coefs[j]);
    const bool affine = J >= 2 && f.is_zero(f.sub(sum, f.one()));
    if (field == 0) {
        if (affine) SRS_LAUNCH((k_lincomb<Fr, true>), (blocks), (256), 0, st, out, a, n, rank, world);
        else SRS_LAUNCH((k_lincomb<Fr, false>), (blocks), (256), 0, st, out, a, n, rank, world);
    } else {
        if (affine) SRS_LAUNCH((k_lincomb<Fq, true>), (blocks), (256), 0, st, out, a, n, rank, world);
        else SRS_LAUNCH((k_lincomb<Fq, false>), (blocks), (256), 0, st, out, a, n, rank, world);
    }
    return 0;
}

int lincomb_rows(int field, fe_t *out, const fe_t *const *w_dev, const fe_t *coefs, size_t J, const uint32_t *rows_dev, size_t n_rows, size_t cols,
                 size_t col_len, hipStream_t st, std::string &err) {
    if (J == 0 || J > JMAX) { err = "unsupported number of witnesses"; return 4; }
    if (!n_rows || !cols) return 0;
    LincombArgs a;
    a.J = (uint32_t)J;
    for (uint32_t j = 0; j < JMAX; ++j) {
        a.w[j] = j < J ? w_dev[j] : nullptr;
        a.coef[j] = j < J ? coefs[j] : Fr::zero();
    }
    const uint32_t blocks = (uint32_t)((n_rows * cols + 255) / 256);
    if (field == 0) SRS_LAUNCH((k_lincomb_rows<Fr>), (blocks), (256), 0, st, out, a, rows_dev, n_rows, cols, col_len);
    else SRS_LAUNCH((k_lincomb_rows<Fq>), (blocks), (256), 0, st, out, a, rows_dev, n_rows, cols, col_len);
    return 0;
}

}  // namespace rowprog
}  // namespace srs
