/*
 * oracle.h -- CPU restatement of the Sirius folding-prover hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product (sirius_amd/, include/) may include, link or call this.  Allowed users:
 * tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * Field / group arithmetic itself lives in the un-vendored third-party dependency
 *   halo2_proofs = git snarkify/halo2 @ branch snarkify/dev.scroll.alpha.2 (Cargo.toml:44-46)
 *   -> halo2curves (bn256, grumpkin); no rev pinned, Cargo.lock git-ignored.
 * Its published algorithms (Montgomery 4x64 fields, short-Weierstrass a=0 group law,
 * `best_multiexp` = chunk-per-thread serial windowed Pippenger) are restated here.
 *
 * PINNING STATUS
 *   fft / ifft / coset_* : pinned by src/fft.rs:241-260 (fft_simple_input_test KAT)
 *   lagrange / inversion : pinned by src/polynomial/lagrange.rs:116-127 (basic_lagrange_test KAT)
 *   scalar-mul on bn256  : pinned by src/digest.rs:100-114 ([r-1]G = -G)
 *   MSM (commit)         : PARITY UNPINNED -- the reference holds no known-answer vector for any
 *                          MSM output; anchored on the group law, the KAT above,
 *                          an independent Python big-int implementation (oracle/pyref.py) and
 *                          the homomorphism property the reference itself tests
 *                          (src/nifs/sangria/mod.rs:455-474).
 *   coset ZETA constant  : [3P] from memory of halo2curves; verified to be a primitive cube root.
 *
 * Data layout (identical to the product C-ABI so bytes can be compared directly):
 *   field element = 4 x u64 little-endian limbs, Montgomery form (R = 2^256)  [halo2curves in-memory]
 *   affine point  = x || y, identity = all-zero bytes.
 */
#ifndef SIRIUS_ORACLE_H
#define SIRIUS_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } ofe;        /* field element (Montgomery) */
typedef struct { ofe x, y; } oaffine;         /* affine point, identity = (0,0) */
typedef struct { ofe x, y, z; } ojac;         /* Jacobian, identity = z == 0 */

enum { O_FR = 0, O_FQ = 1 };                  /* bn256 scalar field / bn256 base field */
enum { O_BN256 = 0, O_GRUMPKIN = 1 };         /* curve ids */

/* ---- field ---- */
void o_fe_to_mont(int field, const ofe *canon, ofe *out, size_t n);
void o_fe_from_mont(int field, const ofe *mont, ofe *out, size_t n);
void o_fe_mul(int field, const ofe *a, const ofe *b, ofe *out, size_t n);
void o_fe_add(int field, const ofe *a, const ofe *b, ofe *out, size_t n);
void o_fe_sub(int field, const ofe *a, const ofe *b, ofe *out, size_t n);
void o_fe_inv(int field, const ofe *a, ofe *out, size_t n);

/* ---- curve ---- */
void o_point_add(int curve, const oaffine *a, const oaffine *b, oaffine *out);
void o_point_mul(int curve, const ofe *scalar_mont, const oaffine *p, oaffine *out);
int  o_point_is_on_curve(int curve, const oaffine *p);
/* bases[i] = [k_i] G for a cheap deterministic key (NOT the reference's hash-to-curve key):
 * k_0 = seed-derived, P_{i+1} = P_i + [step]G chains per thread; see oracle.c */
void o_make_bases(int curve, uint64_t seed, oaffine *out, size_t n, int threads);

/* ---- commit (src/commitment.rs:81-90) ---- */
/* naive double-and-add sum; O(256 n) group ops, for small n */
void o_msm_naive(int curve, const ofe *scalars, const oaffine *bases, size_t n, oaffine *out);
/* halo2 best_multiexp: `threads` chunks, serial Pippenger each, summed; -> affine */
void o_msm(int curve, const ofe *scalars, const oaffine *bases, size_t n, int threads, oaffine *out);

/* ---- fft (src/fft.rs) : Fr only ---- */
int o_fft(ofe *a, size_t n, int threads);          /* :160-165 */
int o_ifft(ofe *a, size_t n, int threads);         /* :168-182 */
int o_coset_fft(ofe *a, size_t n, int threads);    /* :186-190 */
int o_coset_ifft(ofe *a, size_t n, int threads);   /* :194-198 */

/* ---- row program interpreter (src/polynomial/graph_evaluator.rs:93-149,361-388) ----
 * The Expression -> GroupedPoly -> GraphEvaluator compile steps are restated in oracle/expr.py;
 * the calculation list arrives here flattened:  each calc = 8 x int64
 *   [op, a_kind, a_idx, a_rot, b_kind, b_idx, b_rot, target]
 * kinds: 0 Constant(idx) 1 Intermediate(idx) 2 Fixed{idx,rot} 3 Poly{idx,rot} 4 Challenge{idx}
 * ops:   0 Add 1 Sub 2 Mul 3 Square 4 Double 5 Negate 6 Store   (Horner is never emitted by
 * add_expression, graph_evaluator.rs:261-351)
 * Data (= PlonkEvalDomain, src/plonk/eval.rs:93-104,153-228):
 *   selectors: n_sel byte columns of `rows`; fixed: n_fixed columns of `rows` field elements;
 *   W1s / W2s: the witness ROUNDS of the two instances (PlonkWitness::W), n_w1 / n_w2 of them
 *   (1 without lookups, 2 or 3 with), each column-major with w*_len[i] elements.
 *   Column j >= num_advice is lookup variable (j - num_advice) / 5, sub-index % 5 of (l, t, m, h, g),
 *   mapped to (round, column) exactly as eval.rs:169-201 does.
 */
typedef struct {
    int field;
    size_t rows;
    size_t n_sel, n_fixed, num_advice, num_lookup;
    const uint8_t *const *selectors;
    const ofe *const *fixed;
    size_t n_w1, n_w2;
    const ofe *W1s[3], *W2s[3];
    size_t w1_len[3], w2_len[3];
    const ofe *challenges; size_t n_challenges;
} o_eval_domain;

int o_eval_program(const o_eval_domain *d, const int64_t *calcs, size_t n_calcs,
                   const ofe *constants, size_t n_constants,
                   const int32_t *rotations, size_t n_rot, size_t n_intermediates,
                   ofe *out /* rows */, int threads);

/* ---- folds (src/nifs/sangria/accumulator.rs:364-404) ---- */
void o_fold_w(int field, const ofe *w1, const ofe *w2, const ofe *r, ofe *out, size_t n, int threads);
void o_fold_e(int field, const ofe *e, const ofe *const *t, size_t n_terms, const ofe *r,
              ofe *out, size_t n, int threads);

/* ---- ProtoGalaxy (src/nifs/protogalaxy/) ---- */
/* out[i] = sum_j coef[j] * w[j][i]   (fold_witness, mod.rs:176-210; FoldedWitness::new, poly/folded_witness.rs:20-180) */
void o_lincomb(int field, const ofe *const *w, const ofe *coef, size_t J, ofe *out, size_t n, int threads);
/* the weighted binary reduction tree of compute_F / compute_G (poly/mod.rs:68-203, 308-425); see oracle.c */
void o_pg_tree(int field, const ofe *leaves, size_t leaves_stride, size_t n, size_t count, const ofe *weights, size_t P, size_t t,
               ofe *out, int threads);

#ifdef __cplusplus
}
#endif
#endif
