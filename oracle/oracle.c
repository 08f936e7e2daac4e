/*
 * oracle.c -- CPU restatement of the Sirius hot path.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain C11 + OpenMP, no dependencies.  Field arithmetic = 4x64 Montgomery (R = 2^256), the
 * representation halo2curves keeps in memory [3P].  Everything derived (R, R^2, roots of unity)
 * is computed at init from the two moduli and the generator 7, so the only literals are the
 * moduli, the generator, the grumpkin generator's y and ZETA (all checked in oracle/pyref.py
 * against the reference's KATs).
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

typedef struct {
    uint64_t p[4];
    uint64_t inv;      /* -p^{-1} mod 2^64 */
    ofe r;             /* R mod p  (= Montgomery ONE) */
    ofe r2;            /* R^2 mod p */
} fparams;

static fparams F[2] = {
    { {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}, 0, {{0}}, {{0}} },
    { {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}, 0, {{0}}, {{0}} },
};
static int g_init = 0;

/* ------------------------------------------------------------------ raw 256-bit helpers */
static inline int geq(const uint64_t *a, const uint64_t *b) {
    for (int i = 3; i >= 0; --i) { if (a[i] != b[i]) return a[i] > b[i]; }
    return 1;
}
static inline uint64_t add4(uint64_t *o, const uint64_t *a, const uint64_t *b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a[i] + b[i]; o[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static inline uint64_t sub4(uint64_t *o, const uint64_t *a, const uint64_t *b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - b[i] - br; o[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
static inline int is_zero(const ofe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const ofe *a, const ofe *b) { return memcmp(a, b, sizeof(ofe)) == 0; }

static inline void fadd(const fparams *f, ofe *o, const ofe *a, const ofe *b) {
    uint64_t t[4]; uint64_t c = add4(t, a->l, b->l);
    if (c || geq(t, f->p)) sub4(t, t, f->p);
    memcpy(o->l, t, 32);
}
static inline void fsub(const fparams *f, ofe *o, const ofe *a, const ofe *b) {
    uint64_t t[4];
    if (sub4(t, a->l, b->l)) add4(t, t, f->p);
    memcpy(o->l, t, 32);
}
static inline void fneg(const fparams *f, ofe *o, const ofe *a) {
    if (is_zero(a)) { *o = *a; return; }
    uint64_t t[4]; sub4(t, f->p, a->l); memcpy(o->l, t, 32);
}
static inline void fdbl(const fparams *f, ofe *o, const ofe *a) { fadd(f, o, a, a); }

/* Montgomery product a*b*R^{-1} mod p (CIOS) */
static inline void fmul(const fparams *f, ofe *o, const ofe *a, const ofe *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64;
        }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * f->inv;
        c = (u128)m * f->p[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)m * f->p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64;
        }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || geq(t, f->p)) sub4(t, t, f->p);
    memcpy(o->l, t, 32);
}
static inline void fsqr(const fparams *f, ofe *o, const ofe *a) { fmul(f, o, a, a); }

static void fpow(const fparams *f, ofe *o, const ofe *a, const uint64_t e[4]) {
    ofe acc = f->r, base = *a;
    for (int i = 0; i < 256; ++i) {
        if ((e[i >> 6] >> (i & 63)) & 1) fmul(f, &acc, &acc, &base);
        fsqr(f, &base, &base);
    }
    *o = acc;
}
static void finv(const fparams *f, ofe *o, const ofe *a) {   /* a^(p-2); 0 -> 0 */
    uint64_t e[4]; uint64_t two[4] = {2, 0, 0, 0};
    sub4(e, f->p, two);
    fpow(f, o, a, e);
}
static void fto_mont(const fparams *f, ofe *o, const ofe *a) { fmul(f, o, a, &f->r2); }
static void ffrom_mont(const fparams *f, ofe *o, const ofe *a) {
    ofe one = {{1, 0, 0, 0}}; fmul(f, o, a, &one);
}
static void ffrom_u64(const fparams *f, ofe *o, uint64_t v) {
    ofe t = {{v, 0, 0, 0}}; fto_mont(f, o, &t);
}

/* Fr-only derived constants (src/fft.rs uses F::ROOT_OF_UNITY, ROOT_OF_UNITY_INV, TWO_INV, ZETA, S) */
#define FR_S 28
static ofe FR_ROOT, FR_ROOT_INV, FR_TWO_INV, FR_ZETA, FR_ZETA2;

static void oracle_init(void) {
    if (g_init) return;
    for (int k = 0; k < 2; ++k) {
        fparams *f = &F[k];
        uint64_t inv = 1;                       /* Newton: inv = p^{-1} mod 2^64 */
        for (int i = 0; i < 6; ++i) inv *= 2 - f->p[0] * inv;
        f->inv = (uint64_t)0 - inv;
        /* R mod p and R^2 mod p by repeated doubling of 1 */
        uint64_t t[4] = {1, 0, 0, 0};
        for (int i = 0; i < 512; ++i) {
            uint64_t c = add4(t, t, t);
            if (c || geq(t, f->p)) sub4(t, t, f->p);
            if (i == 255) memcpy(f->r.l, t, 32);
        }
        memcpy(f->r2.l, t, 32);
    }
    const fparams *fr = &F[O_FR];
    /* ROOT_OF_UNITY = 7^((r-1)/2^28) [3P halo2curves bn256::Fr; value pinned by src/fft.rs:241-260] */
    ofe g; ffrom_u64(fr, &g, 7);
    uint64_t e[4]; uint64_t one[4] = {1, 0, 0, 0};
    sub4(e, fr->p, one);
    for (int s = 0; s < FR_S; ++s) {            /* e >>= 1 */
        for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 1) | (i < 3 ? e[i + 1] << 63 : 0);
    }
    fpow(fr, &FR_ROOT, &g, e);
    finv(fr, &FR_ROOT_INV, &FR_ROOT);
    ofe two; ffrom_u64(fr, &two, 2); finv(fr, &FR_TWO_INV, &two);
    ofe zc = {{0xb8ca0b2d36636f23ULL, 0xcc37a73fec2bc5e9ULL, 0x048b6e193fd84104ULL, 0x30644e72e131a029ULL}};
    fto_mont(fr, &FR_ZETA, &zc);
    fsqr(fr, &FR_ZETA2, &FR_ZETA);
    g_init = 1;
}

static int clamp_threads(int threads) {
    if (threads <= 0) {
#ifdef _OPENMP
        threads = omp_get_max_threads();
#else
        threads = 1;
#endif
    }
    return threads;
}

/* ------------------------------------------------------------------ exported field ops */
#define FIELD_LOOP(body) oracle_init(); const fparams *f = &F[field]; for (size_t i = 0; i < n; ++i) { body; }
void o_fe_to_mont(int field, const ofe *a, ofe *o, size_t n) { FIELD_LOOP(fto_mont(f, &o[i], &a[i])) }
void o_fe_from_mont(int field, const ofe *a, ofe *o, size_t n) { FIELD_LOOP(ffrom_mont(f, &o[i], &a[i])) }
void o_fe_mul(int field, const ofe *a, const ofe *b, ofe *o, size_t n) { FIELD_LOOP(fmul(f, &o[i], &a[i], &b[i])) }
void o_fe_add(int field, const ofe *a, const ofe *b, ofe *o, size_t n) { FIELD_LOOP(fadd(f, &o[i], &a[i], &b[i])) }
void o_fe_sub(int field, const ofe *a, const ofe *b, ofe *o, size_t n) { FIELD_LOOP(fsub(f, &o[i], &a[i], &b[i])) }
void o_fe_inv(int field, const ofe *a, ofe *o, size_t n) { FIELD_LOOP(finv(f, &o[i], &a[i])) }

/* ------------------------------------------------------------------ curves
 * y^2 = x^3 + b, a = 0.  bn256 G1: base field Fq, b = 3, G = (1,2);
 * grumpkin: base field Fr, b = -17, G = (1, sqrt(-16))  [3P halo2curves; SURVEY.md 8b].
 * Jacobian coordinates, identity z = 0.  Formulas: EFD dbl-2009-l, add-2007-bl, madd-2007-bl.
 */
typedef struct { const fparams *f; const fparams *sf; ofe b; oaffine g; } cparams;
static cparams C[2];
static int c_init = 0;
static void curve_init(void) {
    oracle_init();
    if (c_init) return;
    C[O_BN256].f = &F[O_FQ]; C[O_BN256].sf = &F[O_FR];
    ffrom_u64(&F[O_FQ], &C[O_BN256].b, 3);
    ffrom_u64(&F[O_FQ], &C[O_BN256].g.x, 1); ffrom_u64(&F[O_FQ], &C[O_BN256].g.y, 2);
    C[O_GRUMPKIN].f = &F[O_FR]; C[O_GRUMPKIN].sf = &F[O_FQ];
    ofe t; ffrom_u64(&F[O_FR], &t, 17); fneg(&F[O_FR], &C[O_GRUMPKIN].b, &t);
    ffrom_u64(&F[O_FR], &C[O_GRUMPKIN].g.x, 1);
    /* 17631683881184975370165255887551781615748388533673675138860 */
    ofe gy = {{0x833fc48d823f272cULL, 0x2d270d45f1181294ULL, 0xcf135e7506a45d63ULL, 0x0000000000000002ULL}};
    fto_mont(&F[O_FR], &C[O_GRUMPKIN].g.y, &gy);
    c_init = 1;
}

static inline int aff_is_id(const oaffine *p) { return is_zero(&p->x) && is_zero(&p->y); }
static inline void jac_set_id(ojac *p) { memset(p, 0, sizeof(*p)); }
static inline int jac_is_id(const ojac *p) { return is_zero(&p->z); }

static void jac_from_affine(const cparams *c, ojac *o, const oaffine *a) {
    if (aff_is_id(a)) { jac_set_id(o); return; }
    o->x = a->x; o->y = a->y; o->z = c->f->r;
}
static void jac_dbl(const cparams *c, ojac *o, const ojac *p) {
    const fparams *f = c->f;
    if (jac_is_id(p)) { *o = *p; return; }
    ofe A, B, Cc, D, E, Fv, t, x3, y3, z3;
    fsqr(f, &A, &p->x); fsqr(f, &B, &p->y); fsqr(f, &Cc, &B);
    fadd(f, &t, &p->x, &B); fsqr(f, &t, &t); fsub(f, &t, &t, &A); fsub(f, &t, &t, &Cc); fdbl(f, &D, &t);
    fdbl(f, &E, &A); fadd(f, &E, &E, &A);
    fsqr(f, &Fv, &E);
    fdbl(f, &t, &D); fsub(f, &x3, &Fv, &t);
    fsub(f, &t, &D, &x3); fmul(f, &y3, &E, &t);
    fdbl(f, &t, &Cc); fdbl(f, &t, &t); fdbl(f, &t, &t); fsub(f, &y3, &y3, &t);
    fmul(f, &z3, &p->y, &p->z); fdbl(f, &z3, &z3);
    o->x = x3; o->y = y3; o->z = z3;
}
static void jac_add(const cparams *c, ojac *o, const ojac *p, const ojac *q) {
    const fparams *f = c->f;
    if (jac_is_id(p)) { *o = *q; return; }
    if (jac_is_id(q)) { *o = *p; return; }
    ofe z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
    fsqr(f, &z1z1, &p->z); fsqr(f, &z2z2, &q->z);
    fmul(f, &u1, &p->x, &z2z2); fmul(f, &u2, &q->x, &z1z1);
    fmul(f, &s1, &p->y, &q->z); fmul(f, &s1, &s1, &z2z2);
    fmul(f, &s2, &q->y, &p->z); fmul(f, &s2, &s2, &z1z1);
    if (fe_eq(&u1, &u2)) {
        if (fe_eq(&s1, &s2)) { jac_dbl(c, o, p); return; }
        jac_set_id(o); return;
    }
    fsub(f, &h, &u2, &u1);
    fdbl(f, &i, &h); fsqr(f, &i, &i);
    fmul(f, &j, &h, &i);
    fsub(f, &r, &s2, &s1); fdbl(f, &r, &r);
    fmul(f, &v, &u1, &i);
    fsqr(f, &x3, &r); fsub(f, &x3, &x3, &j); fdbl(f, &t, &v); fsub(f, &x3, &x3, &t);
    fsub(f, &t, &v, &x3); fmul(f, &y3, &r, &t);
    fmul(f, &t, &s1, &j); fdbl(f, &t, &t); fsub(f, &y3, &y3, &t);
    fadd(f, &z3, &p->z, &q->z); fsqr(f, &z3, &z3); fsub(f, &z3, &z3, &z1z1); fsub(f, &z3, &z3, &z2z2);
    fmul(f, &z3, &z3, &h);
    o->x = x3; o->y = y3; o->z = z3;
}
/* mixed addition (madd-2007-bl, 7M + 4S), what halo2curves' `Curve + Affine` amounts to */
static void jac_add_affine(const cparams *c, ojac *o, const ojac *p, const oaffine *q) {
    const fparams *f = c->f;
    if (aff_is_id(q)) { *o = *p; return; }
    if (jac_is_id(p)) { jac_from_affine(c, o, q); return; }
    ofe z1z1, u2, s2, h, hh, i, j, r, v, t, x3, y3, z3;
    fsqr(f, &z1z1, &p->z);
    fmul(f, &u2, &q->x, &z1z1);
    fmul(f, &s2, &q->y, &p->z); fmul(f, &s2, &s2, &z1z1);
    if (fe_eq(&p->x, &u2)) {
        if (fe_eq(&p->y, &s2)) { jac_dbl(c, o, p); return; }
        jac_set_id(o); return;
    }
    fsub(f, &h, &u2, &p->x);
    fsqr(f, &hh, &h);
    fdbl(f, &i, &hh); fdbl(f, &i, &i);
    fmul(f, &j, &h, &i);
    fsub(f, &r, &s2, &p->y); fdbl(f, &r, &r);
    fmul(f, &v, &p->x, &i);
    fsqr(f, &x3, &r); fsub(f, &x3, &x3, &j); fdbl(f, &t, &v); fsub(f, &x3, &x3, &t);
    fsub(f, &t, &v, &x3); fmul(f, &y3, &r, &t);
    fmul(f, &t, &p->y, &j); fdbl(f, &t, &t); fsub(f, &y3, &y3, &t);
    fadd(f, &z3, &p->z, &h); fsqr(f, &z3, &z3); fsub(f, &z3, &z3, &z1z1); fsub(f, &z3, &z3, &hh);
    o->x = x3; o->y = y3; o->z = z3;
}
static void jac_to_affine(const cparams *c, oaffine *o, const ojac *p) {
    const fparams *f = c->f;
    if (jac_is_id(p)) { memset(o, 0, sizeof(*o)); return; }
    ofe zi, zi2, zi3;
    finv(f, &zi, &p->z); fsqr(f, &zi2, &zi); fmul(f, &zi3, &zi2, &zi);
    fmul(f, &o->x, &p->x, &zi2); fmul(f, &o->y, &p->y, &zi3);
}

void o_point_add(int curve, const oaffine *a, const oaffine *b, oaffine *out) {
    curve_init(); const cparams *c = &C[curve];
    ojac p; jac_from_affine(c, &p, a); jac_add_affine(c, &p, &p, b); jac_to_affine(c, out, &p);
}
static void jac_mul_canon(const cparams *c, ojac *o, const uint64_t k[4], const oaffine *p) {
    ojac acc; jac_set_id(&acc);
    for (int i = 255; i >= 0; --i) {
        jac_dbl(c, &acc, &acc);
        if ((k[i >> 6] >> (i & 63)) & 1) jac_add_affine(c, &acc, &acc, p);
    }
    *o = acc;
}
void o_point_mul(int curve, const ofe *s, const oaffine *p, oaffine *out) {
    curve_init(); const cparams *c = &C[curve];
    ofe k; ffrom_mont(c->sf, &k, s);
    ojac r; jac_mul_canon(c, &r, k.l, p); jac_to_affine(c, out, &r);
}
int o_point_is_on_curve(int curve, const oaffine *p) {
    curve_init(); const cparams *c = &C[curve]; const fparams *f = c->f;
    if (aff_is_id(p)) return 1;
    ofe l, r; fsqr(f, &l, &p->y); fsqr(f, &r, &p->x); fmul(f, &r, &r, &p->x); fadd(f, &r, &r, &c->b);
    return fe_eq(&l, &r);
}

static uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
/* Synthetic commitment key for tests/benches: NOT CommitmentKey::setup (src/commitment.rs:55-79,
 * hash_to_curve is [3P] and not restatable).  Block j (4096 points) starts at [h_j]G with h_j from
 * splitmix64(seed, j) and walks P_{i+1} = P_i + [d_j]G; blocks run in parallel.  Any valid points
 * with no known small relation serve for parity and timing. */
void o_make_bases(int curve, uint64_t seed, oaffine *out, size_t n, int threads) {
    curve_init(); const cparams *c = &C[curve];
    threads = clamp_threads(threads);
    const size_t BLK = 4096; size_t nblk = (n + BLK - 1) / BLK;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (size_t j = 0; j < nblk; ++j) {
        uint64_t s = seed ^ (0xA5A5A5A5ULL + j * 0x100000001b3ULL);
        uint64_t h[4] = {splitmix64(&s), splitmix64(&s), splitmix64(&s), splitmix64(&s) >> 4};
        uint64_t d[4] = {splitmix64(&s), splitmix64(&s), splitmix64(&s), splitmix64(&s) >> 4};
        ojac P, D; jac_mul_canon(c, &P, h, &c->g); jac_mul_canon(c, &D, d, &c->g);
        oaffine Da; jac_to_affine(c, &Da, &D);
        size_t lo = j * BLK, hi = lo + BLK < n ? lo + BLK : n;
        /* batch-normalise the block: collect Jacobians then one inversion (Montgomery trick) */
        ojac *buf = (ojac *)malloc((hi - lo) * sizeof(ojac));
        ofe *pre = (ofe *)malloc((hi - lo) * sizeof(ofe));
        for (size_t i = lo; i < hi; ++i) { buf[i - lo] = P; jac_add_affine(c, &P, &P, &Da); }
        const fparams *f = c->f; ofe acc = f->r;
        for (size_t i = 0; i < hi - lo; ++i) { pre[i] = acc; if (!jac_is_id(&buf[i])) fmul(f, &acc, &acc, &buf[i].z); }
        ofe ai; finv(f, &ai, &acc);
        for (size_t i = hi - lo; i-- > 0;) {
            if (jac_is_id(&buf[i])) { memset(&out[lo + i], 0, sizeof(oaffine)); continue; }
            ofe zi, zi2, zi3; fmul(f, &zi, &ai, &pre[i]); fmul(f, &ai, &ai, &buf[i].z);
            fsqr(f, &zi2, &zi); fmul(f, &zi3, &zi2, &zi);
            fmul(f, &out[lo + i].x, &buf[i].x, &zi2); fmul(f, &out[lo + i].y, &buf[i].y, &zi3);
        }
        free(buf); free(pre);
    }
}

/* ------------------------------------------------------------------ MSM
 * src/commitment.rs:81-90: best_multiexp(v, &ck[..v.len()]).to_affine().
 * best_multiexp / multiexp_serial are [3P] halo2_proofs::arithmetic (not under /root/reference);
 * restated from the published halo2 source:
 *   - scalars -> to_repr() (canonical little-endian bytes)
 *   - c = 1 if n < 4, 3 if n < 32, else ceil(ln n); segments = 256/c + 1
 *   - MSB-first over segments: acc <<= c ; buckets[digit-1] += base (digit 0 skipped);
 *     running-sum bucket reduction (summation by parts)
 *   - if n > threads: chunk = n / threads, one multiexp_serial per chunk (chunks(chunk) may yield
 *     threads+1 chunks), results summed in order
 * Group arithmetic is exact, so the evaluation order cannot change the affine result.
 */
static inline size_t get_at(size_t segment, size_t c, const uint64_t k[4]) {
    size_t skip_bits = segment * c;
    if (skip_bits >= 256) return 0;
    size_t w = skip_bits >> 6, b = skip_bits & 63;
    uint64_t v = k[w] >> b;
    if (b && w + 1 < 4) v |= k[w + 1] << (64 - b);
    return (size_t)(v & ((1ULL << c) - 1));
}
static void multiexp_serial(const cparams *cp, const ofe *canon, const oaffine *bases, size_t n, ojac *acc) {
    size_t c;
    if (n < 4) c = 1; else if (n < 32) c = 3; else c = (size_t)ceil(log((double)(uint32_t)n));
    size_t segments = 256 / c + 1;
    size_t nb = ((size_t)1 << c) - 1;
    ojac *buckets = (ojac *)malloc(nb * sizeof(ojac));
    for (size_t seg = segments; seg-- > 0;) {
        for (size_t i = 0; i < c; ++i) jac_dbl(cp, acc, acc);
        for (size_t i = 0; i < nb; ++i) jac_set_id(&buckets[i]);
        for (size_t i = 0; i < n; ++i) {
            size_t d = get_at(seg, c, canon[i].l);
            if (d) jac_add_affine(cp, &buckets[d - 1], &buckets[d - 1], &bases[i]);
        }
        ojac run; jac_set_id(&run);
        for (size_t i = nb; i-- > 0;) { jac_add(cp, &run, &run, &buckets[i]); jac_add(cp, acc, acc, &run); }
    }
    free(buckets);
}
void o_msm(int curve, const ofe *scalars, const oaffine *bases, size_t n, int threads, oaffine *out) {
    curve_init(); const cparams *cp = &C[curve];
    threads = clamp_threads(threads);
    ojac total; jac_set_id(&total);
    if (n == 0) { jac_to_affine(cp, out, &total); return; }
    ofe *canon = (ofe *)malloc(n * sizeof(ofe));
#pragma omp parallel for num_threads(threads)
    for (size_t i = 0; i < n; ++i) ffrom_mont(cp->sf, &canon[i], &scalars[i]);
    if (n > (size_t)threads) {
        size_t chunk = n / (size_t)threads, nch = (n + chunk - 1) / chunk;
        ojac *res = (ojac *)malloc(nch * sizeof(ojac));
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
        for (size_t j = 0; j < nch; ++j) {
            size_t lo = j * chunk, len = lo + chunk <= n ? chunk : n - lo;
            jac_set_id(&res[j]);
            multiexp_serial(cp, canon + lo, bases + lo, len, &res[j]);
        }
        for (size_t j = 0; j < nch; ++j) jac_add(cp, &total, &total, &res[j]);
        free(res);
    } else {
        multiexp_serial(cp, canon, bases, n, &total);
    }
    free(canon);
    jac_to_affine(cp, out, &total);
}
void o_msm_naive(int curve, const ofe *scalars, const oaffine *bases, size_t n, oaffine *out) {
    curve_init(); const cparams *cp = &C[curve];
    ojac total; jac_set_id(&total);
    for (size_t i = 0; i < n; ++i) {
        ofe k; ffrom_mont(cp->sf, &k, &scalars[i]);
        ojac t; jac_mul_canon(cp, &t, k.l, &bases[i]); jac_add(cp, &total, &total, &t);
    }
    jac_to_affine(cp, out, &total);
}

/* ------------------------------------------------------------------ fft (src/fft.rs) */
static void get_omega_or_inv(ofe *o, uint32_t k, int inverse) {      /* src/fft.rs:12-23 */
    const fparams *f = &F[O_FR];
    *o = inverse ? FR_ROOT_INV : FR_ROOT;
    for (uint32_t i = k; i < FR_S; ++i) fsqr(f, o, o);
}
static size_t bitreverse(size_t x, unsigned bits) {                  /* src/fft.rs:41-49 */
    size_t r = 0; for (unsigned i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; } return r;
}
/* src/fft.rs:118-155 */
static void recursive_butterfly(ofe *a, size_t n, size_t twiddle_chunk, const ofe *tw, int depth) {
    const fparams *f = &F[O_FR];
    if (n == 2) {
        ofe t = a[1]; a[1] = a[0]; fadd(f, &a[0], &a[0], &t); fsub(f, &a[1], &a[1], &t);
        return;
    }
    ofe *left = a, *right = a + n / 2;
    if (depth > 0) {
#pragma omp task
        recursive_butterfly(left, n / 2, twiddle_chunk * 2, tw, depth - 1);
#pragma omp task
        recursive_butterfly(right, n / 2, twiddle_chunk * 2, tw, depth - 1);
#pragma omp taskwait
    } else {
        recursive_butterfly(left, n / 2, twiddle_chunk * 2, tw, 0);
        recursive_butterfly(right, n / 2, twiddle_chunk * 2, tw, 0);
    }
    ofe t = right[0]; right[0] = left[0]; fadd(f, &left[0], &left[0], &t); fsub(f, &right[0], &right[0], &t);
    for (size_t i = 1; i < n / 2; ++i) {
        fmul(f, &t, &right[i], &tw[i * twiddle_chunk]);
        right[i] = left[i]; fadd(f, &left[i], &left[i], &t); fsub(f, &right[i], &right[i], &t);
    }
}
/* src/fft.rs:61-115 */
static void best_fft(ofe *a, size_t n, const ofe *omega, unsigned log_n, int threads) {
    const fparams *f = &F[O_FR];
    for (size_t k = 0; k < n; ++k) { size_t rk = bitreverse(k, log_n); if (k < rk) { ofe t = a[k]; a[k] = a[rk]; a[rk] = t; } }
    size_t nt = n / 2 ? n / 2 : 1;
    ofe *tw = (ofe *)malloc(nt * sizeof(ofe));
    ofe w = f->r;
    for (size_t i = 0; i < n / 2; ++i) { tw[i] = w; fmul(f, &w, &w, omega); }
    unsigned log_threads = 0; while ((2u << log_threads) <= (unsigned)threads) ++log_threads;
    if (log_n <= log_threads) {
        size_t chunk = 2, twc = n / 2;
        for (unsigned s = 0; s < log_n; ++s) {
            for (size_t st = 0; st < n; st += chunk) {
                size_t h = chunk / 2; ofe *l = a + st, *r = a + st + h; ofe t;
                t = r[0]; r[0] = l[0]; fadd(f, &l[0], &l[0], &t); fsub(f, &r[0], &r[0], &t);
                for (size_t i = 1; i < h; ++i) {
                    fmul(f, &t, &r[i], &tw[i * twc]); r[i] = l[i]; fadd(f, &l[i], &l[i], &t); fsub(f, &r[i], &r[i], &t);
                }
            }
            chunk *= 2; twc /= 2;
        }
    } else if (n >= 2) {
#pragma omp parallel num_threads(threads)
#pragma omp single
        recursive_butterfly(a, n, 1, tw, (int)log_threads + 1);
    }
    free(tw);
}
static int ilog2_exact(size_t n, unsigned *k) {
    if (n == 0 || (n & (n - 1))) return 0;
    unsigned r = 0; while (((size_t)1 << r) < n) ++r; *k = r; return 1;
}
int o_fft(ofe *a, size_t n, int threads) {                            /* src/fft.rs:160-165 */
    oracle_init(); threads = clamp_threads(threads);
    unsigned k; if (!ilog2_exact(n, &k)) return 2; if (k > FR_S) return 3;
    ofe w; get_omega_or_inv(&w, k, 0); best_fft(a, n, &w, k, threads); return 0;
}
int o_ifft(ofe *a, size_t n, int threads) {                           /* src/fft.rs:168-182 */
    oracle_init(); threads = clamp_threads(threads);
    const fparams *f = &F[O_FR];
    unsigned k; if (!ilog2_exact(n, &k)) return 2; if (k > FR_S) return 3;
    ofe w; get_omega_or_inv(&w, k, 1);
    ofe d = f->r; for (unsigned i = 0; i < k; ++i) fmul(f, &d, &d, &FR_TWO_INV);   /* TWO_INV^k, :25-27 */
    best_fft(a, n, &w, k, threads);
#pragma omp parallel for num_threads(threads)
    for (size_t i = 0; i < n; ++i) fmul(f, &a[i], &a[i], &d);
    return 0;
}
static void distribute_powers_zeta(ofe *a, size_t n, int into_coset, int threads) {   /* src/fft.rs:206-228 */
    const fparams *f = &F[O_FR];
    const ofe *cp0 = into_coset ? &FR_ZETA : &FR_ZETA2, *cp1 = into_coset ? &FR_ZETA2 : &FR_ZETA;
#pragma omp parallel for num_threads(threads)
    for (size_t i = 0; i < n; ++i) {
        size_t j = i % 3;
        if (j == 1) fmul(f, &a[i], &a[i], cp0); else if (j == 2) fmul(f, &a[i], &a[i], cp1);
    }
}
int o_coset_fft(ofe *a, size_t n, int threads) {
    oracle_init(); threads = clamp_threads(threads);
    unsigned k; if (!ilog2_exact(n, &k)) return 2; if (k > FR_S) return 3;
    distribute_powers_zeta(a, n, 1, threads); return o_fft(a, n, threads);
}
int o_coset_ifft(ofe *a, size_t n, int threads) {
    oracle_init(); threads = clamp_threads(threads);
    int rc = o_ifft(a, n, threads); if (rc) return rc;
    distribute_powers_zeta(a, n, 0, threads); return 0;
}

/* ------------------------------------------------------------------ row-program interpreter
 * src/polynomial/graph_evaluator.rs:93-149 (Calculation::evaluate), :361-388 (evaluate);
 * value fetch = GetDataForEval::eval_column_var (src/plonk/eval.rs:57-69) +
 * PlonkEvalDomain::eval_advice_var (src/plonk/eval.rs:153-228):
 *   Poly{index}: index < n_sel -> selector bit; < n_sel+n_fixed -> fixed; else fold variable
 *   j = index - ..., width = num_advice + 5 * num_lookup: j < width -> first instance (W1s), else second (W2s);
 *   (round, column) = index_map(j), eval.rs:169-201 -- depends on the NUMBER of rounds of that instance.
 * Rotation: (row + rot) rem_euclid rows  (graph_evaluator.rs:51-53).
 */
enum { K_CONST = 0, K_INTER = 1, K_FIXED = 2, K_POLY = 3, K_CHAL = 4 };
enum { OP_ADD = 0, OP_SUB, OP_MUL, OP_SQUARE, OP_DOUBLE, OP_NEGATE, OP_STORE };

static inline int fetch(const o_eval_domain *d, const fparams *f, int64_t kind, int64_t idx, int64_t rot,
                        const size_t *rots, const ofe *consts, const ofe *inter, ofe *o) {
    switch (kind) {
    case K_CONST: *o = consts[idx]; return 0;
    case K_INTER: *o = inter[idx]; return 0;
    case K_FIXED: if ((size_t)idx >= d->n_fixed) return 1; *o = d->fixed[idx][rots[rot]]; return 0;
    case K_CHAL: if ((size_t)idx >= d->n_challenges) return 1; *o = d->challenges[idx]; return 0;
    case K_POLY: {
        size_t row = rots[rot], i = (size_t)idx;
        if (i < d->n_sel) { if (d->selectors[i][row]) *o = f->r; else memset(o, 0, sizeof(*o)); return 0; }
        i -= d->n_sel;
        if (i < d->n_fixed) { *o = d->fixed[i][row]; return 0; }
        i -= d->n_fixed;
        const size_t width = d->num_advice + 5 * d->num_lookup;
        const int first = i < width;
        if (!first) i -= width;
        const size_t nw = first ? d->n_w1 : d->n_w2;
        const ofe *const *Ws = first ? d->W1s : d->W2s;
        const size_t *lens = first ? d->w1_len : d->w2_len;
        size_t round, col;
        if (i < d->num_advice) { round = 0; col = i; }
        else {
            size_t li = (i - d->num_advice) / 5, sub = (i - d->num_advice) % 5;
            int first_round = sub < 3; if (!first_round) sub -= 3;
            if (nw == 2) { if (first_round) { round = 0; col = d->num_advice + li * 3 + sub; } else { round = 1; col = li * 2 + sub; } }
            else if (nw == 3) { if (first_round) { round = 1; col = li * 3 + sub; } else { round = 2; col = li * 2 + sub; } }
            else return 1;
        }
        if (nw <= round || lens[round] <= col * d->rows + row) return 1;
        *o = Ws[round][col * d->rows + row];
        return 0;
        return 1;
    }
    }
    return 1;
}
int o_eval_program(const o_eval_domain *d, const int64_t *calcs, size_t n_calcs, const ofe *constants,
                   size_t n_constants, const int32_t *rotations, size_t n_rot, size_t n_inter, ofe *out, int threads) {
    oracle_init(); threads = clamp_threads(threads);
    (void)n_constants;
    const fparams *f = &F[d->field];
    int err = 0;
#pragma omp parallel num_threads(threads)
    {
        ofe *inter = (ofe *)calloc(n_inter ? n_inter : 1, sizeof(ofe));
        size_t *rots = (size_t *)calloc(n_rot ? n_rot : 1, sizeof(size_t));
#pragma omp for
        for (size_t row = 0; row < d->rows; ++row) {
            for (size_t r = 0; r < n_rot; ++r) {
                int64_t v = ((int64_t)row + rotations[r]) % (int64_t)d->rows; if (v < 0) v += (int64_t)d->rows;
                rots[r] = (size_t)v;
            }
            for (size_t ci = 0; ci < n_calcs; ++ci) {
                const int64_t *c = calcs + ci * 8; ofe a, b, r;
                if (fetch(d, f, c[1], c[2], c[3], rots, constants, inter, &a)) { err = 1; continue; }
                switch (c[0]) {
                case OP_ADD: if (fetch(d, f, c[4], c[5], c[6], rots, constants, inter, &b)) { err = 1; continue; } fadd(f, &r, &a, &b); break;
                case OP_SUB: if (fetch(d, f, c[4], c[5], c[6], rots, constants, inter, &b)) { err = 1; continue; } fsub(f, &r, &a, &b); break;
                case OP_MUL: if (fetch(d, f, c[4], c[5], c[6], rots, constants, inter, &b)) { err = 1; continue; } fmul(f, &r, &a, &b); break;
                case OP_SQUARE: fsqr(f, &r, &a); break;
                case OP_DOUBLE: fdbl(f, &r, &a); break;
                case OP_NEGATE: fneg(f, &r, &a); break;
                default: r = a; break;
                }
                inter[c[7]] = r;
            }
            if (n_calcs) out[row] = inter[calcs[(n_calcs - 1) * 8 + 7]]; else memset(&out[row], 0, sizeof(ofe));
        }
        free(inter); free(rots);
    }
    return err;
}

/* ------------------------------------------------------------------ folds
 * src/nifs/sangria/accumulator.rs:364-376 : W[j][i] = W1[j][i] + r * W2[j][i]
 * src/nifs/sangria/accumulator.rs:385-398 : E[i] = E[i] + sum_k r^(k+1) * T_k[i]   (k = 0..n_terms)
 */
void o_fold_w(int field, const ofe *w1, const ofe *w2, const ofe *r, ofe *out, size_t n, int threads) {
    oracle_init(); threads = clamp_threads(threads); const fparams *f = &F[field];
#pragma omp parallel for num_threads(threads)
    for (size_t i = 0; i < n; ++i) { ofe t; fmul(f, &t, r, &w2[i]); fadd(f, &out[i], &w1[i], &t); }
}
void o_fold_e(int field, const ofe *e, const ofe *const *t, size_t n_terms, const ofe *r, ofe *out, size_t n, int threads) {
    oracle_init(); threads = clamp_threads(threads); const fparams *f = &F[field];
    ofe *pw = (ofe *)malloc((n_terms ? n_terms : 1) * sizeof(ofe));
    ofe acc = *r; for (size_t k = 0; k < n_terms; ++k) { pw[k] = acc; fmul(f, &acc, &acc, r); }
#pragma omp parallel for num_threads(threads)
    for (size_t i = 0; i < n; ++i) {
        ofe a = e[i];
        for (size_t k = 0; k < n_terms; ++k) { ofe m; fmul(f, &m, &pw[k], &t[k][i]); fadd(f, &a, &a, &m); }
        out[i] = a;
    }
    free(pw);
}

/* ------------------------------------------------------------------ ProtoGalaxy
 * src/nifs/protogalaxy/mod.rs:176-210 (fold_witness), poly/folded_witness.rs:20-180 (FoldedWitness::new):
 *   out[i] = sum_j coef[j] * w[j][i]
 */
void o_lincomb(int field, const ofe *const *w, const ofe *coef, size_t J, ofe *out, size_t n, int threads) {
    oracle_init(); threads = clamp_threads(threads); const fparams *f = &F[field];
#pragma omp parallel for num_threads(threads)
    for (size_t i = 0; i < n; ++i) {
        ofe a; fmul(f, &a, &coef[0], &w[0][i]);
        for (size_t j = 1; j < J; ++j) { ofe m; fmul(f, &m, &coef[j], &w[j][i]); fadd(f, &a, &a, &m); }
        out[i] = a;
    }
}

/* The binary reduction tree of compute_F / compute_G (src/nifs/protogalaxy/poly/mod.rs:68-203, 308-425):
 *   node(height h) = left + right * weight[h]        (leaves are height 0; `reducer`, :127-163 and :333-361)
 * over `n` (a power of two) leaves, for P independent points:
 *   leaf i of point p = leaves[p * leaves_stride + i] if i < count else 0     (count = rows * gates; padding leaves are 0,
 *                        get_evaluate_witness_fn returns zero beyond the last gate, src/plonk/mod.rs:705-711)
 *   weights of point p = weights[p * t + h], h < t = log2(n)
 * compute_F: ONE leaf table (leaves_stride = 0), 32 weight vectors beta + X_p * delta; compute_G: one table per point
 * (the folded witnesses), the same weights beta' for every point.  out[p] = root.  Serial recursion per point below
 * 2^12 leaves, OpenMP tasks above (the reference: rayon::join above 2^18, :171-186).
 */
static void pg_tree_rec(const fparams *f, const ofe *leaves, size_t count, const ofe *w, size_t lo, size_t len, unsigned h, ofe *out) {
    if (len == 1) {
        if (lo < count) *out = leaves[lo]; else memset(out, 0, sizeof(*out));
        return;
    }
    ofe l, r, m;
    if (len > 4096) {
#pragma omp task shared(l)
        pg_tree_rec(f, leaves, count, w, lo, len / 2, h - 1, &l);
#pragma omp task shared(r)
        pg_tree_rec(f, leaves, count, w, lo + len / 2, len / 2, h - 1, &r);
#pragma omp taskwait
    } else {
        pg_tree_rec(f, leaves, count, w, lo, len / 2, h - 1, &l);
        pg_tree_rec(f, leaves, count, w, lo + len / 2, len / 2, h - 1, &r);
    }
    fmul(f, &m, &r, &w[h - 1]);
    fadd(f, out, &l, &m);
}
void o_pg_tree(int field, const ofe *leaves, size_t leaves_stride, size_t n, size_t count, const ofe *weights, size_t P, size_t t,
               ofe *out, int threads) {
    oracle_init(); threads = clamp_threads(threads); const fparams *f = &F[field];
    for (size_t p = 0; p < P; ++p) {
#pragma omp parallel num_threads(threads)
#pragma omp single
        pg_tree_rec(f, leaves + p * leaves_stride, count, weights + p * t, 0, n, (unsigned)t, &out[p]);
    }
}

