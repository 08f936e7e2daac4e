// curve.cuh -- group law for the bn256 / grumpkin cycle (y^2 = x^3 + b, a = 0) on gfx950.
//
// The reference reaches the group only through halo2curves [3P] (`best_multiexp`,
// `to_affine`, src/commitment.rs:81-90; single scalar-muls in src/nifs/sangria/accumulator.rs:213,243).
// Affine points use the halo2curves in-memory layout: x || y, Montgomery form, identity = (0,0).
//
// Accumulators use extended-Jacobian XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// a mixed add costs 8M + 2S, the cheapest complete-enough form for bucket accumulation, and the
// identity is ZZ = 0.  Formulas: EFD shortw/xyzz  madd-2008-s, add-2008-s, dbl-2008-s-1, mdbl-2008-s-1.
#pragma once
#include "field.cuh"
#include "lanes.cuh"

namespace srs {

struct alignas(16) affine_t {
    fe_t x, y;
};
struct alignas(16) xyzz_t {
    fe_t x, y, zz, zzz;
};

// Curve parameter packs: F = base field of the coordinates, S = scalar field.
struct Bn256 {
    using F = Fq;
    using S = Fr;
    static constexpr int ID = 0;
};
struct Grumpkin {
    using F = Fr;
    using S = Fq;
    static constexpr int ID = 1;
};

// ---- quad cooperation: 4 adjacent lanes (a DPP "quad") hold identical operands and share one group addition ----
// The tail of an MSM (bucket-part levels, row/column sums, weighted suffix sums) is a chain of ~30 DEPENDENT XYZZ
// additions on a nearly idle chip; one addition is 14 dependent-or-not modmuls that a single lane issues back to back
// (~12 us).  Spread over a quad, the 14 products form 4 levels of <= 4 independent products, exchanged with DPP
// quad_perm moves (register crossbar, no LDS): ~3.5 us per addition for ~20 % more lane-cycles.
template <int K>
SRS_D fe_t quad_bcast(const fe_t &x) {        // quad_bcast_words: lanes.cuh
    fe_t o;
    quad_bcast_words<K, 8>(x.v, o.v);
    return o;
}
// operand of the lane's role: value selects limb by limb on by-value arguments.  (A nested ?: over references, or
// reference parameters here, make the compiler select POINTERS and park the points in scratch memory -- measured:
// 144-336 B of scratch per lane and no latency gain at all; tools/quad_probe.hip.)
SRS_D fe_t quad_select(uint32_t q, fe_t a0, fe_t a1, fe_t a2, fe_t a3) {   // BY VALUE, see below
    fe_t o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t x = a0.v[i];
        x = q == 1 ? a1.v[i] : x;
        x = q == 2 ? a2.v[i] : x;
        x = q == 3 ? a3.v[i] : x;
        o.v[i] = x;
    }
    return o;
}

SRS_HD fe_t fe_select(bool c, const fe_t &x, const fe_t &y) {   // c ? x : y, limb by limb
    fe_t o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = c ? x.v[i] : y.v[i];
    return o;
}

template <class C>
struct Ec {
    using F = typename C::F;

    SRS_HD static bool is_identity(const affine_t &p) { return F::is_zero(p.x) && F::is_zero(p.y); }
    SRS_HD static bool is_identity(const xyzz_t &p) { return F::is_zero(p.zz); }
    SRS_HD static xyzz_t identity() {
        xyzz_t o;
        o.x = F::zero();
        o.y = F::zero();
        o.zz = F::zero();
        o.zzz = F::zero();
        return o;
    }
    SRS_HD static affine_t affine_identity() {
        affine_t o;
        o.x = F::zero();
        o.y = F::zero();
        return o;
    }
    SRS_HD static xyzz_t from_affine(const affine_t &p) {
        xyzz_t o;
        if (is_identity(p)) return identity();
        o.x = p.x;
        o.y = p.y;
        o.zz = F::one();
        o.zzz = F::one();
        return o;
    }
    SRS_HD static affine_t neg(const affine_t &p) {
        affine_t o;
        o.x = p.x;
        o.y = F::neg(p.y);   // neg(0) = 0 keeps the identity encoding
        return o;
    }
    SRS_HD static xyzz_t neg(const xyzz_t &p) {
        xyzz_t o = p;
        o.y = F::neg(p.y);
        return o;
    }

    // 2*P for an affine, non-identity P (mdbl-2008-s-1)
    SRS_HD static xyzz_t dbl_affine(const affine_t &p) {
        xyzz_t o;
        fe_t u = F::dbl(p.y);
        fe_t v = F::sqr(u);
        fe_t w = F::mul(u, v);
        fe_t s = F::mul(p.x, v);
        fe_t xx = F::sqr(p.x);
        fe_t m = F::add(F::dbl(xx), xx);
        o.x = F::sub(F::sqr(m), F::dbl(s));
        o.y = F::sub(F::mul(m, F::sub(s, o.x)), F::mul(w, p.y));
        o.zz = v;
        o.zzz = w;
        return o;
    }
    // 2*P (dbl-2008-s-1)
    SRS_HD static xyzz_t dbl(const xyzz_t &p) {
        if (is_identity(p)) return p;
        xyzz_t o;
        fe_t u = F::dbl(p.y);
        fe_t v = F::sqr(u);
        fe_t w = F::mul(u, v);
        fe_t s = F::mul(p.x, v);
        fe_t xx = F::sqr(p.x);
        fe_t m = F::add(F::dbl(xx), xx);
        o.x = F::sub(F::sqr(m), F::dbl(s));
        o.y = F::sub(F::mul(m, F::sub(s, o.x)), F::mul(w, p.y));
        o.zz = F::mul(v, p.zz);
        o.zzz = F::mul(w, p.zzz);
        return o;
    }
    // acc + Q, Q affine (madd-2008-s), complete: handles acc = O, Q = O, Q = +-acc
    SRS_HD static xyzz_t madd(const xyzz_t &a, const affine_t &q) {
        if (is_identity(q)) return a;
        if (is_identity(a)) {
            xyzz_t o;
            o.x = q.x;
            o.y = q.y;
            o.zz = F::one();
            o.zzz = F::one();
            return o;
        }
        fe_t u2 = F::mul(q.x, a.zz);
        fe_t s2 = F::mul(q.y, a.zzz);
        fe_t p = F::sub(u2, a.x);
        fe_t r = F::sub(s2, a.y);
        if (F::is_zero(p)) {
            if (F::is_zero(r)) return dbl_affine(q);
            return identity();
        }
        fe_t pp = F::sqr(p);
        fe_t ppp = F::mul(p, pp);
        fe_t qv = F::mul(a.x, pp);
        xyzz_t o;
        o.x = F::sub(F::sub(F::sqr(r), ppp), F::dbl(qv));
        o.y = F::sub(F::mul(r, F::sub(qv, o.x)), F::mul(a.y, ppp));
        o.zz = F::mul(a.zz, pp);
        o.zzz = F::mul(a.zzz, ppp);
        return o;
    }
    // a + b, both XYZZ (add-2008-s), complete
    SRS_HD static xyzz_t add(const xyzz_t &a, const xyzz_t &b) {
        if (is_identity(b)) return a;
        if (is_identity(a)) return b;
        fe_t u1 = F::mul(a.x, b.zz);
        fe_t u2 = F::mul(b.x, a.zz);
        fe_t s1 = F::mul(a.y, b.zzz);
        fe_t s2 = F::mul(b.y, a.zzz);
        fe_t p = F::sub(u2, u1);
        fe_t r = F::sub(s2, s1);
        if (F::is_zero(p)) {
            if (F::is_zero(r)) return dbl(a);
            return identity();
        }
        fe_t pp = F::sqr(p);
        fe_t ppp = F::mul(p, pp);
        fe_t qv = F::mul(u1, pp);
        xyzz_t o;
        o.x = F::sub(F::sub(F::sqr(r), ppp), F::dbl(qv));
        o.y = F::sub(F::mul(r, F::sub(qv, o.x)), F::mul(s1, ppp));
        o.zz = F::mul(F::mul(a.zz, b.zz), pp);
        o.zzz = F::mul(F::mul(a.zzz, b.zzz), ppp);
        return o;
    }
    // a + b computed by the 4 lanes of a quad together (add-2008-s split into 4 levels of independent products).
    // Preconditions: the 4 lanes hold the same a and b, q = lane index inside the quad, the quad is convergent.
    // All 4 lanes return the same sum (bit-identical to add(a, b)).
    SRS_D static xyzz_t add_quad(const xyzz_t &a, const xyzz_t &b, uint32_t q) {
        // identity operands: the 4 lanes hold the same a, b, so the branch is quad-uniform and the DPP exchanges
        // below still see whole quads
        const bool ia = is_identity(a), ib = is_identity(b);
        xyzz_t o;
        if (ia || ib) {                            // a + O = a ; O + b = b
            o.x = fe_select(ib, a.x, b.x);
            o.y = fe_select(ib, a.y, b.y);
            o.zz = fe_select(ib, a.zz, b.zz);
            o.zzz = fe_select(ib, a.zzz, b.zzz);
            return o;
        }
        // level 1: u1 = X1 ZZ2 | u2 = X2 ZZ1 | s1 = Y1 ZZZ2 | s2 = Y2 ZZZ1
        fe_t m1 = F::mul(quad_select(q, a.x, b.x, a.y, b.y), quad_select(q, b.zz, a.zz, b.zzz, a.zzz));
        fe_t u1 = quad_bcast<0>(m1), u2 = quad_bcast<1>(m1), s1 = quad_bcast<2>(m1), s2 = quad_bcast<3>(m1);
        fe_t p = F::sub(u2, u1), r = F::sub(s2, s1);
        // level 2: PP = P^2 | RR = R^2 | ZZ1 ZZ2 | ZZZ1 ZZZ2
        fe_t m2 = F::mul(quad_select(q, p, r, a.zz, a.zzz), quad_select(q, p, r, b.zz, b.zzz));
        fe_t pp = quad_bcast<0>(m2), rr = quad_bcast<1>(m2), zzz12 = quad_bcast<3>(m2);
        // level 3: PPP = P PP | Q = U1 PP | ZZ3 = ZZ1 ZZ2 PP | (lane 3 repeats lane 0)
        fe_t m3 = F::mul(quad_select(q, p, u1, m2, p), pp);
        fe_t ppp = quad_bcast<0>(m3), qv = quad_bcast<1>(m3), zz3 = quad_bcast<2>(m3);
        fe_t x3 = F::sub(F::sub(rr, ppp), F::dbl(qv));
        // level 4: R (Q - X3) | S1 PPP | ZZZ3 = ZZZ1 ZZZ2 PPP | (lane 3 repeats lane 1)
        fe_t m4 = F::mul(quad_select(q, r, s1, zzz12, s1), quad_select(q, F::sub(qv, x3), ppp, ppp, ppp));
        fe_t t2 = quad_bcast<0>(m4), t1 = quad_bcast<1>(m4), zzz3 = quad_bcast<2>(m4);
        o.x = x3;
        o.y = F::sub(t2, t1);
        o.zz = zz3;
        o.zzz = zzz3;
        if (F::is_zero(p)) o = F::is_zero(r) ? dbl(a) : identity();   // P2 = +-P1 (quad-uniform as well)
        return o;
    }
    // -> affine (one field inversion): x = X/ZZ, y = Y/ZZZ
    SRS_HD static affine_t to_affine(const xyzz_t &p) {
        if (is_identity(p)) return affine_identity();
        fe_t i = F::inv(F::mul(p.zz, p.zzz));
        affine_t o;
        o.x = F::mul(p.x, F::mul(i, p.zzz));
        o.y = F::mul(p.y, F::mul(i, p.zz));
        return o;
    }
    // [k]P, k = 256-bit canonical scalar as 8 u32 limbs, MSB-first double-and-add
    SRS_HD static xyzz_t mul_canon(const uint32_t k[8], const affine_t &p) {
        xyzz_t acc = identity();
        for (int i = 255; i >= 0; --i) {
            acc = dbl(acc);
            if ((k[i >> 5] >> (i & 31)) & 1u) acc = madd(acc, p);
        }
        return acc;
    }
    // y^2 == x^3 + b   (b = 3 on bn256, -17 on grumpkin)
    SRS_HD static bool is_on_curve(const affine_t &p) {
        if (is_identity(p)) return true;
        fe_t b = (C::ID == 0) ? F::from_u64(3) : F::neg(F::from_u64(17));
        fe_t l = F::sqr(p.y);
        fe_t r = F::add(F::mul(F::sqr(p.x), p.x), b);
        return F::eq(l, r);
    }
};

using EcBn = Ec<Bn256>;
using EcGr = Ec<Grumpkin>;

}  // namespace srs
