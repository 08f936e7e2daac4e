// curve29.cuh -- the mixed addition of the MSM bucket accumulation on the 9 x 29-bit limb form (field29.cuh).
//
// Same formulas as Ec<C>::madd (curve.cuh; EFD shortw/xyzz madd-2008-s, mdbl-2008-s-1), same completeness (identity
// operands, Q = +-acc), exact arithmetic -> the same group element.  Coordinates are in R' = 2^261 Montgomery form and
// are kept LAZILY: a product returns a value < 2P, sums and differences are small multiples of P above the canonical
// value and are not reduced.  The bound of every intermediate is stated beside it; the multiplier needs
// A * B < 2^261 * P (> 165 P^2), one operand with limbs < 2^29 and the other < 2^31 (or both < 2^30).
// "norm" = limbs < 2^29 (after Fp29::normalize or out of a product).
#pragma once
#include "curve.cuh"
#include "field29.cuh"

namespace srs {

struct aff29_t {
    f29_t x, y;            // norm, < P (table entries); identity = (0, 0) exactly
};
struct xyzz29_t {
    f29_t x, y, zz, zzz;   // x < 9P norm; y < 5P, limbs < 2^31; zz, zzz < 2P norm; identity: zz == 0 exactly
};

// quad exchange of 9-limb values (see quad_bcast / quad_select in curve.cuh)
template <int K>
SRS_D f29_t quad_bcast29(const f29_t &x) {
    f29_t o;
    quad_bcast_words<K, 9>(x.v, o.v);
    return o;
}
SRS_D f29_t quad_select29(uint32_t q, f29_t a0, f29_t a1, f29_t a2, f29_t a3) {   // BY VALUE (see quad_select)
    f29_t o;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint32_t x = a0.v[i];
        x = q == 1 ? a1.v[i] : x;
        x = q == 2 ? a2.v[i] : x;
        x = q == 3 ? a3.v[i] : x;
        o.v[i] = x;
    }
    return o;
}

template <class C>
struct Ec29 {
    using P = typename C::F::Params;
    using F = Fp29<P>;

    // 2^261 mod p (the Montgomery one of this form), limb i -- folded at compile time
    SRS_HD static constexpr uint32_t one_limb(int i) {
        uint32_t w[8] = {P::r(0), P::r(1), P::r(2), P::r(3), P::r(4), P::r(5), P::r(6), P::r(7)};   // 2^256 mod p
        for (int d = 0; d < 5; ++d) {
            uint32_t c = 0;
            for (int j = 0; j < 8; ++j) {                       // w <- 2 w   (w < p < 2^254: no carry out)
                uint32_t n = (w[j] << 1) | c;
                c = w[j] >> 31;
                w[j] = n;
            }
            uint32_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            uint32_t borrow = 0;
            for (int j = 0; j < 8; ++j) {                       // t <- w - p
                uint64_t x = (uint64_t)w[j] - P::p(j) - borrow;
                t[j] = (uint32_t)x;
                borrow = (uint32_t)(x >> 32) & 1u;
            }
            if (!borrow)
                for (int j = 0; j < 8; ++j) w[j] = t[j];
        }
        const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t two = (uint64_t)w[wi] | (wi + 1 < 8 ? (uint64_t)w[wi + 1] << 32 : 0);
        uint32_t limb = (uint32_t)(two >> sh);
        return i < 8 ? (limb & F::MASK) : limb;
    }
    // 2^256 mod p as a plain integer in limbs: product with it takes R'-form back to the ABI's 2^256 form
    SRS_HD static constexpr uint32_t r256_limb(int i) {
        const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t two = (uint64_t)P::r(wi) | (wi + 1 < 8 ? (uint64_t)P::r(wi + 1) << 32 : 0);
        uint32_t limb = (uint32_t)(two >> sh);
        return i < 8 ? (limb & F::MASK) : limb;
    }
    SRS_HD static f29_t one() {
        f29_t o;
#pragma unroll
        for (int i = 0; i < 9; ++i) o.v[i] = one_limb(i);
        return o;
    }

    SRS_HD static xyzz29_t identity() {
        xyzz29_t o;
        o.x = F::zero();
        o.y = F::zero();
        o.zz = F::zero();
        o.zzz = F::zero();
        return o;
    }
    SRS_HD static bool is_identity(const xyzz29_t &a) { return F::is_zero_exact(a.zz); }
    SRS_HD static bool is_identity(const aff29_t &q) { return F::is_zero_exact(q.x) && F::is_zero_exact(q.y); }

    // table entry (8 x u32, R'-form, canonical) -> registers; negate: -Q  (y -> P - y; the identity stays (0, 0))
    SRS_HD static aff29_t load(const affine_t &q, bool negate) {
        aff29_t o;
        o.x = F::unpack(q.x);
        f29_t y = F::unpack(q.y);
        f29_t ny = F::normalize(F::template neg_lazy<1, 0>(y));       // P - y  (top limb may pass through -1: wraps back)
        o.y = F::select(negate && !F::is_zero_exact(y), ny, y);
        return o;
    }

    // 2 Q for affine, non-identity Q (mdbl-2008-s-1)
    SRS_HD static xyzz29_t dbl_affine(const aff29_t &q) {
        xyzz29_t o;
        f29_t u = F::add_lazy(q.y, q.y);                                        // 2y < 2P, limbs < 2^30
        f29_t v = F::sqr(u);                                                    // < 2P norm
        f29_t w = F::mul(u, v);                                                 // < 2P norm
        f29_t s = F::mul(q.x, v);                                               // < 2P norm
        f29_t xx = F::sqr(q.x);                                                 // < 2P norm
        f29_t m = F::normalize(F::add_lazy(F::add_lazy(xx, xx), xx));           // 3 xx < 6P norm
        f29_t mm = F::sqr(m);                                                   // 36 P^2: < 2P norm
        f29_t x3 = F::normalize(F::template sub_lazy<5, 1>(mm, F::add_lazy(s, s)));   // mm - 2s + 5P in (P, 7P) norm
        f29_t t = F::template sub_lazy<8, 0>(s, x3);                            // s - x3 + 8P in (P, 10P), limbs < 2^31
        f29_t m1 = F::mul(t, m);                                                // 60 P^2
        f29_t m2 = F::mul(w, q.y);
        o.x = x3;
        o.y = F::template sub_lazy<3, 0>(m1, m2);                               // (P, 5P), limbs < 2^31
        o.zz = v;
        o.zzz = w;
        return o;
    }

    // acc + Q (madd-2008-s), complete: acc = O, Q = O, Q = +-acc
    SRS_HD static xyzz29_t madd(const xyzz29_t &a, const aff29_t &q) {
        if (is_identity(q)) return a;
        if (is_identity(a)) {
            xyzz29_t o;
            o.x = q.x;
            o.y = q.y;
            o.zz = one();
            o.zzz = o.zz;
            return o;
        }
        f29_t u2 = F::mul(a.zz, q.x);                                           // < 2P norm
        f29_t s2 = F::mul(a.zzz, q.y);                                          // < 2P norm
        f29_t p = F::normalize(F::template sub_lazy<10, 0>(u2, a.x));           // u2 - X1 + 10P in (P, 12P) norm
        f29_t r = F::normalize(F::template sub_lazy<6, 2>(s2, a.y));            // s2 - Y1 + 6P  in (P, 8P)  norm
        f29_t pp = F::sqr(p);                                                   // 144 P^2: < 2P norm
        f29_t rr = F::sqr(r);                                                   // 64 P^2
        if (F::is_zero_mod(pp)) {                                               // P == 0: same x
            if (F::is_zero_mod(rr)) return dbl_affine(q);
            return identity();
        }
        f29_t ppp = F::mul(p, pp);                                              // 24 P^2
        f29_t qv = F::mul(a.x, pp);                                             // 18 P^2
        xyzz29_t o;
        f29_t sub = F::add_lazy(ppp, F::add_lazy(qv, qv));                      // ppp + 2 qv < 6P, limbs < 3 * 2^29
        o.x = F::normalize(F::template sub_lazy<7, 2>(rr, sub));                // (P, 9P) norm
        f29_t t = F::template sub_lazy<10, 0>(qv, o.x);                         // qv - X3 + 10P in (P, 12P), limbs < 2^31
        f29_t m1 = F::mul(t, r);                                                // 96 P^2
        f29_t m2 = F::mul(a.y, ppp);                                            // 10 P^2
        o.y = F::template sub_lazy<3, 0>(m1, m2);                               // (P, 5P), limbs < 2^31
        o.zz = F::mul(a.zz, pp);
        o.zzz = F::mul(a.zzz, ppp);
        return o;
    }

    // The same addition for the bucket accumulation (k_accum0), r03: acc + Q or acc - Q straight from the RAW table entry.
    //   * the sign is folded into the formula -- r = +-s2 - Y1 -- instead of negating y when the entry is loaded (a negation,
    //     a carry normalisation, a zero test and a select per gathered point);
    //   * Y3 = t r - Y1 ppp is ONE double-width sum with ONE Montgomery reduction (Fp29::mul2): 9 reductions per addition, not 10.
    // Invariant of the accumulator along a chain of these: y NORMALISED (limbs < 2^29), value < 5P; x, zz, zzz as before.
    SRS_HD static xyzz29_t madd_signed(const xyzz29_t &a, const aff29_t &q, bool neg) {
        if (is_identity(q)) return a;
        if (is_identity(a)) {
            xyzz29_t o;
            o.x = q.x;
            o.y = neg ? F::normalize(F::template neg_lazy<1, 0>(q.y)) : q.y;     // P - y: y != 0 on a curve of odd order
            o.zz = one();
            o.zzz = o.zz;
            return o;
        }
        f29_t u2 = F::mul(a.zz, q.x);                                           // < 2P norm
        f29_t s2 = F::mul(a.zzz, q.y);                                          // < 2P norm
        f29_t p = F::normalize(F::template sub_lazy<10, 0>(u2, a.x));           // u2 - X1 + 10P in (P, 12P) norm
        f29_t rpos = F::template sub_lazy<6, 0>(s2, a.y);                       //  s2 - Y1 + 6P in (P, 8P), limbs < 3 * 2^29
        f29_t rneg = F::template neg_lazy<8, 1>(F::add_lazy(s2, a.y));          // -s2 - Y1 + 8P in (P, 8P], limbs < 3 * 2^29
        f29_t r = F::normalize(F::select(neg, rneg, rpos));
        f29_t pp = F::sqr(p);                                                   // 144 P^2: < 2P norm
        f29_t rr = F::sqr(r);                                                   // 64 P^2
        if (F::is_zero_mod(pp)) {                                               // same x: acc = +-(+-Q)
            if (F::is_zero_mod(rr)) {
                aff29_t qq = q;
                if (neg) qq.y = F::normalize(F::template neg_lazy<1, 0>(q.y));
                xyzz29_t d = dbl_affine(qq);
                d.y = F::normalize(d.y);                                        // keep the chain's invariant
                return d;
            }
            return identity();
        }
        f29_t ppp = F::mul(p, pp);                                              // 24 P^2
        f29_t qv = F::mul(a.x, pp);                                             // 18 P^2
        xyzz29_t o;
        f29_t sub = F::add_lazy(ppp, F::add_lazy(qv, qv));                      // ppp + 2 qv < 6P, limbs < 3 * 2^29
        o.x = F::normalize(F::template sub_lazy<7, 2>(rr, sub));                // (P, 9P) norm
        f29_t t = F::normalize(F::template sub_lazy<10, 0>(qv, o.x));           // qv - X3 + 10P in (P, 12P) norm
        f29_t yn = F::template neg_lazy<6, 0>(a.y);                             // 6P - Y1 in (P, 6P], limbs < 2^30
        o.y = F::mul2(t, r, yn, ppp);                                           // 96 P^2 + 12 P^2: < 2P norm
        o.zz = F::mul(a.zz, pp);
        o.zzz = F::mul(a.zzz, ppp);
        return o;
    }
    // r06: the same addition WITHOUT its exceptional cases, for the inner loop of the bucket accumulation.  Identity operands and equal
    // x coordinates (Q = +-acc) are not handled but DETECTED: `exc` is set and the caller recomputes its whole chain with madd_signed
    // (one sticky flag per chain instead of two operand tests, a branch and the doubling path inside every addition: VERDICT r05 item 2).
    // The value returned in an exceptional case is garbage within the limb bounds (every intermediate keeps its stated bound whatever
    // the operands hold, so nothing overflows before the chain is redone).
    //   Q = O   <=>  y == 0 (no point of a curve of odd order has y = 0; the table's identity is (0, 0))
    //   acc = O <=>  zz == 0 exactly; it makes u2 = 0, p = -X1 ...: not detected by pp -- tested on its own
    SRS_HD static xyzz29_t madd_signed_fast(const xyzz29_t &a, const aff29_t &q, bool neg, bool &exc) {
        f29_t u2 = F::mul(a.zz, q.x);                                           // < 2P norm
        f29_t s2 = F::mul(a.zzz, q.y);                                          // < 2P norm
        f29_t p = F::normalize(F::template sub_lazy<10, 0>(u2, a.x));           // u2 - X1 + 10P in (P, 12P) norm
        f29_t rpos = F::template sub_lazy<6, 0>(s2, a.y);                       //  s2 - Y1 + 6P in (P, 8P), limbs < 3 * 2^29
        f29_t rneg = F::template neg_lazy<8, 1>(F::add_lazy(s2, a.y));          // -s2 - Y1 + 8P in (P, 8P], limbs < 3 * 2^29
        f29_t r = F::normalize(F::select(neg, rneg, rpos));
        f29_t pp = F::sqr(p);                                                   // 144 P^2: < 2P norm
        f29_t rr = F::sqr(r);                                                   // 64 P^2
        exc = exc || F::is_zero_mod(pp) || F::is_zero_exact(q.y) || F::is_zero_exact(a.zz);
        f29_t ppp = F::mul(p, pp);                                              // 24 P^2
        f29_t qv = F::mul(a.x, pp);                                             // 18 P^2
        xyzz29_t o;
        f29_t sub = F::add_lazy(ppp, F::add_lazy(qv, qv));                      // ppp + 2 qv < 6P, limbs < 3 * 2^29
        o.x = F::normalize(F::template sub_lazy<7, 2>(rr, sub));                // (P, 9P) norm
        f29_t t = F::normalize(F::template sub_lazy<10, 0>(qv, o.x));           // qv - X3 + 10P in (P, 12P) norm
        f29_t yn = F::template neg_lazy<6, 0>(a.y);                             // 6P - Y1 in (P, 6P], limbs < 2^30
        o.y = F::mul2(t, r, yn, ppp);                                           // 96 P^2 + 12 P^2: < 2P norm
        o.zz = F::mul(a.zz, pp);
        o.zzz = F::mul(a.zzz, ppp);
        return o;
    }
    // RAW table entry (8 x u32, R'-form, canonical) -> registers, no sign handling (madd_signed)
    SRS_HD static aff29_t load_raw(const affine_t &q) {
        aff29_t o;
        o.x = F::unpack(q.x);
        o.y = F::unpack(q.y);
        return o;
    }

    // ---- the partial sums after level 0 stay in R'-form end to end (msm.hip): in memory a point is the usual 4 x 8 x u32 record
    // holding CANONICAL R'-form coordinates; in registers it is lazy as above
    SRS_HD static xyzz29_t unpack(const xyzz_t &p) {
        xyzz29_t o;
        o.x = F::unpack(p.x);
        o.y = F::unpack(p.y);
        o.zz = F::unpack(p.zz);
        o.zzz = F::unpack(p.zzz);
        return o;
    }
    SRS_HD static xyzz_t pack(const xyzz29_t &a) {      // x < 9P and y < 5P (limbs < 2^31) are folded below 2P by F::reduce_lazy (r05; until then by a
        xyzz_t o;                                        // product with the radix' one: ~230 instructions each against ~75)
        o.x = F::to_canonical_fe(F::reduce_lazy(F::normalize(a.x)));
        o.y = F::to_canonical_fe(F::reduce_lazy(F::normalize(a.y)));
        o.zz = F::to_canonical_fe(a.zz);
        o.zzz = F::to_canonical_fe(a.zzz);
        return o;
    }
    // 2 A for a lazy accumulator (dbl-2008-s-1)
    SRS_HD static xyzz29_t dbl(const xyzz29_t &a) {
        if (is_identity(a)) return a;
        xyzz29_t o;
        const f29_t yn = F::normalize(a.y);                                      // < 5P norm
        const f29_t u = F::add_lazy(yn, yn);                                     // < 10P, limbs < 2^30
        const f29_t v = F::sqr(u);                                               // 100 P^2
        const f29_t w = F::mul(u, v);
        const f29_t s = F::mul(a.x, v);                                          // 18 P^2
        const f29_t xx = F::sqr(a.x);                                            // 81 P^2
        const f29_t m = F::normalize(F::add_lazy(F::add_lazy(xx, xx), xx));      // < 6P norm
        const f29_t mm = F::sqr(m);
        o.x = F::normalize(F::template sub_lazy<5, 1>(mm, F::add_lazy(s, s)));   // (P, 7P) norm
        const f29_t t = F::template sub_lazy<8, 0>(s, o.x);                      // (P, 10P), limbs < 2^31
        const f29_t m1 = F::mul(t, m);
        const f29_t m2 = F::mul(yn, w);
        o.y = F::template sub_lazy<3, 0>(m1, m2);                                // (P, 5P), limbs < 2^31
        o.zz = F::mul(v, a.zz);
        o.zzz = F::mul(w, a.zzz);
        return o;
    }
    // a + b, both lazy accumulators (add-2008-s), complete.  Products put the operand with wide limbs (a y) first.
    SRS_HD static xyzz29_t add(const xyzz29_t &a, const xyzz29_t &b) {
        if (is_identity(b)) return a;
        if (is_identity(a)) return b;
        const f29_t u1 = F::mul(a.x, b.zz);                                      // 18 P^2
        const f29_t u2 = F::mul(b.x, a.zz);
        const f29_t s1 = F::mul(a.y, b.zzz);                                     // 10 P^2
        const f29_t s2 = F::mul(b.y, a.zzz);
        const f29_t p = F::normalize(F::template sub_lazy<3, 0>(u2, u1));        // (P, 5P) norm
        const f29_t r = F::normalize(F::template sub_lazy<3, 0>(s2, s1));
        const f29_t pp = F::sqr(p);
        const f29_t rr = F::sqr(r);
        if (F::is_zero_mod(pp)) {
            if (F::is_zero_mod(rr)) return dbl(a);
            return identity();
        }
        const f29_t ppp = F::mul(p, pp);
        const f29_t qv = F::mul(u1, pp);
        xyzz29_t o;
        o.x = F::normalize(F::template sub_lazy<7, 2>(rr, F::add_lazy(ppp, F::add_lazy(qv, qv))));   // (P, 9P) norm
        const f29_t t = F::template sub_lazy<10, 0>(qv, o.x);                    // (P, 12P), limbs < 2^31
        const f29_t m1 = F::mul(t, r);                                           // 60 P^2
        const f29_t m2 = F::mul(s1, ppp);
        o.y = F::template sub_lazy<3, 0>(m1, m2);                                // (P, 5P), limbs < 2^31
        o.zz = F::mul(F::mul(a.zz, b.zz), pp);
        o.zzz = F::mul(F::mul(a.zzz, b.zzz), ppp);
        return o;
    }

    // a + b by the 4 lanes of a DPP quad (see Ec::add_quad, curve.cuh): the 14 products in 4 levels of <= 4 independent ones.
    // Preconditions: the 4 lanes hold the same a and b, q = lane index inside the quad, the quad is convergent.
    SRS_D static xyzz29_t add_quad(const xyzz29_t &a, const xyzz29_t &b, uint32_t q) {
        const bool ia = is_identity(a), ib = is_identity(b);
        xyzz29_t o;
        if (ia || ib) {                            // a + O = a ; O + b = b   (quad-uniform)
            o.x = F::select(ib, a.x, b.x);
            o.y = F::select(ib, a.y, b.y);
            o.zz = F::select(ib, a.zz, b.zz);
            o.zzz = F::select(ib, a.zzz, b.zzz);
            return o;
        }
        // level 1: u1 = X1 ZZ2 | u2 = X2 ZZ1 | s1 = Y1 ZZZ2 | s2 = Y2 ZZZ1      (first operand: the one that may have wide limbs)
        const f29_t m1 = F::mul(quad_select29(q, a.x, b.x, a.y, b.y), quad_select29(q, b.zz, a.zz, b.zzz, a.zzz));
        const f29_t u1 = quad_bcast29<0>(m1), u2 = quad_bcast29<1>(m1), s1 = quad_bcast29<2>(m1), s2 = quad_bcast29<3>(m1);
        const f29_t p = F::normalize(F::template sub_lazy<3, 0>(u2, u1)), r = F::normalize(F::template sub_lazy<3, 0>(s2, s1));
        // level 2: PP = P^2 | RR = R^2 | ZZ1 ZZ2 | ZZZ1 ZZZ2
        const f29_t m2 = F::mul(quad_select29(q, p, r, a.zz, a.zzz), quad_select29(q, p, r, b.zz, b.zzz));
        const f29_t pp = quad_bcast29<0>(m2), rr = quad_bcast29<1>(m2), zzz12 = quad_bcast29<3>(m2);
        // level 3: PPP = P PP | Q = U1 PP | ZZ3 = ZZ1 ZZ2 PP | (lane 3 repeats lane 0)
        const f29_t m3 = F::mul(quad_select29(q, p, u1, m2, p), pp);
        const f29_t ppp = quad_bcast29<0>(m3), qv = quad_bcast29<1>(m3), zz3 = quad_bcast29<2>(m3);
        const f29_t x3 = F::normalize(F::template sub_lazy<7, 2>(rr, F::add_lazy(ppp, F::add_lazy(qv, qv))));
        // level 4: (Q - X3) R | S1 PPP | ZZZ3 = ZZZ1 ZZZ2 PPP | (lane 3 repeats lane 1)
        const f29_t m4 = F::mul(quad_select29(q, F::template sub_lazy<10, 0>(qv, x3), s1, zzz12, s1), quad_select29(q, r, ppp, ppp, ppp));
        const f29_t t2 = quad_bcast29<0>(m4), t1 = quad_bcast29<1>(m4), zzz3 = quad_bcast29<2>(m4);
        o.x = x3;
        o.y = F::template sub_lazy<3, 0>(t2, t1);
        o.zz = zz3;
        o.zzz = zzz3;
        if (F::is_zero_mod(pp)) o = F::is_zero_mod(rr) ? dbl(a) : identity();   // P2 = +-P1 (quad-uniform as well)
        return o;
    }

    // R'-form lazy coordinates -> the ABI's canonical 2^256-form point (one product per coordinate)
    SRS_HD static xyzz_t to_xyzz(const xyzz29_t &a) {
        f29_t c;
#pragma unroll
        for (int i = 0; i < 9; ++i) c.v[i] = r256_limb(i);
        xyzz_t o;
        o.x = F::to_canonical_fe(F::mul(a.x, c));
        o.y = F::to_canonical_fe(F::mul(a.y, c));
        o.zz = F::to_canonical_fe(F::mul(a.zz, c));
        o.zzz = F::to_canonical_fe(F::mul(a.zzz, c));
        return o;
    }
    // canonical 2^256-form affine point -> canonical R'-form (table build): x * 2^261 = Fp::mul(x~, 2^261 mod p)
    SRS_HD static affine_t table_form(const affine_t &q) {
        using G = typename C::F;
        fe_t c = G::one();
        for (int d = 0; d < 5; ++d) c = G::dbl(c);            // (2^256 mod p) * 32 mod p = 2^261 mod p, as an integer
        affine_t o;
        o.x = G::mul(q.x, c);
        o.y = G::mul(q.y, c);
        return o;
    }
};

}  // namespace srs
