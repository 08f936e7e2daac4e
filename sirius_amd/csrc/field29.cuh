// field29.cuh -- carry-free 9 x 29-bit limb form of the two 254-bit fields, for the throughput-bound kernels.
//
// Why a second limb form.  gfx950's integer multiplier is v_mad_u64_u32 (32 x 32 + 64 -> 64, roughly half rate).  With
// 8 x 32-bit limbs (field.cuh) a column of the Montgomery product overflows 64 bits, so every multiply-accumulate needs a
// second, equally slow carry instruction (v_addc_co_u32): 128 + 128 issue slots per product, 405 instructions in all.
// With 29-bit limbs a column holds 9 + 9 products of < 2^60 / 2^58 and never leaves 64 bits: one v_mad_u64_u32 per
// term and nothing else -- 162 multiply-accumulates, 9 v_mul_lo for the Montgomery quotients, two shift/mask
// operations per column, no final conditional subtraction (9 x 29 = 261 bits leave 7 bits of headroom above p, so
// sums and differences are kept lazily as small multiples of p).  tools/ubench29.hip measures both forms.
//
// The price: the Montgomery radix of this form is R' = 2^261, not the 2^256 of the in-memory ABI (halo2curves).  It is
// therefore used only where every operand is library-internal or can be brought into R'-form for free:
//   * MSM bucket accumulation (msm.hip): the window table is built by the library and stored in R'-form; the partial
//     sums are converted back to the 2^256 form by one extra product per coordinate when a thread stores its result.
// Values in memory stay 8 x u32 (fe_t); f29_t exists in registers only.
//
// Bounds (P = modulus < 2^254; 2^261 / P > 165):
//   mul(a, b)  needs  A * B < 2^261 * P  (e.g. A, B <= 12 P)  and limbs  a_i < 2^31, b_j < 2^29  or both < 2^30;
//              returns a value < 2P with limbs < 2^29 ("normalised").
//   The lazy add / sub helpers state their own bounds.
#pragma once
#include "field.cuh"

namespace srs {

struct f29_t {
    uint32_t v[9];
};

template <class P>
struct Fp29 {
    static constexpr uint32_t B = 29;
    static constexpr uint32_t MASK = (1u << B) - 1u;
    static constexpr uint32_t INV = P::INV & MASK;   // -p^-1 mod 2^29

    // limb i of c * p in radix 2^29 (c <= 64), plain positional digits; limb 8 takes everything above bit 232
    SRS_HD static constexpr uint32_t kp(uint32_t c, int i) {
        // c * p as 9 x 32-bit words, then bits [29 i, 29 i + 29) (or [232, ...) for i = 8)
        uint32_t w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t carry = 0;
        for (int j = 0; j < 8; ++j) {
            uint64_t t = (uint64_t)P::p(j) * c + carry;
            w[j] = (uint32_t)t;
            carry = t >> 32;
        }
        w[8] = (uint32_t)carry;
        const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t two = (uint64_t)w[wi] | (wi + 1 < 9 ? (uint64_t)w[wi + 1] << 32 : 0);
        uint32_t limb = (uint32_t)(two >> sh);
        return i < 8 ? (limb & MASK) : limb;
    }
    SRS_HD static constexpr uint32_t p(int i) { return kp(1, i); }
    // c * p with every limb below the top one raised by 2^(29+e) and the limb above lowered by 2^e: same value, but
    // limb-wise  a_i + K_i - b_i  cannot go negative for b_i < 2^(29+e)   (e = 0: b normalised; e = 2: b_i < 2^31)
    SRS_HD static constexpr uint32_t kpb(uint32_t c, uint32_t e, int i) {
        uint32_t k = kp(c, i);
        if (i < 8) k += 1u << (B + e);
        if (i > 0) k -= 1u << e;
        return k;
    }

    SRS_HD static f29_t zero() {
        f29_t o;
#pragma unroll
        for (int i = 0; i < 9; ++i) o.v[i] = 0;
        return o;
    }
    SRS_HD static bool is_zero_exact(const f29_t &a) {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) t |= a.v[i];
        return t == 0;
    }
    // a == 0 (mod p) for a NORMALISED value < 2P: the only representatives are 0 and P
    SRS_HD static bool is_zero_mod(const f29_t &a) {
        uint32_t z = 0, e = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            z |= a.v[i];
            e |= a.v[i] ^ p(i);
        }
        return z == 0 || e == 0;
    }

    // 8 x u32 (value < 2^256) -> 9 x 29
    SRS_HD static f29_t unpack(const fe_t &a) {
        f29_t o;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
            uint32_t lo = a.v[wi] >> sh;
            if (sh > 3 && wi + 1 < 8) lo |= a.v[wi + 1] << (32 - sh);      // the limb straddles two words
            o.v[i] = i < 8 ? (lo & MASK) : lo;
        }
        return o;
    }
    // normalised 9 x 29 with value < 2^256 -> 8 x u32
    SRS_HD static fe_t pack(const f29_t &a) {
        fe_t o;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int bit = 32 * w, li = bit / 29, sh = bit % 29;      // word w starts inside limb li at bit sh
            uint32_t x = a.v[li] >> sh;
            x |= a.v[li + 1] << (29 - sh);
            if (29 - sh + 29 < 32 && li + 2 < 9) x |= a.v[li + 2] << (58 - sh);
            o.v[w] = x;
        }
        return o;
    }

    // carry propagation: limbs < 2^32 in, limbs < 2^29 out (top limb takes the rest); value unchanged
    SRS_HD static f29_t normalize(const f29_t &a) {
        f29_t o;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t t = a.v[i] + c;          // a_i < 2^32 - 8: no wrap (callers keep limbs < 2^31.6)
            o.v[i] = t & MASK;
            c = t >> B;
        }
        o.v[8] = a.v[8] + c;
        return o;
    }

#if defined(SRS_F29_CHAIN) && defined(__HIP_DEVICE_COMPILE__)
    // r04: the same three products with every column's multiply-accumulate chain STARTED FROM the carry of the column before (the addend
    // operand of v_mad_u64_u32) -- hipcc builds the next column's sum on the side and joins the two with a v_lshl_add_u64 per column (16
    // per product, 144 per mixed addition, 4.2 cycles each).  Same values, same bounds.  Generated (tools/gen_field29_chain.py): one asm
    // statement per run of multiply-accumulates, because every statement costs an s_nop.
#include "field29_chain.inc"
#else
    // Montgomery product a * b / 2^261 mod p -- see the bounds in the header comment
    SRS_HD static f29_t mul(const f29_t &a, const f29_t &b) {
        uint64_t acc = 0;
        uint32_t m[9];
        f29_t o;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int i = 0; i <= k; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * p(k - i);
            m[k] = ((uint32_t)acc * INV) & MASK;
            acc += (uint64_t)m[k] * p(0);
            acc >>= B;
        }
#pragma unroll
        for (int k = 9; k < 17; ++k) {
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * p(k - i);
            o.v[k - 9] = (uint32_t)acc & MASK;
            acc >>= B;
        }
        o.v[8] = (uint32_t)acc;
        return o;
    }
    // (a * b + c * d) / 2^261 mod p with ONE Montgomery reduction: the two products are accumulated column by column into the same
    // 64-bit sums (27 terms per column instead of 18), which needs the tighter limb bounds  a_i, c_i < 2^30,  b_j, d_j < 2^29
    // (18 * 2^59 + 9 * 2^58 < 2^64) and  A * B + C * D < 2^261 * P;  returns a value < 2P with limbs < 2^29.
    // Saves the 81 + 9 multiplier instructions of the second reduction (r03: the Y coordinate of a mixed addition is t r - y ppp).
    SRS_HD static f29_t mul2(const f29_t &a, const f29_t &b, const f29_t &c, const f29_t &d) {
        uint64_t acc = 0;
        uint32_t m[9];
        f29_t o;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int i = 0; i <= k; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = 0; i <= k; ++i) acc += (uint64_t)c.v[i] * d.v[k - i];
#pragma unroll
            for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * p(k - i);
            m[k] = ((uint32_t)acc * INV) & MASK;
            acc += (uint64_t)m[k] * p(0);
            acc >>= B;
        }
#pragma unroll
        for (int k = 9; k < 17; ++k) {
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)c.v[i] * d.v[k - i];
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * p(k - i);
            o.v[k - 9] = (uint32_t)acc & MASK;
            acc >>= B;
        }
        o.v[8] = (uint32_t)acc;
        return o;
    }
    SRS_HD static f29_t sqr(const f29_t &a) {
        // the doubled cross products a_i a_j (i < j) are formed once: 45 + 81 multiply-accumulates instead of 81 + 81
        uint64_t acc = 0;
        uint32_t m[9], a2[9];
        f29_t o;
#pragma unroll
        for (int i = 0; i < 9; ++i) a2[i] = a.v[i] << 1;        // limbs < 2^30 in -> < 2^31
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int i = 0; 2 * i < k; ++i) acc += (uint64_t)a2[i] * a.v[k - i];
            if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
            for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * p(k - i);
            m[k] = ((uint32_t)acc * INV) & MASK;
            acc += (uint64_t)m[k] * p(0);
            acc >>= B;
        }
#pragma unroll
        for (int k = 9; k < 17; ++k) {
#pragma unroll
            for (int i = k - 8; 2 * i < k; ++i) acc += (uint64_t)a2[i] * a.v[k - i];
            if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * p(k - i);
            o.v[k - 9] = (uint32_t)acc & MASK;
            acc >>= B;
        }
        o.v[8] = (uint32_t)acc;
        return o;
    }

#endif   // SRS_F29_CHAIN

    // a - b + c p, limb-wise, NOT normalised.  Needs b_i < 2^(29+e) (i < 8), b_8 <= top limb of c p - 2^e,
    // result limbs < a_i + 2^(29+e) + 2^29.
    template <uint32_t CP, uint32_t E>
    SRS_HD static f29_t sub_lazy(const f29_t &a, const f29_t &b) {
        f29_t o;
#pragma unroll
        for (int i = 0; i < 9; ++i) o.v[i] = a.v[i] + kpb(CP, E, i) - b.v[i];
        return o;
    }
    // c p - b  (negation), same conditions
    template <uint32_t CP, uint32_t E>
    SRS_HD static f29_t neg_lazy(const f29_t &b) {
        f29_t o;
#pragma unroll
        for (int i = 0; i < 9; ++i) o.v[i] = kpb(CP, E, i) - b.v[i];
        return o;
    }
    SRS_HD static f29_t add_lazy(const f29_t &a, const f29_t &b) {
        f29_t o;
#pragma unroll
        for (int i = 0; i < 9; ++i) o.v[i] = a.v[i] + b.v[i];
        return o;
    }
    SRS_HD static f29_t select(bool c, const f29_t &x, const f29_t &y) {
        f29_t o;
#pragma unroll
        for (int i = 0; i < 9; ++i) o.v[i] = c ? x.v[i] : y.v[i];
        return o;
    }

    // r05: a normalised LAZY value (limbs < 2^29, top limb < 2^31: any value < 2^263, i.e. several hundred P -- what the NTT's tile holds after its
    // last stage) -> the same residue below 2P, normalised, WITHOUT a Montgomery product: q = floor(top limb * floor(2^282 / P) / 2^50) is
    // floor(a / P) or one less (the top limb is a >> 232; the truncations lose < 2^-19 of a unit), then a - q P limb by limb with a signed
    // running carry: ~50 instructions against the ~230 of the product with the radix' one that did this before.
    SRS_HD static constexpr uint32_t quotient_const() {
        const uint64_t p_top = ((uint64_t)P::p(7) << 32) | P::p(6);              // P >> 192, truncated: the quotient below is an OVER-estimate by < 2^-60 ...
        return (uint32_t)((((unsigned __int128)1) << 90) / p_top) - 1u;         // ... so one less: never above 2^282 / P
    }
    SRS_HD static f29_t reduce_lazy(const f29_t &a) {
        constexpr uint32_t MQ = quotient_const();
        static_assert(P::p(7) >> 28 != 0 && MQ < (1u << 30), "reduce_lazy: P must have 253-254 bits");
        const uint32_t q = (uint32_t)(((uint64_t)a.v[8] * MQ) >> 50);
        f29_t o;
        int64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc += (int64_t)a.v[i] - (int64_t)((uint64_t)q * p(i));
            o.v[i] = (uint32_t)acc & MASK;
            acc >>= B;                          // arithmetic: the borrow travels as a negative carry
        }
        o.v[8] = (uint32_t)(acc + (int64_t)a.v[8] - (int64_t)((uint64_t)q * p(8)));      // a - q P is in [0, 2P): the top limb ends below 2^23
        return o;
    }

    // normalised value < 4P  ->  the canonical representative in [0, P), packed
    SRS_HD static fe_t to_canonical_fe(const f29_t &a) {
        fe_t x = pack(a);                       // < 4P < 2^256
        using F = Fp<P>;
        // two conditional subtractions of 2P / P on the 8 x 32 form (reduce_once expects < 2p after the first)
        fe_t twop;
        {
            uint32_t c = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint64_t t = ((uint64_t)P::p(i) << 1) + c;
                twop.v[i] = (uint32_t)t;
                c = (uint32_t)(t >> 32);
            }
        }
        fe_t d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)x.v[i] - twop.v[i] - borrow;
            d.v[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 32) & 1u;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) x.v[i] = borrow ? x.v[i] : d.v[i];
        return F::reduce_once(x);
    }

    // constants of the two Montgomery forms, as plain integers in limbs (computed on the host side of the call sites
    // through Fp<P>, see msm.hip): nothing here.
};

using Fr29 = Fp29<FrP>;
using Fq29 = Fp29<FqP>;

}  // namespace srs
