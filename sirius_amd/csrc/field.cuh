// field.cuh -- 256-bit Montgomery prime-field arithmetic for gfx950 (and the host, for tables).
//
// Two moduli, the bn256 / grumpkin cycle used everywhere in the reference (src/lib.rs:29-48):
//   Fr = bn256 scalar field = grumpkin base field   (2-adicity 28 -> the NTT field, src/fft.rs:13)
//   Fq = bn256 base field   = grumpkin scalar field
// In-memory form is the one halo2curves keeps [3P]: 4 x u64 little-endian limbs, Montgomery
// R = 2^256.  On the device the same 32 bytes are viewed as 8 x u32 limbs: CDNA4's integer
// multiplier is v_mad_u64_u32 (32x32+64 -> 64), so 32-bit limbs are the native width.
//
// Both moduli are < 2^254, so the CIOS accumulator never exceeds 2p < 2^255: the row carry fits
// one limb and the classic extra accumulator words vanish ("no-carry" CIOS).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SRS_HD __host__ __device__ __forceinline__
#define SRS_D __device__ __forceinline__
#else
#define SRS_HD inline __attribute__((always_inline))
#define SRS_D inline __attribute__((always_inline))
#endif

namespace srs {

struct alignas(16) fe_t {
    uint32_t v[8];
};

// ---- modulus parameter packs (constexpr accessors fold to literals after unrolling) ----
struct FrP {
    static constexpr int ID = 0;
    SRS_HD static constexpr uint32_t p(int i) {
        constexpr uint32_t m[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                   0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return m[i];
    }
    SRS_HD static constexpr uint32_t r(int i) {   // R mod p  (Montgomery one)
        constexpr uint32_t m[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                   0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return m[i];
    }
    SRS_HD static constexpr uint32_t r2(int i) {  // R^2 mod p
        constexpr uint32_t m[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                   0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return m[i];
    }
    static constexpr uint32_t INV = 0xefffffffu;  // -p^{-1} mod 2^32
};
struct FqP {
    static constexpr int ID = 1;
    SRS_HD static constexpr uint32_t p(int i) {
        constexpr uint32_t m[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                   0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return m[i];
    }
    SRS_HD static constexpr uint32_t r(int i) {
        constexpr uint32_t m[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                   0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return m[i];
    }
    SRS_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t m[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                   0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return m[i];
    }
    static constexpr uint32_t INV = 0xe4866389u;
};

template <class P>
struct Fp {
    using Params = P;
    SRS_HD static fe_t zero() {
        fe_t o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[i] = 0;
        return o;
    }
    SRS_HD static fe_t one() {
        fe_t o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[i] = P::r(i);
        return o;
    }
    SRS_HD static bool is_zero(const fe_t &a) {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t |= a.v[i];
        return t == 0;
    }
    SRS_HD static bool eq(const fe_t &a, const fe_t &b) {
        uint32_t t = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t |= a.v[i] ^ b.v[i];
        return t == 0;
    }
    // o = a - p if a >= p else a   (a < 2p)
    SRS_HD static fe_t reduce_once(const fe_t &a) {
        fe_t d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)a.v[i] - P::p(i) - borrow;
            d.v[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 32) & 1u;
        }
        fe_t o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[i] = borrow ? a.v[i] : d.v[i];
        return o;
    }
    // a / 2 mod p: (a + (a odd ? p : 0)) >> 1.  Same on Montgomery representatives ((aR)/2 = (a/2)R).
    SRS_HD static fe_t halve(const fe_t &a) {
        const uint32_t mask = 0u - (a.v[0] & 1u);
        fe_t s;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)a.v[i] + (P::p(i) & mask) + c;
            s.v[i] = (uint32_t)t;
            c = (uint32_t)(t >> 32);
        }
        fe_t o;                                   // a + p < 2p < 2^255: no carry out of limb 7
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[i] = (s.v[i] >> 1) | (i < 7 ? (s.v[i + 1] << 31) : 0u);
        return o;
    }
    SRS_HD static fe_t add(const fe_t &a, const fe_t &b) {
        fe_t s;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)a.v[i] + b.v[i] + c;
            s.v[i] = (uint32_t)t;
            c = (uint32_t)(t >> 32);
        }
        return reduce_once(s);   // a+b < 2p < 2^255: no carry out of limb 7
    }
    SRS_HD static fe_t sub(const fe_t &a, const fe_t &b) {
        fe_t d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)a.v[i] - b.v[i] - borrow;
            d.v[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 32) & 1u;
        }
        uint32_t mask = 0u - borrow, c = 0;
        fe_t o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)d.v[i] + (P::p(i) & mask) + c;
            o.v[i] = (uint32_t)t;
            c = (uint32_t)(t >> 32);
        }
        return o;
    }
    SRS_HD static fe_t neg(const fe_t &a) { return sub(zero(), a); }
    SRS_HD static fe_t dbl(const fe_t &a) { return add(a, a); }

    // Montgomery product a*b/R mod p.  (device body generated by tools/gen_field_fips.py: one asm
    // statement per column, because hipcc pads every asm statement with an s_nop)
    //
    // Device: finely-integrated product scanning (FIPS).  Column k accumulates sum a_i*b_(k-i) and
    // sum m_i*p_(k-i) in a 96-bit accumulator; every term is ONE v_mad_u64_u32 whose carry-out
    // (vcc) is folded into the third limb by ONE v_addc_co_u32 -- 2 half-rate VALU ops per 32x32
    // product, no zero-extension moves (a C CIOS form compiles to 575 instructions, 250 of them
    // v_mov; this form to 405).  Measured rates: profiles/r01_ubench_gfx950*.txt.
    // Host (and the CPU logic emulator): portable 4x64 CIOS, identical results.
#if defined(__HIP_DEVICE_COMPILE__)
    static __device__ __forceinline__ fe_t mul(const fe_t &a, const fe_t &b) {
#include "field_fips.inc"
    }
#else
    // host: 4 x 64-bit limbs with 128-bit products (same CIOS, 4x fewer steps than the 32-bit form)
    static inline fe_t mul(const fe_t &a, const fe_t &b) {
        typedef unsigned __int128 u128;
        uint64_t A[4], B[4], Pm[4], t[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            A[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
            B[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
            Pm[i] = (uint64_t)P::p(2 * i) | ((uint64_t)P::p(2 * i + 1) << 32);
        }
        // -p^-1 mod 2^64 from the 32-bit constant by one Newton step
        uint64_t inv = (uint64_t)P::INV;                 // -p^-1 mod 2^32
        inv = inv * (2 + Pm[0] * inv);                   // x' = x (2 + p x) for x = -p^-1
        for (int i = 0; i < 4; ++i) {
            u128 c = 0;
            for (int j = 0; j < 4; ++j) {
                c += (u128)A[j] * B[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            uint64_t hi = (uint64_t)c + t[4];
            uint64_t m = t[0] * inv;
            c = (u128)m * Pm[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; ++j) {
                c += (u128)m * Pm[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += hi;
            t[3] = (uint64_t)c;
            t[4] = (uint64_t)(c >> 64);
        }
        fe_t o;
        for (int i = 0; i < 4; ++i) {
            o.v[2 * i] = (uint32_t)t[i];
            o.v[2 * i + 1] = (uint32_t)(t[i] >> 32);
        }
        return reduce_once(o);                           // t < 2p < 2^255, t[4] == 0
    }
#endif
    SRS_HD static fe_t sqr(const fe_t &a) { return mul(a, a); }

    SRS_HD static fe_t from_mont(const fe_t &a) {
        fe_t one_raw = zero();
        one_raw.v[0] = 1;
        return mul(a, one_raw);
    }
    SRS_HD static fe_t to_mont(const fe_t &a) {
        fe_t r2;
#pragma unroll
        for (int i = 0; i < 8; ++i) r2.v[i] = P::r2(i);
        return mul(a, r2);
    }
    // a^e for a 256-bit exponent given as 8 u32 limbs (square-and-multiply, MSB first)
    SRS_HD static fe_t pow(const fe_t &a, const uint32_t e[8]) {
        fe_t acc = one();
        for (int i = 255; i >= 0; --i) {
            acc = sqr(acc);
            if ((e[i >> 5] >> (i & 31)) & 1u) acc = mul(acc, a);
        }
        return acc;
    }
    SRS_HD static fe_t pow_u64(const fe_t &a, uint64_t e) {
        fe_t acc = one();
        for (int i = 63; i >= 0; --i) {
            acc = sqr(acc);
            if ((e >> i) & 1u) acc = mul(acc, a);
        }
        return acc;
    }
    SRS_HD static fe_t inv(const fe_t &a) {   // a^(p-2); inv(0) = 0
        uint32_t e[8];
        uint32_t borrow = 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)P::p(i) - borrow;
            e[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 32) & 1u;
        }
        return pow(a, e);
    }
    SRS_HD static fe_t from_u64(uint64_t x) {
        fe_t t = zero();
        t.v[0] = (uint32_t)x;
        t.v[1] = (uint32_t)(x >> 32);
        return to_mont(t);
    }
};

using Fr = Fp<FrP>;
using Fq = Fp<FqP>;

}  // namespace srs
