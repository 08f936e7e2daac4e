// poseidon.hip -- the reference's off-circuit random oracle (src/poseidon/poseidon_hash.rs:16-237), host code.
//
// Why it is here although it is not data-parallel: between `commit_cross_terms` and the folds the
// reference squeezes the challenge r out of this sponge; with the oracle inside the library a shim (or bench.py) derives r
// without going back to its own field arithmetic, and the challenge in the benchmark is a real function of the commitments.
// One permutation is ~60 field multiplications: the CPU does it in microseconds, a GPU launch would only add latency.
//
// Parameters: `Spec::new(r_f, r_p)` comes from the third-party crate privacy-scaling-explorations/poseidon @ 807f8f55
// (Cargo.toml:40-42), which is not part of the reference tree; its published construction is restated: Grain LFSR of the
// Poseidon paper -> round constants (rejection sampling, MSB first) -> Cauchy MDS 1 / (x_i + y_j) (x, y without rejection);
// initial state (2^64, 0, ...).  oracle/poseidon.py holds the same restatement in Python and is pinned by the reference's
// known answer (poseidon_hash.rs:248-266); tests/test_poseidon.py compares the two on both fields.
#include "poseidon.h"

#include <cstring>

namespace srs {
namespace poseidon {
namespace {

struct Grain {
    bool s[80];
    int head = 0;                        // ring buffer: s[(head + i) % 80] is bit i
    bool at(int i) const { return s[(head + i) % 80]; }
    bool new_bit() {
        bool b = at(62) ^ at(51) ^ at(38) ^ at(23) ^ at(13) ^ at(0);
        s[head] = b;                     // drop bit 0, append b
        head = (head + 1) % 80;
        return b;
    }
    bool next() {                        // self-shrinking output
        bool b = new_bit();
        while (!b) {
            new_bit();
            b = new_bit();
        }
        return new_bit();
    }
    Grain(uint32_t num_bits, uint32_t t, uint32_t r_f, uint32_t r_p) {
        int n = 0;
        auto app = [&](int bits, uint64_t v) { for (int i = bits - 1; i >= 0; --i) s[n++] = (v >> i) & 1u; };
        app(2, 1); app(4, 0); app(12, num_bits); app(12, t); app(10, r_f); app(10, r_p); app(30, (1u << 30) - 1);
        for (int i = 0; i < 160; ++i) new_bit();
    }
    // `num_bits` bits, first one most significant, as 8 little-endian u32 limbs (canonical integer, NOT Montgomery)
    fe_t bits(uint32_t num_bits) {
        fe_t v;
        std::memset(&v, 0, sizeof v);
        for (uint32_t i = 0; i < num_bits; ++i) {
            uint32_t pos = num_bits - 1 - i;
            if (next()) v.v[pos >> 5] |= 1u << (pos & 31);
        }
        return v;
    }
};

template <class F, class PP>
bool less_than_p(const fe_t &a) {
    for (int i = 7; i >= 0; --i) {
        if (a.v[i] < PP::p(i)) return true;
        if (a.v[i] > PP::p(i)) return false;
    }
    return false;
}

template <class F, class PP>
void build(Hash &h) {
    const uint32_t nbits = 254;          // F::NUM_BITS of both bn256 fields
    Grain g(nbits, (uint32_t)h.t, (uint32_t)h.r_f, (uint32_t)h.r_p);
    h.rc.resize((h.r_f + h.r_p) * h.t);
    for (auto &c : h.rc) {
        fe_t v;
        do { v = g.bits(nbits); } while (!less_than_p<F, PP>(v));
        c = F::to_mont(v);
    }
    std::vector<fe_t> xs(h.t), ys(h.t);
    for (auto &x : xs) x = F::to_mont(F::reduce_once(g.bits(nbits)));        // < 2^254 < 2p: one conditional subtraction
    for (auto &y : ys) y = F::to_mont(F::reduce_once(g.bits(nbits)));
    h.mds.resize(h.t * h.t);
    for (size_t i = 0; i < h.t; ++i)
        for (size_t j = 0; j < h.t; ++j) h.mds[i * h.t + j] = F::inv(F::add(xs[i], ys[j]));
}

// One row of the MDS product, sum_j a[j] b[j] mod p (t <= 16 terms), with ONE Montgomery reduction: the t double-width products
// are added up as a 576-bit integer first (4 x 64-bit limbs, 128-bit partial products), then reduced like a single product.  The
// value is the same field element as the chain of F::add(F::mul(..)) it replaces; 25 of the 40 products of a full round (25 of 28
// of a partial one) are row products, so a permutation costs ~30 % less host time.
template <class PP>
fe_t dot_row(const fe_t *a, const fe_t *b, size_t t) {
    typedef unsigned __int128 u128;
    uint64_t P[4];
    for (int i = 0; i < 4; ++i) P[i] = (uint64_t)PP::p(2 * i) | ((uint64_t)PP::p(2 * i + 1) << 32);
    uint64_t inv = (uint64_t)PP::INV;                    // -p^-1 mod 2^32 -> mod 2^64 by one Newton step (as Fp::mul does)
    inv = inv * (2 + P[0] * inv);
    uint64_t T[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t k = 0; k < t; ++k) {
        uint64_t A[4], B[4];
        for (int i = 0; i < 4; ++i) {
            A[i] = (uint64_t)a[k].v[2 * i] | ((uint64_t)a[k].v[2 * i + 1] << 32);
            B[i] = (uint64_t)b[k].v[2 * i] | ((uint64_t)b[k].v[2 * i + 1] << 32);
        }
        for (int i = 0; i < 4; ++i) {
            u128 c = 0;
            for (int j = 0; j < 4; ++j) {
                c += (u128)A[j] * B[i] + T[i + j];
                T[i + j] = (uint64_t)c;
                c >>= 64;
            }
            for (int j = i + 4; c && j < 10; ++j) {         // carry into the upper limbs
                c += T[j];
                T[j] = (uint64_t)c;
                c >>= 64;
            }
        }
    }
    for (int i = 0; i < 4; ++i) {                          // REDC: 4 steps, each clears the lowest live limb
        const uint64_t m = T[i] * inv;
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)m * P[j] + T[i + j];
            T[i + j] = (uint64_t)c;
            c >>= 64;
        }
        for (int j = i + 4; c && j < 10; ++j) {
            c += T[j];
            T[j] = (uint64_t)c;
            c >>= 64;
        }
    }
    // T[4..9) = (sum + m p) / 2^256 < (16 p^2 + p 2^256) / 2^256 < 5 p: a few conditional subtractions
    uint64_t R[5] = {T[4], T[5], T[6], T[7], T[8]};
    for (;;) {
        uint64_t D[5];
        u128 bw = 0;
        for (int i = 0; i < 5; ++i) {
            const u128 d = (u128)R[i] - (i < 4 ? P[i] : 0) - (uint64_t)bw;
            D[i] = (uint64_t)d;
            bw = (d >> 64) ? 1 : 0;
        }
        if (bw) break;                                     // R < p
        for (int i = 0; i < 5; ++i) R[i] = D[i];
    }
    fe_t o;
    for (int i = 0; i < 4; ++i) {
        o.v[2 * i] = (uint32_t)R[i];
        o.v[2 * i + 1] = (uint32_t)(R[i] >> 32);
    }
    return o;
}

template <class F>
void permute(const Hash &h, std::vector<fe_t> &st) {
    const size_t t = h.t, half = h.r_f / 2;
    std::vector<fe_t> nx(t);
    auto pow5 = [](const fe_t &x) { fe_t x2 = F::sqr(x); return F::mul(F::sqr(x2), x); };
    for (size_t r = 0; r < h.r_f + h.r_p; ++r) {
        for (size_t i = 0; i < t; ++i) st[i] = F::add(st[i], h.rc[r * t + i]);
        if (r < half || r >= half + h.r_p) {
            for (size_t i = 0; i < t; ++i) st[i] = pow5(st[i]);
        } else {
            st[0] = pow5(st[0]);
        }
        for (size_t i = 0; i < t; ++i) nx[i] = dot_row<typename F::Params>(&h.mds[i * t], st.data(), t);
        st.swap(nx);
    }
}

template <class F>
fe_t run(Hash &h) {
    // The sponge state after the full RATE-sized chunks absorbed so far depends on those chunks only (padding touches the
    // last, partial chunk): it is cached across squeezes, so a transcript that squeezes after every absorb (delta, alpha,
    // gamma of ProtoGalaxy::prove: 2 + 32 + 256 elements) permutes every chunk once, not once per squeeze.  Same output
    // as re-running the sponge over the whole buffer, which is what the reference does (poseidon_hash.rs:190-212).
    if (h.state.empty()) {
        h.state.assign(h.t, F::zero());
        fe_t cap;
        std::memset(&cap, 0, sizeof cap);
        cap.v[2] = 1;                                        // 2^64  (poseidon::State::default())
        h.state[0] = F::to_mont(cap);
        h.done = 0;
    }
    const size_t n = h.buf.size();
    while (n - h.done > h.rate) {                            // chunks that are certainly not the last one
        for (size_t i = 0; i < h.rate; ++i) h.state[1 + i] = F::add(h.state[1 + i], h.buf[h.done + i]);
        permute<F>(h, h.state);
        h.done += h.rate;
    }
    // the tail: at most one full chunk (then followed by the empty, padded chunk) or one partial chunk, on a copy of the state
    std::vector<fe_t> st(h.state);
    for (size_t at = h.done; at <= n; at += h.rate) {
        const size_t len = n - at < h.rate ? n - at : h.rate;
        if (at == n && n % h.rate != 0) break;               // the empty chunk exists only when the buffer is exact
        for (size_t i = 0; i < len; ++i) st[1 + i] = F::add(st[1 + i], h.buf[at + i]);
        if (len < h.rate) st[1 + len] = F::add(st[1 + len], F::one());      // pre_round padding, :52-64
        permute<F>(h, st);
        if (len < h.rate) break;
    }
    return F::from_mont(st[1]);
}

}  // namespace

Hash *create(int field, size_t t, size_t rate, size_t r_f, size_t r_p, std::string &err) {
    if (t < 2 || t > 16 || rate != t - 1) { err = "RATE must be T - 1 (poseidon_hash.rs:41), 2 <= T <= 16"; return nullptr; }
    if (r_f == 0 || (r_f & 1) || r_f >= 1024 || r_p >= 1024) { err = "R_F must be even and positive; R_F, R_P < 1024"; return nullptr; }
    Hash *h = new Hash();
    h->field = field;
    h->t = t;
    h->rate = rate;
    h->r_f = r_f;
    h->r_p = r_p;
    if (field == 0) build<Fr, FrP>(*h); else build<Fq, FqP>(*h);
    return h;
}

void absorb(Hash &h, const fe_t *v, size_t n) { h.buf.insert(h.buf.end(), v, v + n); }

bool squeeze(Hash &h, size_t num_bits, int out_field, fe_t &out, std::string &err) {
    if (num_bits == 0 || num_bits > 256) { err = "num_bits must be in 1..256"; return false; }
    fe_t canon = h.field == 0 ? run<Fr>(h) : run<Fq>(h);
    for (size_t b = num_bits; b < 256; ++b) canon.v[b >> 5] &= ~(1u << (b & 31));     // bits[..num_bits] (little endian)
    // bits_to_fe_le -> F1::from_repr(..).unwrap() (src/util/mod.rs:58-64): up to 253 bits fit both moduli; the reference's own
    // MAX_BITS = 255 (src/constants.rs:4) keeps the whole element, which must then be a residue of the output field
    if (num_bits > 253 && !(out_field == 0 ? less_than_p<Fr, FrP>(canon) : less_than_p<Fq, FqP>(canon))) {
        err = "squeezed value is not a canonical element of the output field (the reference's from_repr(..).unwrap() panics here)";
        return false;
    }
    out = out_field == 0 ? Fr::to_mont(canon) : Fq::to_mont(canon);
    return true;
}

}  // namespace poseidon
}  // namespace srs
