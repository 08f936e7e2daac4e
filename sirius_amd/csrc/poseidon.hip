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

#include "devrt.h"
#include "field29.cuh"

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

// ---- the permutation on 4 x 64-bit limbs (the host's native width; field.cuh's 8 x 32-bit code is written for the GPU's
// multiplier and costs ~4x more here).  Same field elements, Montgomery R = 2^256, canonical representatives throughout.
template <class PP>
struct Host64 {
    typedef unsigned __int128 u128;
    struct E { uint64_t l[4]; };
    static uint64_t P(int i) { return (uint64_t)PP::p(2 * i) | ((uint64_t)PP::p(2 * i + 1) << 32); }
    static uint64_t inv64() {
        uint64_t inv = (uint64_t)PP::INV;                // -p^-1 mod 2^32 -> mod 2^64 by one Newton step
        return inv * (2 + P(0) * inv);
    }
    static E load(const fe_t &a) {
        E o;
        for (int i = 0; i < 4; ++i) o.l[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
        return o;
    }
    static fe_t store(const E &a) {
        fe_t o;
        for (int i = 0; i < 4; ++i) { o.v[2 * i] = (uint32_t)a.l[i]; o.v[2 * i + 1] = (uint32_t)(a.l[i] >> 32); }
        return o;
    }
    static E cond_sub(const uint64_t r[4], uint64_t top) {      // r + top * 2^256 < 2p  ->  canonical
        const uint64_t p0 = P(0), p1 = P(1), p2 = P(2), p3 = P(3);
        uint64_t d[4];
        u128 x = (u128)r[0] - p0;
        d[0] = (uint64_t)x;
        x = (u128)r[1] - p1 - (uint64_t)((x >> 64) & 1);
        d[1] = (uint64_t)x;
        x = (u128)r[2] - p2 - (uint64_t)((x >> 64) & 1);
        d[2] = (uint64_t)x;
        x = (u128)r[3] - p3 - (uint64_t)((x >> 64) & 1);
        d[3] = (uint64_t)x;
        const bool borrow = ((x >> 64) & 1) != 0;
        E o;
        const bool take = top || !borrow;
        for (int i = 0; i < 4; ++i) o.l[i] = take ? d[i] : r[i];
        return o;
    }
    static E add(const E &a, const E &b) {
        uint64_t r[4];
        u128 c = 0;
        for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r[i] = (uint64_t)c; c >>= 64; }
        return cond_sub(r, (uint64_t)c);
    }
    static E mul(const E &a, const E &b) {                     // CIOS, both moduli < 2^254: the running value stays < 2p
        const uint64_t inv = inv64();
        const uint64_t p[4] = {P(0), P(1), P(2), P(3)};
        uint64_t t[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            u128 c = 0;
            for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[4];
            t[4] = (uint64_t)c;
            const uint64_t t5 = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * inv;
            c = (u128)m * p[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; ++j) { c += (u128)m * p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = t5 + (uint64_t)(c >> 64);
        }
        return cond_sub(t, t[4]);
    }
    static E pow5(const E &x) { const E x2 = mul(x, x); return mul(mul(x2, x2), x); }
    // sum_j a[j] b[j] with ONE reduction (see dot_row)
    static E dot(const E *a, const E *b, size_t n) {
        const uint64_t inv = inv64();
        const uint64_t p[4] = {P(0), P(1), P(2), P(3)};
        uint64_t T[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t k = 0; k < n; ++k) {
            for (int i = 0; i < 4; ++i) {
                u128 c = 0;
                for (int j = 0; j < 4; ++j) { c += (u128)a[k].l[j] * b[k].l[i] + T[i + j]; T[i + j] = (uint64_t)c; c >>= 64; }
                for (int j = i + 4; c && j < 10; ++j) { c += T[j]; T[j] = (uint64_t)c; c >>= 64; }
            }
        }
        for (int i = 0; i < 4; ++i) {
            const uint64_t m = T[i] * inv;
            u128 c = 0;
            for (int j = 0; j < 4; ++j) { c += (u128)m * p[j] + T[i + j]; T[i + j] = (uint64_t)c; c >>= 64; }
            for (int j = i + 4; c && j < 10; ++j) { c += T[j]; T[j] = (uint64_t)c; c >>= 64; }
        }
        uint64_t R[5] = {T[4], T[5], T[6], T[7], T[8]};       // < (n p^2 + p 2^256) / 2^256 < (n + 1) p: a few subtractions
        for (;;) {
            uint64_t D[5];
            u128 bw = 0;
            for (int i = 0; i < 5; ++i) {
                const u128 d = (u128)R[i] - (i < 4 ? p[i] : 0) - (uint64_t)bw;
                D[i] = (uint64_t)d;
                bw = (d >> 64) ? 1 : 0;
            }
            if (bw) break;
            for (int i = 0; i < 5; ++i) R[i] = D[i];
        }
        E o;
        for (int i = 0; i < 4; ++i) o.l[i] = R[i];
        return o;
    }
};

// Hades permutation as the reference's PoseidonHash runs it (poseidon_hash.rs:66-105 through the poseidon crate): R_F / 2 full
// rounds, R_P partial rounds (s-box on state[0] only), R_F / 2 full rounds; every round = add constants, s-box x^5, MDS.
template <class F>
void permute(const Hash &h, std::vector<fe_t> &st_io) {
    using H = Host64<typename F::Params>;
    using E = typename H::E;
    const size_t t = h.t, half = h.r_f / 2;
    if (h.ifma) {                                        // AVX-512 IFMA: the T elements in the lanes of one register
        uint64_t buf[16 * 4];
        for (size_t i = 0; i < t; ++i) { const E e = H::load(st_io[i]); std::memcpy(buf + 4 * i, e.l, 32); }
        ifma_permute(h.ifma, buf);
        for (size_t i = 0; i < t; ++i) { E e; std::memcpy(e.l, buf + 4 * i, 32); st_io[i] = H::store(e); }
        return;
    }
    E st[16], nx[16];
    for (size_t i = 0; i < t; ++i) st[i] = H::load(st_io[i]);
    const E *rc = reinterpret_cast<const E *>(h.rc64.data()), *mds = reinterpret_cast<const E *>(h.mds64.data());
    for (size_t r = 0; r < h.r_f + h.r_p; ++r) {
        for (size_t i = 0; i < t; ++i) st[i] = H::add(st[i], rc[r * t + i]);
        if (r < half || r >= half + h.r_p) {
            for (size_t i = 0; i < t; ++i) st[i] = H::pow5(st[i]);
        } else {
            st[0] = H::pow5(st[0]);
        }
        for (size_t i = 0; i < t; ++i) nx[i] = H::dot(&mds[i * t], st, t);
        for (size_t i = 0; i < t; ++i) st[i] = nx[i];
    }
    for (size_t i = 0; i < t; ++i) st_io[i] = H::store(st[i]);
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

// ---------------------------------------------------------------------------------------------------------------------
// The same sponge ON THE DEVICE, kept for the measured comparison with the host code above (DESIGN.md 4.8;
// tools/poseidon_probe.py): one workgroup of 64 lanes = one wavefront, because a permutation is a chain --
//   lanes 0..t-1      : add the round constants, s-box x^5 (3 dependent products; a partial round keeps only lane 0 busy)
//   lanes (i, j) < t^2 : the t^2 products M[i][j] * s[j] of the MDS step, one each, summed per row through LDS
// on the 9 x 29-bit multiplier (the shortest dependent product the chip has).  The state is kept in the ABI's 2^256 Montgomery
// form as lazy 9-limb values; every product on that multiplier loses 2^-5, which the MDS constants give back: a column whose
// element went through the s-box (x^5 = 4 products) is stored times 2^25, any other column times 2^5.
struct DevSpongeArgs {
    const fe_t *in;        // padded input: n_chunks * rate elements (the sponge's padding rule applied by the host)
    const fe_t *rc;        // [(r_f + r_p)][t]   canonical ABI form
    const fe_t *mds_sbox;  // [t][t] times 2^25  (ABI form)
    const fe_t *mds_plain; // [t][t] times 2^5
    fe_t cap;              // initial state[0] = 2^64 (ABI form)
    fe_t one261;           // 2^261 in ABI form: product with it leaves a lazy value below 2p unchanged in value
    uint32_t n_chunks, t, rate, r_f, r_p;
    fe_t *out;             // state[1], canonical ABI (Montgomery) form
};

template <class PP>
__global__ void SRS_KERNEL_BOUNDS(64, 1) k_poseidon_sponge(DevSpongeArgs A) {
    using G = Fp29<PP>;
    __shared__ uint32_t st[16][9];          // state, lazy 9 x 29 (value < 12 p, limbs < 2^29 after normalize)
    __shared__ uint32_t prod[16 * 16][9];   // MDS products
    const uint32_t lane = threadIdx.x, t = A.t, half = A.r_f / 2;
    const uint32_t mi = lane / t, mj = lane % t;                      // MDS role (lane < t * t)
    f29_t m_sbox = G::zero(), m_plain = G::zero();
    if (lane < t * t) {
        m_sbox = G::unpack(A.mds_sbox[lane]);
        m_plain = G::unpack(A.mds_plain[lane]);
    }
    if (lane < t) {
        const f29_t s0 = lane == 0 ? G::unpack(A.cap) : G::zero();
        for (int w = 0; w < 9; ++w) st[lane][w] = s0.v[w];
    }
    __syncthreads();
    for (uint32_t c = 0; c < A.n_chunks; ++c) {
        if (lane >= 1 && lane <= A.rate) {                            // absorb: state[1 + i] += chunk[i]
            f29_t s;
            for (int w = 0; w < 9; ++w) s.v[w] = st[lane][w];
            s = G::normalize(G::add_lazy(s, G::unpack(A.in[(size_t)c * A.rate + lane - 1])));
            for (int w = 0; w < 9; ++w) st[lane][w] = s.v[w];
        }
        __syncthreads();
        for (uint32_t r = 0; r < A.r_f + A.r_p; ++r) {
            const bool full = r < half || r >= half + A.r_p;
            if (lane < t) {
                f29_t s;
                for (int w = 0; w < 9; ++w) s.v[w] = st[lane][w];
                s = G::normalize(G::add_lazy(s, G::unpack(A.rc[r * t + lane])));          // < 12 p
                if (full || lane == 0) {
                    const f29_t s2 = G::sqr(s);
                    s = G::mul(G::sqr(s2), s);                                         // x^5 * 2^-20 (4 products), < 2 p
                }
                for (int w = 0; w < 9; ++w) st[lane][w] = s.v[w];
            }
            __syncthreads();
            if (lane < t * t) {
                f29_t x;
                for (int w = 0; w < 9; ++w) x.v[w] = st[mj][w];
                const f29_t pr = G::mul(x, (full || mj == 0) ? m_sbox : m_plain);        // < 2 p each, ABI form again
                for (int w = 0; w < 9; ++w) prod[lane][w] = pr.v[w];
            }
            __syncthreads();
            if (lane < t) {
                f29_t acc = G::zero();
                for (uint32_t j = 0; j < t; ++j)
                    for (int w = 0; w < 9; ++w) acc.v[w] += prod[lane * t + j][w];     // t <= 16 values < 2 p, limbs < 2^29: no wrap
                acc = G::normalize(acc);
                // t * 2 p can exceed what the next product may take (<= 12 p) for t > 5: fold with 2^261 (value unchanged, < 2 p)
                if (t > 5) acc = G::mul(acc, G::unpack(A.one261));
                for (int w = 0; w < 9; ++w) st[lane][w] = acc.v[w];
            }
            __syncthreads();
        }
    }
    if (lane == 1) {
        f29_t s;
        for (int w = 0; w < 9; ++w) s.v[w] = st[1][w];
        *A.out = G::to_canonical_fe(G::mul(s, G::unpack(A.one261)));
    }
}

template <class F, class PP>
static bool run_device(Hash &h, fe_t &canon, double *kernel_ms, std::string &err) {
    const size_t t = h.t, rate = h.rate, n = h.buf.size();
    std::vector<fe_t> in(h.buf);
    in.push_back(F::one());                                            // pre_round padding (:52-64) == absorbing a 1, then zeros
    while (in.size() % rate) in.push_back(F::zero());
    fe_t two5 = F::one(), two25;
    for (int d = 0; d < 5; ++d) two5 = F::add(two5, two5);
    two25 = two5;
    for (int d = 0; d < 20; ++d) two25 = F::add(two25, two25);
    std::vector<fe_t> ms(t * t), mp(t * t);
    for (size_t i = 0; i < t * t; ++i) { ms[i] = F::mul(h.mds[i], two25); mp[i] = F::mul(h.mds[i], two5); }
    DevSpongeArgs a;
    fe_t cap;
    std::memset(&cap, 0, sizeof cap);
    cap.v[2] = 1;
    a.cap = F::to_mont(cap);
    a.one261 = two5;                                                   // 2^5 in ABI form = 2^261 as an integer factor
    a.n_chunks = (uint32_t)(in.size() / rate);
    a.t = (uint32_t)t; a.rate = (uint32_t)rate; a.r_f = (uint32_t)h.r_f; a.r_p = (uint32_t)h.r_p;
    fe_t *d = nullptr;
    const size_t words = in.size() + h.rc.size() + 2 * t * t + 1;
    SRS_HIP_CHECK(hipMalloc((void **)&d, words * sizeof(fe_t)));
    bool ok = true;
    try {
        fe_t *d_in = d, *d_rc = d_in + in.size(), *d_ms = d_rc + h.rc.size(), *d_mp = d_ms + t * t, *d_out = d_mp + t * t;
        SRS_HIP_CHECK(hipMemcpy(d_in, in.data(), in.size() * sizeof(fe_t), hipMemcpyHostToDevice));
        SRS_HIP_CHECK(hipMemcpy(d_rc, h.rc.data(), h.rc.size() * sizeof(fe_t), hipMemcpyHostToDevice));
        SRS_HIP_CHECK(hipMemcpy(d_ms, ms.data(), t * t * sizeof(fe_t), hipMemcpyHostToDevice));
        SRS_HIP_CHECK(hipMemcpy(d_mp, mp.data(), t * t * sizeof(fe_t), hipMemcpyHostToDevice));
        a.in = d_in; a.rc = d_rc; a.mds_sbox = d_ms; a.mds_plain = d_mp; a.out = d_out;
        hipEvent_t e0, e1;
        SRS_HIP_CHECK(hipEventCreate(&e0));
        SRS_HIP_CHECK(hipEventCreate(&e1));
        SRS_HIP_CHECK(hipEventRecord(e0, nullptr));
        SRS_LAUNCH((k_poseidon_sponge<PP>), (1), (64), 0, nullptr, a);
        SRS_HIP_CHECK(hipEventRecord(e1, nullptr));
        fe_t mont;
        SRS_HIP_CHECK(hipMemcpy(&mont, d_out, sizeof(fe_t), hipMemcpyDeviceToHost));
        float ms_f = 0;
        SRS_HIP_CHECK(hipEventElapsedTime(&ms_f, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (kernel_ms) *kernel_ms = ms_f;
        canon = F::from_mont(mont);
    } catch (...) {
        (void)hipFree(d);
        throw;
    }
    (void)hipFree(d);
    (void)n;
    (void)err;
    return ok;
}

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
    auto pack64 = [](const std::vector<fe_t> &in, std::vector<uint64_t> &out) {      // the same constants as 4 x u64 limbs
        out.resize(in.size() * 4);
        for (size_t i = 0; i < in.size(); ++i)
            for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint64_t)in[i].v[2 * j] | ((uint64_t)in[i].v[2 * j + 1] << 32);
    };
    pack64(h->rc, h->rc64);
    pack64(h->mds, h->mds64);
    if (ifma_available() && t <= 8) {
        uint64_t p4[4];
        for (int i = 0; i < 4; ++i)
            p4[i] = field == 0 ? Host64<FrP>::P(i) : Host64<FqP>::P(i);
        const uint64_t inv = field == 0 ? Host64<FrP>::inv64() : Host64<FqP>::inv64();
        h->ifma = ifma_prepare(p4, inv, h->rc64.data(), h->mds64.data(), t, r_f, r_p);
    }
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

bool squeeze_device(Hash &h, size_t num_bits, int out_field, fe_t &out, double *kernel_ms, std::string &err) {
    if (num_bits == 0 || num_bits > 256) { err = "num_bits must be in 1..256"; return false; }
    if (h.t * h.t > 64) { err = "device sponge: T <= 8 (one wavefront holds the T^2 MDS products)"; return false; }
    fe_t canon;
    if (!(h.field == 0 ? run_device<Fr, FrP>(h, canon, kernel_ms, err) : run_device<Fq, FqP>(h, canon, kernel_ms, err))) return false;
    for (size_t b = num_bits; b < 256; ++b) canon.v[b >> 5] &= ~(1u << (b & 31));
    if (num_bits > 253 && !(out_field == 0 ? less_than_p<Fr, FrP>(canon) : less_than_p<Fq, FqP>(canon))) {
        err = "squeezed value is not a canonical element of the output field";
        return false;
    }
    out = out_field == 0 ? Fr::to_mont(canon) : Fq::to_mont(canon);
    return true;
}

}  // namespace poseidon
}  // namespace srs
