// poseidon.h -- off-circuit Poseidon random oracle (host code; see poseidon.hip).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "field.cuh"

namespace srs {
namespace poseidon {

// AVX-512 IFMA permutation (poseidon_x86.hip): T state elements in the lanes of one register; chosen at run time
struct IfmaConsts;
bool ifma_available();
IfmaConsts *ifma_prepare(const uint64_t p4[4], uint64_t inv64, const uint64_t *rc64, const uint64_t *mds64, size_t t, size_t r_f, size_t r_p);
void ifma_release(IfmaConsts *K);
void ifma_permute(const IfmaConsts *K, uint64_t *state /* t x 4 limbs, canonical 2^256-Montgomery, in place */);

struct Hash {
    int field = 0;                       // 0 bn256::Fr, 1 bn256::Fq
    size_t t = 0, rate = 0, r_f = 0, r_p = 0;
    std::vector<fe_t> rc;                // [(r_f + r_p)][t] round constants, Montgomery
    std::vector<fe_t> mds;               // [t][t]
    std::vector<uint64_t> rc64, mds64;   // the same constants as 4 x u64 limbs (the host permutation's form)
    IfmaConsts *ifma = nullptr;          // non-null: permutations run on AVX-512 IFMA (same values)
    ~Hash() { if (ifma) ifma_release(ifma); }
    std::vector<fe_t> buf;               // absorbed elements (kept across squeezes, like the reference)
    std::vector<fe_t> state;             // sponge state after the first `done` elements (full chunks only); empty = fresh
    size_t done = 0;
};

// Spec::new(r_f, r_p) (src/poseidon/spec.rs:14-16).  nullptr + err on bad parameters.
Hash *create(int field, size_t t, size_t rate, size_t r_f, size_t r_p, std::string &err);
void absorb(Hash &h, const fe_t *v, size_t n);
// PoseidonHash::output (src/poseidon/poseidon_hash.rs:190-212): low `num_bits` bits of state[1], as an element of `out_field`
bool squeeze(Hash &h, size_t num_bits, int out_field, fe_t &out, std::string &err);

// The same value from the device sponge (one wavefront; poseidon.hip): exists for the measured host-vs-device comparison of
// DESIGN.md 4.8.  kernel_ms (optional): HIP-event time of the kernel.  Throws DeviceError on HIP failures.
bool squeeze_device(Hash &h, size_t num_bits, int out_field, fe_t &out, double *kernel_ms, std::string &err);

}  // namespace poseidon
}  // namespace srs
