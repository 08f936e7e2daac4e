// poseidon.h -- off-circuit Poseidon random oracle (host code; see poseidon.hip).
#pragma once
#include <cstddef>
#include <string>
#include <vector>

#include "field.cuh"

namespace srs {
namespace poseidon {

struct Hash {
    int field = 0;                       // 0 bn256::Fr, 1 bn256::Fq
    size_t t = 0, rate = 0, r_f = 0, r_p = 0;
    std::vector<fe_t> rc;                // [(r_f + r_p)][t] round constants, Montgomery
    std::vector<fe_t> mds;               // [t][t]
    std::vector<fe_t> buf;               // absorbed elements (kept across squeezes, like the reference)
    std::vector<fe_t> state;             // sponge state after the first `done` elements (full chunks only); empty = fresh
    size_t done = 0;
};

// Spec::new(r_f, r_p) (src/poseidon/spec.rs:14-16).  nullptr + err on bad parameters.
Hash *create(int field, size_t t, size_t rate, size_t r_f, size_t r_p, std::string &err);
void absorb(Hash &h, const fe_t *v, size_t n);
// PoseidonHash::output (src/poseidon/poseidon_hash.rs:190-212): low `num_bits` bits of state[1], as an element of `out_field`
bool squeeze(Hash &h, size_t num_bits, int out_field, fe_t &out, std::string &err);

}  // namespace poseidon
}  // namespace srs
