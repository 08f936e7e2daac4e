// rowprog.h -- internal interface of the row-program engine (see rowprog.hip).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "devrt.h"
#include "field.cuh"
#include "rowprog_dev.cuh"

namespace srs {
namespace rowprog {

struct Structure;

// rc: 0 ok, 4 invalid, 5 device, 7 index out of range
Structure *create(int field, uint32_t k, size_t num_selectors, size_t num_fixed, size_t num_advice,
                  const uint8_t *const *selectors, const fe_t *const *fixed, int space_device,
                  const uint64_t *gates, size_t gates_words, size_t num_gates, size_t num_lookups, bool has_vector_lookup,
                  const uint64_t *lookup_exprs, size_t lookup_words, int &rc, std::string &err);
void destroy(Structure *S);
// multi-GPU: cross terms (mode 0 of evaluate) are computed only on the rows of this rank's block-cyclic stripes
void set_shard(Structure *S, uint32_t rank, uint32_t world);
uint32_t shard_world(const Structure *S);
uint32_t shard_rank(const Structure *S);
void rotation_range(const Structure *S, int32_t *lo, int32_t *hi);   // min / max rotation over every column query (0, 0: none)
uint32_t log_rows(const Structure *S);
size_t degree(const Structure *S);            // homogeneous degree d = number of cross terms
size_t num_challenges(const Structure *S);    // PlonkStructure::num_challenges
size_t num_advice(const Structure *S);
size_t num_witness_columns(const Structure *S);   // num_advice + 5 * num_lookups: columns of W[0] || W[1] || ...
size_t num_lookups(const Structure *S);
size_t rows(const Structure *S);
int field(const Structure *S);

// mode 0: cross terms T_1..T_d (W1, W2, challenges = U1.ch || U1.u || U2.ch || 1)
// mode 1: compressed gate value per row on W1   (challenges = U.ch)
// mode 2: homogeneous gate value per row on W1  (challenges = U.ch || U.u)
// W*, outputs: DEVICE pointers (out_dev_ptrs_host = host array of device pointers).
int evaluate(Structure *S, int mode, const fe_t *W1_dev, const fe_t *W2_dev, const fe_t *challenges_host, size_t n_ch,
             fe_t *const *out_dev_ptrs_host, hipStream_t st, std::string &err, bool sync = true);

// ---- ProtoGalaxy (Fr only) ----
struct PgSizes {   // PolyContext, src/nifs/protogalaxy/poly/mod.rs:205-269
    size_t count_with_padding = 0, betas_count = 0, points_F = 0, points_G = 0, instances_to_fold = 0, lagrange_domain = 0;
    uint32_t log_domain_K = 0;
};
bool pg_sizes(const Structure *S, size_t traces_len, PgSizes &out);
std::vector<fe_t> lagrange_eval(const fe_t &X, uint32_t log_n);
fe_t poly_eval(const fe_t *coeffs, size_t n, const fe_t &x);
// compute_G's values left ON THE DEVICE for pg_K_from_G_device (r06): G at the integer nodes node0 .. node0 + n_vals - 1 (the first one
// is `g_at_one`, a host value, when skip_one), valid until the structure's next pg_* call
struct PgGValues {
    const fe_t *vals_dev = nullptr;      // n_dev evaluated values
    uint32_t n_dev = 0, degree = 0;      // degree = d_G: degree + 1 nodes in all
    bool skip_one = false;               // nodes 1 .. d_G + 1, value at node 1 = F(alpha) (not in vals_dev); else nodes 0 .. d_G
};
int pg_sum(Structure *S, int mode, const fe_t *const *W_dev, const fe_t *const *challenges_host, size_t n_ch, size_t J,
           const fe_t *weights_in, size_t n_weights, const fe_t *delta, int compat, hipStream_t st, fe_t *out_host,
           size_t *n_out, std::string &err, const fe_t *g_at_one = nullptr, PgGValues *keep_on_device = nullptr);
// compute_K_from_G straight from compute_G's device values (one incoming trace, K domain <= 4096 points): interpolation, the K points and the
// coset ifft in one chain of launches, ONE synchronisation (the host route: D2H of G, host interpolation and Horner, H2D of the points)
bool pg_K_device_ok(const PgGValues &g, uint32_t log_domain_K);
int pg_K_from_G_device(Structure *S, const PgGValues &g, const fe_t &f_alpha, size_t instances_to_fold, uint32_t log_domain_K, hipStream_t st,
                       fe_t *out_host, std::string &err);
int pg_K_from_G(const fe_t *polyG_host, size_t nG, const fe_t &f_alpha, size_t instances_to_fold, uint32_t log_domain_K,
                hipStream_t st, fe_t *out_host, std::string &err);
// world > 1: only the elements of rank's block-cyclic stripes (2^10 each) of out[0 .. n) are written
int lincomb(int field, fe_t *out, const fe_t *const *w_dev, const fe_t *coefs, size_t J, size_t n, hipStream_t st, std::string &err,
            uint32_t rank = 0, uint32_t world = 1);
// the same sum on `n_rows` listed rows (device array) of each of `cols` columns of length col_len
int lincomb_rows(int field, fe_t *out, const fe_t *const *w_dev, const fe_t *coefs, size_t J, const uint32_t *rows_dev, size_t n_rows, size_t cols,
                 size_t col_len, hipStream_t st, std::string &err);

// straight-line C++ of the structure's row program (tools/gen_rowprog_spec.py), its fingerprint and the
// ahead-of-time kernel it maps to (-1: interpreter)
const char *spec_source(Structure *S, int which, uint64_t *fingerprint, int *spec_id, std::string &buf);
int kernel_kind(const Structure *S, int which);      // which: 0 cross terms, 1 gates, 2 ProtoGalaxy leaves -> 0 / 1 / 2
// hiprtc compiles a small program in the emitted form against the embedded device headers (host only, no device needed)
bool jit_selfcheck(size_t *code_bytes, std::string &log);

// lookup arguments (src/plonk/lookup.rs): all pointers DEVICE; ls / ts / ms HOST arrays of num_lookups DEVICE vectors
int lookup_coeff_1(Structure *S, const fe_t *advice_dev, const fe_t &r, fe_t *const *ls, fe_t *const *ts, fe_t *const *ms,
                   hipStream_t st, std::string &err);
void lookup_coeff_2(int field, const fe_t *l, const fe_t *t, const fe_t *m, const fe_t &r, size_t n, fe_t *h, fe_t *g, hipStream_t st);
size_t log_derivative_mismatches(Structure *S, const fe_t *W_dev, hipStream_t st);
// batch_invert_assigned (src/util/mod.rs:119-153); has_den may be nullptr (every element has a denominator)
void assigned_invert(int field, const fe_t *num, const fe_t *den, const uint8_t *has_den, size_t n, fe_t *out, hipStream_t st);

size_t count_mismatch(const fe_t *a_dev, const fe_t *b_dev /* or nullptr: compare with 0 */, size_t n, hipStream_t st);

void fold_w(int field, fe_t *out, const fe_t *w1, const fe_t *w2, const fe_t &r, size_t n, hipStream_t st);
int fold_e(int field, fe_t *out, const fe_t *e, const fe_t *const *t_dev_ptrs_host, size_t n_terms, const fe_t &r, size_t n,
           hipStream_t st, std::string &err);

}  // namespace rowprog
}  // namespace srs
