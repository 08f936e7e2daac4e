// decider.hip -- the copy-constraint (permutation) check of the Sangria deciders on gfx950.
//
// Reference: VanillaFS::is_sat_permutation (src/nifs/sangria/mod.rs:385-453) builds
//   Z = consistency markers || padding || W[0][.. num_advice * 2^k]
// and counts the rows where (P * Z)[row] != Z[row], P = the sparse permutation matrix of the copy constraints
// (PermutationData::matrix, src/plonk/permutation.rs; sparse::matrix_multiply, src/polynomial/sparse.rs:7-19:
// result[row] += value * Z[col], serial over the COO list).
//
// Here the COO list is sorted by row once (CSR, device resident); one thread per row gathers its entries.  Field
// addition is exact and commutative, so the order of accumulation inside a row does not change the bits.
// Entries equal to 1 (every entry of a permutation matrix) skip the multiplication.  HBM-bound gather:
// 32 B (Z[col]) + 32 B (Z[row]) + 8 B index per row.
#include "decider.h"

#include <algorithm>
#include <cstring>
#include <memory>
#include <numeric>
#include <vector>

namespace srs {
namespace decider {

struct Sparse {
    int field = 0;
    size_t n = 0, nnz = 0;
    uint32_t *row_ptr = nullptr;   // n + 1
    uint32_t *col = nullptr;       // nnz
    fe_t *val = nullptr;           // nnz, or nullptr when every value is 1
    uint32_t *d_count = nullptr;
};

template <class F>
__global__ void k_spmv(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col, const fe_t *__restrict__ val,
                       const fe_t *__restrict__ z, uint32_t n, fe_t *__restrict__ y, uint32_t *__restrict__ mismatch) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    fe_t acc = F::zero();
    for (uint32_t e = row_ptr[row], end = row_ptr[row + 1]; e < end; ++e) {
        fe_t x = z[col[e]];
        acc = F::add(acc, val ? F::mul(val[e], x) : x);
    }
    if (y) y[row] = acc;
    if (mismatch && !F::eq(acc, z[row])) atomicAdd(mismatch, 1u);
}

Sparse *create(int field, size_t n, const uint64_t *rows, const uint64_t *cols, const fe_t *values, size_t nnz, int &rc,
               std::string &err) {
    rc = 4;
    if (n >= 0xFFFFFFFFull || nnz >= 0xFFFFFFFFull) { err = "matrix too large"; return nullptr; }
    const fe_t one = field == 0 ? Fr::one() : Fq::one();
    bool all_one = true;
    for (size_t e = 0; e < nnz; ++e) {
        if (rows[e] >= n) { err = "row index " + std::to_string(rows[e]) + " outside the matrix"; return nullptr; }
        if (cols[e] >= n) { err = "invalid matrix multiply"; return nullptr; }      // sparse.rs:15-17
        if (all_one && !Fr::eq(values[e], one)) all_one = false;
    }
    std::vector<uint32_t> ptr(n + 1, 0), c(nnz);
    std::vector<fe_t> v(all_one ? 0 : nnz);
    for (size_t e = 0; e < nnz; ++e) ptr[rows[e] + 1]++;
    for (size_t i = 0; i < n; ++i) ptr[i + 1] += ptr[i];
    std::vector<uint32_t> fill(ptr.begin(), ptr.end() - 1);
    for (size_t e = 0; e < nnz; ++e) {
        uint32_t at = fill[rows[e]]++;
        c[at] = (uint32_t)cols[e];
        if (!all_one) v[at] = values[e];
    }
    std::unique_ptr<Sparse> M(new Sparse());
    M->field = field;
    M->n = n;
    M->nnz = nnz;
    rc = 5;
    SRS_HIP_CHECK(hipMalloc((void **)&M->row_ptr, (n + 1) * sizeof(uint32_t)));
    SRS_HIP_CHECK(hipMalloc((void **)&M->col, (nnz + 1) * sizeof(uint32_t)));
    SRS_HIP_CHECK(hipMalloc((void **)&M->d_count, sizeof(uint32_t)));
    SRS_HIP_CHECK(hipMemcpy(M->row_ptr, ptr.data(), (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (nnz) SRS_HIP_CHECK(hipMemcpy(M->col, c.data(), nnz * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (!all_one) {
        SRS_HIP_CHECK(hipMalloc((void **)&M->val, nnz * sizeof(fe_t)));
        SRS_HIP_CHECK(hipMemcpy(M->val, v.data(), nnz * sizeof(fe_t), hipMemcpyHostToDevice));
    }
    rc = 0;
    return M.release();
}

void destroy(Sparse *M) {
    if (!M) return;
    (void)hipFree(M->row_ptr);
    (void)hipFree(M->col);
    (void)hipFree(M->val);
    (void)hipFree(M->d_count);
    delete M;
}

size_t dim(const Sparse *M) { return M->n; }

static void launch(Sparse *M, const fe_t *z, fe_t *y, uint32_t *cnt, hipStream_t st) {
    if (!M->n) return;
    const uint32_t n = (uint32_t)M->n, blocks = (n + 255) / 256;
    if (M->field == 0) SRS_LAUNCH((k_spmv<Fr>), (blocks), (256), 0, st, (const uint32_t *)M->row_ptr, (const uint32_t *)M->col, (const fe_t *)M->val, z, n, y, cnt);
    else SRS_LAUNCH((k_spmv<Fq>), (blocks), (256), 0, st, (const uint32_t *)M->row_ptr, (const uint32_t *)M->col, (const fe_t *)M->val, z, n, y, cnt);
}

void matvec(Sparse *M, const fe_t *z_dev, fe_t *y_dev, hipStream_t st) {
    launch(M, z_dev, y_dev, nullptr, st);
    SRS_HIP_CHECK(hipStreamSynchronize(st));
    SRS_HIP_CHECK(hipGetLastError());
}

size_t permutation_mismatches(Sparse *M, const fe_t *z_dev, hipStream_t st) {
    uint32_t h = 0;
    SRS_HIP_CHECK(hipMemsetAsync(M->d_count, 0, sizeof(uint32_t), st));
    launch(M, z_dev, nullptr, M->d_count, st);
    SRS_HIP_CHECK(hipMemcpyAsync(&h, M->d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SRS_HIP_CHECK(hipStreamSynchronize(st));
    SRS_HIP_CHECK(hipGetLastError());
    return h;
}

}  // namespace decider
}  // namespace srs
