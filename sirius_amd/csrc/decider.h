// decider.h -- sparse matrix-vector product for the permutation check of the deciders (see decider.hip).
#pragma once
#include <string>

#include "devrt.h"
#include "field.cuh"

namespace srs {
namespace decider {

struct Sparse;   // device-resident CSR copy of the reference's SparseMatrix<F> = Vec<(row, col, value)>

// rc: 0 ok, 4 invalid (row / column index outside n), 5 device
Sparse *create(int field, size_t n, const uint64_t *rows, const uint64_t *cols, const fe_t *values, size_t nnz, int &rc,
               std::string &err);
void destroy(Sparse *M);
size_t dim(const Sparse *M);
// y = M * z  (z, y: DEVICE vectors of dim(M) elements)
void matvec(Sparse *M, const fe_t *z_dev, fe_t *y_dev, hipStream_t st);
// number of rows with (M * z)[row] != z[row]
size_t permutation_mismatches(Sparse *M, const fe_t *z_dev, hipStream_t st);

}  // namespace decider
}  // namespace srs
