// ntt.h -- internal interface of the NTT engine (see ntt.hip).
#pragma once
#include "devrt.h"
#include "field.cuh"

namespace srs {
namespace ntt {
constexpr uint32_t FR_S = 28;   // 2-adicity of bn256::Fr (F::S, reference src/fft.rs:13)
// `a`: DEVICE pointer to `batch` vectors of 2^log_n Fr elements (Montgomery), `stride` apart; in place.
void run(fe_t *a, uint32_t log_n, size_t stride, uint32_t batch, bool inverse, bool coset, hipStream_t stream);
void release_plans();
fe_t omega(uint32_t k, bool inverse);   // get_omega_or_inv (src/fft.rs:12-23), Montgomery
fe_t zeta();                            // WithSmallOrderMulGroup<3>::ZETA
void set_max_radix_bits(uint32_t bits);   // 4..8; drops cached plans
}  // namespace ntt
}  // namespace srs
