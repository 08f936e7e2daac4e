// lanes.cuh -- the cross-lane primitives the kernels are written against: DPP quad broadcasts and wavefront shuffles of multi-word
// values, plus the storage class of the sweep kernels' per-point accumulators.
//
// This header is the ONLY place in the device code that knows about the CPU logic emulator (tests/emu/hipemu.h, test
// infrastructure: the dev container has no GPU): with SRS_EMU the same four primitives come from the emulator's fiber model.
// Product builds (hipcc, gfx950) never define SRS_EMU; every kernel source is branch-free in that respect.
#pragma once
#include <stdint.h>

#include "field.cuh"      // SRS_D

#if defined(SRS_EMU)
#include "hipemu.h"
#endif

namespace srs {

// ---- quad cooperation: 4 adjacent lanes (a DPP "quad") hold identical operands and share one group addition (curve.cuh) ----
// quad_bcast_words<K, N>: every lane of a quad receives the N words of the quad's lane K.
#if defined(SRS_EMU)
template <int K, int N>
SRS_D void quad_bcast_words(const uint32_t *in, uint32_t *out) { __emu_quad_bcast_n(in, K, out, N); }
template <int N>
SRS_D void shfl_down_words(const uint32_t *in, uint32_t *out, unsigned delta, int width) { __emu_shfl_down_bulk(in, out, delta, width, N); }
#define SRS_SWEEP_ACC(name) static thread_local uint32_t name[(DMAX + 1) * SW_WORDS * RP_THREADS]
#define SRS_DYN_LDS(type, name, max_elems) static thread_local type name[max_elems]
#else
template <int K>
SRS_D uint32_t quad_bcast_u32(uint32_t v) {   // quad_perm:[K,K,K,K]: register crossbar, no LDS
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, K * 0x55, 0xF, 0xF, true);
#else
    return v;                                 // host pass of hipcc: never executed
#endif
}
template <int K, int N>
SRS_D void quad_bcast_words(const uint32_t *in, uint32_t *out) {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = quad_bcast_u32<K>(in[i]);
}
// shfl_down_words<N>: lane l receives the N words of lane l + delta inside groups of `width` lanes (64-wide wavefronts)
template <int N>
SRS_D void shfl_down_words(const uint32_t *in, uint32_t *out, unsigned delta, int width) {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = __shfl_down(in[i], delta, width);
}
// per-point accumulators of the sweep kernels: dynamic LDS sized by the launch (rowprog_dev.cuh: sweep_smem_bytes)
#define SRS_SWEEP_ACC(name) extern __shared__ uint32_t name[]
// dynamic LDS of `type`, sized by the launch (the emulator build reserves max_elems)
#define SRS_DYN_LDS(type, name, max_elems) extern __shared__ type name[]
#endif

}  // namespace srs
