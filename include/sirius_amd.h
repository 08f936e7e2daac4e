/*
 * sirius_amd.h -- C-ABI of the MI355X-native folding-prover hot path for snarkify/sirius.
 *
 * This is the drop-in boundary: a thin Rust shim replaces only the BODIES of the reference
 * functions cited at each entry point (paths relative to the sirius repository) with calls into
 * libsirius_amd.so; the StepCircuit / IVC / nifs API above stays untouched (INTEGRATION.md shows
 * the `extern "C"` block and the shim).  Plain pointers and sizes only; no C++/torch types.
 *
 * Data layout (what halo2curves keeps in memory, so the shim can pass `&[F]` / `&[C]` as-is):
 *   srs_fe     = 4 x u64 little-endian limbs, Montgomery form (R = 2^256)
 *   srs_affine = x || y, identity = all-zero bytes
 * `repr` arguments select SRS_REPR_MONT (default, zero-copy from Rust) or SRS_REPR_CANON for
 * scalars; srs_layout_selftest() lets the shim verify the Montgomery assumption at start-up.
 *
 * Memory spaces: every data pointer of a call is either a host pointer (SRS_SPACE_HOST, what the
 * Rust shim passes; the library stages H2D/D2H itself) or a device pointer (SRS_SPACE_DEVICE,
 * used when witnesses already live in HBM).  Small outputs (commitments) are always written to
 * host memory.  Outputs are written only when the call returns SRS_OK.
 *
 * Calls are synchronous from the caller's point of view and thread-compatible (distinct handles
 * may be used from distinct threads).  Errors: int return code + srs_last_error() (thread-local).
 * There is NO CPU fallback: without a gfx950 device every compute entry returns SRS_ERR_DEVICE.
 */
#ifndef SIRIUS_AMD_H
#define SIRIUS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } srs_fe;
typedef struct { srs_fe x, y; } srs_affine;

typedef struct srs_ck srs_ck;                 /* device-resident commitment key */
typedef struct srs_structure srs_structure;   /* device-resident PlonkStructure slice (gates + fixed columns) */

enum {
    SRS_OK = 0,
    SRS_ERR_TOO_LONG_INPUT = 1,   /* commitment::Error::TooLongInput   (src/commitment.rs:82-88) */
    SRS_ERR_NOT_POW2 = 2,         /* assert!(a.len().is_power_of_two()) (src/fft.rs:161,169)      */
    SRS_ERR_K_TOO_LARGE = 3,      /* assert!(k <= F::S)                 (src/fft.rs:13)           */
    SRS_ERR_INVALID = 4,          /* bad argument (null pointer, unknown curve/field id, ...)     */
    SRS_ERR_DEVICE = 5,           /* HIP runtime failure / no gfx950 device                        */
    SRS_ERR_LAYOUT = 6,           /* srs_layout_selftest mismatch                                  */
    SRS_ERR_EVAL_INDEX = 7        /* plonk::eval::Error::*OutOfBoundary (src/plonk/eval.rs:3-25)   */
};

enum { SRS_CURVE_BN256 = 0, SRS_CURVE_GRUMPKIN = 1 };   /* src/lib.rs:29-48 (C1 / C2 of the cycle) */
enum { SRS_FIELD_FR = 0, SRS_FIELD_FQ = 1 };            /* bn256::Fr (= grumpkin::Fq) / bn256::Fq (= grumpkin::Fr) */
enum { SRS_SPACE_HOST = 0, SRS_SPACE_DEVICE = 1 };
enum { SRS_REPR_MONT = 0, SRS_REPR_CANON = 1 };

/* ---- library ---- */
int srs_init(int device_ordinal);            /* binds the calling thread's HIP device; checks gfx950 */
const char *srs_last_error(void);
const char *srs_version(void);
/* Scalar field of a curve (bn256 -> Fr, grumpkin -> Fq). */
int srs_scalar_field_of(int curve);
/* Layout self-test: the shim passes the raw bytes of F::ONE and F::from(2); rc SRS_ERR_LAYOUT if they
 * are not R mod p and 2R mod p (i.e. the build of halo2curves is not Montgomery-4x64). */
int srs_layout_selftest(int field, const srs_fe *one, const srs_fe *two);

/* ---- CommitmentKey (src/commitment.rs:29-32) ----
 * srs_ck_create: device cache of `CommitmentKey<C>::ck` (Rust keeps owning the Box<[C]>).
 * The key is expanded into its 16-bit-window table (16x the size) in HBM; see DESIGN.md.
 * Multi-GPU: rank `rank` of `world` keeps only its block-cyclic stripes (2^10 bases) of `bases`
 * (which is always the FULL key); commits then return this rank's PARTIAL sum, to be combined
 * with srs_point_sum after an all-gather of the 64-byte partials (world = 1: the full result). */
int srs_ck_create(int curve, const srs_affine *bases, size_t len, int space, srs_ck **out);
int srs_ck_create_sharded(int curve, const srs_affine *bases, size_t len, int space,
                          uint32_t rank, uint32_t world, srs_ck **out);
void srs_ck_free(srs_ck *ck);
size_t srs_ck_len(const srs_ck *ck);         /* CommitmentKey::len (src/commitment.rs:47-49) */

/* CommitmentKey::commit (src/commitment.rs:81-90):  out = sum_{i<n} scalars[i] * ck[i]  -> affine.
 * n > len -> SRS_ERR_TOO_LONG_INPUT; n == 0 -> identity.  `stream` = hipStream_t or NULL. */
int srs_commit(srs_ck *ck, const srs_fe *scalars, size_t n, int space, int repr, void *stream,
               srs_affine *out);
/* The d-1 cross-term commitments share one base prefix (src/nifs/sangria/mod.rs:151-154):
 * out[m] = commit(scalars[m][0..n[m]]) for m < batch, one set of launches. */
int srs_commit_batch(srs_ck *ck, const srs_fe *const *scalars, const size_t *n, size_t batch,
                     int space, int repr, void *stream, srs_affine *out);
/* out = sum of `n` affine points (host); combines per-rank partial commitments. */
int srs_point_sum(int curve, const srs_affine *points, size_t n, srs_affine *out);
/* out = [scalar] P  (the 1-element best_multiexp of src/nifs/sangria/accumulator.rs:213,243), host. */
int srs_point_mul(int curve, const srs_fe *scalar, int repr, const srs_affine *p, srs_affine *out);

#ifdef __cplusplus
}
#endif
#endif /* SIRIUS_AMD_H */
