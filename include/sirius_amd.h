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
 * Calls are synchronous from the caller's point of view (the two streaming folds on device-resident
 * operands are stream-ordered instead, see srs_fold_witness) and thread-compatible (distinct handles
 * may be used from distinct threads; the library rebinds the calling thread to the process's GPU; commits on ONE key handle from several threads are
 * serialised inside the library -- the key's scratch memory is per handle).  Errors: int return code + srs_last_error() (thread-local).
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
    SRS_ERR_EVAL_INDEX = 7,       /* plonk::eval::Error::*OutOfBoundary (src/plonk/eval.rs:3-25)   */
    SRS_ERR_IO = 8,               /* io::Error from File::open / read_exact / write_all (src/commitment.rs:99-127) */
    SRS_ERR_INVALID_DATA = 9,     /* io::ErrorKind::InvalidData "Wrong file in cache, some ptr out of curve" (:152-158) */
    SRS_ERR_UNSUPPORTED = 10      /* part of the reference that lives in an un-vendored third-party crate (srs_ck_setup) */
};

enum { SRS_CURVE_BN256 = 0, SRS_CURVE_GRUMPKIN = 1 };   /* src/lib.rs:29-48 (C1 / C2 of the cycle) */
enum { SRS_FIELD_FR = 0, SRS_FIELD_FQ = 1 };            /* bn256::Fr (= grumpkin::Fq) / bn256::Fq (= grumpkin::Fr) */
enum { SRS_SPACE_HOST = 0, SRS_SPACE_DEVICE = 1 };
enum { SRS_REPR_MONT = 0, SRS_REPR_CANON = 1 };

/* ---- library ---- */
int srs_init(int device_ordinal);            /* binds the PROCESS to a device (-1: the current one); checks gfx950 */
/* Binds the CALLING THREAD to a device (-1: back to the process's device): one process may drive one GPU per host thread, every thread
 * with its own handles -- e.g. sharded keys and structures (srs_ck_create_sharded, srs_structure_set_shard) with rank = thread index;
 * the caller adds the ranks' partial commitments / polynomials itself (srs_point_sum, field additions).  A handle must be used by
 * threads bound to the device it was created on.  No reference counterpart (the reference is single-device CPU code). */
int srs_init_thread(int device_ordinal);
const char *srs_last_error(void);
const char *srs_version(void);
/* Run-time tunables (csrc/tuning.h holds the table): each selects among code paths the library takes by default for SOME input size
 * (or moves the size threshold between them); none changes a result.  Tests use them to run a large-input path on an input the CPU
 * oracle can check; a deployer may set "msm_wide" = 0 to save 13/16 of a large key's HBM (also: environment SRS_MSM_WIDE=0).
 * Names (srs_tuning_name(i), i = 0, 1, ... until NULL): msm_sort, msm_l0, msm_wide, msm_wide_min, msm_slots, msm_slot_log,
 * msm_expect_ovf, msm_quad_max, commit_chunks, pg_f_eval, pg_g_fft, jit_always, no_jit.  Process-wide; set before the calls they
 * affect (a value is read when a call starts).  rc SRS_ERR_INVALID for an unknown name.  No reference counterpart. */
int srs_tuning_set(const char *name, int64_t value);
int srs_tuning_get(const char *name, int64_t *value);        /* *value = INT64_MIN while unset */
void srs_tuning_reset(void);
const char *srs_tuning_name(int index);
/* Scalar field of a curve (bn256 -> Fr, grumpkin -> Fq). */
int srs_scalar_field_of(int curve);
/* Layout self-test: the shim passes the raw bytes of F::ONE and F::from(2); rc SRS_ERR_LAYOUT if they
 * are not R mod p and 2R mod p (i.e. the build of halo2curves is not Montgomery-4x64). */
int srs_layout_selftest(int field, const srs_fe *one, const srs_fe *two);
/* The same for the point layout: the shim passes the raw bytes of `C::generator()`; rc SRS_ERR_LAYOUT
 * unless they are x || y in Montgomery form of (1, 2) on bn256 G1 / (1, sqrt(-16)) on grumpkin. */
int srs_layout_selftest_point(int curve, const srs_affine *generator);

/* ---- memory the shim may own ----
 * Device-resident vectors are what keeps a fold step off PCIe: the running accumulator (W, E), the cross terms and the
 * incoming witness stay in HBM between calls, every compute entry takes them with space = SRS_SPACE_DEVICE, and only the
 * new witness goes up (srs_commit_upload) and commitments come down.
 *   srs_dev_alloc / srs_dev_free   : HBM (hipMalloc on the process's device)
 *   srs_host_alloc / srs_host_free : page-locked host memory (uploads from it are asynchronous to the caller)
 *   srs_upload   : host -> device, ordered on `stream`
 *   srs_download : device -> host, returns when the bytes are in `dst_host` */
int srs_dev_alloc(size_t bytes, void **out);
void srs_dev_free(void *p);
int srs_host_alloc(size_t bytes, void **out);
void srs_host_free(void *p);
int srs_upload(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int srs_download(void *dst_host, const void *src_dev, size_t bytes, void *stream);

/* ---- CommitmentKey (src/commitment.rs:29-32) ----
 * srs_ck_create: device cache of `CommitmentKey<C>::ck` (Rust keeps owning the Box<[C]>).
 * The key is expanded into its 16-bit-window table (16x the size) in HBM; see DESIGN.md.
 * Multi-GPU: rank `rank` of `world` keeps only its block-cyclic stripes (2^10 bases) of `bases`
 * (which is always the FULL key); commits then return this rank's PARTIAL sum, to be combined
 * with srs_point_sum after an all-gather of the 64-byte partials (world = 1: the full result). */
int srs_ck_create(int curve, const srs_affine *bases, size_t len, int space, srs_ck **out);
int srs_ck_create_sharded(int curve, const srs_affine *bases, size_t len, int space,
                          uint32_t rank, uint32_t world, srs_ck **out);
/* Synthetic key for tests and benches, generated on the device: P_i = [h(seed, i)] G.  This is NOT
 * CommitmentKey::setup (src/commitment.rs:55-79: SHAKE256 + hash_to_curve, the latter inside the
 * un-vendored halo2curves); real keys enter through srs_ck_create (e.g. from the reference's cache file). */
int srs_ck_setup_synthetic(int curve, size_t len, uint64_t seed, uint32_t rank, uint32_t world, srs_ck **out);
/* CommitmentKey::setup(k, label) (src/commitment.rs:55-79), the restatable half: chunk i of its uniform byte stream is bytes
 * [32 i, 32 i + 32) of SHAKE256(label) (`Shake256::default().chain(label).finalize_xof()`, read 32 bytes at a time, :61-67).
 * Writes chunks first .. first + count - 1 (count * 32 bytes) to `out`.  Host code (FIPS 202 restated in csrc/keygen.hip). */
int srs_ck_setup_uniform_bytes(const uint8_t *label, size_t label_len, size_t first, size_t count, uint8_t *out);
/* The other half maps each chunk to a point with halo2curves' `hash_to_curve("from_uniform_bytes")` (:69-71) -- third-party code
 * outside the reference tree, pulled by an unpinned branch: NOT restated.  Always returns SRS_ERR_UNSUPPORTED (after the
 * reference's own argument check, k < 32) with a message naming the alternatives: srs_ck_load_file on the reference's cache
 * file, or srs_ck_create on bases the Rust side generated.  (`par_bridge()` at :69 does not preserve order, so the reference
 * itself does not reproduce a key's order from the label; the cache file is what pins a key.) */
int srs_ck_setup(int curve, uint32_t k, const uint8_t *label, size_t label_len, srs_ck **out);
/* Copies this rank's bases (`srs_ck_local_len` points, window 0 of the table) to host memory. */
int srs_ck_get_bases(const srs_ck *ck, srs_affine *out);
size_t srs_ck_local_len(const srs_ck *ck);
/* Key cache file of the reference (src/commitment.rs:99-170): `{cache}/{label}/{k}.bin` = the raw memory of `[C; 2^k]`
 * (x || y, Montgomery, identity all-zero).  srs_ck_load_file = load_from_file (reads exactly 2^k points; a short file is
 * SRS_ERR_IO like read_exact) + the is_on_curve validation of load_or_setup_cache (any point off the curve ->
 * SRS_ERR_INVALID_DATA) + the device table; srs_ck_save_file = save_to_file (unsharded keys). */
int srs_ck_load_file(int curve, const char *path, size_t k, uint32_t rank, uint32_t world, srs_ck **out);
int srs_ck_save_file(const srs_ck *ck, const char *path);
/* number of bases of this rank's shard that fail y^2 = x^3 + b */
int srs_ck_count_off_curve(const srs_ck *ck, size_t *bad);
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
/* `ck.commit(&W)` for a witness that was just produced on the host (run_sps_protocol_*, src/plonk/mod.rs:441-447), leaving
 * a device copy behind for the prover calls that follow (VanillaFS::prove / ProtoGalaxy::prove take it with
 * SRS_SPACE_DEVICE).  The vector goes up in a few chunks on an internal copy stream and the MSM of chunk j runs while chunk
 * j+1 is still on the bus; the partial sums are added on the host.  dev_copy: n elements of HBM (srs_dev_alloc) or NULL
 * (library staging, no copy kept).  Same result and errors as srs_commit.  On a key sharded over processes (srs_ck_create_sharded)
 * only THIS rank's stripes of the vector go up (n * 32 / world bytes) and only they are written in dev_copy -- the rank's partial
 * MSM, its rows of the cross terms and its tiles of the ProtoGalaxy leaves read nothing else. */
int srs_commit_upload(srs_ck *ck, const srs_fe *scalars_host, size_t n, srs_fe *dev_copy, int repr, void *stream,
                      srs_affine *out);

/* util::concatenate_with_padding (src/util/mod.rs:214-218) into HBM: out = for every column its `lens[c]` elements followed by
 * zeros up to `pad_size` (a column longer than the pad is kept whole), columns back to back; srs_concat_len = the length of
 * that vector.  The columns are host vectors (the advice columns CircuitRunner::try_collect_witness leaves behind,
 * src/table/circuit_runner.rs:71-107), the result is device memory: the host never builds the concatenated copy.
 * srs_commit_upload_columns = `ck.commit(&concatenate_with_padding(advice, 2^k))` of run_sps_protocol_* (src/plonk/mod.rs:441-447)
 * in one call: groups of columns go up while the MSM of the previous group runs; dev_copy (srs_concat_len elements, or NULL)
 * receives the assembled witness for the prover calls that follow.  Sharded keys (srs_ck_create_multi / srs_ck_create_sharded) stream
 * per shard: a device is sent only its stripes of the columns and zeroes its stripes of the padding (the sharded form returns the
 * rank's PARTIAL commitment and fills the rank's stripes of dev_copy, as srs_commit_upload does). */
size_t srs_concat_len(const size_t *lens, size_t n_columns, size_t pad_size);
int srs_concat_with_padding(srs_fe *out_dev, const srs_fe *const *columns_host, const size_t *lens, size_t n_columns, size_t pad_size,
                            void *stream);
int srs_commit_upload_columns(srs_ck *ck, const srs_fe *const *columns_host, const size_t *lens, size_t n_columns, size_t pad_size,
                              srs_fe *dev_copy, int repr, void *stream, srs_affine *out);

/* ---- one process, several GPUs (`srs_ck_create(.., n_devices, ..)`) ----
 * The Rust IVC driver is a single process (src/ivc/sangria/incrementally_verifiable_computation.rs:429), so the library
 * itself spreads a key over `n_devices` GPUs (0 = all visible): device d keeps the block-cyclic stripes s % n_devices == d
 * of the window table, every commit PARTITIONS the scalars the same way (device d receives only its stripes: n * 32 /
 * n_devices bytes over its own link, straight from the caller's buffer), the n_devices partial sums land in page-locked
 * host memory and are added there -- the caller sees an ordinary key: srs_commit / srs_commit_batch / srs_commit_upload
 * return the full commitment.  More logical shards than physical devices are allowed (shard d runs on device d % count):
 * that is how the path is tested on a one-GPU box.  Scalars with space = SRS_SPACE_DEVICE must live on the process's
 * device (shard 0's); the other shards fetch their stripes with peer copies. */
int srs_ck_create_multi(int curve, const srs_affine *bases, size_t len, int space, int n_devices, srs_ck **out);
int srs_ck_setup_synthetic_multi(int curve, size_t len, uint64_t seed, int n_devices, srs_ck **out);
int srs_ck_num_shards(const srs_ck *ck);      /* 1 for an ordinary key */
/* Diagnostics of the MSM engine behind `ck` (no reference counterpart; tests and tuning): out[0] = sets of launches that ran in
 * slot mode (persistent per-bucket partial sums, csrc/msm.h), out[1] = those that met hot buckets (parts beyond the slots),
 * out[2] = commits that were run a second time because such parts had not been expected, out[3] = sets on the other flows.
 * Multi-device keys report the sum over their shards. */
int srs_ck_msm_stats(const srs_ck *ck, uint64_t *out4);
/* 1 when the key also holds the second, 13-window table of the 20-bit pipeline (keys of >= 2^23 bases: + 81 % key memory; whole
 * device-resident MSMs of >= 2^23 scalars use it), 0 when it does not -- by size, by SRS_MSM_WIDE=0, or because the device could not hold
 * it (the key then works on the 16-bit windows alone).  Multi-device keys: 1 when every shard holds it. */
int srs_ck_has_wide_table(const srs_ck *ck);
/* Diagnostics of ONE shard of a multi-device key (shard < srs_ck_num_shards): out[0] = bytes its streamed commits brought up from host
 * memory over the shard's own link (srs_commit_upload on a multi-device key sends every shard ITS block-cyclic stripes only: n * 32 /
 * shards bytes per commit and link, in chunks that overlap the shard's MSM), out[1] = bytes it forwarded to the process's device to
 * assemble the device copy the caller asked for (peer copies; 0 for the shard that lives there), out[2] = streamed commits it took part
 * in, out[3] = the HIP device ordinal it runs on.  A single-device key has one shard with zeros.  Lets a one-GPU box (logical shards)
 * assert the traffic a multi-GPU node would see. */
int srs_ck_shard_stats(const srs_ck *ck, int shard, uint64_t *out4);

/* out = sum of `n` affine points (host); combines per-rank partial commitments. */
int srs_point_sum(int curve, const srs_affine *points, size_t n, srs_affine *out);
/* out = [scalar] P  (the 1-element best_multiexp of src/nifs/sangria/accumulator.rs:213,243), host. */
int srs_point_mul(int curve, const srs_fe *scalar, int repr, const srs_affine *p, srs_affine *out);
/* out = acc + sum_{i<n} scalars[i] * points[i] on the host (acc may be NULL = identity): the group half of
 * RelaxedPlonkInstance::fold -- W_commitments (n = 1) and E_commitment (n = d) -- src/nifs/sangria/accumulator.rs:201-264. */
int srs_point_lincomb(int curve, const srs_affine *acc, const srs_affine *points, const srs_fe *scalars, size_t n,
                      int repr, srs_affine *out);
/* out[i] = r^(i+1), i < n (Montgomery in, Montgomery out): the powers of the folding challenge the instance fold multiplies the
 * cross-term commitments with (src/nifs/sangria/accumulator.rs:240-244).  Host code. */
int srs_fe_powers(int field, const srs_fe *r, size_t n, srs_fe *out);
/* The same fold, off the caller's thread: nothing on the device waits for the folded instance, so a prover enqueues its next
 * commitment while this runs on the library's host workers (jobs complete in submission order).  acc / points / scalars are
 * copied before the call returns; `out` is written by the job and must stay valid until srs_job_wait(*job) returns.
 * srs_job_wait: 0 once the job has finished (its output is then visible to the caller); SRS_ERR_INVALID for an unknown or
 * already waited-for job. */
int srs_point_lincomb_async(int curve, const srs_affine *acc, const srs_affine *points, const srs_fe *scalars, size_t n,
                            int repr, srs_affine *out, uint64_t *job);
int srs_job_wait(uint64_t job);

/* ---- per-kernel timing (HIP events on the launch stream), used by bench.py's roofline leg ----
 * names: "msm_accum0" (units = scalars), "rowprog_cross_terms" (rows), "rowprog_eval" (rows), "ntt_transform" (elements) */
void srs_profile_enable(int on);
void srs_profile_reset(void);
/* Events on every `every`-th launch of a name only (1 = all, the default): a launch that carries its own event pair costs ~8 us of idle
 * device, which a per-launch measurement of the dominant kernel adds to every step it measures.  srs_profile_get then reports the SAMPLED
 * launches (total_ms, launches and units of the same launches, so units / total_ms stays exact). */
void srs_profile_sampling(unsigned every);
int srs_profile_get(const char *name, double *total_ms, uint64_t *launches, uint64_t *units);

/* ---- fft (src/fft.rs:160-198) ----
 * In place, natural order in and out, `n` elements of bn256::Fr (the only 2-adic field of the cycle).
 *   inverse = 0, coset = 0 : fft        (src/fft.rs:160-165)
 *   inverse = 1, coset = 0 : ifft       (src/fft.rs:168-182)   (includes the * TWO_INV^k)
 *   inverse = 0, coset = 1 : coset_fft  (src/fft.rs:186-190)
 *   inverse = 1, coset = 1 : coset_ifft (src/fft.rs:194-198)
 * n not a power of two -> SRS_ERR_NOT_POW2 (the reference asserts, :161,169; the shim re-panics);
 * log2(n) > F::S = 28 -> SRS_ERR_K_TOO_LARGE (:13).  field must be SRS_FIELD_FR. */
int srs_ntt(int field, srs_fe *a, size_t n, int inverse, int coset, int space, void *stream);
/* `batch` independent transforms of length n, `stride` elements apart (ProtoGalaxy F/G/K vectors). */
int srs_ntt_batch(int field, srs_fe *a, size_t n, size_t stride, size_t batch, int inverse, int coset,
                  int space, void *stream);
/* Tuning knob: bits per Cooley-Tukey digit (4..8, default 8) for n > 2^10; changes the pass
 * structure (and drops cached twiddle plans), never the result.  Returns the value in effect. */
int srs_ntt_set_max_radix_bits(int bits);

/* ---- PlonkStructure slice for the row programs (src/plonk/mod.rs:127-157) ----
 * What the kernels need of `PlonkStructure<F>`: k (rows = 2^k), selectors (bool columns), fixed
 * columns, num_advice_columns and the gate expressions `S.gates`.  Structures with lookup arguments
 * are created with srs_structure_create_lookup below.
 *
 * `gates`: the `Vec<Expression<F>>` (src/polynomial/expression.rs:112-120) serialised as a postfix
 * stream of u64 words, one SRS_EX_END per gate:
 *   SRS_EX_CONST c0 c1 c2 c3 | SRS_EX_POLY index rotation(i64) | SRS_EX_CHALLENGE index |
 *   SRS_EX_NEG | SRS_EX_SUM | SRS_EX_PRODUCT | SRS_EX_SCALED c0 c1 c2 c3 | SRS_EX_END
 * (constants in the same Montgomery layout as srs_fe).  The library then derives, like
 * ConstraintSystemMetainfo::build / CompressedGates::new (src/table/constraint_system_metainfo.rs:81-104,
 * src/plonk/mod.rs:84-107): compress_expression (src/plonk/util.rs:34-56), num_challenges,
 * Expression::homogeneous (src/polynomial/expression.rs:356-429) and its degree d.
 * Query indices out of range -> SRS_ERR_EVAL_INDEX (validated once here, not per row). */
enum { SRS_EX_CONST = 0, SRS_EX_POLY = 1, SRS_EX_CHALLENGE = 2, SRS_EX_NEG = 3, SRS_EX_SUM = 4,
       SRS_EX_PRODUCT = 5, SRS_EX_SCALED = 6, SRS_EX_END = 7 };
int srs_structure_create(int field, uint32_t k, size_t num_selectors, size_t num_fixed, size_t num_advice,
                         const uint8_t *const *selectors, const srs_fe *const *fixed, int space,
                         const uint64_t *gates, size_t gates_words, size_t num_gates, srs_structure **out);
/* PlonkStructure WITH lookup arguments (log-derivative lookups, src/plonk/lookup.rs:72-133):
 *   gates        = S.gates as ConstraintSystemMetainfo::build leaves them: the custom gates FOLLOWED by
 *                  Arguments::to_expressions (vanishing L_i - l_i, T_i - t_i, then the log-derivative lhs/rhs);
 *   lookup_exprs = Arguments::lookup_polys (num_lookups expressions) then Arguments::table_polys (num_lookups),
 *                  same postfix stream; they see the advice COLUMNS only (LookupEvalDomain, src/plonk/eval.rs:106-134)
 *                  and the single challenge r.
 * Each lookup adds the five fold variables (l, t, m, h, g) as query indices after the advice columns; the
 * gate-compression challenge comes after r1 [, r2] (constraint_system_metainfo.rs:81-97).
 * WITNESS LAYOUT for every entry point below that takes W of such a structure: the ROUNDS of PlonkWitness::W
 * CONCATENATED, W[0] || W[1] (|| W[2]) -- (num_advice + 5 * num_lookups) * 2^k elements
 * (srs_structure_num_witness_columns); query index -> column is PlonkEvalDomain::eval_advice_var's index_map
 * (src/plonk/eval.rs:169-201), reproduced literally (for num_lookups > 1 it addresses (l,t,m) / (h,g)
 * interleaved per lookup although run_sps_protocol_* concatenates them grouped -- DESIGN.md quirk Q5). */
int srs_structure_create_lookup(int field, uint32_t k, size_t num_selectors, size_t num_fixed, size_t num_advice,
                                const uint8_t *const *selectors, const srs_fe *const *fixed, int space,
                                const uint64_t *gates, size_t gates_words, size_t num_gates,
                                size_t num_lookups, int has_vector_lookup,
                                const uint64_t *lookup_exprs, size_t lookup_words, srs_structure **out);
void srs_structure_free(srs_structure *S);
/* Multi-GPU (one process per GPU, keys from srs_ck_create_sharded): with a shard set, srs_cross_terms /
 * srs_commit_cross_terms evaluate the cross terms only on the rows of THIS rank's block-cyclic stripes (2^10 rows each, the
 * stripes of the sharded key) -- the only rows the rank's partial commitment and its part of the error fold read.  Rows of
 * T_out outside those stripes read as ZERO afterwards: device buffers are cleared before the rank's rows are written (do not
 * accumulate into a pre-filled T_out), host buffers are returned as zero.  The deciders (srs_eval_gates, srs_is_sat_gates)
 * always cover every row.
 * The ProtoGalaxy sums of a sharded structure (srs_pg_compute_F / _G / srs_pg_evaluate_e) cover the leaves of the rank's tiles
 * only (a 1024-leaf tile of the leaf kernels is a stripe; structures of fewer than 2^10 rows: rank 0 evaluates everything):
 * every rank gets a PARTIAL polynomial / value, the results of all ranks add up (coefficient-wise, e.g. srs_fold_lincomb on
 * host vectors) to the polynomial of the reference.  The whole-prove entries (srs_pg_prove, srs_sangria_prove) refuse sharded
 * handles: the challenges depend on the exchanged sums.
 * WHICH WITNESS ROWS A RANK READS.  With the WHOLE witness resident the sharded calls are correct for any circuit.  After a
 * sharded srs_commit_upload only the rank's key stripes are resident; that suffices only if ALL of these hold:
 *   (1) 2^k / 2^10 is a multiple of `world` and k >= 10 (the row stripes of every column are key stripes of the rank);
 *   (2) every column query of the gates has rotation 0 (a query at (row + rot) leaves the stripe);
 *   (3) reference_compat = 0 (with reference_compat every ProtoGalaxy leaf reads row 0, rank 0's stripe -- src/plonk/mod.rs:714).
 * When (2) or (3) fails call srs_structure_upload_shard_halo after the upload: it brings up exactly the missing rows (the
 * rotation halo around the rank's stripes; row 0 + rotations of every column for reference_compat).  When (1) fails it returns
 * SRS_ERR_INVALID: upload the whole vector instead (srs_upload). */
int srs_structure_set_shard(srs_structure *S, uint32_t rank, uint32_t world);
/* witness_host: the n = srs_structure_num_witness_columns * 2^k elements handed to the sharded srs_commit_upload; dev_copy: its
 * device copy.  No-op for world == 1 and for rotation-free circuits with reference_compat = 0. */
int srs_structure_upload_shard_halo(const srs_structure *S, const srs_fe *witness_host, srs_fe *dev_copy, size_t n, int reference_compat,
                                    void *stream);
size_t srs_structure_num_witness_columns(const srs_structure *S);   /* num_advice + 5 * num_lookups */
size_t srs_structure_num_cross_terms(const srs_structure *S);   /* d = grouped().len() - 1 */
size_t srs_structure_num_challenges(const srs_structure *S);    /* PlonkStructure::num_challenges */
/* Developer hook: the straight-line C++ of a compiled row program (which: 0 cross terms, 1 compressed,
 * 2 homogeneous), its fingerprint and the kernel it runs on: >= 0 ahead-of-time specialised kernel, -1 interpreter,
 * -2 straight-line kernel compiled at structure creation with hiprtc (structures of >= 2^14 rows without an ahead-of-time kernel).
 * Returns the source length (truncated to cap-1).  Used by tools/gen_rowprog_spec.py. */
size_t srs_structure_program_source(srs_structure *S, int which, char *buf, size_t cap, uint64_t *fingerprint, int *spec_id);
/* Which kernel a structure's row programs run on (a failed run-time compilation silently leaves the interpreter in place:
 * this makes it observable).  which: 0 = cross terms, 1 = gates (deciders), 2 = ProtoGalaxy leaves.
 * Returns SRS_KERNEL_INTERPRETER, SRS_KERNEL_AHEAD_OF_TIME or SRS_KERNEL_RUNTIME_COMPILED; -1 for a bad argument. */
enum { SRS_KERNEL_INTERPRETER = 0, SRS_KERNEL_AHEAD_OF_TIME = 1, SRS_KERNEL_RUNTIME_COMPILED = 2 };
int srs_structure_kernel_kind(const srs_structure *S, int which);
/* Developer hook, host only (needs no device): compiles a small row program in the emitted form with hiprtc against the device
 * headers embedded in the library -- the run-time compilation path of srs_structure_create minus the module load.  0 and the
 * size of the gfx950 code object, or SRS_ERR_INVALID with the compiler log (log may be NULL).  A structure whose program fails to
 * compile silently stays on the interpreter, so this is the check that the path is alive. */
int srs_jit_selfcheck(size_t *code_bytes, char *log, size_t log_cap);

/* Evaluation half of VanillaFS::commit_cross_terms (src/nifs/sangria/mod.rs:102-148):
 *   T_out[k-1][row] = coefficient of X^k in P_homogeneous(fixed, W1 + X*W2, ch1 + X*ch2)[row],  k = 1..d
 * W1, W2: witness vectors, column-major num_advice * 2^k (PlonkWitness::W[0]; all rounds concatenated with lookups);
 * challenges = U1.challenges || U1.u || U2.challenges || 1  (src/nifs/sangria/mod.rs:113-118);
 * a challenge index outside that vector -> SRS_ERR_EVAL_INDEX (ChallengeIndexOutOfBoundary). */
int srs_cross_terms(srs_structure *S, const srs_fe *W1, const srs_fe *W2, const srs_fe *challenges,
                    size_t n_challenges, int space, void *stream, srs_fe *const *T_out);
/* Whole commit_cross_terms: the evaluation above, then commits_out[k-1] = ck.commit(T_k) as one
 * batched MSM over the shared base prefix (src/nifs/sangria/mod.rs:150-155).  T_out may be NULL
 * when the caller does not need the vectors themselves (they stay on the device otherwise). */
int srs_commit_cross_terms(srs_structure *S, srs_ck *ck, const srs_fe *W1, const srs_fe *W2,
                           const srs_fe *challenges, size_t n_challenges, int space, void *stream,
                           srs_fe *const *T_out, srs_affine *commits_out);
/* Per-row gate value used by the deciders: homogeneous = 0 -> compressed gate with challenges =
 * U.challenges (PlonkStructure::is_sat, src/plonk/mod.rs:304-361); homogeneous = 1 -> homogeneous
 * gate with challenges = U.challenges || U.u (is_sat_accumulation, src/nifs/sangria/mod.rs:334-383). */
int srs_eval_gates(srs_structure *S, int homogeneous, const srs_fe *W, const srs_fe *challenges,
                   size_t n_challenges, int space, void *stream, srs_fe *out);

/* Deciders' gate check: number of rows where the gate value differs from the expected one.
 * homogeneous = 0: compressed gate vs 0   (PlonkStructure::is_sat, src/plonk/mod.rs:329-346; E must be NULL)
 * homogeneous = 1: homogeneous gate vs E[row] (is_sat_accumulation, src/nifs/sangria/mod.rs:352-376)
 * *mismatch_count == 0 <=> satisfied; the reference reports EvaluationMismatch { mismatch_count, total_row }. */
int srs_is_sat_gates(srs_structure *S, int homogeneous, const srs_fe *W, const srs_fe *challenges, size_t n_challenges,
                     const srs_fe *E, int space, void *stream, size_t *mismatch_count);

/* ---- off-circuit Poseidon random oracle (src/poseidon/poseidon_hash.rs:16-237) -- host code, no device ----
 * srs_poseidon_new        = PoseidonHash::new(Spec::new(r_f, r_p)) over `field` (ROConstantsTrait::new, :99-106); RATE == T - 1.
 *                           Constants: the construction of the PSE `poseidon` crate the reference depends on (Grain LFSR,
 *                           Cauchy MDS), pinned by the reference's known answer poseidon_hash.rs:248-266 (tests/test_poseidon.py).
 * srs_poseidon_absorb_field / _absorb_point = ROTrait::absorb_field / absorb_point (:118-141): a point contributes (x, y),
 *                           the identity (0, 0); the oracle's field must be the curve's base field.
 * srs_poseidon_squeeze     = ROTrait::squeeze::<D>(num_bits) (:149-152, output :190-212): the low num_bits bits of state[1] as
 *                           an element of `out_field` (Montgomery).  As in the reference the absorbed buffer is KEPT. */
typedef struct srs_poseidon srs_poseidon;
int srs_poseidon_new(int field, size_t t, size_t rate, size_t r_f, size_t r_p, srs_poseidon **out);
void srs_poseidon_free(srs_poseidon *H);
void srs_poseidon_reset(srs_poseidon *H);   /* forget everything absorbed (a fresh oracle with the same constants) */
int srs_poseidon_absorb_field(srs_poseidon *H, const srs_fe *v, size_t n);
int srs_poseidon_absorb_point(srs_poseidon *H, int curve, const srs_affine *p);
int srs_poseidon_squeeze(srs_poseidon *H, size_t num_bits, int out_field, srs_fe *out);
/* The same squeeze with the sponge run ON THE DEVICE (one wavefront: lanes = state elements / MDS products, the 9 x 29-bit
 * multiplier).  Same value as srs_poseidon_squeeze; T <= 8.  A permutation is a chain of dependent products, so this is SLOWER
 * than the host code (measured: DESIGN.md 4.8, profiles/r03_poseidon_device_vs_host.txt) -- it exists so that the comparison is
 * a measurement; the proves use the host sponge.  *kernel_ms (optional) receives the kernel's HIP-event time. */
int srs_poseidon_squeeze_device(srs_poseidon *H, size_t num_bits, int out_field, srs_fe *out, double *kernel_ms);

/* ---- deciders: permutation (copy-constraint) check and witness-commitment check ----
 * srs_sparse = the reference's SparseMatrix<F> = Vec<(row, col, value)> (src/polynomial/sparse.rs:5) of an n x n matrix,
 * kept on the device in CSR form.  A column index >= n is the reference's panic "invalid matrix multiply" (:15-17) and is
 * rejected at creation (SRS_ERR_INVALID), as is a row index >= n.
 * srs_sparse_matvec        = sparse::matrix_multiply (sparse.rs:7-19): y = M * Z.
 * srs_is_sat_permutation   = the counting half of VanillaFS::is_sat_permutation (src/nifs/sangria/mod.rs:425-452) /
 *                            PlonkStructure permutation check: number of rows with (M * Z)[row] != Z[row]; the caller builds
 *                            Z = instances (with padding) || W[0][.. num_advice * 2^k]  (:415-423) and M = permutation matrix.
 * srs_is_sat_witness_commit = VanillaFS::is_sat_witness_commit (src/nifs/sangria/mod.rs:455-474; src/plonk/mod.rs:350-358 with
 *                            E == NULL): *w_mismatch_count = #{ i : ck.commit(W[i]) != W_commitments[i] } over the n_rounds
 *                            witness rounds (CommitmentMismatch), *e_mismatch = (ck.commit(E) != *E_commitment) (ECommitmentMismatch).
 *                            One batched MSM.  A round longer than the key -> SRS_ERR_TOO_LONG_INPUT (the reference unwrap()s). */
typedef struct srs_sparse srs_sparse;
int srs_sparse_create(int field, size_t n, const uint64_t *rows, const uint64_t *cols, const srs_fe *values, size_t nnz,
                      srs_sparse **out);
void srs_sparse_free(srs_sparse *M);
int srs_sparse_matvec(srs_sparse *M, const srs_fe *Z, int space, void *stream, srs_fe *y);
int srs_is_sat_permutation(srs_sparse *M, const srs_fe *Z, int space, void *stream, size_t *mismatch_count);
int srs_is_sat_witness_commit(srs_ck *ck, const srs_fe *const *W, const size_t *n, size_t n_rounds,
                              const srs_affine *W_commitments, const srs_fe *E, size_t n_E, const srs_affine *E_commitment,
                              int space, void *stream, size_t *w_mismatch_count, int *e_mismatch);

/* ---- lookup arguments: prover coefficients and the decider's log-derivative check ----
 * srs_lookup_coeff_1 = Arguments::evaluate_coefficient_1 (src/plonk/lookup.rs:319-341): for every lookup i
 *   ls[i][row] = L_i(advice, r)[row], ts[i][row] = T_i(fixed, r)[row]               (evaluate_ls / _ts, :209-272)
 *   ms[i][row] = #{ j : ls[i][j] == ts[i][row] } on the first row holding that table value, 0 on repeats (:275-303)
 *   advice: the advice columns, column-major num_advice * 2^k; ls / ts / ms: arrays of num_lookups vectors of 2^k.
 * srs_lookup_coeff_2 = Arguments::evaluate_h_g (:305-317) for ONE lookup: h = 1 / (l + r), g = m / (t + r), 1/0 := 0.
 * srs_is_sat_log_derivative = PlonkStructure::is_sat_log_derivative (src/plonk/mod.rs:366-398): number of lookups
 *   whose sum_row (h_i - g_i) is non-zero (0 <=> satisfied; the reference returns LogDerivativeNotSat otherwise);
 *   W is the concatenated witness described at srs_structure_create_lookup. */
int srs_lookup_coeff_1(srs_structure *S, const srs_fe *advice, const srs_fe *r, int space, void *stream,
                       srs_fe *const *ls, srs_fe *const *ts, srs_fe *const *ms);
int srs_lookup_coeff_2(int field, const srs_fe *l, const srs_fe *t, const srs_fe *m, const srs_fe *r, size_t n,
                       int space, void *stream, srs_fe *h, srs_fe *g);
int srs_is_sat_log_derivative(srs_structure *S, const srs_fe *W, int space, void *stream, size_t *mismatch_count);
/* batch_invert_assigned (src/util/mod.rs:119-153) for ONE column of halo2 `Assigned<F>` cells flattened by the caller:
 *   Zero -> (0, no denominator), Trivial(x) -> (x, no denominator), Rational(n, d) -> (n, d)
 * out[i] = numerators[i] * (has_denominator[i] ? denominators[i]^-1 : 1), a zero denominator giving 0 (ff::BatchInvert skips
 * zeros).  has_denominator == NULL: every cell is Rational.  One Fermat inversion per 8 cells (Montgomery's trick). */
int srs_batch_invert_assigned(int field, const srs_fe *numerators, const srs_fe *denominators, const uint8_t *has_denominator,
                              size_t n, int space, void *stream, srs_fe *out);

/* ---- RelaxedPlonkWitness::fold (src/nifs/sangria/accumulator.rs:364-404) ----
 * srs_fold_witness: out[i] = w1[i] + r * w2[i]                                   (:366-376)
 * srs_fold_error  : out[i] = e[i] + sum_{k<n_terms} r^(k+1) * T[k][i]            (:380-398)
 * out may alias w1 / e.  With SRS_SPACE_DEVICE operands both calls are STREAM-ORDERED: they return once the kernel is
 * enqueued on `stream` and the result is valid in stream order (host operands: blocking, as everywhere else). */
int srs_fold_witness(int field, srs_fe *out, const srs_fe *w1, const srs_fe *w2, const srs_fe *r, size_t n,
                     int space, void *stream);
int srs_fold_error(int field, srs_fe *out, const srs_fe *e, const srs_fe *const *T, size_t n_terms,
                   const srs_fe *r, size_t n, int space, void *stream);

/* VanillaFS::prove (src/nifs/sangria/mod.rs:253-277) as ONE call on device-resident traces: commit_cross_terms (T_dev = d device
 * vectors of 2^k elements, kept in HBM), the challenge -- with a random oracle `ro` (over the curve's base field, already holding
 * pp_digest, U1, U2) r = ro.absorb_point_iter(commits).squeeze(128) (:162-179) is derived inside and returned in *r_io, with
 * ro = NULL *r_io is used as given --, RelaxedPlonkWitness::fold IN PLACE (W1 <- W1 + r W2, E <- E + sum r^k T_k,
 * accumulator.rs:364-404) and the group half of RelaxedPlonkInstance::fold on the library's host workers:
 * folded_commitments[0] = W_commitments[0] + r W_commitments[1], folded_commitments[1] = E_commitment + sum r^k commits[k]
 * (accumulator.rs:201-264), valid after srs_job_wait(jobs[0]) / srs_job_wait(jobs[1]).  `challenges` as in srs_commit_cross_terms.
 * Failure: every argument check happens before anything is written; a failure after that point is a device failure
 * (SRS_ERR_DEVICE), after which W1 / E may be partly folded and must be considered lost.  No job is left running on an error
 * return (a job already queued is waited for). */
int srs_sangria_prove(srs_structure *S, srs_ck *ck, srs_poseidon *ro, const srs_fe *challenges, size_t n_challenges, srs_fe *W1, const srs_fe *W2,
                      srs_fe *E, void *stream, srs_fe *r_io, srs_fe *const *T_dev, srs_affine *cross_term_commits,
                      const srs_affine *W_commitments, const srs_affine *E_commitment, srs_affine *folded_commitments, uint64_t *jobs);

/* The same prove for an incoming trace that has just been synthesised and is NOT committed yet -- CyclefoldIVC::next's support
 * circuit (src/ivc/cyclefold/incrementally_verifiable_computation/mod.rs:255-300: the trace is generated, committed by
 * run_sps_protocol (src/plonk/mod.rs:441-447) and folded at once).  The reference commits it and then, inside prove, the cross
 * terms: two multi-exponentiations one after the other.  For a structure WITHOUT challenges (num_challenges == 0: one gate, no
 * lookups -- the support circuit, the Sangria secondary) neither depends on the other's result, so here they are ONE batched MSM
 * over d+1 vectors (W2 || T_0..T_{d-1}).  A structure with challenges is refused (SRS_ERR_INVALID): run_sps_protocol_1
 * (src/plonk/mod.rs:465-495) squeezes U2's challenges from a transcript that has absorbed the trace's commitment, so the commitment
 * must exist before the cross terms can be evaluated -- commit first, then srs_sangria_prove.
 * W2_host (may be NULL when W2 already holds the trace) is uploaded into the device
 * vector W2 on `stream`, the cross terms are evaluated, all d+1 commitments come out of one chain of launches.
 * W_commitments[0] = U1's commitment (in), W_commitments[1] = the incoming trace's commitment (OUT).  Transcript: `ro` holds
 * pp_digest and U1; the call absorbs W_commitments[1], then u2_tail[0..n_u2_tail) (the rest of U2 -- its instances and
 * challenges -- as elements of the oracle's field, Montgomery form), then the cross-term commitments, and squeezes r: the same
 * absorb order as generate_challenge (sangria/mod.rs:162-179).  Everything else as srs_sangria_prove. */
int srs_sangria_prove_incoming(srs_structure *S, srs_ck *ck, srs_poseidon *ro, const srs_fe *challenges, size_t n_challenges, srs_fe *W1, srs_fe *W2,
                               const srs_fe *W2_host, const srs_fe *u2_tail, size_t n_u2_tail, srs_fe *E, void *stream, srs_fe *r_io,
                               srs_fe *const *T_dev, srs_affine *cross_term_commits, srs_affine *W_commitments, const srs_affine *E_commitment,
                               srs_affine *folded_commitments, uint64_t *jobs);

/* ---- ProtoGalaxy NIFS polynomials (src/nifs/protogalaxy/poly/mod.rs), bn256::Fr structures only ----
 * Leaves f_i = S.gates[i / 2^k] at row(i) (get_evaluate_witness_fn, src/plonk/mod.rs:683-718), i < n =
 * (gates * 2^k).next_power_of_two(), zero beyond gates * 2^k; pow_i(c) = prod_{b in bits(i)} c_b.
 * `reference_compat` != 0 reproduces the reference bit for bit INCLUDING its row-index quirk
 * `row_index = index & total_row` (src/plonk/mod.rs:714), i.e. every leaf is evaluated at row 0;
 * 0 evaluates the mathematically intended row `index % 2^k`.  Callers that must match the
 * reference (proof transcripts!) pass 1.
 * Witnesses W are round-0 vectors (num_advice * 2^k, column-major); `challenges` the trace's challenges. */
typedef struct {                       /* PolyContext (poly/mod.rs:205-269) */
    size_t count_of_evaluation_with_padding;
    size_t betas_count;                /* log2 of the above */
    size_t fft_points_count_F;         /* (betas_count + 1).next_power_of_two() */
    size_t fft_points_count_G;         /* (traces_len * max_gate_degree + 1).next_power_of_two() */
    size_t instances_to_fold;          /* traces_len + 1 */
    size_t lagrange_domain;            /* log2(instances_to_fold) */
    uint32_t fft_log_domain_size_K;    /* (points_G + 1 - instances_to_fold).next_power_of_two() -- a count used as a log (:263-268) */
} srs_pg_context;
int srs_pg_context_new(const srs_structure *S, size_t traces_len, srs_pg_context *out);
/* ProtoGalaxy::prove (src/nifs/protogalaxy/mod.rs:400-481) as ONE call on device-resident witnesses W[0] = accumulator, W[1..] =
 * incoming traces: compute_F -> alpha -> betas_stroke -> compute_G -> compute_K_from_G -> gamma -> L_j(gamma), calculate_e,
 * fold_witness.  `delta` comes from the caller (Challenges::generate_one absorbs instance data the shim owns, :409-414); with a
 * random oracle `ro` (over bn256::Fr, already holding that transcript) alpha = ro.absorb(poly_F).squeeze(MAX_BITS = 255) and
 * gamma = ro.absorb(poly_K).squeeze(255) are derived inside (:424-427,445-448) and returned in alpha_gamma[0..1]; with ro = NULL
 * alpha_gamma holds them on entry.  Outputs: poly_F[fft_points_count_F], poly_K[2^fft_log_domain_size_K], betas_stroke[betas_count],
 * e, lagrange[n_instances] = L_j(gamma) (for fold_instance, which stays with the caller), W_folded (device, stream-ordered).
 * W_folded = NULL leaves fold_witness to the caller (srs_fold_lincomb with `lagrange`): nothing in the rest of an IVC step reads
 * the folded witness before the next prove, so the 3-vector pass can be queued where the device would otherwise wait for the next
 * witness's upload. */
int srs_pg_prove(srs_structure *S, srs_poseidon *ro, const srs_fe *betas, size_t n_betas, const srs_fe *delta,
                 const srs_fe *const *W, const srs_fe *const *challenges, size_t n_challenges, size_t n_instances, int reference_compat,
                 void *stream, srs_fe *alpha_gamma, srs_fe *poly_F, srs_fe *poly_K, srs_fe *betas_stroke, srs_fe *e, srs_fe *lagrange,
                 srs_fe *W_folded);
/* PolyChallenges::iter_beta_stroke (poly/mod.rs:432-462): out[i] = betas[i] + alpha * delta^(2^i), i < n.  Host code (bn256::Fr). */
int srs_pg_beta_stroke(const srs_fe *betas, size_t n, const srs_fe *alpha, const srs_fe *delta, srs_fe *out);
/* compute_F (:68-203): poly_F[fft_points_count_F] = ifft_X( sum_i pow_i(betas + X*deltas) f_i(w) ), deltas_b = delta^(2^b) */
int srs_pg_compute_F(srs_structure *S, const srs_fe *betas, size_t n_betas, const srs_fe *delta, const srs_fe *W,
                     const srs_fe *challenges, size_t n_challenges, int space, int reference_compat, void *stream,
                     srs_fe *poly_F);
/* compute_G (:308-425) with FoldedWitness (folded_witness.rs:20-180) fused (never materialised):
 * poly_G[fft_points_count_G] = ifft_X( sum_i pow_i(betas_stroke) f_i( sum_j L_j(X) w_j ) ); W[0] / challenges[0] = accumulator,
 * W[1..] = incoming traces; n_instances = L + 1 (a power of two, <= 16: `const L` is generic in the reference). */
int srs_pg_compute_G(srs_structure *S, const srs_fe *betas_stroke, size_t n_betas, const srs_fe *const *W,
                     const srs_fe *const *challenges, size_t n_challenges, size_t n_instances, int space,
                     int reference_compat, void *stream, srs_fe *poly_G);
/* compute_K_from_G (:475-509): K = (G - F(alpha) L_0) / Z evaluated on ZETA * <w> of size 2^fft_log_domain_size_K, then coset_ifft */
int srs_pg_compute_K_from_G(const srs_fe *poly_G, size_t n_G, const srs_fe *poly_F_in_alpha, size_t instances_to_fold,
                            uint32_t fft_log_domain_size_K, void *stream, srs_fe *poly_K);
/* evaluate_e_from_trace (src/nifs/protogalaxy/mod.rs:571-640): e = sum_i pow_i(betas) f_i(w) */
int srs_pg_evaluate_e(srs_structure *S, const srs_fe *betas, size_t n_betas, const srs_fe *W, const srs_fe *challenges,
                      size_t n_challenges, int space, int reference_compat, void *stream, srs_fe *e);
/* calculate_e (protogalaxy/mod.rs:748-764): F(alpha) * L_0(gamma) + Z(gamma) * K(gamma); host */
int srs_pg_calculate_e(const srs_fe *poly_F, size_t n_F, const srs_fe *poly_K, size_t n_K, const srs_fe *gamma,
                       const srs_fe *alpha, uint32_t log_n, srs_fe *out);
/* iter_eval_lagrange_poly_for_cyclic_group (src/polynomial/lagrange.rs:50-75): out[i] = L_i(X), i < 2^log_n; host */
int srs_lagrange_eval(const srs_fe *X, uint32_t log_n, srs_fe *out);
/* UnivariatePoly::eval (src/polynomial/univariate.rs:67-75); host */
int srs_poly_eval(const srs_fe *coeffs, size_t n, const srs_fe *x, srs_fe *out);
/* ProtoGalaxy::fold_witness (protogalaxy/mod.rs:176-210): out[i] = sum_{j<J} coefs[j] * W[j][i]  (J <= 16) */
int srs_fold_lincomb(int field, srs_fe *out, const srs_fe *const *W, const srs_fe *coefs, size_t J, size_t n,
                     int space, void *stream);
/* The same fold on a process-per-GPU rank: DEVICE vectors; only the elements of the rank's block-cyclic stripes (2^10 elements
 * each, stripe s belongs to rank s % world -- the key's stripes) of out[0 .. n) are computed and written, everything else is
 * left as it is.  Enough ONLY when the rank's kernels read nothing else of the folded vector: gates without rotated queries and
 * leaves at their own rows.  For the accumulator of a row-sharded structure use srs_structure_fold_sharded, which also folds
 * the halo rows (rotations; row 0 under reference_compat). */
int srs_fold_lincomb_sharded(int field, srs_fe *out, const srs_fe *const *W, const srs_fe *coefs, size_t J, size_t n, uint32_t rank,
                             uint32_t world, void *stream);
/* ProtoGalaxy::fold_witness (protogalaxy/mod.rs:176-210) / RelaxedPlonkWitness::fold for the witness vectors of a ROW-SHARDED
 * structure (srs_structure_set_shard; W[j] = num_witness_columns * 2^k elements each, DEVICE): folds the rank's stripes and the
 * rows its leaf / cross-term kernels read beyond them -- the rotation halo of every stripe and, with reference_compat, row 0 of
 * every column (+ rotations): exactly the rows srs_structure_upload_shard_halo brings up for the incoming trace, so that after
 * any number of folds every row the rank reads of the accumulator is the reference's.  out may be W[0].  Unsharded: the whole vector. */
int srs_structure_fold_sharded(srs_structure *S, srs_fe *out, const srs_fe *const *W, const srs_fe *coefs, size_t J, int reference_compat,
                               void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SIRIUS_AMD_H */
