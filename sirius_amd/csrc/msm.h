// msm.h -- internal interface of the MSM engine (see msm.hip for the design notes).
#pragma once
#include "curve.cuh"
#include "devrt.h"
#include "tuning.h"

#include <cstdlib>

namespace srs {
namespace msm {

constexpr int WBITS = 16;                 // signed-digit window width
constexpr int NWIN = 16;                  // ceil(256 / WBITS); scalars are < 2^254
constexpr uint32_t NBUCKET = 1u << (WBITS - 1);   // |digit| in 1..2^15 -> bucket |digit|-1
constexpr uint32_t STRIPE_LOG = 10;       // multi-GPU block-cyclic stripe (entries)

constexpr uint32_t SORT_THREADS = 1024;   // one workgroup per CU: 128 KiB LDS histogram
constexpr uint32_t SORT_TILE_MIN = 8192;       // digits per workgroup (lower bound)
constexpr uint32_t SORT_TARGET_BLOCKS = 256;   // ~one 128-KiB-LDS workgroup per CU
constexpr uint32_t SEG = 128, SEG_BUCKETS = NBUCKET / SEG;   // two-pass scatter: segments of 256 consecutive buckets
constexpr uint32_t SORT_TILE2 = 8192;           // entries per workgroup of the second pass (8 per thread, sorted inside LDS)
constexpr uint64_t TWO_PASS_MIN_SLOTS = 1ull << 23;   // digit slots (16 n x batch) from which the two-pass scatter wins (r03, LDS-staged passes: from ~0.5 M scalars;
                                                      // profiles/r03_ab_two_pass_sort.txt -- was 2^28 with lane-per-entry stores)
constexpr uint32_t PLAN_THREADS = 1024;
constexpr uint32_t ACC_THREADS = 256;        // r03 A/B (profiles/r03_ab_accum0_variants.txt): 256 beats 128 and 64 by 2-3 % of a k = 20 step
constexpr uint32_t ACC_L0_LOG = 4, ACC_L0 = 1u << ACC_L0_LOG;   // gathered mixed adds per level-0 thread
constexpr uint32_t ACC_L1_LOG = 3, ACC_L1 = 1u << ACC_L1_LOG;   // full adds per thread on later levels
constexpr uint32_t FINAL_THREADS = 256;   // 4 wavefronts = 4 buckets per workgroup
constexpr uint32_t FINAL_FANIN = 512;     // worst-case parts per bucket left for the wave-level pass
constexpr int MAX_LEVELS = 8;
constexpr uint32_t RED_ROWS = 256;        // bucket index = hi * RED_COLS + lo
constexpr uint32_t RED_COLS = NBUCKET / RED_ROWS;
constexpr uint32_t RED_THREADS = 4 * RED_COLS;   // k_reduce_final: one quad per element, 128 elements
constexpr uint32_t ACC1_QUAD_MAX = 1u << 17;     // k_accum1 runs one quad per output when a level has at most this many outputs (x batch); r04: 2^16 -> 2^17 (the support circuit's 3 x 2^15 outputs: 138 -> 109 us)
constexpr uint32_t BATCH_ARGS = 16;    // MSMs per set of launches (batch descriptor = kernel argument); larger batches are chunked
constexpr uint32_t LANDING_SLOTS = 16;  // sets of launches whose results may be in flight at once (chunked commits)
constexpr uint32_t NORM_G = 16;           // points per inversion in the key-expansion normalise
// Slot mode (r04): every bucket owns 2^SLOT_LOG persistent partial sums ("slots") in HBM.  Part p of bucket b adds its entries INTO
// slot (b, p) -- across all the chunks of a streamed commit -- so the accumulation levels, the wave-level pass and the bucket fold that
// every chunk used to run are replaced by ONE reduction of the slots per commit.  The last slot of a bucket takes the (rare) parts
// beyond 2^SLOT_LOG - 1, which go through the level kernels as before.  tuning msm_slot_log = 2..8 overrides (tests force overflow).
constexpr uint32_t SLOT_LOG = 6;
constexpr uint32_t SLOT_L0_MIN_LOG = 2, SLOT_L0_MAX_LOG = 16;   // part length 2^l0, chosen on the device from the mean bucket load (k_plan_s)
// Wide windows for large MSMs: 13 signed 20-bit digits per scalar instead of 16 signed 16-bit ones (19 % fewer bucket additions).
// The 2^19 buckets are NSEG_W "virtual MSMs" of NBUCKET buckets each (bucket = seg * NBUCKET + lo): one MSD pass groups the
// entries by segment, then every stage of the 16-bit pipeline runs on the 16 segments as a batch.
constexpr int WBITS_W = 20;
constexpr int NWIN_W = 13;                // ceil(254 / 20); the top digit has 14 bits + carry
constexpr uint32_t NSEG_W = 16;           // 2^(WBITS_W - 1) / NBUCKET
constexpr uint32_t WIDE_THREADS = 256, WIDE_PER = 2, WIDE_TILE = WIDE_THREADS * WIDE_PER;   // scalars per workgroup of the segment passes
// Measured (profiles/r02_wide_windows.txt): the wide pipeline wins from ~8 M scalars (12 * 2^20 uniform: 20.3 vs 22.2 ms);
// below, its 16 bucket reductions and the extra grouping pass cost more than the 3 / 16 of the additions it saves.
constexpr uint32_t WIDE_MIN_KEY_LOG = 23; // keys from 2^23 bases get the second, 13-window table (r04, see wants_wide_table in msm.hip; tuning msm_wide / environment SRS_MSM_WIDE = 0 / 1)
constexpr uint32_t WIDE_MIN_N_LOG = 23;   // whole device-resident MSMs from 2^23 scalars take the wide path (tuning msm_wide_min = <log2> overrides)

static_assert(RED_ROWS == 256 && RED_COLS == 128, "k_rowcol lane layout");
static_assert(NBUCKET % PLAN_THREADS == 0, "k_plan tiling");

// a new key's hot-bucket prediction: expected (tuning msm_expect_ovf = 0: not expected -- the tests of the redo path start cold)
inline bool expect_ovf_initial() { return tuning::get_or(tuning::MSM_EXPECT_OVF, 1) != 0; }

// Device-resident commitment key: window-expanded table T[w * len + i] = 2^(16 w) P_i, coordinates in the
// R' = 2^261 Montgomery form of the 9 x 29-bit multiplier (field29.cuh) once build_table has run.
struct Key {
    int curve = 0;            // 0 bn256 G1, 1 grumpkin
    size_t len = 0;           // number of bases held by THIS rank
    size_t global_len = 0;    // length of the whole key (== len when world == 1)
    uint32_t rank = 0, world = 1;
    bool compact_scalars = false;   // world > 1: the scalar vectors handed to run() hold ONLY this rank's stripes, gathered (multi-device keys)
    affine_t *table = nullptr;
    affine_t *table_w = nullptr;   // T_w[w][i] = 2^(20 w) P_i, w < NWIN_W (keys of >= 2^WIDE_MIN_KEY_LOG bases; owned by the key: release())
    xyzz_t *fold_buckets = nullptr;   // running bucket sums of a chunked commit (enqueue(.., fold)); owned by the key
    bool slot_wide[LANDING_SLOTS] = {};       // landing slot -> which pipeline produced it (finish() combines 3 or 4 partial sums)
    // slot mode (see SLOT_LOG): owned by the key, grow-only
    xyzz_t *slots = nullptr;          // [batch][NBUCKET][S] persistent partial sums of the running commit
    size_t slots_pts = 0;
    uint8_t *used = nullptr;          // [2][BATCH_ARGS][NBUCKET]: slots of a bucket that hold a sum (parity = set index inside the commit)
    uint32_t *h_ovf = nullptr;        // page-locked [2][LANDING_SLOTS][BATCH_ARGS]: parts beyond the slots, then non-zero digits, reported by k_plan_s
    uint64_t last_entries = 0;        // non-zero digits (= bucket additions) of the last commit that ran in slot mode (note_commit) ...
    uint64_t last_scalars = 0;        // ... and its scalars (both set by note_commit, both 0 when the commit had no slot-mode set): the density the next streamed commit's chunk cuts are chosen for
    uint32_t slot_s = 0;              // S of the running commit
    uint32_t seq = 0;                 // sets enqueued in the running commit
    bool commit_ovf = false;          // the running commit launches the overflow kernels
    bool expect_ovf = expect_ovf_initial();   // prediction for the next commit: hot buckets seen in one of the last few commits (note_commit); a new key expects them
    uint32_t cold_streak = 0;         // commits in a row without hot buckets while they were expected
    bool slot_mode[LANDING_SLOTS] = {};     // landing slot -> the set ran in slot mode ...
    bool slot_ovf_on[LANDING_SLOTS] = {};   // ... with its overflow kernels launched
    uint32_t slot_batch[LANDING_SLOTS] = {};
    uint64_t stat_slot_sets = 0, stat_hot_sets = 0, stat_redo = 0, stat_other_sets = 0;   // srs_ck_msm_stats
    Arena arena;              // per-key scratch (grow-only)
    void *h_result = nullptr; // page-locked landing buffer of the 3 partial sums per MSM (direct copy, no staging hop)
};

// fills table[len .. 16*len) from table[0 .. len); keys that want it (wants_wide_table) also get table_w
void build_table(Key &k, hipStream_t stream);
// frees what the key owns besides `table`: table_w, the scratch arena, the landing buffer
void release(Key &k);
// fills table[0 .. len) with the synthetic key (see k_gen_bases)
void generate_bases(Key &k, uint64_t seed, hipStream_t stream);

// table[0 .. len) back in the ABI's 2^256 Montgomery form (the table itself is kept in the form of field29.cuh)
void read_bases(const Key &k, affine_t *out_dev, hipStream_t stream);

// number of bases in table[0 .. len) that are not on the curve (identity counts as on-curve)
size_t count_off_curve(const Key &k, hipStream_t stream);

size_t workspace_bytes(uint32_t n_max, uint32_t batch);

// batch of MSMs over the base prefix: result_host[m] = sum_{i < n[m]} scalars[m][i] * P_i  (XYZZ).
// scalars_dev: HOST array of DEVICE pointers; with world > 1 they point at the FULL vectors and
// only this rank's stripes are read (n[m] is then the LOCAL count).
void run(Key &k, const fe_t *const *scalars_dev, const uint32_t *n_host, uint32_t batch, int is_mont,
         hipStream_t stream, xyzz_t *result_host);

// The two halves of run() for callers that keep several sets of launches in flight (chunked commits whose uploads overlap
// the previous chunk's MSM):  enqueue() only launches (<= BATCH_ARGS MSMs; MSM m reads bases [base[m], base[m] + n[m]),
// base_host == nullptr: the usual prefix); the partial sums land in page-locked slot `slot` in stream order.  After the
// caller has synchronised the stream, finish() does the host end.  The scratch arena is shared: sets of launches on ONE
// stream reuse it in stream order; reserve() sizes it up front (growing it later would free memory still in use).
// `fold`: the sets of a chunked commit share ONE bucket set (the buckets are the digit values, whatever the base offset), and the
// bucket reduction is linear: a set with FOLD_FIRST / FOLD_MIDDLE only adds its bucket sums into the key's running buckets (no
// reduction, no result), the FOLD_LAST set adds its own and reduces the total -- one k_rowcol + k_reduce_final + host finish per commit
// instead of one per chunk.  batch == 1, 16-bit-window sets only (may_fold()).
enum Fold { FOLD_NONE = 0, FOLD_FIRST = 1, FOLD_MIDDLE = 2, FOLD_LAST = 3 };
bool may_fold(const Key &k, uint32_t n);     // always true since r04 (the sets of a chunked commit never take the wide-window pipeline)
bool enqueue(Key &k, const fe_t *const *scalars_dev, const uint32_t *n_host, const uint32_t *base_host, uint32_t batch, int is_mont,
             hipStream_t stream, uint32_t slot, Fold fold = FOLD_NONE);
void finish(Key &k, uint32_t batch, uint32_t slot, bool launched, xyzz_t *result_host);
void reserve(Key &k, uint32_t n_max, uint32_t batch);
// Slot mode predicts per key whether a commit has parts beyond the slots (hot buckets: 0 / 1 / small witnesses) and launches the
// overflow kernels only then.  After the stream has been synchronised: overflow_missed(slot) says that the set in `slot` HAD such parts
// while its overflow kernels were not launched -- its result is incomplete and the caller must run the MSM again (the prediction has
// been switched, so the second run is complete); note_commit() records what the finished commit saw for the next prediction (on at
// once -- and from a key's first commit --, off after three commits in a row without hot buckets).
bool overflow_missed(const Key &k, uint32_t slot);
void note_commit(Key &k, const uint32_t *slots_used, uint32_t n_slots, uint64_t scalars);   // scalars: what the commit accumulated (with the entry count: its density)

}  // namespace msm
}  // namespace srs
