// tuning.h -- the library's run-time tunables: ONE table, set through the C-ABI (srs_tuning_set), never through the environment.
//
// Every tunable selects among code paths that the library takes BY DEFAULT for some input size (single- vs two-pass sort, the
// part length of level 0, 16- vs 20-bit windows, the number of chunks of a streamed commit, ...), or moves the size threshold
// between them; none changes a result.  They exist so that a test can run a large-input path on an input small enough for the
// CPU oracle to check, and so that a deployer can trade memory for speed (msm_wide).  Variants that were measured slower and
// are no default anywhere are not selectable: they were removed from the library (records: profiles/*_ab_*.txt).
#pragma once
#include <cstdint>

namespace srs {
namespace tuning {

enum Id : int {
    MSM_SORT = 0,       // 0 auto | 1 single-pass scatter (k_scatter) | 2 two-pass sort (k_group + k_scatter2)
    MSM_L0,             // 0 auto | 1..7: log2 of the level-0 part length of whole MSMs
    MSM_WIDE,           // -1 auto (keys >= 2^23 bases) | 0 never build / use the 13 x 20-bit window table (saves 13/16 of the key's HBM) | 1 always
    MSM_WIDE_MIN,       // log2 of the smallest whole MSM that takes the 20-bit windows (default 23)
    MSM_SLOTS,          // 1 auto (the sets of a streamed commit) | 0 never | 2 every 16-bit-window set
    MSM_SLOT_LOG,       // 0 auto | 2..8: log2 of the slots per bucket (small values make every bucket "hot")
    MSM_EXPECT_OVF,     // 1 a new key expects hot buckets | 0 it does not (the redo path then runs on the first hot commit)
    MSM_QUAD_MAX,       // 0 auto | log2 of the largest level k_accum1 gives to quads of lanes
    COMMIT_CHUNKS,      // 0 auto | n: a streamed commit is cut into n equal chunks
    PG_F_EVAL,          // 1: compute_F by evaluation + ifft (the path of tiles that are not 8 leaves) instead of the polynomial tree
    PG_G_FFT,           // 1: compute_G on the roots of unity + ifft (the path of L >= 2 incoming traces) instead of integer points
    JIT_ALWAYS,         // 1: run-time compile the sweep kernels below k = 14 as well
    NO_JIT,             // 1: never call hiprtc (also: environment SRS_NO_JIT, for deployments without the hiprtc library)
    N_TUNABLES
};

constexpr int64_t UNSET = INT64_MIN;

int64_t get(Id id);                        // UNSET when never set
inline int64_t get_or(Id id, int64_t dflt) {
    const int64_t v = get(id);
    return v == UNSET ? dflt : v;
}
int set(const char *name, int64_t value);  // 0 ok, -1 unknown name; value == UNSET clears
int64_t get_by_name(const char *name, int *found);
void reset();
const char *name_of(int id);

}  // namespace tuning
}  // namespace srs
