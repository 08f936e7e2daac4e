// keygen.hip -- the restatable half of `CommitmentKey::setup` (reference src/commitment.rs:55-79).  Host code.
//
//   let mut reader = Shake256::default().chain(label).finalize_xof();                 (:61)
//   repeat_with(|| { let mut buffer = [0u8; 32]; reader.read_exact(&mut buffer); buffer }).take(n)   (:63-68)
//       .par_bridge().map(|uniform_byte| C::CurveExt::hash_to_curve("from_uniform_bytes")(&uniform_byte))   (:69-71)
//
// The byte stream is FIPS 202 SHAKE256 (Keccak-f[1600], rate 136, domain suffix 0x1F) and is restated here
// (srs_ck_setup_uniform_bytes; pinned against Python's hashlib.shake_256 and the FIPS 202 empty-message answer in
// tests/test_keygen.py).  The map from a 32-byte chunk to a curve point is `hash_to_curve` of halo2curves, a THIRD-PARTY
// dependency absent from the reference tree and pulled by a branch name without a lock file (Cargo.toml:44-46): its
// algorithm (hash suite, domain string, map) cannot be pinned from here, so srs_ck_setup refuses with SRS_ERR_UNSUPPORTED
// rather than guess -- real keys enter through the reference's cache file (srs_ck_load_file).  Note that `par_bridge()` does
// not preserve order: even the reference does not produce the same key order on two runs, the cache file is what pins a key.
#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/sirius_amd.h"
#include "devrt.h"

namespace srs {
namespace {

inline uint64_t rotl(uint64_t x, int s) { return s ? (x << s) | (x >> (64 - s)) : x; }

void keccak_f1600(uint64_t a[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
        0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
        0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
        0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};   // [x + 5 y]
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];                   // theta
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; ++x)                                                                            // rho + pi
            for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], RHO[x + 5 * y]);
        for (int y = 0; y < 5; ++y)                                                                            // chi
            for (int x = 0; x < 5; ++x) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];                                                                                     // iota
    }
}

struct Shake256 {
    static constexpr size_t RATE = 136;
    uint64_t st[25] = {};
    size_t pos = 0;
    void xor_byte(size_t i, uint8_t v) { st[i / 8] ^= (uint64_t)v << (8 * (i % 8)); }
    void absorb(const uint8_t *p, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            xor_byte(pos++, p[i]);
            if (pos == RATE) { keccak_f1600(st); pos = 0; }
        }
    }
    void finalize() {
        xor_byte(pos, 0x1F);
        xor_byte(RATE - 1, 0x80);
        keccak_f1600(st);
        pos = 0;
    }
    void skip(size_t n) {                       // whole blocks are permuted away without copying
        while (n) {
            const size_t take = n < RATE - pos ? n : RATE - pos;
            pos += take;
            n -= take;
            if (pos == RATE) { keccak_f1600(st); pos = 0; }
        }
    }
    void squeeze(uint8_t *out, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            out[i] = (uint8_t)(st[pos / 8] >> (8 * (pos % 8)));
            if (++pos == RATE) { keccak_f1600(st); pos = 0; }
        }
    }
};

}  // namespace
}  // namespace srs

using namespace srs;

static int fail(int rc, const std::string &msg) {
    set_error(msg);
    return rc;
}

extern "C" {

int srs_ck_setup_uniform_bytes(const uint8_t *label, size_t label_len, size_t first, size_t count, uint8_t *out) {
    if ((label_len && !label) || (count && !out)) return fail(SRS_ERR_INVALID, "srs_ck_setup_uniform_bytes: bad argument");
    Shake256 x;
    x.absorb(label, label_len);
    x.finalize();
    x.skip(first * 32);
    x.squeeze(out, count * 32);
    return SRS_OK;
}

int srs_ck_setup(int curve, uint32_t k, const uint8_t *label, size_t label_len, srs_ck **out) {
    (void)label;
    (void)label_len;
    if (out) *out = nullptr;
    if (curve != SRS_CURVE_BN256 && curve != SRS_CURVE_GRUMPKIN) return fail(SRS_ERR_INVALID, "srs_ck_setup: unknown curve");
    if (k >= 32) return fail(SRS_ERR_INVALID, "srs_ck_setup: assert!(k < 32)  (src/commitment.rs:58)");
    return fail(SRS_ERR_UNSUPPORTED,
                "srs_ck_setup: CommitmentKey::setup maps every 32-byte SHAKE256 chunk with halo2curves' hash_to_curve(\"from_uniform_bytes\"), "
                "third-party code that is not part of the reference tree (Cargo.toml:44-46, unpinned branch): not restated.  The byte stream is "
                "available (srs_ck_setup_uniform_bytes); load the key from the reference's cache file instead (srs_ck_load_file), or "
                "hand the bases over (srs_ck_create)");
}

}  // extern "C"
