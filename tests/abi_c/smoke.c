/* Plain-C client of include/sirius_amd.h: what a cgo / Rust-FFI / JNI shim sees.  No Python, no torch, no C++.
 * Checks through the ABI alone:  commit(e_i) == base_i,  commit(1,1,1) == base_0 + base_1 + base_2 (srs_point_sum),
 * too-long input -> SRS_ERR_TOO_LONG_INPUT with the reference's message, NTT forward + inverse round trip.
 * Built and run by tests/test_abi_c_client.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sirius_amd.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != SRS_OK) {                                                             \
            fprintf(stderr, "%s -> rc %d: %s\n", #call, rc_, srs_last_error());          \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

/* Montgomery one (R mod p) of the scalar fields, little-endian limbs */
static const srs_fe ONE_FR = {{0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full}};
static const srs_fe ONE_FQ = {{0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full}};

int main(void) {
    enum { N = 1000 };
    CHECK(srs_init(-1));
    for (int curve = 0; curve < 2; ++curve) {
        const srs_fe one = curve == SRS_CURVE_BN256 ? ONE_FR : ONE_FQ;   /* scalars of bn256 live in Fr, of grumpkin in Fq */
        srs_ck *ck = NULL;
        CHECK(srs_ck_setup_synthetic(curve, N, 7, 0, 1, &ck));
        srs_affine *bases = (srs_affine *)malloc(N * sizeof(srs_affine));
        srs_fe *v = (srs_fe *)calloc(N, sizeof(srs_fe));
        CHECK(srs_ck_get_bases(ck, bases));
        srs_affine c;
        v[5] = one;
        CHECK(srs_commit(ck, v, N, SRS_SPACE_HOST, SRS_REPR_MONT, NULL, &c));
        if (memcmp(&c, &bases[5], sizeof c) != 0) { fprintf(stderr, "commit(e_5) != base_5 (curve %d)\n", curve); return 1; }
        memset(v, 0, N * sizeof(srs_fe));
        v[0] = v[1] = v[2] = one;
        srs_affine s;
        CHECK(srs_commit(ck, v, 3, SRS_SPACE_HOST, SRS_REPR_MONT, NULL, &c));
        CHECK(srs_point_sum(curve, bases, 3, &s));
        if (memcmp(&c, &s, sizeof c) != 0) { fprintf(stderr, "commit(1,1,1) != b0+b1+b2 (curve %d)\n", curve); return 1; }
        srs_fe *big = (srs_fe *)calloc(N + 1, sizeof(srs_fe));
        int rc = srs_commit(ck, big, N + 1, SRS_SPACE_HOST, SRS_REPR_MONT, NULL, &c);
        if (rc != SRS_ERR_TOO_LONG_INPUT || !strstr(srs_last_error(), "Can't commit too long input")) {
            fprintf(stderr, "too-long input: rc %d, '%s'\n", rc, srs_last_error());
            return 1;
        }
        free(big); free(v); free(bases);
        srs_ck_free(ck);
    }
    {   /* ifft(fft(a)) == a on 2^12 points of Fr */
        enum { M = 1 << 12 };
        srs_fe *a = (srs_fe *)malloc(M * sizeof(srs_fe)), *b = (srs_fe *)malloc(M * sizeof(srs_fe));
        for (int i = 0; i < M; ++i) { a[i] = ONE_FR; a[i].l[0] ^= (uint64_t)i * 0x9e3779b97f4a7c15ull; a[i].l[3] &= 0x0fffffffffffffffull; }
        memcpy(b, a, M * sizeof(srs_fe));
        CHECK(srs_ntt(SRS_FIELD_FR, b, M, 0, 0, SRS_SPACE_HOST, NULL));
        if (memcmp(a, b, M * sizeof(srs_fe)) == 0) { fprintf(stderr, "fft left the input unchanged\n"); return 1; }
        CHECK(srs_ntt(SRS_FIELD_FR, b, M, 1, 0, SRS_SPACE_HOST, NULL));
        if (memcmp(a, b, M * sizeof(srs_fe)) != 0) { fprintf(stderr, "ifft(fft(a)) != a\n"); return 1; }
        if (srs_ntt(SRS_FIELD_FR, b, 3, 0, 0, SRS_SPACE_HOST, NULL) != SRS_ERR_NOT_POW2) { fprintf(stderr, "n = 3 accepted\n"); return 1; }
        free(a); free(b);
    }
    printf("C-ABI OK (%s)\n", srs_version());
    return 0;
}
