/* Plain-C client of include/sirius_amd.h: what a cgo / Rust-FFI / JNI shim sees.  No Python, no torch, no C++.
 * Checks through the ABI alone:  commit(e_i) == base_i,  commit(1,1,1) == base_0 + base_1 + base_2 (srs_point_sum),
 * too-long input -> SRS_ERR_TOO_LONG_INPUT with the reference's message, NTT forward + inverse round trip, the layout
 * self-tests, device / page-locked buffers + srs_commit_upload + a multi-device key, and a structure built from C driven
 * through srs_pg_compute_F / srs_pg_evaluate_e (F(0) = e) and srs_commit_cross_terms (commitments = commits of the vectors).
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
    {   /* layout self-tests the shim runs at start-up: field one / two and the generators */
        srs_fe two_fr = ONE_FR;
        /* 2R mod r = R + R - r ... let the library do the doubling: fold 1 + 1 * 1 */
        CHECK(srs_fold_witness(SRS_FIELD_FR, &two_fr, &ONE_FR, &ONE_FR, &ONE_FR, 1, SRS_SPACE_HOST, NULL));
        CHECK(srs_layout_selftest(SRS_FIELD_FR, &ONE_FR, &two_fr));
        if (srs_layout_selftest(SRS_FIELD_FR, &two_fr, &ONE_FR) != SRS_ERR_LAYOUT) { fprintf(stderr, "selftest accepted swapped one / two\n"); return 1; }
        srs_affine g_bn = {ONE_FQ, ONE_FQ};       /* bn256 G1 generator (1, 2): coordinates live in Fq */
        CHECK(srs_fold_witness(SRS_FIELD_FQ, &g_bn.y, &ONE_FQ, &ONE_FQ, &ONE_FQ, 1, SRS_SPACE_HOST, NULL));
        CHECK(srs_layout_selftest_point(SRS_CURVE_BN256, &g_bn));
        if (srs_layout_selftest_point(SRS_CURVE_GRUMPKIN, &g_bn) != SRS_ERR_LAYOUT) { fprintf(stderr, "bn256 generator accepted for grumpkin\n"); return 1; }
    }
    {   /* device-resident vectors, the streamed witness commit, a single-process multi-device key */
        enum { N2 = 5000 };
        srs_ck *ck = NULL, *mk = NULL;
        CHECK(srs_ck_setup_synthetic(SRS_CURVE_BN256, N2, 9, 0, 1, &ck));
        CHECK(srs_ck_setup_synthetic_multi(SRS_CURVE_BN256, N2, 9, 3, &mk));
        if (srs_ck_num_shards(mk) != 3 || srs_ck_num_shards(ck) != 1) { fprintf(stderr, "srs_ck_num_shards\n"); return 1; }
        void *hp = NULL, *dp = NULL;
        CHECK(srs_host_alloc(N2 * sizeof(srs_fe), &hp));
        CHECK(srs_dev_alloc(N2 * sizeof(srs_fe), &dp));
        srs_fe *hv = (srs_fe *)hp, *back = (srs_fe *)malloc(N2 * sizeof(srs_fe));
        for (int i = 0; i < N2; ++i) { hv[i] = ONE_FR; hv[i].l[0] ^= (uint64_t)(i + 1) * 0x9e3779b97f4a7c15ull; hv[i].l[3] &= 0x0fffffffffffffffull; if (i % 3 == 0) memset(&hv[i], 0, sizeof(srs_fe)); }
        srs_affine c0, c1, c2, c3;
        CHECK(srs_commit(ck, hv, N2, SRS_SPACE_HOST, SRS_REPR_MONT, NULL, &c0));
        CHECK(srs_commit_upload(ck, hv, N2, (srs_fe *)dp, SRS_REPR_MONT, NULL, &c1));
        CHECK(srs_commit(ck, (const srs_fe *)dp, N2, SRS_SPACE_DEVICE, SRS_REPR_MONT, NULL, &c2));      /* the copy it left in HBM */
        CHECK(srs_commit_upload(mk, hv, N2, NULL, SRS_REPR_MONT, NULL, &c3));
        if (memcmp(&c0, &c1, sizeof c0) || memcmp(&c0, &c2, sizeof c0) || memcmp(&c0, &c3, sizeof c0)) { fprintf(stderr, "streamed / device / multi-device commits differ\n"); return 1; }
        CHECK(srs_download(back, dp, N2 * sizeof(srs_fe), NULL));
        if (memcmp(back, hv, N2 * sizeof(srs_fe)) != 0) { fprintf(stderr, "device copy != source\n"); return 1; }
        free(back);
        srs_dev_free(dp); srs_host_free(hp);
        srs_ck_free(ck); srs_ck_free(mk);
    }
    {   /* a structure built from C: gate q * a0 * a1 - a2 (1 fixed, 3 advice columns), k = 6; ProtoGalaxy and Sangria entry points */
        enum { K = 6, ROWS = 1 << K, NADV = 3, WLEN = NADV * ROWS };
        /* query index space: [selectors | fixed | advice]: one fixed column q (index 0), advice a0..a2 (1..3); gate q * a0 * a1 - a2 */
        const uint64_t gate[] = {SRS_EX_POLY, 0, 0, SRS_EX_POLY, 1, 0, SRS_EX_PRODUCT, SRS_EX_POLY, 2, 0, SRS_EX_PRODUCT,
                                 SRS_EX_POLY, 3, 0, SRS_EX_NEG, SRS_EX_SUM, SRS_EX_END};
        srs_fe *q = (srs_fe *)malloc(ROWS * sizeof(srs_fe));
        for (int i = 0; i < ROWS; ++i) { q[i] = ONE_FR; q[i].l[2] ^= (uint64_t)(i + 5) * 0x9e3779b97f4a7c15ull; q[i].l[3] &= 0x0fffffffffffffffull; }
        const srs_fe *fixed_cols[1] = {q};
        srs_structure *S = NULL;
        CHECK(srs_structure_create(SRS_FIELD_FR, K, 0, 1, NADV, NULL, fixed_cols, SRS_SPACE_HOST, gate, sizeof gate / sizeof gate[0], 1, &S));
        if (srs_structure_kernel_kind(S, 0) < 0 || srs_structure_kernel_kind(S, 2) != SRS_KERNEL_INTERPRETER) { fprintf(stderr, "kernel kind\n"); return 1; }
        srs_pg_context ctx;
        CHECK(srs_pg_context_new(S, 1, &ctx));
        if (ctx.count_of_evaluation_with_padding != ROWS || ctx.betas_count != K) { fprintf(stderr, "PolyContext sizes\n"); return 1; }
        srs_fe *W1 = (srs_fe *)malloc(WLEN * sizeof(srs_fe)), *W2 = (srs_fe *)malloc(WLEN * sizeof(srs_fe));
        for (int i = 0; i < WLEN; ++i) {
            W1[i] = ONE_FR; W1[i].l[0] ^= (uint64_t)(i + 7) * 0x9e3779b97f4a7c15ull; W1[i].l[3] &= 0x0fffffffffffffffull;
            W2[i] = ONE_FR; W2[i].l[1] ^= (uint64_t)(i + 3) * 0xbf58476d1ce4e5b9ull; W2[i].l[3] &= 0x0fffffffffffffffull;
        }
        srs_fe betas[K], delta = ONE_FR, e, *F = (srs_fe *)malloc(ctx.fft_points_count_F * sizeof(srs_fe));
        for (int i = 0; i < K; ++i) { betas[i] = ONE_FR; betas[i].l[0] ^= (uint64_t)(i + 11) * 0x94d049bb133111ebull; betas[i].l[3] &= 0x0fffffffffffffffull; }
        delta.l[2] ^= 0x1234567ull;
        for (int compat = 0; compat < 2; ++compat) {
            CHECK(srs_pg_compute_F(S, betas, K, &delta, W1, NULL, 0, SRS_SPACE_HOST, compat, NULL, F));
            CHECK(srs_pg_evaluate_e(S, betas, K, W1, NULL, 0, SRS_SPACE_HOST, compat, NULL, &e));
            if (memcmp(&F[0], &e, sizeof e) != 0) { fprintf(stderr, "F(0) != evaluate_e(betas) (compat %d)\n", compat); return 1; }
        }
        /* VanillaFS::commit_cross_terms: d = 2 cross terms; every commitment equals the commit of the returned vector */
        size_t d = srs_structure_num_cross_terms(S);
        if (d != 2) { fprintf(stderr, "cross terms: %zu\n", d); return 1; }
        srs_ck *ck = NULL;
        CHECK(srs_ck_setup_synthetic(SRS_CURVE_BN256, ROWS, 4, 0, 1, &ck));
        srs_fe *T[2] = {(srs_fe *)malloc(ROWS * sizeof(srs_fe)), (srs_fe *)malloc(ROWS * sizeof(srs_fe))};
        srs_fe ch[2] = {ONE_FR, ONE_FR};       /* U1.u, then DEFAULT_u = 1 of the incoming instance (no gate challenges) */
        ch[0].l[1] ^= 0xabcdefull;
        srs_affine cm[2], one_by_one;
        CHECK(srs_commit_cross_terms(S, ck, W1, W2, ch, 2, SRS_SPACE_HOST, NULL, T, cm));
        for (size_t i = 0; i < d; ++i) {
            CHECK(srs_commit(ck, T[i], ROWS, SRS_SPACE_HOST, SRS_REPR_MONT, NULL, &one_by_one));
            if (memcmp(&one_by_one, &cm[i], sizeof one_by_one) != 0) { fprintf(stderr, "cross-term commitment %zu\n", i); return 1; }
        }
        free(T[0]); free(T[1]); free(F); free(W1); free(W2); free(q);
        srs_ck_free(ck);
        srs_structure_free(S);
    }
    printf("C-ABI OK (%s)\n", srs_version());
    return 0;
}
