/* A plain C99 client of libqilhip.so: the drop-in boundary used the way a foreign host would use it (no Python,
 * no C++ types).  3-site real MPS, 3-site real MPO, apply, all 8 coefficients against a dense contraction done here
 * in plain loops; then the error convention (status code + qil_last_error).  Prints "C ABI client OK".
 * Build: gcc -std=c99 -Iinclude tests/cabi_client.c -Lqilaplace.jl_amd/lib -lqilhip -Wl,-rpath,... -lm            */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qilaplace_hip.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int s_ = (call);                                                                 \
        if (s_ != QIL_OK) {                                                              \
            fprintf(stderr, "%s failed with %d: %s\n", #call, s_, qil_last_error());     \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

static double lcg(unsigned* st) {
    *st = *st * 1664525u + 1013904223u;
    return ((*st >> 8) & 0xFFFF) / 65536.0 - 0.5;
}

int main(void) {
    /* bonds: MPS 1-2-2-1, MPO 1-3-2-1; layouts A[a, s, b], W[a, s_in, s_out, b], column-major */
    const int64_t cb[2] = {2, 2}, db[2] = {3, 2};
    const int cdim[4] = {1, 2, 2, 1}, ddim[4] = {1, 3, 2, 1};
    double *A[3], *W[3];
    unsigned st = 7u;
    int i, j;
    for (i = 0; i < 3; ++i) {
        const int na = cdim[i] * 2 * cdim[i + 1], nw = ddim[i] * 4 * ddim[i + 1];
        A[i] = (double*)malloc(sizeof(double) * na);
        W[i] = (double*)malloc(sizeof(double) * nw);
        for (j = 0; j < na; ++j) A[i][j] = lcg(&st);
        for (j = 0; j < nw; ++j) W[i][j] = lcg(&st);
    }
    qil_context* ctx = NULL;
    qil_mps *psi = NULL, *out = NULL;
    qil_mpo* Wd = NULL;
    CHECK(qil_context_create(0, NULL, &ctx));
    CHECK(qil_mps_create(ctx, 3, QIL_F64, 0, cb, NULL, (const void* const*)A, 1.5, &psi));
    CHECK(qil_mpo_create(ctx, 3, QIL_F64, 0, db, NULL, (const void* const*)W, &Wd));
    CHECK(qil_apply(Wd, psi, &out));
    int64_t ob[2];
    CHECK(qil_mps_bond_dims(out, ob));
    if (ob[0] != 6 || ob[1] != 4) {
        fprintf(stderr, "apply: bonds %lld, %lld (expected the products 6, 4)\n", (long long)ob[0], (long long)ob[1]);
        return 1;
    }
    uint8_t bits[8 * 3];
    double got[8 * 2];
    for (i = 0; i < 8; ++i)
        for (j = 0; j < 3; ++j) bits[3 * i + j] = (uint8_t)((i >> (2 - j)) & 1);
    CHECK(qil_coefficient_batch(out, 8, bits, got));
    /* dense reference: coefficient(t) = amp * sum_{s} prod_i sum W_i[a, s_i, t_i, b] A_i[alpha, s_i, beta] */
    double worst = 0.0;
    for (i = 0; i < 8; ++i) {
        double ref = 0.0;
        int s;
        for (s = 0; s < 8; ++s) {
            /* transfer over the fused bond (alpha, a) */
            double v[6] = {1, 0, 0, 0, 0, 0}, vn[6];
            int site, dl = 1, cl = 1;
            for (site = 0; site < 3; ++site) {
                const int si = (s >> (2 - site)) & 1, ti = (i >> (2 - site)) & 1;
                const int cr = cdim[site + 1], dr = ddim[site + 1];
                int al, a, be, b;
                for (b = 0; b < dr; ++b)
                    for (be = 0; be < cr; ++be) {
                        double acc = 0.0;
                        for (a = 0; a < dl; ++a)
                            for (al = 0; al < cl; ++al)
                                acc += v[al + cl * a] * A[site][al + cl * (si + 2 * be)] *
                                       W[site][a + dl * (si + 2 * (ti + 2 * b))];
                        vn[be + cr * b] = acc;
                    }
                memcpy(v, vn, sizeof(double) * (size_t)(cr * dr));
                cl = cr;
                dl = dr;
            }
            ref += v[0];
        }
        ref *= 1.5;
        {
            const double d = fabs(got[2 * i] - ref) + fabs(got[2 * i + 1]);
            if (d > worst) worst = d;
        }
    }
    if (worst > 1e-13) {
        fprintf(stderr, "coefficient mismatch %.3e\n", worst);
        return 1;
    }
    /* error convention: a 2-site MPO on a 3-site MPS */
    {
        qil_mpo* W2 = NULL;
        qil_mps* bad = NULL;
        CHECK(qil_mpo_create(ctx, 2, QIL_F64, 0, db, NULL, (const void* const*)W, &W2));
        const int s_ = qil_apply(W2, psi, &bad);
        if (s_ != QIL_EINVAL_LENGTH || strlen(qil_last_error()) == 0 || bad != NULL) {
            fprintf(stderr, "expected QIL_EINVAL_LENGTH with a message, got %d '%s'\n", s_, qil_last_error());
            return 1;
        }
        CHECK(qil_mpo_destroy(W2));
    }
    /* build_zt_mpo behind its one verb (r06), against a published output of the reference: build_zt_mpo(n = 2, wr = 2 pi,
     * cutoff = 1e-14) has bonds copy_1 = 2, main_1 = 8, copy_2 = 2 (docs/src/tutorials/zt.md:185-190); n = 0 is the reference's
     * ArgumentError (zt_transformer.jl:49) */
    {
        const double wr[2] = {6.283185307179586, 0.5};
        qil_mpo* Wz[2] = {NULL, NULL};
        int64_t zb[3] = {0, 0, 0}, nz = 0;
        int paired = 0;
        CHECK(qil_build_zt_mpo_batch(ctx, 2, 2, wr, 1e-14, 1000, NULL, Wz));
        CHECK(qil_mpo_nsites(Wz[0], &nz));
        CHECK(qil_mpo_is_paired(Wz[0], &paired));
        CHECK(qil_mpo_bond_dims(Wz[0], zb));
        if (nz != 4 || !paired || zb[0] != 2 || zb[1] != 8 || zb[2] != 2) {
            fprintf(stderr, "build_zt_mpo(2, 2 pi): %lld tensors, paired %d, bonds %lld %lld %lld (published: 2 8 2)\n", (long long)nz, paired,
                    (long long)zb[0], (long long)zb[1], (long long)zb[2]);
            return 1;
        }
        CHECK(qil_mpo_destroy(Wz[0]));
        CHECK(qil_mpo_destroy(Wz[1]));
        if (qil_build_zt_mpo_batch(ctx, 0, 1, wr, 1e-14, 1000, NULL, Wz) != QIL_EINVAL_ARG || strlen(qil_last_error()) == 0) {
            fprintf(stderr, "build_zt_mpo(0, ...): expected QIL_EINVAL_ARG with a message\n");
            return 1;
        }
    }
    CHECK(qil_mps_destroy(out));
    CHECK(qil_mps_destroy(psi));
    CHECK(qil_mpo_destroy(Wd));
    CHECK(qil_context_destroy(ctx));
    for (i = 0; i < 3; ++i) {
        free(A[i]);
        free(W[i]);
    }
    printf("C ABI client OK (max |coefficient - dense| = %.2e)\n", worst);
    return 0;
}
