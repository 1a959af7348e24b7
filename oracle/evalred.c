/* TEST INFRASTRUCTURE ONLY (oracle): N-to-1 evaluation reduction restated from
 * /root/reference/joltworks/src/subprotocols/evaluation_reduction.rs:91-147 (prove), :213-249
 * (eval_on_l, compute_h) with the UniPoly arithmetic it uses (poly/unipoly.rs:55-64 from_evals,
 * :415-436 Add/Sub keep the longer length, :463-476 Mul trims, :540-548 uncompressed append). */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "ra.h"

static size_t trim_len(const fr_t *c, size_t n) {
    fr_t z; fr_zero(&z);
    while (n > 0 && memcmp(&c[n - 1], &z, sizeof z) == 0) n--;
    return n ? n : 1;
}

/* UniPoly::from_evals on 0..N-1 (value only; callers trim where the reference does) */
static void interp(const fr_t *evals, size_t N, fr_t *coeffs) {
    const size_t W = N + 1;
    fr_t *m = (fr_t *)calloc(N * W, sizeof(fr_t));
    for (size_t i = 0; i < N; i++) {
        fr_t x; fr_from_u64(i, &x);
        fr_one(&m[i * W]);
        for (size_t j = 1; j < N; j++) fr_mul(&m[i * W + j - 1], &x, &m[i * W + j]);
        m[i * W + N] = evals[i];
    }
    orc_gauss_solve(m, N, coeffs);
    free(m);
}

int orc_eval_reduction_prove(const fr_t *mle, size_t n, const fr_t *points, const fr_t *claims, size_t N,
                             orc_transcript *t, fr_t *h_out, size_t *h_len, fr_t *r_out, fr_t *claim_out) {
    if (N == 0) return -1;
    if (N == 1) {                                                   /* :113-127 */
        h_out[0] = claims[0]; *h_len = 1;
        memcpy(r_out, points, n * sizeof(fr_t)); *claim_out = claims[0];
        return 0;
    }
    const size_t len = (size_t)1 << n, cap = n * (N - 1) + 1;
    /* var_poly_i = from_evals(ri_vec[i]), ri_vec[i][j] = points[j][i] (group_by_variable) */
    fr_t *var = (fr_t *)calloc(n * N, sizeof(fr_t)), *ev = (fr_t *)malloc(N * sizeof(fr_t));
    for (size_t i = 0; i < n; i++) {
        for (size_t j = 0; j < N; j++) ev[j] = points[j * n + i];
        interp(ev, N, &var[i * N]);
    }
    /* compute_h: fold 2^n constant polynomials */
    fr_t *polys = (fr_t *)calloc(len * cap, sizeof(fr_t));
    size_t *plen = (size_t *)malloc(len * sizeof(size_t));
    for (size_t j = 0; j < len; j++) { polys[j * cap] = mle[j]; plen[j] = 1; }
    fr_t *diff = (fr_t *)malloc(cap * sizeof(fr_t)), *prod = (fr_t *)malloc((cap + N) * sizeof(fr_t));
    for (size_t i = 0; i < n; i++) {
        const size_t half = (size_t)1 << (n - i - 1);
        const fr_t *vp = &var[i * N];
        for (size_t j = 0; j < half; j++) {
            fr_t *L = &polys[j * cap], *R = &polys[(j + half) * cap];
            const size_t ll = plen[j], rl = plen[j + half], dl = ll > rl ? ll : rl;
            for (size_t k = 0; k < dl; k++) {                       /* &right - &left, length max */
                fr_t a, b; if (k < rl) a = R[k]; else fr_zero(&a); if (k < ll) b = L[k]; else fr_zero(&b);
                fr_sub(&a, &b, &diff[k]);
            }
            const size_t pl = (N - 1) + (dl - 1) + 1;               /* Mul: degree sum + 1, then from_coeff */
            for (size_t k = 0; k < pl; k++) fr_zero(&prod[k]);
            for (size_t a = 0; a < N; a++)
                for (size_t b = 0; b < dl; b++) { fr_t x; fr_mul(&vp[a], &diff[b], &x); fr_add(&prod[a + b], &x, &prod[a + b]); }
            const size_t tl = trim_len(prod, pl);
            const size_t nl = ll > tl ? ll : tl;                    /* &left + &product, length max */
            for (size_t k = 0; k < nl; k++) {
                fr_t a, b; if (k < ll) a = L[k]; else fr_zero(&a); if (k < tl) b = prod[k]; else fr_zero(&b);
                fr_add(&a, &b, &L[k]);
            }
            plen[j] = nl;
        }
    }
    *h_len = plen[0];
    memcpy(h_out, polys, plen[0] * sizeof(fr_t));
    /* h.append_to_transcript; x' = challenge_scalar_optimized (:139-140) */
    orc_transcript_append_message(t, "UncompressedUniPoly_begin");
    for (size_t k = 0; k < *h_len; k++) orc_transcript_append_scalar(t, &h_out[k]);
    orc_transcript_append_message(t, "UncompressedUniPoly_end");
    fr_t x; u128 raw; orc_transcript_challenge_optimized(t, &raw, &x);
    for (size_t i = 0; i < n; i++) orc_unipoly_eval(&var[i * N], N, &x, &r_out[i]);   /* eval_on_l */
    orc_unipoly_eval(h_out, *h_len, &x, claim_out);
    free(var); free(ev); free(polys); free(plen); free(diff); free(prod);
    return 0;
}
