/* TEST INFRASTRUCTURE ONLY (oracle): the one-hot "ra" sumcheck instances restated from the
 * reference (paths under /root/reference/joltworks/src):
 *   GruenSplitEqPolynomial            poly/split_eq_poly.rs:86-145,331-429,526-597
 *   RaPolynomial                      poly/ra_poly.rs:21-110 (values only: ra(j) = F[idx_j] or 0;
 *                                     the Round1/2/3 tables are a memory layout of the same MLE)
 *   RaSumcheckProver                  subprotocols/ra_virtual.rs:97-145
 *   compute_mles_product_sum + finish subprotocols/mles_product_sum.rs:15-36,61-131,330-376
 *   UniPoly::from_evals_toom          poly/unipoly.rs:103-134 (+ utils/gaussian_elimination.rs)
 *   BooleanitySumcheckProver          subprotocols/booleanity.rs:169-348
 *   ExpandingTable                    utils/expanding_table.rs:46-88
 *   HammingWeightSumcheckProver       subprotocols/hamming_weight.rs:106-148
 * Indices are int32, negative = None (the reference's Option<u8>/Option<u16>). */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "ra.h"
#include "opening.h"
#include "psshout.h"
#include "elementwise.h"
#include "softmax.h"
#include "shout.h"

/* ------------------------------------------------------------------ small helpers */
static size_t trim(fr_t *c, size_t n) {                 /* UniPoly::from_coeff (unipoly.rs:39-52) */
    fr_t z; fr_zero(&z);
    while (n > 0 && memcmp(&c[n - 1], &z, sizeof z) == 0) n--;
    if (n == 0) { c[0] = z; n = 1; }
    return n;
}

/* solve the n x n system rows[i][0..n) * c = rows[i][n] (unique solution; any pivoting) */
void orc_gauss_solve(fr_t *m, size_t n, fr_t *c) {
    const size_t W = n + 1;
    fr_t z; fr_zero(&z);
    for (size_t col = 0; col < n; col++) {
        size_t piv = col;
        while (piv < n && memcmp(&m[piv * W + col], &z, sizeof z) == 0) piv++;
        if (piv != col) for (size_t k = 0; k < W; k++) { fr_t t = m[col * W + k]; m[col * W + k] = m[piv * W + k]; m[piv * W + k] = t; }
        fr_t inv; fr_inv(&m[col * W + col], &inv);
        for (size_t k = 0; k < W; k++) fr_mul(&m[col * W + k], &inv, &m[col * W + k]);
        for (size_t r = 0; r < n; r++) {
            if (r == col) continue;
            fr_t f = m[r * W + col];
            if (memcmp(&f, &z, sizeof z) == 0) continue;
            for (size_t k = 0; k < W; k++) { fr_t t; fr_mul(&f, &m[col * W + k], &t); fr_sub(&m[r * W + k], &t, &m[r * W + k]); }
        }
    }
    for (size_t i = 0; i < n; i++) c[i] = m[i * W + n];
}

/* UniPoly::from_evals_toom: evals on [0, 1, ..., n-2, inf] -> n coefficients */
void orc_unipoly_from_evals_toom(const fr_t *evals, size_t n, fr_t *coeffs) {
    const size_t W = n + 1;
    fr_t *m = (fr_t *)calloc(n * W, sizeof(fr_t));
    for (size_t i = 0; i + 1 < n; i++) {
        fr_t x; fr_from_u64(i, &x);
        fr_one(&m[i * W]);
        for (size_t j = 1; j < n; j++) fr_mul(&m[i * W + j - 1], &x, &m[i * W + j]);
        m[i * W + n] = evals[i];
    }
    fr_one(&m[(n - 1) * W + n - 1]);
    m[(n - 1) * W + n] = evals[n - 1];
    orc_gauss_solve(m, n, coeffs);
    free(m);
}

/* ------------------------------------------------------------------ GruenSplitEqPolynomial, LowToHigh */
static void eq_cached(const fr_t *w, size_t k, fr_t **tabs) {    /* EqPolynomial::evals_cached (eq_poly.rs:174-192) */
    tabs[0] = (fr_t *)malloc(sizeof(fr_t)); fr_one(&tabs[0][0]);
    for (size_t j = 0; j < k; j++) {
        size_t sz = (size_t)1 << j;
        tabs[j + 1] = (fr_t *)malloc(2 * sz * sizeof(fr_t));
        for (size_t i = 0; i < sz; i++) {
            fr_mul(&tabs[j][i], &w[j], &tabs[j + 1][2 * i + 1]);
            fr_sub(&tabs[j][i], &tabs[j + 1][2 * i + 1], &tabs[j + 1][2 * i]);
        }
    }
}

void gse_init(gse_t *S, const fr_t *w, size_t n) {               /* split_eq_poly.rs:97-121 */
    const size_t m = n / 2;
    S->w = (fr_t *)malloc((n ? n : 1) * sizeof(fr_t)); memcpy(S->w, w, n * sizeof(fr_t));
    S->n = n; S->k_out = m; S->k_in = n ? n - 1 - m : 0;
    S->Eout = (fr_t **)malloc((S->k_out + 1) * sizeof(fr_t *)); S->Ein = (fr_t **)malloc((S->k_in + 1) * sizeof(fr_t *));
    eq_cached(S->w, S->k_out, S->Eout); eq_cached(S->w + m, S->k_in, S->Ein);
    S->out_top = S->k_out; S->in_top = S->k_in; S->current_index = n;
    fr_one(&S->scalar);
}

void gse_free(gse_t *S) {
    for (size_t j = 0; j <= S->k_out; j++) free(S->Eout[j]);
    for (size_t j = 0; j <= S->k_in; j++) free(S->Ein[j]);
    free(S->Eout); free(S->Ein); free(S->w);
}

void gse_bind(gse_t *S, const fr_t *r) {                         /* :331-348 */
    const fr_t *wc = &S->w[S->current_index - 1];
    fr_t wr, f, one; fr_one(&one);
    fr_mul(wc, r, &wr);
    fr_sub(&one, wc, &f); fr_sub(&f, r, &f); fr_add(&f, &wr, &f); fr_add(&f, &wr, &f);
    fr_mul(&S->scalar, &f, &S->scalar);
    S->current_index -= 1;
    if (S->n / 2 < S->current_index && S->in_top > 0) S->in_top--;
    else if (0 < S->current_index && S->out_top > 0) S->out_top--;
}

/* gruen_poly_deg_3 (:379-429): 4 coefficients from q(0), q's quadratic coefficient and the claim */
void gse_gruen_deg3(const gse_t *S, const fr_t *q0, const fr_t *qinf, const fr_t *claim, fr_t coeffs[4]) {
    fr_t eq1, eq0, eqm, eq2, eq3, c0, c1, q1, q2, q3, e2, inv, hint, ev3[3];
    fr_mul(&S->scalar, &S->w[S->current_index - 1], &eq1); fr_sub(&S->scalar, &eq1, &eq0);
    fr_sub(&eq1, &eq0, &eqm); fr_add(&eq1, &eqm, &eq2); fr_add(&eq2, &eqm, &eq3);
    fr_mul(&eq0, q0, &c0); fr_sub(claim, &c0, &c1);
    fr_inv(&eq1, &inv); fr_mul(&c1, &inv, &q1);
    fr_add(qinf, qinf, &e2);
    fr_add(&q1, &q1, &q2); fr_sub(&q2, q0, &q2); fr_add(&q2, &e2, &q2);
    fr_add(&q2, &q1, &q3); fr_sub(&q3, q0, &q3); fr_add(&q3, &e2, &q3); fr_add(&q3, &e2, &q3);
    fr_add(&c0, &c1, &hint); ev3[0] = c0; fr_mul(&eq2, &q2, &ev3[1]); fr_mul(&eq3, &q3, &ev3[2]);
    orc_unipoly_from_evals_and_hint(&hint, ev3, 3, coeffs);      /* UniPoly::from_evals, degree 3, fixed length */
}

/* ------------------------------------------------------------------ RaPolynomial (materialised) */
static fr_t *ra_materialise(const int32_t *idx, size_t T, const fr_t *F) {   /* ra_poly.rs:197-207 */
    fr_t *o = (fr_t *)malloc(T * sizeof(fr_t));
    for (size_t j = 0; j < T; j++) { if (idx[j] < 0) fr_zero(&o[j]); else o[j] = F[idx[j]]; }
    return o;
}

/* ------------------------------------------------------------------ RaSumcheckProver */
void orc_ra_virtual_init(orc_ra_virtual *S, const int32_t *const *H_indices, size_t d, size_t log_k, size_t log_T,
                         const fr_t *r_address_chunks, const fr_t *r_cycle) {
    S->d = d; S->log_T = log_T; S->len = (size_t)1 << log_T;
    S->ra = (fr_t **)malloc(d * sizeof(fr_t *));
    fr_t *F = (fr_t *)malloc(((size_t)1 << log_k) * sizeof(fr_t));
    for (size_t i = 0; i < d; i++) {
        orc_eq_evals(r_address_chunks + i * log_k, log_k, 0, F);  /* ra_virtual.rs:113-116 */
        S->ra[i] = ra_materialise(H_indices[i], S->len, F);
    }
    free(F);
    gse_init(&S->eq, r_cycle, log_T);
}

void orc_ra_virtual_free(orc_ra_virtual *S) {
    for (size_t i = 0; i < S->d; i++) free(S->ra[i]);
    free(S->ra); gse_free(&S->eq);
}

/* finish_mles_product_sum_from_evals (mles_product_sum.rs:330-376); returns the coefficient count */
size_t orc_finish_product_sum(const fr_t *sum_evals, size_t d, const fr_t *claim, const gse_t *eq, fr_t *coeffs) {
    const fr_t r = eq->w[eq->current_index - 1];
    fr_t one, eq0, t, e0; fr_one(&one); fr_sub(&one, &r, &eq0);
    fr_mul(&r, &sum_evals[0], &t); fr_sub(claim, &t, &e0);
    if (d > 1) { fr_t inv; fr_inv(&eq0, &inv); fr_mul(&e0, &inv, &e0); }
    fr_t *toom = (fr_t *)malloc((d + 1) * sizeof(fr_t)), *tmp = (fr_t *)malloc((d + 1) * sizeof(fr_t));
    toom[0] = e0; memcpy(toom + 1, sum_evals, d * sizeof(fr_t));
    orc_unipoly_from_evals_toom(toom, d + 1, tmp);
    fr_t xc; fr_add(&r, &r, &xc); fr_sub(&xc, &one, &xc);        /* eq(X, r) = (1 - r) + (2r - 1) X */
    for (size_t i = 0; i < d + 2; i++) fr_zero(&coeffs[i]);
    for (size_t i = 0; i < d + 1; i++) {
        fr_mul(&tmp[i], &eq0, &t); fr_add(&coeffs[i], &t, &coeffs[i]);
        fr_mul(&tmp[i], &xc, &t); fr_add(&coeffs[i + 1], &t, &coeffs[i + 1]);
    }
    free(toom); free(tmp);
    return trim(coeffs, d + 2);
}

size_t orc_ra_virtual_message(orc_ra_virtual *S, const fr_t *claim, fr_t *coeffs) {
    const size_t d = S->d;
    const gse_t *E = &S->eq;
    const fr_t *e_out = E->Eout[E->out_top], *e_in = E->Ein[E->in_top];
    const size_t out_len = (size_t)1 << E->out_top, in_len = (size_t)1 << E->in_top;
    fr_t *sums = (fr_t *)calloc(d, sizeof(fr_t)), *inner = (fr_t *)malloc(d * sizeof(fr_t));
    fr_t *p0 = (fr_t *)malloc(d * sizeof(fr_t)), *dl = (fr_t *)malloc(d * sizeof(fr_t));
    for (size_t xo = 0; xo < out_len; xo++) {
        for (size_t k = 0; k < d; k++) fr_zero(&inner[k]);
        for (size_t xi = 0; xi < in_len; xi++) {
            const size_t g = (xo << E->in_top) | xi;
            for (size_t i = 0; i < d; i++) { p0[i] = S->ra[i][2 * g]; fr_sub(&S->ra[i][2 * g + 1], &p0[i], &dl[i]); }
            /* eval_linear_prod_assign: P on [1, 2, ..., d-1, inf] */
            for (size_t k = 0; k < d; k++) {
                fr_t prod; fr_one(&prod);
                for (size_t i = 0; i < d; i++) {
                    if (k + 1 < d) fr_add(&p0[i], &dl[i], &p0[i]);   /* p_i(k+1) */
                    fr_mul(&prod, k + 1 < d ? &p0[i] : &dl[i], &prod);
                }
                fr_t t; fr_mul(&e_in[xi], &prod, &t); fr_add(&inner[k], &t, &inner[k]);
            }
        }
        for (size_t k = 0; k < d; k++) { fr_t t; fr_mul(&e_out[xo], &inner[k], &t); fr_add(&sums[k], &t, &sums[k]); }
    }
    for (size_t k = 0; k < d; k++) fr_mul(&sums[k], &E->scalar, &sums[k]);   /* :131 */
    size_t n = orc_finish_product_sum(sums, d, claim, E, coeffs);
    free(sums); free(inner); free(p0); free(dl);
    return n;
}

void orc_ra_virtual_ingest(orc_ra_virtual *S, const fr_t *r) {
    for (size_t i = 0; i < S->d; i++) orc_bind(S->ra[i], S->len, r, ORC_LOW_TO_HIGH);
    S->len /= 2;
    gse_bind(&S->eq, r);
}

/* ------------------------------------------------------------------ BooleanitySumcheckProver */
void orc_booleanity_init(orc_booleanity *S, const fr_t *G, const int32_t *const *H_indices, size_t d, size_t log_k,
                         size_t log_T, const fr_t *gammas, const fr_t *r_address, const fr_t *r_cycle) {
    const size_t K = (size_t)1 << log_k;
    S->d = d; S->log_k = log_k; S->log_T = log_T; S->len = (size_t)1 << log_T;
    S->G = (fr_t *)malloc(d * K * sizeof(fr_t)); memcpy(S->G, G, d * K * sizeof(fr_t));
    S->H_indices = H_indices; S->H = 0;
    S->gammas = (fr_t *)malloc(d * sizeof(fr_t)); memcpy(S->gammas, gammas, d * sizeof(fr_t));
    gse_init(&S->B, r_address, log_k); gse_init(&S->D, r_cycle, log_T);
    S->F = (fr_t *)calloc(K, sizeof(fr_t)); fr_one(&S->F[0]); S->F_len = 1;   /* ExpandingTable::reset(1) */
    fr_zero(&S->eq_r_r);
}

void orc_booleanity_free(orc_booleanity *S) {
    if (S->H) { for (size_t i = 0; i < S->d; i++) free(S->H[i]); free(S->H); }
    free(S->G); free(S->gammas); free(S->F); gse_free(&S->B); gse_free(&S->D);
}

size_t orc_booleanity_message(orc_booleanity *S, size_t round, const fr_t *claim, fr_t *coeffs) {
    fr_t q0, qinf; fr_zero(&q0); fr_zero(&qinf);
    if (round < S->log_k) {                                       /* compute_phase1_message :193-252 */
        const size_t m = round + 1, K = (size_t)1 << S->log_k;
        const gse_t *E = &S->B;
        const fr_t *e_out = E->Eout[E->out_top], *e_in = E->Ein[E->in_top];
        const size_t out_len = (size_t)1 << E->out_top, in_len = (size_t)1 << E->in_top;
        for (size_t xo = 0; xo < out_len; xo++) {
            fr_t i0, i1; fr_zero(&i0); fr_zero(&i1);
            for (size_t xi = 0; xi < in_len; xi++) {
                const size_t kp = (xo << E->in_top) | xi;
                fr_t c0, c1; fr_zero(&c0); fr_zero(&c1);
                for (size_t i = 0; i < S->d; i++) {
                    fr_t s0, s1; fr_zero(&s0); fr_zero(&s1);
                    for (size_t k = 0; k < ((size_t)1 << m); k++) {
                        const fr_t *Gk = &S->G[i * K + (kp << m) + k];
                        const size_t k_m = k >> (m - 1);
                        const fr_t *Fk = &S->F[k % ((size_t)1 << (m - 1))];
                        fr_t gf, ei; fr_mul(Gk, Fk, &gf); fr_mul(&gf, Fk, &ei);
                        if (k_m == 0) { fr_t t; fr_sub(&ei, &gf, &t); fr_add(&s0, &t, &s0); }
                        fr_add(&s1, &ei, &s1);
                    }
                    fr_mul(&S->gammas[i], &s0, &s0); fr_mul(&S->gammas[i], &s1, &s1);
                    fr_add(&c0, &s0, &c0); fr_add(&c1, &s1, &c1);
                }
                fr_mul(&e_in[xi], &c0, &c0); fr_mul(&e_in[xi], &c1, &c1);
                fr_add(&i0, &c0, &i0); fr_add(&i1, &c1, &i1);
            }
            fr_mul(&e_out[xo], &i0, &i0); fr_mul(&e_out[xo], &i1, &i1);
            fr_add(&q0, &i0, &q0); fr_add(&qinf, &i1, &qinf);
        }
        gse_gruen_deg3(E, &q0, &qinf, claim, coeffs);
        return 4;
    }
    /* compute_phase2_message :254-288 */
    const gse_t *E = &S->D;
    const fr_t *e_out = E->Eout[E->out_top], *e_in = E->Ein[E->in_top];
    const size_t out_len = (size_t)1 << E->out_top, in_len = (size_t)1 << E->in_top;
    fr_t one; fr_one(&one);
    for (size_t xo = 0; xo < out_len; xo++) {
        fr_t i0, i1; fr_zero(&i0); fr_zero(&i1);
        for (size_t xi = 0; xi < in_len; xi++) {
            const size_t j = (xo << E->in_top) | xi;
            fr_t ac, ae; fr_zero(&ac); fr_zero(&ae);
            for (size_t i = 0; i < S->d; i++) {
                const fr_t h0 = S->H[i][2 * j], h1 = S->H[i][2 * j + 1];
                fr_t b, t, u; fr_sub(&h1, &h0, &b);
                fr_mul(&S->gammas[i], &h0, &t); fr_sub(&h0, &one, &u); fr_mul(&t, &u, &t); fr_add(&ac, &t, &ac);
                fr_mul(&S->gammas[i], &b, &t); fr_mul(&t, &b, &t); fr_add(&ae, &t, &ae);
            }
            fr_mul(&e_in[xi], &ac, &ac); fr_mul(&e_in[xi], &ae, &ae);
            fr_add(&i0, &ac, &i0); fr_add(&i1, &ae, &i1);
        }
        fr_mul(&e_out[xo], &i0, &i0); fr_mul(&e_out[xo], &i1, &i1);
        fr_add(&q0, &i0, &q0); fr_add(&qinf, &i1, &qinf);
    }
    fr_t inv, adj; fr_inv(&S->eq_r_r, &inv); fr_mul(claim, &inv, &adj);
    gse_gruen_deg3(E, &q0, &qinf, &adj, coeffs);
    for (int k = 0; k < 4; k++) fr_mul(&coeffs[k], &S->eq_r_r, &coeffs[k]);   /* gruen_poly * eq_r_r -> from_coeff */
    return trim(coeffs, 4);
}

void orc_booleanity_ingest(orc_booleanity *S, size_t round, const fr_t *r) {   /* :316-348 */
    if (round < S->log_k) {
        gse_bind(&S->B, r);
        for (size_t x = 0; x < S->F_len; x++) {                   /* ExpandingTable::update, LowToHigh */
            fr_mul(&S->F[x], r, &S->F[S->F_len + x]);
            fr_sub(&S->F[x], &S->F[S->F_len + x], &S->F[x]);
        }
        S->F_len *= 2;
        if (round == S->log_k - 1) {
            S->eq_r_r = S->B.scalar;
            S->H = (fr_t **)malloc(S->d * sizeof(fr_t *));
            for (size_t i = 0; i < S->d; i++) S->H[i] = ra_materialise(S->H_indices[i], S->len, S->F);
        }
    } else {
        gse_bind(&S->D, r);
        for (size_t i = 0; i < S->d; i++) orc_bind(S->H[i], S->len, r, ORC_LOW_TO_HIGH);
        S->len /= 2;
    }
}

/* ------------------------------------------------------------------ HammingWeightSumcheckProver */
void orc_hamming_init(orc_hamming *S, const fr_t *G, size_t d, size_t log_k, const fr_t *gamma_powers) {
    const size_t K = (size_t)1 << log_k;
    S->d = d; S->log_k = log_k; S->len = K;
    S->ra = (fr_t *)malloc(d * K * sizeof(fr_t)); memcpy(S->ra, G, d * K * sizeof(fr_t));
    S->gamma_powers = (fr_t *)malloc(d * sizeof(fr_t)); memcpy(S->gamma_powers, gamma_powers, d * sizeof(fr_t));
    S->stride = K;
}

void orc_hamming_free(orc_hamming *S) { free(S->ra); free(S->gamma_powers); }

size_t orc_hamming_message(orc_hamming *S, const fr_t *claim, fr_t *coeffs) {   /* :118-140 */
    fr_t e0; fr_zero(&e0);
    for (size_t i = 0; i < S->d; i++) {
        fr_t s; fr_zero(&s);
        for (size_t k = 0; k < S->len / 2; k++) fr_add(&s, &S->ra[i * S->stride + 2 * k], &s);
        fr_mul(&s, &S->gamma_powers[i], &s); fr_add(&e0, &s, &e0);
    }
    /* from_evals_and_hint(claim, [e0]) -> from_evals of 2 points -> vandermonde -> from_coeff */
    coeffs[0] = e0;
    fr_t e1; fr_sub(claim, &e0, &e1); fr_sub(&e1, &e0, &coeffs[1]);
    return trim(coeffs, 2);
}

void orc_hamming_ingest(orc_hamming *S, const fr_t *r) {
    for (size_t i = 0; i < S->d; i++) orc_bind(&S->ra[i * S->stride], S->len, r, ORC_LOW_TO_HIGH);
    S->len /= 2;
}

/* ------------------------------------------------------------------ Sumcheck::prove over one such instance
 * (sumcheck.rs:565-599); rows of `stride` Fr hold n_coeffs[i] compressed coefficients */
static size_t inst_message(int kind, void *st, size_t round, const fr_t *claim, fr_t *c) {
    switch (kind) {
        case ORC_INST_RA_VIRTUAL: return orc_ra_virtual_message((orc_ra_virtual *)st, claim, c);
        case ORC_INST_BOOLEANITY: return orc_booleanity_message((orc_booleanity *)st, round, claim, c);
        case ORC_INST_DENSE_OPENING: return orc_dense_opening_message((orc_dense_opening *)st, claim, c);
        case ORC_INST_ONEHOT_OPENING: return orc_onehot_opening_message((orc_onehot_opening *)st, round, claim, c);
        case ORC_INST_PS_RELU: return orc_ps_relu_message((orc_ps_relu *)st, round, claim, c);
        case ORC_INST_PS_IDENTITY: return orc_ps_identity_message((orc_ps_identity *)st, round, claim, c);
        case ORC_INST_PS_CLAMP: return orc_ps_clamp_message((orc_ps_clamp *)st, round, claim, c);
        case ORC_INST_PS_ULT: return orc_ps_ult_message((orc_ps_ult *)st, round, claim, c);
        case ORC_INST_ELEMENTWISE: return orc_elementwise_message((orc_elementwise *)st, claim, c);
        case ORC_INST_SOFTMAX: return orc_softmax_message((orc_softmax *)st, round, claim, c);
        case ORC_INST_SHOUT: return orc_shout_inst_message((orc_shout_inst *)st, claim, c);
        default: return orc_hamming_message((orc_hamming *)st, claim, c);
    }
}
static void inst_ingest(int kind, void *st, size_t round, const fr_t *r) {
    switch (kind) {
        case ORC_INST_RA_VIRTUAL: orc_ra_virtual_ingest((orc_ra_virtual *)st, r); break;
        case ORC_INST_BOOLEANITY: orc_booleanity_ingest((orc_booleanity *)st, round, r); break;
        case ORC_INST_DENSE_OPENING: orc_dense_opening_ingest((orc_dense_opening *)st, r); break;
        case ORC_INST_ONEHOT_OPENING: orc_onehot_opening_ingest((orc_onehot_opening *)st, round, r); break;
        case ORC_INST_PS_RELU: orc_ps_relu_ingest((orc_ps_relu *)st, round, r); break;
        case ORC_INST_PS_IDENTITY: orc_ps_identity_ingest((orc_ps_identity *)st, round, r); break;
        case ORC_INST_PS_CLAMP: orc_ps_clamp_ingest((orc_ps_clamp *)st, round, r); break;
        case ORC_INST_PS_ULT: orc_ps_ult_ingest((orc_ps_ult *)st, round, r); break;
        case ORC_INST_ELEMENTWISE: orc_elementwise_ingest((orc_elementwise *)st, r); break;
        case ORC_INST_SOFTMAX: orc_softmax_ingest((orc_softmax *)st, round, r); break;
        case ORC_INST_SHOUT: orc_shout_inst_ingest((orc_shout_inst *)st, r); break;
        default: orc_hamming_ingest((orc_hamming *)st, r); break;
    }
}

size_t orc_ra_inst_message(int kind, void *st, size_t round, const fr_t *claim, fr_t *c) { return inst_message(kind, st, round, claim, c); }
void orc_ra_inst_ingest(int kind, void *st, size_t round, const fr_t *r) { inst_ingest(kind, st, round, r); }

int orc_ra_inst_prove(int kind, void *st, size_t n_rounds, const fr_t *input_claim, orc_transcript *t, fr_t *compressed,
                      size_t stride, uint32_t *n_coeffs, u128 *challenges) {
    orc_transcript_append_scalar(t, input_claim);
    fr_t prev = *input_claim;
    fr_t coeffs[40], cc[40];
    for (size_t rnd = 0; rnd < n_rounds; rnd++) {
        size_t nc = inst_message(kind, st, rnd, &prev, coeffs);
        size_t ncc = orc_unipoly_compress(coeffs, nc, cc);
        orc_transcript_append_compressed(t, cc, ncc);
        fr_t r; u128 raw; orc_transcript_challenge_optimized(t, &raw, &r);
        challenges[rnd] = raw; n_coeffs[rnd] = (uint32_t)ncc;
        for (size_t k = 0; k < ncc; k++) compressed[rnd * stride + k] = cc[k];
        orc_unipoly_eval(coeffs, nc, &r, &prev);
        inst_ingest(kind, st, rnd, &r);
    }
    return 0;
}
