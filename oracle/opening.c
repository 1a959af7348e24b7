/* TEST INFRASTRUCTURE ONLY (oracle): opening-reduction sumcheck provers restated from
 * /root/reference/joltworks/src:
 *   GruenSplitEqPolynomial (HighToLow)   poly/split_eq_poly.rs:121-145,349-372,379-428,475-493
 *   EqPolynomial::evals_cached_rev       poly/eq_poly.rs:193-217
 *   DensePolynomialProverOpening         subprotocols/opening_reduction.rs:355-425
 *   OneHotPolynomialProverOpening        subprotocols/opening_reduction.rs:532-723
 *   ExpandingTable (HighToLow)           utils/expanding_table.rs:75-85
 * Shared Eq*State objects only avoid re-binding the same table twice; each opening here carries its
 * own copy (same values). */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "ra.h"
#include "opening.h"

/* result[j][x] = eq(r[(n-j)..], x); index bit 0 of x <-> r[n-1]?  No: as written in eq_poly.rs:
 * evals[j+1][i + 2^j] = evals[j][i] * rev_r[j]  — the variable added at step j is bit j of x. */
static fr_t **cached_rev(const fr_t *r, size_t n) {
    fr_t **t = (fr_t **)malloc((n + 1) * sizeof(fr_t *));
    t[0] = (fr_t *)malloc(sizeof(fr_t)); fr_one(&t[0][0]);
    for (size_t j = 0; j < n; j++) {
        const size_t size = (size_t)1 << j;
        t[j + 1] = (fr_t *)malloc(2 * size * sizeof(fr_t));
        const fr_t *rv = &r[n - 1 - j];
        for (size_t i = 0; i < size; i++) {
            fr_mul(&t[j][i], rv, &t[j + 1][i + size]);
            fr_sub(&t[j][i], &t[j + 1][i + size], &t[j + 1][i]);
        }
    }
    return t;
}

void gseh_init(gseh_t *S, const fr_t *w, size_t n) {
    S->n = n; S->current_index = 0; fr_one(&S->scalar);
    S->w = (fr_t *)malloc((n ? n : 1) * sizeof(fr_t)); memcpy(S->w, w, n * sizeof(fr_t));
    const size_t m = n / 2;                      /* [w_first | w_in (m) | w_out (n-1-m)] */
    S->k_in = n ? (m < n - 1 ? m : n - 1) : 0;   /* wprime.split_at(m): wprime has n-1 elements */
    S->k_out = n ? n - 1 - S->k_in : 0;
    S->Ein = cached_rev(w + (n ? 1 : 0), S->k_in);
    S->Eout = cached_rev(w + (n ? 1 : 0) + S->k_in, S->k_out);
    S->in_top = S->k_in; S->out_top = S->k_out;
}

void gseh_free(gseh_t *S) {
    for (size_t j = 0; j <= S->k_in; j++) free(S->Ein[j]);
    for (size_t j = 0; j <= S->k_out; j++) free(S->Eout[j]);
    free(S->Ein); free(S->Eout); free(S->w);
}

void gseh_bind(gseh_t *S, const fr_t *r) {      /* split_eq_poly.rs:349-371 */
    const fr_t *wc = &S->w[S->current_index];
    fr_t wr, f, one; fr_one(&one);
    fr_mul(wc, r, &wr);
    fr_sub(&one, wc, &f); fr_sub(&f, r, &f); fr_add(&f, &wr, &f); fr_add(&f, &wr, &f);
    fr_mul(&S->scalar, &f, &S->scalar);
    S->current_index += 1;
    if (S->current_index <= S->n / 2 && S->in_top > 0) S->in_top--;
    else if (S->current_index <= S->n && S->out_top > 0) S->out_top--;
}

/* gruen_poly_deg_2 (:379-428): 3 coefficients (UniPoly::from_evals of 3 points, fixed length) */
static void gseh_gruen_deg2(const gseh_t *S, const fr_t *q0, const fr_t *claim, fr_t coeffs[3]) {
    fr_t eq1, eq0, eqm, eq2, c0, c1, l1, l2, inv, ev2[2], hint;
    fr_mul(&S->scalar, &S->w[S->current_index], &eq1); fr_sub(&S->scalar, &eq1, &eq0);
    fr_sub(&eq1, &eq0, &eqm); fr_add(&eq1, &eqm, &eq2);
    fr_mul(&eq0, q0, &c0); fr_sub(claim, &c0, &c1);
    fr_inv(&eq1, &inv); fr_mul(&c1, &inv, &l1);
    fr_add(&l1, &l1, &l2); fr_sub(&l2, q0, &l2);
    ev2[0] = c0; fr_mul(&eq2, &l2, &ev2[1]); fr_add(&c0, &c1, &hint);
    orc_unipoly_from_evals_and_hint(&hint, ev2, 2, coeffs);
}

/* sum over the lower half of P weighted by the current split tables (:355-401 / :634-668) */
static void gseh_q0(const gseh_t *S, const fr_t *P, fr_t *out) {
    const fr_t *e_in = S->Ein[S->in_top], *e_out = S->Eout[S->out_top];
    const size_t num_in = (size_t)1 << S->in_top, num_out = (size_t)1 << S->out_top;
    fr_t acc; fr_zero(&acc);
    for (size_t xi = 0; xi < num_in; xi++) {
        fr_t inner; fr_zero(&inner);
        for (size_t xo = 0; xo < num_out; xo++) {
            const size_t j = (xi << S->out_top) | xo;
            fr_t t; fr_mul(&e_out[xo], &P[j], &t); fr_add(&inner, &t, &inner);
        }
        fr_mul(&e_in[xi], &inner, &inner); fr_add(&acc, &inner, &acc);
    }
    *out = acc;
}

/* ------------------------------------------------------------------ dense opening */
void orc_dense_opening_init(orc_dense_opening *S, const fr_t *poly, size_t n, const fr_t *point) {
    S->n = n; S->len = (size_t)1 << n;
    S->P = (fr_t *)malloc(S->len * sizeof(fr_t)); memcpy(S->P, poly, S->len * sizeof(fr_t));
    gseh_init(&S->D, point, n);
}
void orc_dense_opening_free(orc_dense_opening *S) { free(S->P); gseh_free(&S->D); }

size_t orc_dense_opening_message(orc_dense_opening *S, const fr_t *claim, fr_t *coeffs) {
    fr_t q0; gseh_q0(&S->D, S->P, &q0);
    gseh_gruen_deg2(&S->D, &q0, claim, coeffs);
    return 3;
}
void orc_dense_opening_ingest(orc_dense_opening *S, const fr_t *r) {
    gseh_bind(&S->D, r);
    orc_bind(S->P, S->len, r, ORC_HIGH_TO_LOW); S->len /= 2;
}

/* ------------------------------------------------------------------ one-hot opening */
void orc_onehot_opening_init(orc_onehot_opening *S, const int32_t *idx, size_t log_K, size_t log_T,
                             const fr_t *r_address, const fr_t *r_cycle) {
    const size_t K = (size_t)1 << log_K, T = (size_t)1 << log_T;
    S->log_K = log_K; S->log_T = log_T; S->idx = idx; S->H = 0; S->H_len = T;
    S->B = (fr_t *)malloc(K * sizeof(fr_t)); orc_eq_evals(r_address, log_K, 0, S->B); S->B_len = K;   /* EqAddressState::new */
    S->F = (fr_t *)calloc(K, sizeof(fr_t)); fr_one(&S->F[0]); S->F_len = 1;
    gseh_init(&S->D, r_cycle, log_T);
    /* initialize (:532-575): G[k] = sum_{j: idx_j = k} D.merge()[j], merge = eq(r_cycle, .) * scalar(=1) */
    fr_t *E = (fr_t *)malloc(T * sizeof(fr_t)); orc_eq_evals(r_cycle, log_T, 0, E);
    S->G = (fr_t *)calloc(K, sizeof(fr_t));
    for (size_t j = 0; j < T; j++) if (idx[j] >= 0) fr_add(&S->G[idx[j]], &E[j], &S->G[idx[j]]);
    free(E);
}
void orc_onehot_opening_free(orc_onehot_opening *S) { free(S->B); free(S->F); free(S->G); if (S->H) free(S->H); gseh_free(&S->D); }

size_t orc_onehot_opening_message(orc_onehot_opening *S, size_t round, const fr_t *claim, fr_t *coeffs) {
    if (round < S->log_K) {                                     /* :583-633 */
        const size_t unbound = S->log_K - round, K = (size_t)1 << S->log_K, half = S->B_len / 2;
        fr_t e0, e2; fr_zero(&e0); fr_zero(&e2);
        for (size_t kp = 0; kp < half; kp++) {
            fr_t b0 = S->B[kp], b2, m; fr_sub(&S->B[kp + half], &b0, &m); fr_add(&S->B[kp + half], &m, &b2);
            fr_t s0, s2; fr_zero(&s0); fr_zero(&s2);
            for (size_t k = kp; k < K; k += half) {
                const size_t k_m = (k >> (unbound - 1)) & 1;
                fr_t gf; fr_mul(&S->G[k], &S->F[k >> unbound], &gf);
                if (k_m == 0) { fr_add(&s0, &gf, &s0); fr_sub(&s2, &gf, &s2); }
                else { fr_add(&s2, &gf, &s2); fr_add(&s2, &gf, &s2); }
            }
            fr_mul(&b0, &s0, &s0); fr_mul(&b2, &s2, &s2);
            fr_add(&e0, &s0, &e0); fr_add(&e2, &s2, &e2);
        }
        fr_t ev[2] = {e0, e2};
        return orc_unipoly_from_evals_and_hint(claim, ev, 2, coeffs);
    }
    fr_t q0; gseh_q0(&S->D, S->H, &q0);                          /* :634-676 */
    const fr_t eqa = S->B[0];
    fr_t inv, adj; fr_inv(&eqa, &inv); fr_mul(claim, &inv, &adj);
    gseh_gruen_deg2(&S->D, &q0, &adj, coeffs);
    for (int k = 0; k < 3; k++) fr_mul(&coeffs[k], &eqa, &coeffs[k]);
    size_t n = 3; fr_t z; fr_zero(&z);
    while (n > 0 && memcmp(&coeffs[n - 1], &z, sizeof z) == 0) n--;
    if (n == 0) { coeffs[0] = z; n = 1; }
    return n;
}

void orc_onehot_opening_ingest(orc_onehot_opening *S, size_t round, const fr_t *r) {   /* :679-718 */
    if (round < S->log_K) {
        orc_bind(S->B, S->B_len, r, ORC_HIGH_TO_LOW); S->B_len /= 2;
        fr_t *nf = (fr_t *)calloc((size_t)1 << S->log_K, sizeof(fr_t));   /* ExpandingTable::update, HighToLow */
        for (size_t i = 0; i < S->F_len; i++) { fr_mul(r, &S->F[i], &nf[2 * i + 1]); fr_sub(&S->F[i], &nf[2 * i + 1], &nf[2 * i]); }
        free(S->F); S->F = nf; S->F_len *= 2;
        if (round == S->log_K - 1) {
            const size_t T = (size_t)1 << S->log_T;
            S->H = (fr_t *)malloc(T * sizeof(fr_t));
            for (size_t j = 0; j < T; j++) { if (S->idx[j] < 0) fr_zero(&S->H[j]); else S->H[j] = S->F[S->idx[j]]; }
        }
    } else {
        gseh_bind(&S->D, r);
        orc_bind(S->H, S->H_len, r, ORC_HIGH_TO_LOW); S->H_len /= 2;
    }
}
