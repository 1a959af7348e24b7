/* TEST INFRASTRUCTURE ONLY — CPU oracle, never linked into or called by the product.
 * parity unpinned by reference vectors (the reference holds none for these provers); pinned against the dense-table
 * models in oracle/pymodel/ra.py (tests/test_oracle_ra.py).
 *
 * The two-phase provers of jolt-atlas-core/src/onnx_proof/ops/softmax_last_axis/ over tensors laid out [k][j]
 * (K rows, N = last-axis length), all bound LowToHigh — the log_N last-axis variables first:
 *   exp_sum.rs:146-197     ExpSumProver        phase 1: [sum exp_q(2kj) eq_r0_k(k)], k = kj >> (log_N - m)
 *                                               (from_evals_and_hint with ONE evaluation: degree 1, from_coeff);
 *                                               phase 2: Gruen over r0_k, [exp_q(2g)], gruen_poly_deg_2
 *   max.rs:185-261         MaxIndicatorProver  phase 1: eq(k) X e at 0, 2, 3 (from_evals_and_hint, 4 coefficients);
 *                                               phase 2: [i0 X0, i_inf X_inf], gruen_poly_deg_3
 *   recip_mult.rs:196-268  RecipMultProver     one Gruen eq over all log_K + log_N variables; phase 1:
 *                                               [exp_q(2kj) inv_sum(k)] gruen_poly_deg_2; phase 2: [e0 s0, e_inf s_inf]
 *                                               gruen_poly_deg_3; inv_sum is bound in phase 2 only
 * and ops/sum/axis.rs:220-232 SumAxisProver: [sum operand(i), i < half] HighToLow, one evaluation + hint. */
#include <stdlib.h>
#include <string.h>
#include "softmax.h"

static size_t trim2(fr_t *c) {                   /* from_coeff of a 2-coefficient vector (unipoly.rs:39-52) */
    fr_t z; fr_zero(&z); size_t n = 2;
    while (n > 0 && memcmp(&c[n - 1], &z, sizeof z) == 0) n--;
    if (n == 0) { c[0] = z; n = 1; }
    return n;
}
static size_t linear_from_eval0(const fr_t *claim, const fr_t *e0, fr_t *coeffs) {   /* from_evals_and_hint(claim, [e0]) */
    fr_t e1; coeffs[0] = *e0; fr_sub(claim, e0, &e1); fr_sub(&e1, e0, &coeffs[1]);
    return trim2(coeffs);
}
static size_t gruen_deg2(const gse_t *E, const fr_t *q0, const fr_t *claim, fr_t *coeffs) {   /* split_eq_poly.rs:379-428 */
    fr_t eq1, eq0, eqm, eq2, c0, c1, l1, l2, inv, ev2[2], hint;
    fr_mul(&E->scalar, &E->w[E->current_index - 1], &eq1); fr_sub(&E->scalar, &eq1, &eq0);
    fr_sub(&eq1, &eq0, &eqm); fr_add(&eq1, &eqm, &eq2);
    fr_mul(&eq0, q0, &c0); fr_sub(claim, &c0, &c1);
    fr_inv(&eq1, &inv); fr_mul(&c1, &inv, &l1);
    fr_add(&l1, &l1, &l2); fr_sub(&l2, q0, &l2);
    ev2[0] = c0; fr_mul(&eq2, &l2, &ev2[1]); fr_add(&c0, &c1, &hint);
    return orc_unipoly_from_evals_and_hint(&hint, ev2, 2, coeffs);
}
static void evals023(const fr_t *z, size_t i, fr_t o[3]) {     /* sumcheck_evals(i, 3, LowToHigh): values at 0, 2, 3 */
    fr_t m; o[0] = z[2 * i]; fr_sub(&z[2 * i + 1], &z[2 * i], &m); fr_add(&z[2 * i + 1], &m, &o[1]); fr_add(&o[1], &m, &o[2]);
}

void orc_softmax_init(orc_softmax *S, int kind, const fr_t *a, const fr_t *b, size_t log_K, size_t log_N, const fr_t *r) {
    memset(S, 0, sizeof *S);
    S->kind = kind; S->log_K = log_K; S->log_N = log_N; S->len = (size_t)1 << (log_K + log_N);
    S->a = (fr_t *)malloc(S->len * sizeof(fr_t)); memcpy(S->a, a, S->len * sizeof(fr_t));
    if (kind == ORC_SM_MAX_INDICATOR) { S->len2 = S->len; S->b = (fr_t *)malloc(S->len * sizeof(fr_t)); memcpy(S->b, b, S->len * sizeof(fr_t)); }
    if (kind == ORC_SM_RECIP_MULT) { S->len2 = (size_t)1 << log_K; S->b = (fr_t *)malloc(S->len2 * sizeof(fr_t)); memcpy(S->b, b, S->len2 * sizeof(fr_t)); }
    if (kind == ORC_SM_EXP_SUM || kind == ORC_SM_MAX_INDICATOR) {
        S->eq_k = (fr_t *)malloc(((size_t)1 << log_K) * sizeof(fr_t)); orc_eq_evals(r, log_K, 0, S->eq_k);
        S->r_k = (fr_t *)malloc((log_K ? log_K : 1) * sizeof(fr_t)); memcpy(S->r_k, r, log_K * sizeof(fr_t));
    }
    if (kind == ORC_SM_RECIP_MULT) { gse_init(&S->gs, r, log_K + log_N); S->gs_live = 1; }
}

void orc_softmax_free(orc_softmax *S) {
    free(S->a); if (S->b) free(S->b); if (S->eq_k) free(S->eq_k); if (S->r_k) free(S->r_k);
    if (S->gs_live) gse_free(&S->gs);
}

/* sum_{x_out} E_out sum_{x_in} E_in f(g) for up to two outputs */
typedef void (*fold_fn)(const orc_softmax *S, size_t g, fr_t out[2]);
static void fold(const orc_softmax *S, fold_fn f, int n_out, fr_t q[2]) {
    const gse_t *E = &S->gs;
    const fr_t *e_out = E->Eout[E->out_top], *e_in = E->Ein[E->in_top];
    const size_t out_len = (size_t)1 << E->out_top, in_len = (size_t)1 << E->in_top;
    fr_zero(&q[0]); fr_zero(&q[1]);
    for (size_t xo = 0; xo < out_len; xo++) {
        fr_t in[2]; fr_zero(&in[0]); fr_zero(&in[1]);
        for (size_t xi = 0; xi < in_len; xi++) {
            fr_t v[2], t; f(S, (xo << E->in_top) | xi, v);
            for (int k = 0; k < n_out; k++) { fr_mul(&e_in[xi], &v[k], &t); fr_add(&in[k], &t, &in[k]); }
        }
        for (int k = 0; k < n_out; k++) { fr_t t; fr_mul(&e_out[xo], &in[k], &t); fr_add(&q[k], &t, &q[k]); }
    }
}
static void f_exp2(const orc_softmax *S, size_t g, fr_t o[2]) { o[0] = S->a[2 * g]; }
static void f_prod(const orc_softmax *S, size_t g, fr_t o[2]) {             /* [a0 b0, a_inf b_inf] */
    fr_t da, db; fr_mul(&S->a[2 * g], &S->b[2 * g], &o[0]);
    fr_sub(&S->a[2 * g + 1], &S->a[2 * g], &da); fr_sub(&S->b[2 * g + 1], &S->b[2 * g], &db); fr_mul(&da, &db, &o[1]);
}
static void f_recip1(const orc_softmax *S, size_t kj, fr_t o[2]) {           /* exp_q(2kj) inv_sum(k), k = kj >> (log_N - m) */
    const size_t m = S->round + 1;
    fr_mul(&S->a[2 * kj], &S->b[kj >> (S->log_N - m)], &o[0]);
}

size_t orc_softmax_message(orc_softmax *S, size_t round, const fr_t *claim, fr_t *coeffs) {
    S->round = round;
    const size_t half = S->len / 2;
    if (S->kind == ORC_SM_SUM_AXIS) {
        fr_t e0; fr_zero(&e0);
        for (size_t i = 0; i < half; i++) fr_add(&e0, &S->a[i], &e0);
        return linear_from_eval0(claim, &e0, coeffs);
    }
    const size_t m = round + 1;
    if (round < S->log_N) {
        if (S->kind == ORC_SM_EXP_SUM) {
            fr_t e0, t; fr_zero(&e0);
            for (size_t kj = 0; kj < half; kj++) { fr_mul(&S->a[2 * kj], &S->eq_k[kj >> (S->log_N - m)], &t); fr_add(&e0, &t, &e0); }
            return linear_from_eval0(claim, &e0, coeffs);
        }
        if (S->kind == ORC_SM_MAX_INDICATOR) {
            fr_t ev[3], t; for (int k = 0; k < 3; k++) fr_zero(&ev[k]);
            for (size_t kj = 0; kj < half; kj++) {
                fr_t x[3], e[3]; evals023(S->a, kj, x); evals023(S->b, kj, e);
                for (int k = 0; k < 3; k++) { fr_mul(&S->eq_k[kj >> (S->log_N - m)], &x[k], &t); fr_mul(&t, &e[k], &t); fr_add(&ev[k], &t, &ev[k]); }
            }
            return orc_unipoly_from_evals_and_hint(claim, ev, 3, coeffs);
        }
        fr_t q[2]; fold(S, f_recip1, 1, q);
        return gruen_deg2(&S->gs, &q[0], claim, coeffs);
    }
    if (S->kind == ORC_SM_EXP_SUM) { fr_t q[2]; fold(S, f_exp2, 1, q); return gruen_deg2(&S->gs, &q[0], claim, coeffs); }
    fr_t q[2]; fold(S, f_prod, 2, q);
    gse_gruen_deg3(&S->gs, &q[0], &q[1], claim, coeffs);
    return 4;
}

void orc_softmax_ingest(orc_softmax *S, size_t round, const fr_t *r) {
    if (S->kind == ORC_SM_SUM_AXIS) { orc_bind(S->a, S->len, r, ORC_HIGH_TO_LOW); S->len /= 2; return; }
    orc_bind(S->a, S->len, r, ORC_LOW_TO_HIGH); S->len /= 2;
    if (S->kind == ORC_SM_MAX_INDICATOR) { orc_bind(S->b, S->len2, r, ORC_LOW_TO_HIGH); S->len2 /= 2; }
    if (S->kind == ORC_SM_RECIP_MULT) {
        gse_bind(&S->gs, r);
        if (round >= S->log_N) { orc_bind(S->b, S->len2, r, ORC_LOW_TO_HIGH); S->len2 /= 2; }
        return;
    }
    if (round + 1 == S->log_N) { gse_init(&S->gs, S->r_k, S->log_K); S->gs_live = 1; }
    if (round >= S->log_N) gse_bind(&S->gs, r);
}

void orc_softmax_finals(const orc_softmax *S, fr_t *out) { out[0] = S->a[0]; if (S->b) out[1] = S->b[0]; }
