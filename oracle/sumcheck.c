/* ORACLE — TEST INFRASTRUCTURE ONLY (see bn254.h header note; parity unpinned).
 * CPU restatement of the sumcheck hot loop.  Reference (under /root/reference):
 *   joltworks/src/poly/eq_poly.rs:149-167,225-252          EQ tables
 *   joltworks/src/poly/dense_mlpoly.rs:84-141,209-305      bind / evaluate
 *   joltworks/src/poly/compact_polynomial.rs:272-353       small-scalar first bind
 *   joltworks/src/poly/multilinear_polynomial.rs:873-905   sumcheck_evals
 *   joltworks/src/poly/unipoly.rs:55-133,219-245,307-318,502-558
 *   joltworks/src/subprotocols/sumcheck.rs:565-599,653-686 driver / verifier
 *   jolt-atlas-core/src/onnx_proof/ops/einsum/dot.rs:290-375   dot-product instance
 * Rayon's chunked map-reduce is mirrored with OpenMP; all reductions are over exact
 * field elements so the association order cannot change a result. */
#include "oracle.h"
#include "ra.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ EQ */
void orc_eq_evals(const fr_t *r, size_t n, const fr_t *scaling, fr_t *out) {
    /* evals_parallel (eq_poly.rs:225-252): iterate r reversed, doubling; r[0] = MSB */
    if (scaling) out[0] = *scaling; else fr_one(&out[0]);
    size_t size = 1;
    for (size_t k = n; k-- > 0;) {
        const fr_t rk = r[k];
#pragma omp parallel for schedule(static) if (size >= 4096)
        for (size_t i = 0; i < size; i++) {
            fr_t y; fr_mul(&out[i], &rk, &y);
            out[i + size] = y;
            fr_sub(&out[i], &y, &out[i]);
        }
        size *= 2;
    }
}

/* ------------------------------------------------------------------ bind */
void orc_bind(fr_t *z, size_t len, const fr_t *r, int order) {
    size_t n = len / 2;
    if (order == ORC_HIGH_TO_LOW) {
        /* bound_poly_var_top (dense_mlpoly.rs:91-101): a += r*(b-a) in place */
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t i = 0; i < n; i++) {
            fr_t m; fr_sub(&z[i + n], &z[i], &m);
            fr_mul(&m, r, &m);
            fr_add(&z[i], &m, &z[i]);
        }
    } else {
        /* bound_poly_var_bot_01_optimized (dense_mlpoly.rs:219-239) writes a new
         * vector; in place is safe serially, so compute into scratch when parallel. */
        fr_t *tmp = (fr_t *)malloc(n * sizeof(fr_t));
#pragma omp parallel for schedule(static) if (n >= 512)
        for (size_t i = 0; i < n; i++) {
            fr_t m; fr_sub(&z[2 * i + 1], &z[2 * i], &m);
            fr_mul(&m, r, &m);
            fr_add(&z[2 * i], &m, &tmp[i]);
        }
        memcpy(z, tmp, n * sizeof(fr_t));
        free(tmp);
    }
}

void orc_i32_to_fr(const int32_t *z, size_t len, fr_t *out) {
#pragma omp parallel for schedule(static) if (len >= 4096)
    for (size_t i = 0; i < len; i++) fr_from_i64(z[i], &out[i]);
}

void orc_bind_i32(const int32_t *z, size_t len, const fr_t *r, int order, fr_t *out) {
    /* CompactPolynomial::bind_parallel, unbound branch (compact_polynomial.rs:305-350):
     * a (+|-) r*|b-a| with diff_mul_field = r.mul_u64(abs_diff) (small_scalar.rs:184-186) */
    size_t n = len / 2;
#pragma omp parallel for schedule(static) if (n >= 4096)
    for (size_t i = 0; i < n; i++) {
        int32_t a = order == ORC_HIGH_TO_LOW ? z[i] : z[2 * i];
        int32_t b = order == ORC_HIGH_TO_LOW ? z[i + n] : z[2 * i + 1];
        fr_t fa; fr_from_i64(a, &fa);
        if (a == b) { out[i] = fa; continue; }
        uint64_t d = a < b ? (uint64_t)((int64_t)b - a) : (uint64_t)((int64_t)a - b);
        fr_t fd, m; fr_from_u64(d, &fd); fr_mul(r, &fd, &m);
        if (a < b) fr_add(&fa, &m, &out[i]); else fr_sub(&fa, &m, &out[i]);
    }
}

void orc_evaluate(const fr_t *z, size_t n_vars, const fr_t *r, fr_t *out) {
    /* DensePolynomial::evaluate (dense_mlpoly.rs:265-305): split-eq double sum */
    size_t m = n_vars / 2;
    size_t n1 = (size_t)1 << m, n2 = (size_t)1 << (n_vars - m);
    fr_t *eq1 = (fr_t *)malloc(n1 * sizeof(fr_t)), *eq2 = (fr_t *)malloc(n2 * sizeof(fr_t));
    orc_eq_evals(r, m, 0, eq1);          /* r2 = r[..m]  -> eq_one (outer) */
    orc_eq_evals(r + m, n_vars - m, 0, eq2);
    fr_t acc; fr_zero(&acc);
    for (size_t x1 = 0; x1 < n1; x1++) {
        fr_t part; fr_zero(&part);
        for (size_t x2 = 0; x2 < n2; x2++) {
            fr_t t; fr_mul(&eq2[x2], &z[x1 * n2 + x2], &t); fr_add(&part, &t, &part);
        }
        fr_mul(&eq1[x1], &part, &part); fr_add(&acc, &part, &acc);
    }
    *out = acc; free(eq1); free(eq2);
}

/* ------------------------------------------------------------------ UniPoly */
static void fr_small(uint64_t v, fr_t *o) { fr_from_u64(v, o); }

size_t orc_unipoly_from_evals_and_hint(const fr_t *hint, const fr_t *evals, size_t n_evals, fr_t *c) {
    /* unipoly.rs:91-98 then from_evals_degree2 / degree3 (:66-89): fixed length, no trim */
    fr_t e0 = evals[0], e1; fr_sub(hint, &e0, &e1);
    fr_t two, two_inv; fr_small(2, &two); fr_inv(&two, &two_inv);
    if (n_evals == 2) {
        fr_t e2 = evals[1], t;
        c[0] = e0;
        fr_sub(&e0, &e1, &t); fr_sub(&t, &e1, &t); fr_add(&t, &e2, &t); fr_mul(&t, &two_inv, &c[2]);
        fr_sub(&e1, &e0, &t); fr_sub(&t, &c[2], &c[1]);
        return 3;
    }
    fr_t e2 = evals[1], e3 = evals[2], t, u, three, six, six_inv;
    fr_small(3, &three); fr_small(6, &six); fr_inv(&six, &six_inv);
    c[0] = e0;
    fr_sub(&e1, &e2, &u); fr_mul(&u, &three, &u);
    fr_sub(&e3, &e0, &t); fr_add(&t, &u, &t); fr_mul(&t, &six_inv, &c[3]);
    fr_sub(&e0, &e1, &t); fr_sub(&t, &e1, &t); fr_add(&t, &e2, &t); fr_mul(&t, &two_inv, &t);
    fr_sub(&t, &c[3], &t); fr_sub(&t, &c[3], &t); fr_sub(&t, &c[3], &c[2]);
    fr_sub(&e1, &e0, &t); fr_sub(&t, &c[2], &t); fr_sub(&t, &c[3], &c[1]);
    return 4;
}

void orc_unipoly_eval(const fr_t *c, size_t n, const fr_t *x, fr_t *out) {
    /* eval_with_coeffs (unipoly.rs:229-245) */
    fr_t ev = c[0], pw = *x;
    for (size_t i = 1; i < n; i++) {
        fr_t t; fr_mul(&pw, &c[i], &t); fr_add(&ev, &t, &ev);
        fr_mul(&pw, x, &pw);
    }
    *out = ev;
}

size_t orc_unipoly_compress(const fr_t *c, size_t n, fr_t *out) { /* unipoly.rs:307-318 */
    if (n < 2) { if (n) out[0] = c[0]; return n; }
    out[0] = c[0];
    for (size_t i = 2; i < n; i++) out[i - 1] = c[i];
    return n - 1;
}

void orc_compressed_eval_from_hint(const fr_t *cc, size_t n, const fr_t *hint, const fr_t *x, fr_t *out) {
    /* unipoly.rs:519-533 */
    fr_t lin; fr_sub(hint, &cc[0], &lin); fr_sub(&lin, &cc[0], &lin);
    for (size_t i = 1; i < n; i++) fr_sub(&lin, &cc[i], &lin);
    fr_t rp = *x, sum, t;
    fr_mul(x, &lin, &t); fr_add(&cc[0], &t, &sum);
    for (size_t i = 1; i < n; i++) {
        fr_mul(&rp, x, &rp);
        fr_mul(&cc[i], &rp, &t); fr_add(&sum, &t, &sum);
    }
    *out = sum;
}

void orc_transcript_append_compressed(orc_transcript *t, const fr_t *cc, size_t n) { /* unipoly.rs:550-558 */
    orc_transcript_append_message(t, "UniPoly_begin");
    for (size_t i = 0; i < n; i++) orc_transcript_append_scalar(t, &cc[i]);
    orc_transcript_append_message(t, "UniPoly_end");
}

/* ------------------------------------------------------------------ dot instance */
static inline void evals3(const fr_t *z, size_t i, size_t half, fr_t e[3], int deg) {
    /* sumcheck_evals HighToLow: values at 0, 2, 3 */
    e[0] = z[i];
    fr_t m, v = z[i + half]; fr_sub(&v, &e[0], &m);
    for (int k = 1; k < deg; k++) { fr_add(&v, &m, &v); e[k] = v; }
}

typedef struct { size_t round, half; int schedule; size_t a, b; const fr_t *eq; fr_t eq_bound; size_t eq_len; } dot_ctx;

static void dot_message(const fr_t *L, const fr_t *R, const dot_ctx *cx, fr_t out[3]) {
    const int deg = cx->schedule == 0 ? 2 : 3;
    const size_t half = cx->half;
    int nt = 1;
#ifdef _OPENMP
    nt = half >= 2048 ? omp_get_max_threads() : 1;
#endif
    fr_t *part = (fr_t *)calloc((size_t)nt * 3, sizeof(fr_t));
#pragma omp parallel num_threads(nt)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        fr_t acc[3]; fr_zero(&acc[0]); fr_zero(&acc[1]); fr_zero(&acc[2]);
#pragma omp for schedule(static)
        for (size_t i = 0; i < half; i++) {
            fr_t l[3], r[3], q[3], t;
            evals3(L, i, half, l, deg); evals3(R, i, half, r, deg);
            if (cx->schedule == 1) {
                if (cx->round < cx->a) evals3(cx->eq, i >> cx->b, cx->eq_len / 2, q, 3);
                else q[0] = q[1] = q[2] = cx->eq_bound;
            } else if (cx->schedule == 2) {
                if (cx->round < cx->a) q[0] = q[1] = q[2] = cx->eq[i & (((size_t)1 << cx->b) - 1)];
                else evals3(cx->eq, i, cx->eq_len / 2, q, 3);
            }
            for (int k = 0; k < deg; k++) {
                fr_mul(&l[k], &r[k], &t);
                if (cx->schedule) fr_mul(&t, &q[k], &t);
                fr_add(&acc[k], &t, &acc[k]);
            }
        }
        part[3 * tid] = acc[0]; part[3 * tid + 1] = acc[1]; part[3 * tid + 2] = acc[2];
    }
    fr_zero(&out[0]); fr_zero(&out[1]); fr_zero(&out[2]);
    for (int t = 0; t < nt; t++)
        for (int k = 0; k < 3; k++) fr_add(&out[k], &part[3 * t + k], &out[k]);
    free(part);
}

void orc_dot_claim(const fr_t *l, const fr_t *r, const fr_t *eq, size_t len, int schedule,
                   size_t a, size_t b, fr_t *out) {
    fr_t acc; fr_zero(&acc);
    for (size_t i = 0; i < len; i++) {
        fr_t t; fr_mul(&l[i], &r[i], &t);
        if (schedule == 1) fr_mul(&t, &eq[i >> b], &t);
        if (schedule == 2) fr_mul(&t, &eq[i & (((size_t)1 << b) - 1)], &t);
        fr_add(&acc, &t, &acc);
    }
    (void)a; *out = acc;
}

typedef struct { fr_t *left, *right, *eq; size_t len, sa; dot_ctx cx; } dot_state;

static void dot_init(dot_state *S, fr_t *left, fr_t *right, fr_t *eq, size_t n_vars, int schedule, size_t sa, size_t sb) {
    S->left = left; S->right = right; S->eq = eq; S->len = (size_t)1 << n_vars; S->sa = sa;
    memset(&S->cx, 0, sizeof S->cx);
    S->cx.schedule = schedule; S->cx.a = sa; S->cx.b = sb; S->cx.eq = eq;
    S->cx.eq_len = schedule == 1 ? ((size_t)1 << sa) : schedule == 2 ? ((size_t)1 << sb) : 0;
    if (schedule == 1 && sa == 0) S->cx.eq_bound = eq[0];
}

/* compute_message (dot.rs:290-350) + UniPoly::from_evals_and_hint */
static size_t dot_round_message(dot_state *S, size_t rnd, const fr_t *prev, fr_t coeffs[4]) {
    fr_t ev[3];
    S->cx.round = rnd; S->cx.half = S->len / 2;
    dot_message(S->left, S->right, &S->cx, ev);
    return orc_unipoly_from_evals_and_hint(prev, ev, S->cx.schedule == 0 ? 2 : 3, coeffs);
}

/* ingest_challenge (dot.rs:352-375) */
static void dot_ingest(dot_state *S, size_t rnd, const fr_t *r) {
    const int schedule = S->cx.schedule;
    orc_bind(S->left, S->len, r, ORC_HIGH_TO_LOW);
    orc_bind(S->right, S->len, r, ORC_HIGH_TO_LOW);
    if (schedule == 1 && rnd < S->sa) {
        orc_bind(S->eq, S->cx.eq_len, r, ORC_HIGH_TO_LOW); S->cx.eq_len /= 2;
        if (rnd == S->sa - 1) S->cx.eq_bound = S->eq[0];
    } else if (schedule == 2 && rnd >= S->sa) {
        orc_bind(S->eq, S->cx.eq_len, r, ORC_HIGH_TO_LOW); S->cx.eq_len /= 2;
    }
    S->len /= 2;
}

static void dot_finals(const dot_state *S, fr_t *final_claims) {
    final_claims[0] = S->left[0]; final_claims[1] = S->right[0];
    if (S->cx.schedule == 1) final_claims[2] = S->cx.eq_bound;
    else if (S->cx.schedule == 2) final_claims[2] = S->eq[0];
    else fr_one(&final_claims[2]);
}

int orc_sumcheck_dot_prove(fr_t *left, fr_t *right, fr_t *eq, size_t n_vars, int schedule,
                           size_t sa, size_t sb, const fr_t *input_claim, orc_transcript *t,
                           fr_t *proof, u128 *challenges, fr_t *final_claims) {
    const size_t deg = schedule == 0 ? 2 : 3;
    dot_state S; dot_init(&S, left, right, eq, n_vars, schedule, sa, sb);
    orc_transcript_append_scalar(t, input_claim);            /* sumcheck.rs:573-574 */
    fr_t prev = *input_claim;
    for (size_t rnd = 0; rnd < n_vars; rnd++) {
        fr_t coeffs[4], cc[3], r;
        size_t nc = dot_round_message(&S, rnd, &prev, coeffs);
        size_t ncc = orc_unipoly_compress(coeffs, nc, cc);   /* sumcheck.rs:581 */
        orc_transcript_append_compressed(t, cc, ncc);
        u128 raw; orc_transcript_challenge_optimized(t, &raw, &r);  /* :583 */
        challenges[rnd] = raw;
        orc_unipoly_eval(coeffs, nc, &r, &prev);             /* :587 */
        dot_ingest(&S, rnd, &r);
        for (size_t k = 0; k < ncc; k++) proof[rnd * deg + k] = cc[k];
    }
    dot_finals(&S, final_claims);
    return 0;
}

int orc_sumcheck_dot_prove_i32(const int32_t *left, const int32_t *right, size_t n_vars,
                               const fr_t *input_claim, orc_transcript *t, fr_t *proof,
                               u128 *challenges, fr_t *final_claims) {
    /* I32Scalars operands: sumcheck_evals go through get_bound_coeff -> to_field
     * (multilinear_polynomial.rs:242-319), first bind is CompactPolynomial's. Values are
     * those of the dense instance on the converted operands. */
    size_t len = (size_t)1 << n_vars;
    fr_t *L = (fr_t *)malloc(len * sizeof(fr_t)), *R = (fr_t *)malloc(len * sizeof(fr_t));
    orc_i32_to_fr(left, len, L); orc_i32_to_fr(right, len, R);
    int rc = orc_sumcheck_dot_prove(L, R, 0, n_vars, 0, 0, 0, input_claim, t, proof, challenges, final_claims);
    free(L); free(R);
    return rc;
}

int orc_sumcheck_verify(const fr_t *proof, size_t n_rounds, size_t degree, const fr_t *claim,
                        orc_transcript *t, fr_t *e_out, u128 *challenges) {
    fr_t e = *claim;
    for (size_t i = 0; i < n_rounds; i++) {
        const fr_t *cc = &proof[i * degree];
        orc_transcript_append_compressed(t, cc, degree);
        fr_t r; u128 raw; orc_transcript_challenge_optimized(t, &raw, &r);
        if (challenges) challenges[i] = raw;
        orc_compressed_eval_from_hint(cc, degree, &e, &r, &e);
    }
    *e_out = e;
    return 0;
}

/* ------------------------------------------------------------------ Mul (Gruen split-eq)
 * MulProver (jolt-atlas-core/src/onnx_proof/ops/mul.rs:125-185) over
 * GruenSplitEqPolynomial (joltworks/src/poly/split_eq_poly.rs): LowToHigh binding,
 * w = [w_out | w_in | w_last] (:97-121), prefix tables evals_cached (eq_poly.rs:174-192),
 * par_fold_out_in (:526-597), gruen_poly_deg_3 with the division by eq(1) (:379-429),
 * bind (:331-348). */
static void eq_cached(const fr_t *w, size_t k, fr_t **tabs) {
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

typedef struct {
    fr_t *left, *right; const fr_t *w; size_t n, len, out_top, in_top, k_out, k_in, current_index;
    fr_t **Eout, **Ein; fr_t scalar;
} mul_state;

static void mul_init(mul_state *S, fr_t *left, fr_t *right, const fr_t *w, size_t n) {
    const size_t m = n / 2;
    S->left = left; S->right = right; S->w = w; S->n = n; S->len = (size_t)1 << n;
    S->k_out = m; S->k_in = n - 1 - m;
    S->Eout = (fr_t **)malloc((S->k_out + 1) * sizeof(fr_t *)); S->Ein = (fr_t **)malloc((S->k_in + 1) * sizeof(fr_t *));
    eq_cached(w, S->k_out, S->Eout); eq_cached(w + m, S->k_in, S->Ein);
    S->out_top = S->k_out; S->in_top = S->k_in;     /* index of the current (last) table */
    S->current_index = n;
    fr_one(&S->scalar);
}

static void mul_free(mul_state *S) {
    for (size_t j = 0; j <= S->k_out; j++) free(S->Eout[j]);
    for (size_t j = 0; j <= S->k_in; j++) free(S->Ein[j]);
    free(S->Eout); free(S->Ein);
}

/* compute_message (mul.rs:160-176): 4 coefficients of the round polynomial */
static size_t mul_message(const mul_state *S, const fr_t *prev, fr_t coeffs[4]) {
    const fr_t *left = S->left, *right = S->right, *w = S->w;
    const size_t in_top = S->in_top;
    const fr_t *e_out = S->Eout[S->out_top], *e_in = S->Ein[in_top];
    const size_t out_len = (size_t)1 << S->out_top, in_len = (size_t)1 << in_top;
    fr_t qc, qe; fr_zero(&qc); fr_zero(&qe);
#pragma omp parallel if (out_len * in_len >= 4096)
    {
        fr_t pc, pe; fr_zero(&pc); fr_zero(&pe);
#pragma omp for schedule(static)
        for (size_t xo = 0; xo < out_len; xo++) {
            fr_t ic, ie; fr_zero(&ic); fr_zero(&ie);
            for (size_t xi = 0; xi < in_len; xi++) {
                size_t gidx = (xo << in_top) | xi;
                fr_t lo0 = left[2 * gidx], loi, ro0 = right[2 * gidx], roi, c0, e;
                fr_sub(&left[2 * gidx + 1], &lo0, &loi); fr_sub(&right[2 * gidx + 1], &ro0, &roi);
                fr_mul(&lo0, &ro0, &c0); fr_mul(&loi, &roi, &e);
                fr_mul(&e_in[xi], &c0, &c0); fr_mul(&e_in[xi], &e, &e);
                fr_add(&ic, &c0, &ic); fr_add(&ie, &e, &ie);
            }
            fr_mul(&e_out[xo], &ic, &ic); fr_mul(&e_out[xo], &ie, &ie);
            fr_add(&pc, &ic, &pc); fr_add(&pe, &ie, &pe);
        }
#pragma omp critical
        { fr_add(&qc, &pc, &qc); fr_add(&qe, &pe, &qe); }
    }
    /* gruen_poly_deg_3 */
    fr_t eq1, eq0, eqm, eq2, eq3, c_ev0, c_ev1, q1, q2, q3, e2, inv, evals[4];
    fr_mul(&S->scalar, &w[S->current_index - 1], &eq1); fr_sub(&S->scalar, &eq1, &eq0);
    fr_sub(&eq1, &eq0, &eqm); fr_add(&eq1, &eqm, &eq2); fr_add(&eq2, &eqm, &eq3);
    fr_mul(&eq0, &qc, &c_ev0); fr_sub(prev, &c_ev0, &c_ev1);
    fr_inv(&eq1, &inv); fr_mul(&c_ev1, &inv, &q1);
    fr_add(&qe, &qe, &e2);
    fr_add(&q1, &q1, &q2); fr_sub(&q2, &qc, &q2); fr_add(&q2, &e2, &q2);
    fr_add(&q2, &q1, &q3); fr_sub(&q3, &qc, &q3); fr_add(&q3, &e2, &q3); fr_add(&q3, &e2, &q3);
    evals[0] = c_ev0; evals[1] = c_ev1; fr_mul(&eq2, &q2, &evals[2]); fr_mul(&eq3, &q3, &evals[3]);
    /* UniPoly::from_evals degree 3: reuse the hint form (hint = e0 + e1) */
    fr_t hint, ev3[3];
    fr_add(&evals[0], &evals[1], &hint); ev3[0] = evals[0]; ev3[1] = evals[2]; ev3[2] = evals[3];
    return orc_unipoly_from_evals_and_hint(&hint, ev3, 3, coeffs);
}

/* ingest_challenge (mul.rs:178-185): eq.bind(r); left/right bind LowToHigh */
static void mul_ingest(mul_state *S, const fr_t *r) {
    const fr_t *w = S->w;
    fr_t wr, f, one; fr_one(&one);
    fr_mul(&w[S->current_index - 1], r, &wr);
    fr_sub(&one, &w[S->current_index - 1], &f); fr_sub(&f, r, &f); fr_add(&f, &wr, &f); fr_add(&f, &wr, &f);
    fr_mul(&S->scalar, &f, &S->scalar);
    S->current_index -= 1;
    if (S->n / 2 < S->current_index && S->in_top > 0) S->in_top--;
    else if (0 < S->current_index && S->out_top > 0) S->out_top--;
    orc_bind(S->left, S->len, r, ORC_LOW_TO_HIGH); orc_bind(S->right, S->len, r, ORC_LOW_TO_HIGH);
    S->len /= 2;
}

int orc_sumcheck_mul_prove(fr_t *left, fr_t *right, const fr_t *w, size_t n, const fr_t *input_claim,
                           orc_transcript *t, fr_t *proof, u128 *challenges, fr_t *final_claims) {
    mul_state S; mul_init(&S, left, right, w, n);
    orc_transcript_append_scalar(t, input_claim);
    fr_t prev = *input_claim;
    for (size_t rnd = 0; rnd < n; rnd++) {
        fr_t coeffs[4], cc[3], r;
        size_t nc = mul_message(&S, &prev, coeffs);
        size_t ncc = orc_unipoly_compress(coeffs, nc, cc);
        orc_transcript_append_compressed(t, cc, ncc);
        u128 raw; orc_transcript_challenge_optimized(t, &raw, &r);
        challenges[rnd] = raw;
        orc_unipoly_eval(coeffs, nc, &r, &prev);
        for (size_t k = 0; k < 3; k++) proof[rnd * 3 + k] = cc[k];
        mul_ingest(&S, &r);
    }
    final_claims[0] = left[0]; final_claims[1] = right[0]; final_claims[2] = S.scalar;
    mul_free(&S);
    return 0;
}

/* sum_x eq(w, x) L(x) R(x) */
void orc_mul_claim(const fr_t *l, const fr_t *r, const fr_t *w, size_t n, fr_t *out) {
    size_t len = (size_t)1 << n;
    fr_t *eq = (fr_t *)malloc(len * sizeof(fr_t));
    orc_eq_evals(w, n, 0, eq);
    fr_t acc; fr_zero(&acc);
    for (size_t i = 0; i < len; i++) { fr_t t; fr_mul(&l[i], &r[i], &t); fr_mul(&t, &eq[i], &t); fr_add(&acc, &t, &acc); }
    *out = acc; free(eq);
}

/* ------------------------------------------------------------------ BatchedSumcheck::prove
 * joltworks/src/subprotocols/sumcheck.rs:30-184.  Instances are described by orc_batched_inst
 * (kind 0 = dot instance, 1 = Mul instance); operand arrays are bound in place. */
static void fr_mul_pow2(const fr_t *x, size_t pow, fr_t *o) {          /* field/mod.rs:274-284 */
    fr_t v = *x;
    for (size_t i = 0; i < pow; i++) fr_add(&v, &v, &v);
    *o = v;
}

static int fr_eq(const fr_t *a, const fr_t *b) { return memcmp(a, b, sizeof(fr_t)) == 0; }

static size_t scale_trim(const fr_t *p, size_t n, const fr_t *c, fr_t *o) {   /* &UniPoly * F -> from_coeff (unipoly.rs:39-52,454-461) */
    for (size_t i = 0; i < n; i++) fr_mul(&p[i], c, &o[i]);
    fr_t z; fr_zero(&z);
    while (n > 0 && fr_eq(&o[n - 1], &z)) n--;
    if (n == 0) { o[0] = z; n = 1; }
    return n;
}

#define ORC_BATCH_MAXC 40
int orc_batched_prove(orc_batched_inst *inst, size_t n_inst, orc_transcript *t, fr_t *compressed, size_t stride,
                      uint32_t *n_coeffs, u128 *challenges, size_t *max_rounds_out) {
    size_t max_rounds = 0;
    dot_state *D = (dot_state *)calloc(n_inst, sizeof(dot_state));
    mul_state *M = (mul_state *)calloc(n_inst, sizeof(mul_state));
    fr_t *coeff = (fr_t *)malloc(n_inst * sizeof(fr_t)), *claim = (fr_t *)malloc(n_inst * sizeof(fr_t));
    fr_t (*polys)[ORC_BATCH_MAXC] = (fr_t (*)[ORC_BATCH_MAXC])malloc(n_inst * sizeof(fr_t[ORC_BATCH_MAXC]));
    size_t *plen = (size_t *)malloc(n_inst * sizeof(size_t));
    for (size_t i = 0; i < n_inst; i++) {
        orc_batched_inst *I = &inst[i];
        if (I->kind == 0) dot_init(&D[i], I->left, I->right, I->eq, I->n_vars, I->schedule, I->sa, I->sb);
        else if (I->kind == 1) mul_init(&M[i], I->left, I->right, I->w, I->n_vars);
        /* kinds >= 2: an initialised oracle/ra.c state in I->state, n_vars = its round count */
        if (I->n_vars > max_rounds) max_rounds = I->n_vars;
    }
    for (size_t i = 0; i < n_inst; i++) orc_transcript_append_scalar(t, &inst[i].input_claim);   /* :42-45 */
    for (size_t i = 0; i < n_inst; i++) orc_transcript_challenge_scalar(t, &coeff[i]);           /* :47 */
    for (size_t i = 0; i < n_inst; i++) fr_mul_pow2(&inst[i].input_claim, max_rounds - inst[i].n_vars, &claim[i]);  /* :58-66 */
    for (size_t round = 0; round < max_rounds; round++) {
        const size_t remaining = max_rounds - round;
        for (size_t i = 0; i < n_inst; i++) {
            const size_t nr = inst[i].n_vars;
            if (remaining > nr) { fr_mul_pow2(&inst[i].input_claim, remaining - nr - 1, &polys[i][0]); plen[i] = 1; }   /* :91-98 */
            else {
                const size_t local = round - (max_rounds - nr);
                plen[i] = inst[i].kind == 0 ? dot_round_message(&D[i], local, &claim[i], polys[i])
                        : inst[i].kind == 1 ? mul_message(&M[i], &claim[i], polys[i])
                                            : orc_ra_inst_message(inst[i].kind, inst[i].state, local, &claim[i], polys[i]);
            }
        }
        fr_t batched[ORC_BATCH_MAXC], tmp[ORC_BATCH_MAXC]; size_t blen = 1; fr_zero(&batched[0]);   /* :109-116 */
        for (size_t i = 0; i < n_inst; i++) {
            size_t tl = scale_trim(polys[i], plen[i], &coeff[i], tmp);
            for (size_t k = 0; k < tl; k++) {
                if (k < blen) fr_add(&batched[k], &tmp[k], &batched[k]);
                else batched[blen++] = tmp[k];
            }
        }
        fr_t cc[ORC_BATCH_MAXC]; size_t ncc = orc_unipoly_compress(batched, blen, cc);
        orc_transcript_append_compressed(t, cc, ncc);
        fr_t r; u128 raw; orc_transcript_challenge_optimized(t, &raw, &r);
        challenges[round] = raw; n_coeffs[round] = (uint32_t)ncc;
        for (size_t k = 0; k < ncc; k++) compressed[round * stride + k] = cc[k];
        for (size_t i = 0; i < n_inst; i++) orc_unipoly_eval(polys[i], plen[i], &r, &claim[i]);    /* :123-126 */
        for (size_t i = 0; i < n_inst; i++) {
            const size_t nr = inst[i].n_vars;
            if (remaining <= nr) {
                const size_t local = round - (max_rounds - nr);
                if (inst[i].kind == 0) dot_ingest(&D[i], local, &r);
                else if (inst[i].kind == 1) mul_ingest(&M[i], &r);
                else orc_ra_inst_ingest(inst[i].kind, inst[i].state, local, &r);
            }
        }
    }
    for (size_t i = 0; i < n_inst; i++) {
        if (inst[i].kind == 0) dot_finals(&D[i], inst[i].final_claims);
        else if (inst[i].kind == 1) {
            inst[i].final_claims[0] = inst[i].left[0]; inst[i].final_claims[1] = inst[i].right[0];
            inst[i].final_claims[2] = M[i].scalar; mul_free(&M[i]);
        }
    }
    *max_rounds_out = max_rounds;
    free(D); free(M); free(coeff); free(claim); free(polys); free(plen);
    return 0;
}

int orc_batched_verify(const fr_t *compressed, size_t stride, const uint32_t *n_coeffs, size_t max_rounds, const fr_t *input_claims,
                       const size_t *rounds, size_t n_inst, orc_transcript *t, fr_t *e_out, fr_t *coeffs_out,
                       u128 *challenges) {
    for (size_t i = 0; i < n_inst; i++) orc_transcript_append_scalar(t, &input_claims[i]);      /* :205-208 */
    for (size_t i = 0; i < n_inst; i++) orc_transcript_challenge_scalar(t, &coeffs_out[i]);    /* :210 */
    fr_t e; fr_zero(&e);
    for (size_t i = 0; i < n_inst; i++) {                                                       /* :221-229 */
        fr_t c; fr_mul_pow2(&input_claims[i], max_rounds - rounds[i], &c);
        fr_mul(&c, &coeffs_out[i], &c); fr_add(&e, &c, &e);
    }
    for (size_t i = 0; i < max_rounds; i++) {                                                   /* proof.verify :653-686 */
        const fr_t *cc = &compressed[i * stride];
        orc_transcript_append_compressed(t, cc, n_coeffs[i]);
        fr_t r; u128 raw; orc_transcript_challenge_optimized(t, &raw, &r);
        if (challenges) challenges[i] = raw;
        orc_compressed_eval_from_hint(cc, n_coeffs[i], &e, &r, &e);
    }
    *e_out = e;
    return 0;
}
