/* TEST INFRASTRUCTURE ONLY — CPU oracle, never linked into or called by the product.
 * parity unpinned by reference vectors (the reference holds none for these provers); pinned against the
 * brute-force models in oracle/pymodel/ra.py (tests/test_oracle_ra.py).
 *
 * The element-wise operator provers of jolt-atlas-core/src/onnx_proof/ops/: one GruenSplitEqPolynomial over
 * r_node_output (LowToHigh), operands bound LowToHigh, and per round
 *   add.rs:283-296     [lo0 + ro0]                              gruen_poly_deg_2
 *   sub.rs:267-280     [lo0 - ro0]                              gruen_poly_deg_2
 *   neg.rs:206-217     [-o0]                                    gruen_poly_deg_2
 *   square.rs:163-178  [o0 o0, o_inf o_inf]                     gruen_poly_deg_3
 *   mul.rs:160-185     [lo0 ro0, lo_inf ro_inf]                 gruen_poly_deg_3
 *   iff.rs:189-216     [m0 a0 + (1 - m0) b0, m_inf a_inf + f b_inf],  f = (1 - m1) - (1 - m0)   gruen_poly_deg_3
 *   cube.rs:159-166    compute_mle_product_sum(3, operand)      (mles_product_sum.rs:41-55, degree 4)
 *   div.rs:329-351     [ro0 q0 + R0 - lo0, ro_inf q_inf]                                          gruen_poly_deg_3
 *   rsqrt.rs:390-421   [x0 quot0 + dr0 - S^3 + gamma (out0^2 + sr0 - quot0), x_inf quot_inf + gamma out_inf^2]   gruen_poly_deg_3
 *   (scalar_const_div.rs:227-241 [lo0 - R0] is the Sub fold)
 *   joltworks/src/subprotocols/hamming_booleanity.rs:131-156   sum_d gamma_d [hw_d0 (hw_d0 - 1), hw_d_inf^2]   gruen_poly_deg_3
 *   neural_teleport/division.rs:231-256   [tau q0 + r0 - inp0]   (input, quotient, remainder; tau)   gruen_poly_deg_2
 * and the selector-style provers without an eq factor (LowToHigh sumcheck_evals at 0 and 2, from_evals_and_hint):
 *   reshape.rs:286-308, slice.rs:254-274   input * selector;  concat.rs:290-317   sum over terms of input_t * selector_t
 *   gather/mod.rs:232-261                  ra * (dictionary + gamma * identity)
 * with par_fold_out_in_unreduced (split_eq_poly.rs:526-597) = sum_{x_out} E_out sum_{x_in} E_in f(g). */
#include <stdlib.h>
#include <string.h>
#include "elementwise.h"

void orc_elementwise_init(orc_elementwise *S, int op, const fr_t *const *operands, size_t n_ops, size_t n_vars, const fr_t *r_node_output,
                          const fr_t *constants, size_t n_constants) {
    memset(S, 0, sizeof *S);
    for (size_t i = 0; i < n_constants && i < 16; i++) S->k[i] = constants[i];
    S->op = op; S->n_ops = n_ops; S->n_vars = n_vars; S->len = (size_t)1 << n_vars;
    for (size_t i = 0; i < n_ops; i++) { S->x[i] = (fr_t *)malloc(S->len * sizeof(fr_t)); memcpy(S->x[i], operands[i], S->len * sizeof(fr_t)); }
    gse_init(&S->eq, r_node_output, n_vars);
}

void orc_elementwise_free(orc_elementwise *S) { for (size_t i = 0; i < S->n_ops; i++) free(S->x[i]); gse_free(&S->eq); }

/* gruen_poly_deg_2 (split_eq_poly.rs:379-428), LowToHigh */
static size_t gruen_deg2(const gse_t *E, const fr_t *q0, const fr_t *claim, fr_t *coeffs) {
    fr_t eq1, eq0, eqm, eq2, c0, c1, l1, l2, inv, ev2[2], hint;
    fr_mul(&E->scalar, &E->w[E->current_index - 1], &eq1); fr_sub(&E->scalar, &eq1, &eq0);
    fr_sub(&eq1, &eq0, &eqm); fr_add(&eq1, &eqm, &eq2);
    fr_mul(&eq0, q0, &c0); fr_sub(claim, &c0, &c1);
    fr_inv(&eq1, &inv); fr_mul(&c1, &inv, &l1);
    fr_add(&l1, &l1, &l2); fr_sub(&l2, q0, &l2);
    ev2[0] = c0; fr_mul(&eq2, &l2, &ev2[1]); fr_add(&c0, &c1, &hint);
    return orc_unipoly_from_evals_and_hint(&hint, ev2, 2, coeffs);
}

/* MultilinearPolynomial::sumcheck_evals(i, 2, LowToHigh): values at 0 and 2 of the pair (2i, 2i+1) */
static void evals02(const fr_t *z, size_t i, fr_t o[2]) {
    fr_t m; o[0] = z[2 * i]; fr_sub(&z[2 * i + 1], &z[2 * i], &m); fr_add(&z[2 * i + 1], &m, &o[1]);
}

size_t orc_elementwise_message(orc_elementwise *S, const fr_t *claim, fr_t *coeffs) {
    if (S->op == ORC_EW_DOT || S->op == ORC_EW_GATHER) {
        fr_t ev[2], t; fr_zero(&ev[0]); fr_zero(&ev[1]);
        for (size_t i = 0; i < S->len / 2; i++) {
            if (S->op == ORC_EW_DOT) {
                for (size_t term = 0; term + 1 < S->n_ops; term += 2) {
                    fr_t a[2], s[2]; evals02(S->x[term], i, a); evals02(S->x[term + 1], i, s);
                    for (int k = 0; k < 2; k++) { fr_mul(&a[k], &s[k], &t); fr_add(&ev[k], &t, &ev[k]); }
                }
            } else {
                fr_t ra[2], dc[2], id[2]; evals02(S->x[0], i, ra); evals02(S->x[1], i, dc); evals02(S->x[2], i, id);
                for (int k = 0; k < 2; k++) { fr_mul(&id[k], &S->k[0], &t); fr_add(&dc[k], &t, &t); fr_mul(&ra[k], &t, &t); fr_add(&ev[k], &t, &ev[k]); }
            }
        }
        return orc_unipoly_from_evals_and_hint(claim, ev, 2, coeffs);
    }
    const gse_t *E = &S->eq;
    const fr_t *e_out = E->Eout[E->out_top], *e_in = E->Ein[E->in_top];
    const size_t out_len = (size_t)1 << E->out_top, in_len = (size_t)1 << E->in_top;
    if (S->op == ORC_EW_CUBE) {                     /* grid [1, 2, inf] of the product of three copies, then the shared finish */
        fr_t sums[3], inner[3];
        for (int k = 0; k < 3; k++) fr_zero(&sums[k]);
        for (size_t xo = 0; xo < out_len; xo++) {
            for (int k = 0; k < 3; k++) fr_zero(&inner[k]);
            for (size_t xi = 0; xi < in_len; xi++) {
                const size_t g = (xo << E->in_top) | xi;
                fr_t p0 = S->x[0][2 * g], dl, p1, p2, v[3], t;
                fr_sub(&S->x[0][2 * g + 1], &p0, &dl); fr_add(&p0, &dl, &p1); fr_add(&p1, &dl, &p2);
                fr_mul(&p1, &p1, &v[0]); fr_mul(&v[0], &p1, &v[0]);
                fr_mul(&p2, &p2, &v[1]); fr_mul(&v[1], &p2, &v[1]);
                fr_mul(&dl, &dl, &v[2]); fr_mul(&v[2], &dl, &v[2]);
                for (int k = 0; k < 3; k++) { fr_mul(&e_in[xi], &v[k], &t); fr_add(&inner[k], &t, &inner[k]); }
            }
            for (int k = 0; k < 3; k++) { fr_t t; fr_mul(&e_out[xo], &inner[k], &t); fr_add(&sums[k], &t, &sums[k]); }
        }
        for (int k = 0; k < 3; k++) fr_mul(&sums[k], &E->scalar, &sums[k]);
        return orc_finish_product_sum(sums, 3, claim, E, coeffs);
    }
    fr_t q0, qe; fr_zero(&q0); fr_zero(&qe);
    for (size_t xo = 0; xo < out_len; xo++) {
        fr_t in0, ine; fr_zero(&in0); fr_zero(&ine);
        for (size_t xi = 0; xi < in_len; xi++) {
            const size_t g = (xo << E->in_top) | xi;
            fr_t c0, e, t, u, one; fr_zero(&e); fr_one(&one);
            const fr_t *a = S->x[0], *b = S->n_ops > 1 ? S->x[1] : 0, *c = S->n_ops > 2 ? S->x[2] : 0;
            switch (S->op) {
                case ORC_EW_ADD: fr_add(&a[2 * g], &b[2 * g], &c0); break;
                case ORC_EW_SUB: fr_sub(&a[2 * g], &b[2 * g], &c0); break;
                case ORC_EW_NEG: fr_zero(&t); fr_sub(&t, &a[2 * g], &c0); break;
                case ORC_EW_SQUARE: fr_mul(&a[2 * g], &a[2 * g], &c0); fr_sub(&a[2 * g + 1], &a[2 * g], &t); fr_mul(&t, &t, &e); break;
                case ORC_EW_MUL:
                    fr_mul(&a[2 * g], &b[2 * g], &c0);
                    fr_sub(&a[2 * g + 1], &a[2 * g], &t); fr_sub(&b[2 * g + 1], &b[2 * g], &u); fr_mul(&t, &u, &e); break;
                case ORC_EW_HAMMING_BOOL: {         /* x = hw_0 .. hw_{d-1}; k = gamma_powers */
                    fr_zero(&c0);
                    for (size_t d = 0; d < S->n_ops; d++) {
                        const fr_t *hw = S->x[d]; fr_t a2, cc;
                        fr_sub(&hw[2 * g + 1], &hw[2 * g], &t); fr_mul(&t, &t, &a2);
                        fr_sub(&hw[2 * g], &one, &u); fr_mul(&hw[2 * g], &u, &cc);
                        fr_mul(&cc, &S->k[d], &cc); fr_add(&c0, &cc, &c0);
                        fr_mul(&a2, &S->k[d], &a2); fr_add(&e, &a2, &e);
                    }
                    break;
                }
                case ORC_EW_TELEPORT_DIV:           /* x = input, quotient, remainder; k[0] = divisor */
                    fr_mul(&S->k[0], &S->x[1][2 * g], &c0); fr_add(&c0, &S->x[2][2 * g], &c0); fr_sub(&c0, &S->x[0][2 * g], &c0); break;
                case ORC_EW_DIV: {                  /* x = left, right, q, R */
                    const fr_t *lo = S->x[0], *ro = S->x[1], *q = S->x[2], *R = S->x[3];
                    fr_mul(&ro[2 * g], &q[2 * g], &c0); fr_add(&c0, &R[2 * g], &c0); fr_sub(&c0, &lo[2 * g], &c0);
                    fr_sub(&ro[2 * g + 1], &ro[2 * g], &t); fr_sub(&q[2 * g + 1], &q[2 * g], &u); fr_mul(&t, &u, &e); break;
                }
                case ORC_EW_RSQRT: {                /* x = input, quotient, output, div_remainder, sqrt_remainder; k = S^3, gamma */
                    const fr_t *x = S->x[0], *qt = S->x[1], *o = S->x[2], *dr = S->x[3], *sr = S->x[4];
                    fr_t div0, sqrt0, dq, sq;
                    fr_mul(&x[2 * g], &qt[2 * g], &div0); fr_add(&div0, &dr[2 * g], &div0); fr_sub(&div0, &S->k[0], &div0);
                    fr_mul(&o[2 * g], &o[2 * g], &sqrt0); fr_add(&sqrt0, &sr[2 * g], &sqrt0); fr_sub(&sqrt0, &qt[2 * g], &sqrt0);
                    fr_sub(&x[2 * g + 1], &x[2 * g], &t); fr_sub(&qt[2 * g + 1], &qt[2 * g], &u); fr_mul(&t, &u, &dq);
                    fr_sub(&o[2 * g + 1], &o[2 * g], &t); fr_mul(&t, &t, &sq);
                    fr_mul(&S->k[1], &sqrt0, &t); fr_add(&div0, &t, &c0);
                    fr_mul(&S->k[1], &sq, &t); fr_add(&dq, &t, &e); break;
                }
                default: {                          /* Iff: a = mask, b = a_operand, c = b_operand */
                    fr_t m_inf, a_inf, b_inf, nm0, nm1, f;
                    fr_sub(&a[2 * g + 1], &a[2 * g], &m_inf); fr_sub(&b[2 * g + 1], &b[2 * g], &a_inf); fr_sub(&c[2 * g + 1], &c[2 * g], &b_inf);
                    fr_sub(&one, &a[2 * g], &nm0); fr_sub(&one, &a[2 * g + 1], &nm1);
                    fr_mul(&a[2 * g], &b[2 * g], &t); fr_mul(&nm0, &c[2 * g], &u); fr_add(&t, &u, &c0);
                    fr_sub(&nm1, &nm0, &f);
                    fr_mul(&m_inf, &a_inf, &t); fr_mul(&f, &b_inf, &u); fr_add(&t, &u, &e);
                }
            }
            fr_mul(&e_in[xi], &c0, &t); fr_add(&in0, &t, &in0);
            fr_mul(&e_in[xi], &e, &t); fr_add(&ine, &t, &ine);
        }
        fr_t t; fr_mul(&e_out[xo], &in0, &t); fr_add(&q0, &t, &q0); fr_mul(&e_out[xo], &ine, &t); fr_add(&qe, &t, &qe);
    }
    if (S->op == ORC_EW_ADD || S->op == ORC_EW_SUB || S->op == ORC_EW_NEG || S->op == ORC_EW_TELEPORT_DIV) return gruen_deg2(E, &q0, claim, coeffs);
    gse_gruen_deg3(E, &q0, &qe, claim, coeffs);
    return 4;
}

void orc_elementwise_ingest(orc_elementwise *S, const fr_t *r) {
    for (size_t i = 0; i < S->n_ops; i++) orc_bind(S->x[i], S->len, r, ORC_LOW_TO_HIGH);
    S->len /= 2;
    if (S->op != ORC_EW_DOT && S->op != ORC_EW_GATHER) gse_bind(&S->eq, r);
}

void orc_elementwise_finals(const orc_elementwise *S, fr_t *out) { for (size_t i = 0; i < S->n_ops; i++) out[i] = S->x[i][0]; }
