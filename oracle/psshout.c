/* TEST INFRASTRUCTURE ONLY (oracle): prefix-suffix Shout read-raf sumcheck, unary flavour with the
 * ReLU table, restated from /root/reference/joltworks/src:
 *   ReadRafSumcheckProver             subprotocols/ps_shout/mod.rs:227-560 (new_inner, init_phase,
 *                                     init_suffix_polys, prover_msg_read_checking, init_log_t_rounds,
 *                                     compute_message, ingest_challenge)
 *   UnaryRafPS / ps_read_raf_prover   subprotocols/ps_shout/unary.rs:45-148
 *   PrefixSuffixDecomposition         poly/prefix_suffix.rs:237-330 (init_Q, SIGNED), :433-492
 *   SignedIdentityPoly prefix/suffix  poly/signed_identity_poly.rs:136-209, 379-402
 *   ReluTable                         lookup_tables/relu.rs:46-60
 *   NotMsb / WordNoMsb prefixes       lookup_tables/prefixes/not_msb.rs, word_no_msb.rs
 *   One / WordNoMSB suffixes          lookup_tables/suffixes/one.rs, word_no_msb.rs
 *   LookupBits::split                 utils/lookup_bits.rs:32-38;  ExpandingTable HighToLow
 * LOG_K = N = X_LEN (16, 32 or 64), NUM_PHASES = 8, log_m = N / 8. */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "ra.h"
#include "psshout.h"

static void fr_pow2(unsigned k, fr_t *o) { fr_t two; fr_from_u64(2, &two); fr_one(o); for (unsigned i = 0; i < k; i++) fr_mul(o, &two, o); }
static uint64_t split_prefix(uint64_t k, size_t suffix_len) { return suffix_len >= 64 ? 0 : k >> suffix_len; }
static uint64_t split_suffix(uint64_t k, size_t suffix_len) { return suffix_len >= 64 ? k : k % ((uint64_t)1 << suffix_len); }

static void init_phase(orc_ps_relu *S, size_t phase) {                      /* mod.rs:269-301 */
    const size_t log_m = S->log_m, m = S->m, m_mask = m - 1, T = S->T;
    if (phase != 0)
        for (size_t t = 0; t < T; t++) {
            const uint64_t k_bound = split_prefix(S->idx[t], (8 - phase) * log_m) & m_mask;
            fr_mul(&S->u[t], &S->v[phase - 1][k_bound], &S->u[t]);
        }
    const size_t suffix_len = (8 - 1 - phase) * log_m;
    /* raf_state.init_Q: suffixes [One, Identity], SIGNED accumulation (prefix_suffix.rs:237-330) */
    for (int s = 0; s < 2; s++) { for (size_t y = 0; y < m; y++) { fr_zero(&S->RQ[s][y]); fr_zero(&S->Q[s][y]); } }
    for (size_t t = 0; t < T; t++) {
        const uint64_t y = split_prefix(S->idx[t], suffix_len) & m_mask, sb = split_suffix(S->idx[t], suffix_len);
        fr_add(&S->RQ[0][y], &S->u[t], &S->RQ[0][y]);                       /* One: t = 1 */
        if (sb) { fr_t w, x; fr_from_i64((int64_t)sb, &w); fr_mul(&S->u[t], &w, &x); fr_add(&S->RQ[1][y], &x, &S->RQ[1][y]); }
        /* init_suffix_polys (mod.rs:304-335): suffixes [One, WordNoMSB], t as u32 */
        fr_add(&S->Q[0][y], &S->u[t], &S->Q[0][y]);
        /* table 1: suffixes [One, TrigRightShift], suffix_mle = bits >> D (suffixes/right_shift.rs:12-17) */
        const uint32_t tw = S->table == 1 ? (uint32_t)(sb >> S->shift) : (uint32_t)(sb % ((uint64_t)1 << (S->N - 1)));
        if (tw) { fr_t w, x; fr_from_u64(tw, &w); fr_mul(&S->u[t], &w, &x); fr_add(&S->Q[1][y], &x, &S->Q[1][y]); }
    }
    S->Q_len = m;
    /* raf_state.init_P: SignedIdentity prefix polynomial of this phase (signed_identity_poly.rs:181-209) */
    {
        const size_t xlen = S->N, sfx = xlen - log_m * (phase + 1);
        fr_t bound; if (S->has_sid) bound = S->sid_cp; else fr_zero(&bound);
        fr_t penalty; fr_pow2((unsigned)xlen, &penalty);
        for (size_t i = 0; i < m; i++) {
            /* (i << sfx) % 2^xlen as a field element: i < 2^log_m, sfx + log_m <= xlen */
            fr_t sh, vi; fr_pow2((unsigned)sfx, &sh); fr_from_u64(i, &vi); fr_mul(&vi, &sh, &vi);
            if (phase == 0 && sfx + log_m == xlen) { /* % 2^xlen never wraps: i << sfx < 2^xlen */ }
            fr_add(&bound, &vi, &S->RP[i]);
            if (phase == 0 && ((i >> (log_m - 1)) & 1)) fr_sub(&S->RP[i], &penalty, &S->RP[i]);   /* sign_correction */
        }
        S->RP_len = m;
    }
    fr_one(&S->v[phase][0]); S->v_len[phase] = 1;                           /* v[phase].reset(1) */
}

static void ps_unary_init(orc_ps_relu *S, const uint64_t *idx, size_t N, size_t log_T, const fr_t *r_node, const fr_t *gamma, int table, size_t shift) {
    memset(S, 0, sizeof *S);
    S->table = table; S->shift = shift;
    S->N = N; S->log_T = log_T; S->log_m = N / 8; S->m = (size_t)1 << S->log_m; S->T = (size_t)1 << log_T;
    S->idx = idx; S->gamma = *gamma;
    S->u = (fr_t *)malloc(S->T * sizeof(fr_t)); orc_eq_evals(r_node, log_T, 0, S->u);        /* mod.rs:234 */
    for (int s = 0; s < 2; s++) { S->Q[s] = (fr_t *)malloc(S->m * sizeof(fr_t)); S->RQ[s] = (fr_t *)malloc(S->m * sizeof(fr_t)); }
    S->RP = (fr_t *)malloc(S->m * sizeof(fr_t));
    for (int p = 0; p < 8; p++) S->v[p] = (fr_t *)calloc(S->m, sizeof(fr_t));
    gse_init(&S->eq, r_node, log_T);
    init_phase(S, 0);
}

void orc_ps_relu_init(orc_ps_relu *S, const uint64_t *idx, size_t N, size_t log_T, const fr_t *r_node, const fr_t *gamma) {
    ps_unary_init(S, idx, N, log_T, r_node, gamma, 0, 0);
}
void orc_ps_rshift_init(orc_ps_relu *S, const uint64_t *idx, size_t N, size_t shift, size_t log_T, const fr_t *r_node, const fr_t *gamma) {
    ps_unary_init(S, idx, N, log_T, r_node, gamma, 1, shift);
}

void orc_ps_relu_free(orc_ps_relu *S) {
    free(S->u); for (int s = 0; s < 2; s++) { free(S->Q[s]); free(S->RQ[s]); } free(S->RP);
    for (int p = 0; p < 8; p++) free(S->v[p]);
    if (S->ra) free(S->ra);
    gse_free(&S->eq);
}

/* prefix_mle of the two ReLU prefixes (not_msb.rs:12-33, word_no_msb.rs:12-50); bits b of length blen */
static void prefix_not_msb(const orc_ps_relu *S, const fr_t *r_x, uint32_t c, size_t j, fr_t *o) {
    fr_t one; fr_one(&one);
    if (j == 0) { fr_t cc; fr_from_u64(c, &cc); fr_sub(&one, &cc, o); }
    else if (j == 1) fr_sub(&one, r_x, o);
    else *o = S->cp_notmsb;
}
static void prefix_word_no_msb(const orc_ps_relu *S, const fr_t *r_x, uint32_t c, uint64_t b, size_t blen, size_t j, fr_t *o) {
    fr_t word; if (S->has_word) word = S->cp_word; else fr_zero(&word);
    const size_t XLEN = S->N;
    if (j >= XLEN) { *o = word; return; }
    const size_t suffix_len = XLEN - j - blen - 1;
    fr_t w, t, cc; fr_from_u64(c, &cc);
    if (!r_x && j == 0) {
    } else if (!r_x) {
        const size_t x_shift = XLEN - j - 1, y_shift = x_shift - 1;
        fr_pow2((unsigned)x_shift, &w); fr_mul(&w, &cc, &t); fr_add(&word, &t, &word);
        const uint64_t msb = (b >> (blen - 1)) & 1; b %= (uint64_t)1 << (blen - 1); blen -= 1;      /* pop_msb */
        fr_pow2((unsigned)y_shift, &w); fr_from_u64(msb, &t); fr_mul(&w, &t, &t); fr_add(&word, &t, &word);
    } else {
        const size_t x_shift = XLEN - j, y_shift = x_shift - 1;
        fr_t rx; if (j == 1) fr_zero(&rx); else rx = *r_x;
        fr_pow2((unsigned)x_shift, &w); fr_mul(&w, &rx, &t); fr_add(&word, &t, &word);
        fr_pow2((unsigned)y_shift, &w); fr_mul(&w, &cc, &t); fr_add(&word, &t, &word);
    }
    fr_from_u64(b << suffix_len, &t); fr_add(&word, &t, &word);
    *o = word;
}

/* RightShiftPrefix::prefix_mle (prefixes/right_shift.rs:16-64), D = S->shift */
static void prefix_rshift(const orc_ps_relu *S, const fr_t *r_x, uint32_t c, uint64_t b, size_t blen, size_t j, fr_t *o) {
    const size_t XLEN = S->N, D = S->shift;
    if (j + blen >= XLEN || D >= XLEN) { fr_zero(o); return; }
    const size_t ubound_index = XLEN - D - 1;
    fr_t result, w, t, cc; fr_from_u64(c, &cc);
    if (S->has_rs) result = S->cp_rs; else fr_zero(&result);
    if (r_x) {
        if (j > 0) {
            const size_t prev_index = j - 1;
            if (prev_index <= ubound_index) { fr_pow2((unsigned)(ubound_index - prev_index), &w); fr_mul(&w, r_x, &t); fr_add(&result, &t, &result); }
        }
        if (j <= ubound_index) { fr_pow2((unsigned)(ubound_index - j), &w); fr_mul(&w, &cc, &t); fr_add(&result, &t, &result); }
    } else if (j <= ubound_index) { fr_pow2((unsigned)(ubound_index - j), &w); fr_mul(&w, &cc, &t); fr_add(&result, &t, &result); }
    const size_t left_shift = XLEN - 1 - j - blen;
    fr_from_u64((b << left_shift) >> D, &t); fr_add(&result, &t, &result);
    *o = result;
}
static void combine_rshift(const fr_t *prefix, const fr_t *s_one, const fr_t *s_rs, fr_t *o) {      /* right_shift.rs:54-58 */
    fr_t a; fr_mul(prefix, s_one, &a); fr_add(&a, s_rs, o);
}

static void combine(const fr_t *p_notmsb, const fr_t *p_word, const fr_t *s_one, const fr_t *s_relu, fr_t *o) {   /* relu.rs:55-59 */
    fr_t a, b; fr_mul(p_notmsb, p_word, &a); fr_mul(&a, s_one, &a); fr_mul(p_notmsb, s_relu, &b); fr_add(&a, &b, o);
}

size_t orc_ps_relu_message(orc_ps_relu *S, size_t round, const fr_t *claim, fr_t *coeffs) {
    if (round < S->N) {                                                      /* mod.rs:337-352 */
        const size_t j = round, half = S->Q_len / 2;
        size_t blen = 0; while (((size_t)1 << blen) < half) blen++;
        const fr_t *r_x = (j % 2 == 1) ? &S->r[S->n_r - 1] : 0;
        fr_t e0, e2l, e2h; fr_zero(&e0); fr_zero(&e2l); fr_zero(&e2h);
        for (size_t i = 0; i < half; i++) {                                  /* prover_msg_read_checking :354-417 */
            fr_t n0, w0, n2, w2, t;
            if (S->table == 1) {
                prefix_rshift(S, r_x, 0, i, blen, j, &w0); prefix_rshift(S, r_x, 2, i, blen, j, &w2);
                combine_rshift(&w0, &S->Q[0][i], &S->Q[1][i], &t); fr_add(&e0, &t, &e0);
                combine_rshift(&w2, &S->Q[0][i], &S->Q[1][i], &t); fr_add(&e2l, &t, &e2l);
                combine_rshift(&w2, &S->Q[0][i + half], &S->Q[1][i + half], &t); fr_add(&e2h, &t, &e2h);
                continue;
            }
            prefix_not_msb(S, r_x, 0, j, &n0); prefix_word_no_msb(S, r_x, 0, i, blen, j, &w0);
            prefix_not_msb(S, r_x, 2, j, &n2); prefix_word_no_msb(S, r_x, 2, i, blen, j, &w2);
            combine(&n0, &w0, &S->Q[0][i], &S->Q[1][i], &t); fr_add(&e0, &t, &e0);
            combine(&n2, &w2, &S->Q[0][i], &S->Q[1][i], &t); fr_add(&e2l, &t, &e2l);
            combine(&n2, &w2, &S->Q[0][i + half], &S->Q[1][i + half], &t); fr_add(&e2h, &t, &e2h);
        }
        fr_t rc0 = e0, rc2; fr_add(&e2h, &e2h, &rc2); fr_sub(&rc2, &e2l, &rc2);
        /* raf_state.prover_msg (unary.rs:55-79): sum_b identity_ps.sumcheck_evals(b), times gamma */
        fr_t o0, o2l, o2r; fr_zero(&o0); fr_zero(&o2l); fr_zero(&o2r);
        for (size_t b = 0; b < half; b++) {
            /* P[0] = SignedIdentity prefix poly (HighToLow sumcheck evals at 0 and 2), P[1] = None -> (1, 1) */
            fr_t p0 = S->RP[b], p2, m, t; fr_sub(&S->RP[b + S->RP_len / 2], &p0, &m); fr_add(&S->RP[b + S->RP_len / 2], &m, &p2);
            fr_mul(&p0, &S->RQ[0][b], &t); fr_add(&o0, &t, &o0);
            fr_mul(&p2, &S->RQ[0][b], &t); fr_add(&o2l, &t, &o2l);
            fr_mul(&p2, &S->RQ[0][b + half], &t); fr_add(&o2r, &t, &o2r);
            fr_add(&o0, &S->RQ[1][b], &o0); fr_add(&o2l, &S->RQ[1][b], &o2l); fr_add(&o2r, &S->RQ[1][b + half], &o2r);
        }
        fr_t op2; fr_add(&o2r, &o2r, &op2); fr_sub(&op2, &o2l, &op2);
        fr_mul(&o0, &S->gamma, &o0); fr_mul(&op2, &S->gamma, &op2);
        fr_t ev[2]; fr_add(&rc0, &o0, &ev[0]); fr_add(&rc2, &op2, &ev[1]);
        return orc_unipoly_from_evals_and_hint(claim, ev, 2, coeffs);
    }
    /* log T rounds (mod.rs:461-487): eval_at_0 = sum_j E_out E_in ra[2j]; gruen_poly_deg_2 (LowToHigh) */
    const gse_t *E = &S->eq;
    const fr_t *e_out = E->Eout[E->out_top], *e_in = E->Ein[E->in_top];
    const size_t out_len = (size_t)1 << E->out_top, in_len = (size_t)1 << E->in_top;
    fr_t acc; fr_zero(&acc);
    for (size_t xo = 0; xo < out_len; xo++) {
        fr_t inner; fr_zero(&inner);
        for (size_t xi = 0; xi < in_len; xi++) { const size_t jj = (xo << E->in_top) | xi; fr_t t; fr_mul(&e_in[xi], &S->ra[2 * jj], &t); fr_add(&inner, &t, &inner); }
        fr_mul(&e_out[xo], &inner, &inner); fr_add(&acc, &inner, &acc);
    }
    fr_t vv, q0; fr_add(&S->val, &S->raf_val, &vv); fr_mul(&acc, &vv, &q0);
    /* gruen_poly_deg_2 (split_eq_poly.rs:379-428), LowToHigh: w[current_index - 1] */
    fr_t eq1, eq0, eqm, eq2, c0, c1, l1, l2, inv, ev2[2], hint;
    fr_mul(&E->scalar, &E->w[E->current_index - 1], &eq1); fr_sub(&E->scalar, &eq1, &eq0);
    fr_sub(&eq1, &eq0, &eqm); fr_add(&eq1, &eqm, &eq2);
    fr_mul(&eq0, &q0, &c0); fr_sub(claim, &c0, &c1);
    fr_inv(&eq1, &inv); fr_mul(&c1, &inv, &l1);
    fr_add(&l1, &l1, &l2); fr_sub(&l2, &q0, &l2);
    ev2[0] = c0; fr_mul(&eq2, &l2, &ev2[1]); fr_add(&c0, &c1, &hint);
    return orc_unipoly_from_evals_and_hint(&hint, ev2, 2, coeffs);
}

static void bind_h2l(fr_t *z, size_t *len, const fr_t *r) { orc_bind(z, *len, r, ORC_HIGH_TO_LOW); *len /= 2; }

void orc_ps_relu_ingest(orc_ps_relu *S, size_t round, const fr_t *r) {      /* mod.rs:490-560 */
    const size_t log_m = S->log_m, LOG_K = S->N;
    S->r[S->n_r++] = *r;
    if (round < LOG_K) {
        const size_t phase = round / log_m;
        size_t ql = S->Q_len;
        for (int s = 0; s < 2; s++) { size_t l = ql; bind_h2l(S->Q[s], &l, r); l = ql; bind_h2l(S->RQ[s], &l, r); }
        S->Q_len = ql / 2;
        bind_h2l(S->RP, &S->RP_len, r);                                      /* identity_ps.bind: P and Q HighToLow */
        {                                                                    /* v[phase].update(r), HighToLow */
            const size_t n = S->v_len[phase];
            fr_t *nv = (fr_t *)calloc(S->m, sizeof(fr_t));
            for (size_t i = 0; i < n; i++) { fr_mul(r, &S->v[phase][i], &nv[2 * i + 1]); fr_sub(&S->v[phase][i], &nv[2 * i + 1], &nv[2 * i]); }
            free(S->v[phase]); S->v[phase] = nv; S->v_len[phase] = 2 * n;
        }
        if (S->n_r % 2 == 0) {                                               /* Prefixes::update_checkpoints, j = round */
            const fr_t *r_x = &S->r[S->n_r - 2], *r_y = &S->r[S->n_r - 1];
            const size_t j = round;
            fr_t one; fr_one(&one);
            if (j == 1) { fr_sub(&one, r_x, &S->cp_notmsb); S->has_notmsb = 1; }          /* not_msb.rs:35-50 */
            {                                                                /* word_no_msb.rs:52-70 */
                fr_t word; if (S->has_word) word = S->cp_word; else fr_zero(&word);
                const size_t x_shift = LOG_K >= j ? LOG_K - j : 0, y_shift = x_shift ? x_shift - 1 : 0;
                fr_t rx, w, t; if (j == 1) fr_zero(&rx); else rx = *r_x;
                fr_pow2((unsigned)x_shift, &w); fr_mul(&w, &rx, &t); fr_add(&word, &t, &word);
                fr_pow2((unsigned)y_shift, &w); fr_mul(&w, r_y, &t); fr_add(&word, &t, &word);
                S->cp_word = word; S->has_word = 1;
            }
            if (S->table == 1 && S->shift < LOG_K) {                         /* prefixes/right_shift.rs:66-98 (suffix_len guard never fires) */
                const size_t ubound_index = LOG_K - S->shift - 1;
                fr_t res, w, t; if (S->has_rs) res = S->cp_rs; else fr_zero(&res);
                if (j > 0 && j - 1 <= ubound_index) { fr_pow2((unsigned)(ubound_index - (j - 1)), &w); fr_mul(&w, r_x, &t); fr_add(&res, &t, &res); }
                if (j <= ubound_index) { fr_pow2((unsigned)(ubound_index - j), &w); fr_mul(&w, r_y, &t); fr_add(&res, &t, &res); }
                S->cp_rs = res; S->has_rs = 1;
            }
        }
        if ((round + 1) % log_m == 0) {
            S->sid_cp = S->RP[0]; S->has_sid = 1;                            /* prefix_registry.update_checkpoints */
            if (phase != 7) init_phase(S, phase + 1);
        }
        if (round + 1 == LOG_K) {
            /* val = combine(prefix checkpoints, suffix_mle(empty)) : One -> 1, WordNoMSB -> 0 */
            fr_t one, zero; fr_one(&one); fr_zero(&zero);
            combine(&S->cp_notmsb, &S->cp_word, &one, &zero, &S->val);
            if (S->table == 1) { if (S->has_rs) S->val = S->cp_rs; else fr_zero(&S->val); }   /* One -> 1, RightShift suffix of no bits -> 0 */
            fr_mul(&S->gamma, &S->sid_cp, &S->raf_val);                       /* unary.rs:85-88 */
            S->ra = (fr_t *)malloc(S->T * sizeof(fr_t)); S->ra_len = S->T;    /* init_log_t_rounds :419-446 */
            for (size_t t = 0; t < S->T; t++) {
                fr_t p; fr_one(&p);
                for (size_t ph = 0; ph < 8; ph++) {
                    const uint64_t kb = split_prefix(S->idx[t], (8 - 1 - ph) * log_m) & (S->m - 1);
                    fr_mul(&p, &S->v[ph][kb], &p);
                }
                S->ra[t] = p;
            }
        }
    } else {
        orc_bind(S->ra, S->ra_len, r, ORC_LOW_TO_HIGH); S->ra_len /= 2;
        gse_bind(&S->eq, r);
    }
}

/* ------------------------------------------------------------------ IdentityRCProver
 * subprotocols/identity_range_check.rs:196-420 with IdentityPolynomial's prefix-suffix decomposition
 * (poly/identity_poly.rs:112-165): suffixes [Shift: 2^len(b), Identity: b], prefix polynomial
 * bound * 2^chunk_len + i, UNSIGNED accumulation (prefix_suffix.rs:237-330). */
static void idrc_init_phase(orc_ps_identity *S, size_t phase) {
    const size_t log_m = S->log_m, m = S->m, m_mask = m - 1, T = S->T;
    if (phase != 0)
        for (size_t t = 0; t < T; t++) {
            const uint64_t k_bound = split_prefix(S->idx[t], (S->phases - phase) * log_m) & m_mask;
            fr_mul(&S->u[t], &S->v[phase - 1][k_bound], &S->u[t]);
        }
    const size_t suffix_len = (S->phases - 1 - phase) * log_m;
    for (int s = 0; s < 2; s++) for (size_t y = 0; y < m; y++) fr_zero(&S->Q[s][y]);
    fr_t shift; fr_pow2((unsigned)suffix_len, &shift);                       /* ShiftSuffixPolynomial: 1 << b.len() */
    for (size_t t = 0; t < T; t++) {
        const uint64_t y = split_prefix(S->idx[t], suffix_len) & m_mask, sb = split_suffix(S->idx[t], suffix_len);
        fr_t x; fr_mul(&S->u[t], &shift, &x); fr_add(&S->Q[0][y], &x, &S->Q[0][y]);
        if (sb) { fr_t w; fr_from_u64(sb, &w); fr_mul(&S->u[t], &w, &x); fr_add(&S->Q[1][y], &x, &S->Q[1][y]); }
    }
    S->Q_len = m;
    fr_t bound, sc; if (S->has_cp) bound = S->cp; else fr_zero(&bound);
    fr_pow2((unsigned)log_m, &sc); fr_mul(&bound, &sc, &bound);              /* bound_value.mul_u128(1 << chunk_len) */
    for (size_t i = 0; i < m; i++) { fr_t vi; fr_from_u64(i, &vi); fr_add(&bound, &vi, &S->P[i]); }
    S->P_len = m;
    fr_one(&S->v[phase][0]); S->v_len[phase] = 1;
}

void orc_ps_identity_init(orc_ps_identity *S, const uint64_t *idx, size_t log_K, size_t phases, size_t log_T, const fr_t *r_node) {
    memset(S, 0, sizeof *S);
    S->log_K = log_K; S->phases = phases; S->log_T = log_T; S->log_m = log_K / phases; S->m = (size_t)1 << S->log_m; S->T = (size_t)1 << log_T;
    S->idx = idx;
    S->u = (fr_t *)malloc(S->T * sizeof(fr_t)); orc_eq_evals(r_node, log_T, 0, S->u);
    for (int s = 0; s < 2; s++) S->Q[s] = (fr_t *)malloc(S->m * sizeof(fr_t));
    S->P = (fr_t *)malloc(S->m * sizeof(fr_t));
    for (size_t p = 0; p < phases; p++) S->v[p] = (fr_t *)calloc(S->m, sizeof(fr_t));
    gse_init(&S->eq, r_node, log_T);
    idrc_init_phase(S, 0);
}

void orc_ps_identity_free(orc_ps_identity *S) {
    free(S->u); free(S->Q[0]); free(S->Q[1]); free(S->P);
    for (size_t p = 0; p < S->phases; p++) free(S->v[p]);
    if (S->ra) free(S->ra);
    gse_free(&S->eq);
}

size_t orc_ps_identity_message(orc_ps_identity *S, size_t round, const fr_t *claim, fr_t *coeffs) {
    if (round < S->log_K) {
        const size_t half = S->Q_len / 2;
        fr_t o0, o2l, o2r; fr_zero(&o0); fr_zero(&o2l); fr_zero(&o2r);
        for (size_t b = 0; b < half; b++) {                                  /* identity_ps.sumcheck_evals(b) */
            fr_t p0 = S->P[b], p2, m, t; fr_sub(&S->P[b + S->P_len / 2], &p0, &m); fr_add(&S->P[b + S->P_len / 2], &m, &p2);
            fr_mul(&p0, &S->Q[0][b], &t); fr_add(&o0, &t, &o0);
            fr_mul(&p2, &S->Q[0][b], &t); fr_add(&o2l, &t, &o2l);
            fr_mul(&p2, &S->Q[0][b + half], &t); fr_add(&o2r, &t, &o2r);
            fr_add(&o0, &S->Q[1][b], &o0); fr_add(&o2l, &S->Q[1][b], &o2l); fr_add(&o2r, &S->Q[1][b + half], &o2r);
        }
        fr_t ev[2]; ev[0] = o0; fr_add(&o2r, &o2r, &ev[1]); fr_sub(&ev[1], &o2l, &ev[1]);
        return orc_unipoly_from_evals_and_hint(claim, ev, 2, coeffs);
    }
    const gse_t *E = &S->eq;
    const fr_t *e_out = E->Eout[E->out_top], *e_in = E->Ein[E->in_top];
    const size_t out_len = (size_t)1 << E->out_top, in_len = (size_t)1 << E->in_top;
    fr_t acc; fr_zero(&acc);
    for (size_t xo = 0; xo < out_len; xo++) {
        fr_t inner; fr_zero(&inner);
        for (size_t xi = 0; xi < in_len; xi++) { const size_t jj = (xo << E->in_top) | xi; fr_t t; fr_mul(&e_in[xi], &S->ra[2 * jj], &t); fr_add(&inner, &t, &inner); }
        fr_mul(&e_out[xo], &inner, &inner); fr_add(&acc, &inner, &acc);
    }
    fr_t q0; fr_mul(&acc, &S->raf_val, &q0);
    fr_t eq1, eq0, eqm, eq2, c0, c1, l1, l2, inv, ev2[2], hint;
    fr_mul(&E->scalar, &E->w[E->current_index - 1], &eq1); fr_sub(&E->scalar, &eq1, &eq0);
    fr_sub(&eq1, &eq0, &eqm); fr_add(&eq1, &eqm, &eq2);
    fr_mul(&eq0, &q0, &c0); fr_sub(claim, &c0, &c1);
    fr_inv(&eq1, &inv); fr_mul(&c1, &inv, &l1);
    fr_add(&l1, &l1, &l2); fr_sub(&l2, &q0, &l2);
    ev2[0] = c0; fr_mul(&eq2, &l2, &ev2[1]); fr_add(&c0, &c1, &hint);
    return orc_unipoly_from_evals_and_hint(&hint, ev2, 2, coeffs);
}

void orc_ps_identity_ingest(orc_ps_identity *S, size_t round, const fr_t *r) {
    const size_t log_m = S->log_m;
    if (round < S->log_K) {
        const size_t phase = round / log_m;
        size_t ql = S->Q_len;
        for (int s = 0; s < 2; s++) { size_t l = ql; bind_h2l(S->Q[s], &l, r); }
        S->Q_len = ql / 2;
        bind_h2l(S->P, &S->P_len, r);
        {
            const size_t n = S->v_len[phase];
            fr_t *nv = (fr_t *)calloc(S->m, sizeof(fr_t));
            for (size_t i = 0; i < n; i++) { fr_mul(r, &S->v[phase][i], &nv[2 * i + 1]); fr_sub(&S->v[phase][i], &nv[2 * i + 1], &nv[2 * i]); }
            free(S->v[phase]); S->v[phase] = nv; S->v_len[phase] = 2 * n;
        }
        if ((round + 1) % log_m == 0) {
            S->cp = S->P[0]; S->has_cp = 1;
            if (phase != S->phases - 1) idrc_init_phase(S, phase + 1);
        }
        if (round + 1 == S->log_K) {
            S->raf_val = S->cp;
            S->ra = (fr_t *)malloc(S->T * sizeof(fr_t)); S->ra_len = S->T;
            for (size_t t = 0; t < S->T; t++) {
                fr_t p; fr_one(&p);
                for (size_t ph = 0; ph < S->phases; ph++) {
                    const uint64_t kb = split_prefix(S->idx[t], (S->phases - 1 - ph) * log_m) & (S->m - 1);
                    fr_mul(&p, &S->v[ph][kb], &p);
                }
                S->ra[t] = p;
            }
        }
    } else {
        orc_bind(S->ra, S->ra_len, r, ORC_LOW_TO_HIGH); S->ra_len /= 2;
        gse_bind(&S->eq, r);
    }
}

/* ------------------------------------------------------------------ clamp family (unary read-raf)
 * ClampBoundedTable<XLEN, BOUND, SYMMETRIC> (lookup_tables/clamp.rs:44-121): suffixes
 * [HigherAllZero, HZeroMulLWord, HOneMulLWord, One] restated from suffixes/higher_all_zero.rs,
 * hzero_mul_lword.rs, hone_mul_lword.rs; `combine` as written (clamp.rs:84-109); prefixes
 * HigherAllZero / HigherAllOne / LowerWord / Msb evaluated as the multilinear extensions they stand for
 * (clamp.rs:140-195 evaluate_mle gives the same factors); RAF = SignedIdentity as in the ReLU case. */
static uint32_t suf_haz(uint64_t bits, size_t len, size_t XLEN, size_t BOUND) {
    const size_t bound_index = XLEN - BOUND - 1, start = XLEN - len;
    for (size_t pos = 0; pos < len; pos++) if (start + pos <= bound_index && ((bits >> (len - 1 - pos)) & 1)) return 0;
    return 1;
}
static uint32_t suf_hz_lw(uint64_t bits, size_t len, size_t XLEN, size_t BOUND) {
    const size_t bound_index = XLEN - BOUND - 1, start = XLEN - len;
    uint32_t lw = 0;
    for (size_t pos = 0; pos < len; pos++) {
        const size_t gi = start + pos; const uint32_t bit = (uint32_t)((bits >> (len - 1 - pos)) & 1);
        if (gi <= bound_index && bit == 1) return 0;
        if (gi > bound_index) lw += bit << (XLEN - gi - 1);
    }
    return lw;
}
static uint32_t suf_ho_lw(uint64_t bits, size_t len, size_t XLEN, size_t BOUND) {
    const size_t bound_index = XLEN - BOUND - 1, start = XLEN - len;
    uint32_t lw = 0;
    for (size_t pos = 0; pos < len; pos++) {
        const size_t gi = start + pos; const uint32_t bit = (uint32_t)((bits >> (len - 1 - pos)) & 1);
        if (gi <= bound_index && bit == 0) return 0;
        if (gi > bound_index) lw += bit << (XLEN - gi - 1);
    }
    return lw;
}

static void clamp_init_phase(orc_ps_clamp *S, size_t phase) {
    const size_t log_m = S->log_m, m = S->m, m_mask = m - 1, T = S->T;
    if (phase != 0)
        for (size_t t = 0; t < T; t++) {
            const uint64_t k_bound = split_prefix(S->idx[t], (8 - phase) * log_m) & m_mask;
            fr_mul(&S->u[t], &S->v[phase - 1][k_bound], &S->u[t]);
        }
    const size_t suffix_len = (8 - 1 - phase) * log_m;
    for (int s = 0; s < 6; s++) for (size_t y = 0; y < m; y++) fr_zero(&S->Q[s][y]);
    for (size_t t = 0; t < T; t++) {
        const uint64_t y = split_prefix(S->idx[t], suffix_len) & m_mask, sb = split_suffix(S->idx[t], suffix_len);
        const uint32_t tv[4] = {suf_haz(sb, suffix_len, S->N, S->bound), suf_hz_lw(sb, suffix_len, S->N, S->bound),
                                S->symmetric ? suf_ho_lw(sb, suffix_len, S->N, S->bound) : 1u, 1u};
        for (int s = 0; s < 4; s++) if (tv[s]) { fr_t w, x; fr_from_u64(tv[s], &w); fr_mul(&S->u[t], &w, &x); fr_add(&S->Q[s][y], &x, &S->Q[s][y]); }
        fr_add(&S->Q[4][y], &S->u[t], &S->Q[4][y]);                         /* RAF One */
        if (sb) { fr_t w, x; fr_from_i64((int64_t)sb, &w); fr_mul(&S->u[t], &w, &x); fr_add(&S->Q[5][y], &x, &S->Q[5][y]); }
    }
    S->Q_len = m;
    fr_one(&S->v[phase][0]); S->v_len[phase] = 1;
}

void orc_ps_clamp_init(orc_ps_clamp *S, const uint64_t *idx, size_t N, size_t bound, int symmetric, size_t log_T,
                       const fr_t *r_node, const fr_t *gamma) {
    memset(S, 0, sizeof *S);
    S->N = N; S->bound = bound; S->symmetric = symmetric; S->log_T = log_T; S->log_m = N / 8; S->m = (size_t)1 << S->log_m; S->T = (size_t)1 << log_T;
    S->idx = idx; S->gamma = *gamma;
    S->u = (fr_t *)malloc(S->T * sizeof(fr_t)); orc_eq_evals(r_node, log_T, 0, S->u);
    for (int s = 0; s < 6; s++) S->Q[s] = (fr_t *)malloc(S->m * sizeof(fr_t));
    for (int p = 0; p < 8; p++) S->v[p] = (fr_t *)calloc(S->m, sizeof(fr_t));
    gse_init(&S->eq, r_node, log_T);
    clamp_init_phase(S, 0);
}

void orc_ps_clamp_free(orc_ps_clamp *S) {
    free(S->u); for (int s = 0; s < 6; s++) free(S->Q[s]);
    for (int p = 0; p < 8; p++) free(S->v[p]);
    if (S->ra) free(S->ra);
    gse_free(&S->eq);
}

/* prefix MLEs at (r_0..r_{j-1}, c, b): x = the first j + 1 + blen variables */
static void clamp_prefixes(const orc_ps_clamp *S, size_t j, uint32_t c, uint64_t b, size_t blen,
                           fr_t *haz, fr_t *hao, fr_t *lw, fr_t *msb, fr_t *sid) {
    const size_t XLEN = S->N, h = XLEN - S->bound;                          /* variables 0..h-1 are the high ones */
    fr_t one, x; fr_one(&one); fr_one(haz); fr_one(hao); fr_zero(lw); fr_zero(sid);
    for (size_t i = 0; i < j + 1 + blen; i++) {
        if (i < j) x = S->r[i];
        else if (i == j) fr_from_u64(c, &x);
        else fr_from_u64((b >> (blen - 1 - (i - j - 1))) & 1, &x);
        if (i == 0) *msb = x;
        fr_t w, t; fr_pow2((unsigned)(XLEN - 1 - i), &w);
        if (i < h) { fr_sub(&one, &x, &t); fr_mul(haz, &t, haz); fr_mul(hao, &x, hao); }
        else { fr_mul(&w, &x, &t); fr_add(lw, &t, lw); }
        fr_mul(&w, &x, &t); fr_add(sid, &t, sid);
        if (i == 0) { fr_t pen; fr_pow2((unsigned)XLEN, &pen); fr_mul(&pen, &x, &t); fr_sub(sid, &t, sid); }
    }
    if (!S->symmetric) fr_zero(hao);                                         /* ZeroPrefix */
}

static void clamp_combine(const orc_ps_clamp *S, const fr_t *haz, const fr_t *hao, const fr_t *lw, const fr_t *msb,
                          const fr_t *s_haz, const fr_t *s_hzlw, const fr_t *s_holw, const fr_t *s_one, fr_t *o) {   /* clamp.rs:84-109 */
    fr_t cu, lc, one, t, u, acc; fr_one(&one);
    fr_from_u64(((uint64_t)1 << S->bound) - 1, &cu);
    if (S->symmetric) { fr_add(&cu, &cu, &lc); fr_add(&lc, &one, &lc); } else lc = cu;
    fr_mul(s_one, &cu, &acc);
    fr_mul(msb, s_one, &t); fr_mul(&t, &lc, &t); fr_sub(&acc, &t, &acc);
    fr_mul(lw, s_one, &t); fr_add(s_hzlw, &t, &t); fr_mul(s_haz, &cu, &u); fr_sub(&t, &u, &t); fr_mul(haz, &t, &t); fr_add(&acc, &t, &acc);
    fr_mul(lw, s_one, &t); fr_add(s_holw, &t, &t); fr_mul(hao, &t, &t); fr_add(&acc, &t, &acc);
    *o = acc;
}

size_t orc_ps_clamp_message(orc_ps_clamp *S, size_t round, const fr_t *claim, fr_t *coeffs) {
    if (round < S->N) {
        const size_t j = round, half = S->Q_len / 2;
        size_t blen = 0; while (((size_t)1 << blen) < half) blen++;
        fr_t ev[2];
        for (int ci = 0; ci < 2; ci++) {
            const uint32_t c = ci ? 2 : 0;
            fr_t acc; fr_zero(&acc);
            for (size_t i = 0; i < half; i++) {
                fr_t haz, hao, lw, msb, sid, q[6], t;
                clamp_prefixes(S, j, c, i, blen, &haz, &hao, &lw, &msb, &sid);
                for (int s = 0; s < 6; s++) {
                    if (ci == 0) q[s] = S->Q[s][i];
                    else { fr_add(&S->Q[s][i + half], &S->Q[s][i + half], &q[s]); fr_sub(&q[s], &S->Q[s][i], &q[s]); }
                }
                clamp_combine(S, &haz, &hao, &lw, &msb, &q[0], &q[1], &q[2], &q[3], &t); fr_add(&acc, &t, &acc);
                fr_mul(&sid, &q[4], &t); fr_add(&t, &q[5], &t); fr_mul(&t, &S->gamma, &t); fr_add(&acc, &t, &acc);
            }
            ev[ci] = acc;
        }
        return orc_unipoly_from_evals_and_hint(claim, ev, 2, coeffs);
    }
    const gse_t *E = &S->eq;
    const fr_t *e_out = E->Eout[E->out_top], *e_in = E->Ein[E->in_top];
    const size_t out_len = (size_t)1 << E->out_top, in_len = (size_t)1 << E->in_top;
    fr_t acc; fr_zero(&acc);
    for (size_t xo = 0; xo < out_len; xo++) {
        fr_t inner; fr_zero(&inner);
        for (size_t xi = 0; xi < in_len; xi++) { const size_t jj = (xo << E->in_top) | xi; fr_t t; fr_mul(&e_in[xi], &S->ra[2 * jj], &t); fr_add(&inner, &t, &inner); }
        fr_mul(&e_out[xo], &inner, &inner); fr_add(&acc, &inner, &acc);
    }
    fr_t q0; fr_mul(&acc, &S->wv, &q0);
    fr_t eq1, eq0, eqm, eq2, c0, c1, l1, l2, inv, ev2[2], hint;
    fr_mul(&E->scalar, &E->w[E->current_index - 1], &eq1); fr_sub(&E->scalar, &eq1, &eq0);
    fr_sub(&eq1, &eq0, &eqm); fr_add(&eq1, &eqm, &eq2);
    fr_mul(&eq0, &q0, &c0); fr_sub(claim, &c0, &c1);
    fr_inv(&eq1, &inv); fr_mul(&c1, &inv, &l1);
    fr_add(&l1, &l1, &l2); fr_sub(&l2, &q0, &l2);
    ev2[0] = c0; fr_mul(&eq2, &l2, &ev2[1]); fr_add(&c0, &c1, &hint);
    return orc_unipoly_from_evals_and_hint(&hint, ev2, 2, coeffs);
}

void orc_ps_clamp_ingest(orc_ps_clamp *S, size_t round, const fr_t *r) {
    const size_t log_m = S->log_m, LOG_K = S->N;
    if (round < LOG_K) {
        S->r[S->n_r++] = *r;
        const size_t phase = round / log_m;
        size_t ql = S->Q_len;
        for (int s = 0; s < 6; s++) { size_t l = ql; bind_h2l(S->Q[s], &l, r); }
        S->Q_len = ql / 2;
        {
            const size_t n = S->v_len[phase];
            fr_t *nv = (fr_t *)calloc(S->m, sizeof(fr_t));
            for (size_t i = 0; i < n; i++) { fr_mul(r, &S->v[phase][i], &nv[2 * i + 1]); fr_sub(&S->v[phase][i], &nv[2 * i + 1], &nv[2 * i]); }
            free(S->v[phase]); S->v[phase] = nv; S->v_len[phase] = 2 * n;
        }
        if ((round + 1) % log_m == 0 && phase != 7) clamp_init_phase(S, phase + 1);
        if (round + 1 == LOG_K) {
            /* val = combine(prefixes at r, suffix_mle(empty)): HAZ = 1, HZeroMulLWord = 0, HOneMulLWord = 0 (One for softmax), One = 1 */
            fr_t haz, hao, lw, msb, sid, one, zero, s3; fr_one(&one); fr_zero(&zero);
            S->r[S->n_r] = zero;                                             /* placeholder for the c slot */
            {   /* all N variables bound: evaluate with j = N - 1, c = r_{N-1} handled by a direct loop */
                const size_t XLEN = S->N, h = XLEN - S->bound;
                fr_one(&haz); fr_one(&hao); fr_zero(&lw); fr_zero(&sid);
                for (size_t i = 0; i < XLEN; i++) {
                    const fr_t x = S->r[i]; fr_t w, t; fr_pow2((unsigned)(XLEN - 1 - i), &w);
                    if (i == 0) msb = x;
                    if (i < h) { fr_sub(&one, &x, &t); fr_mul(&haz, &t, &haz); fr_mul(&hao, &x, &hao); }
                    else { fr_mul(&w, &x, &t); fr_add(&lw, &t, &lw); }
                    fr_mul(&w, &x, &t); fr_add(&sid, &t, &sid);
                    if (i == 0) { fr_t pen; fr_pow2((unsigned)XLEN, &pen); fr_mul(&pen, &x, &t); fr_sub(&sid, &t, &sid); }
                }
                if (!S->symmetric) fr_zero(&hao);
            }
            s3 = S->symmetric ? zero : one;
            fr_t val; clamp_combine(S, &haz, &hao, &lw, &msb, &one, &zero, &s3, &one, &val);
            fr_t rv; fr_mul(&S->gamma, &sid, &rv); fr_add(&val, &rv, &S->wv);
            S->ra = (fr_t *)malloc(S->T * sizeof(fr_t)); S->ra_len = S->T;
            for (size_t t = 0; t < S->T; t++) {
                fr_t p; fr_one(&p);
                for (size_t ph = 0; ph < 8; ph++) {
                    const uint64_t kb = split_prefix(S->idx[t], (8 - 1 - ph) * log_m) & (S->m - 1);
                    fr_mul(&p, &S->v[ph][kb], &p);
                }
                S->ra[t] = p;
            }
        }
    } else {
        orc_bind(S->ra, S->ra_len, r, ORC_LOW_TO_HIGH); S->ra_len /= 2;
        gse_bind(&S->eq, r);
    }
}

/* ------------------------------------------------------------------ binary read-raf: UnsignedLessThanTable<32>
 * ps_shout/binary.rs:26-200 (LOG_K = 64 interleaved operand bits, RAF = gamma * SignedLeft + gamma^2 * SignedRight),
 * lookup_tables/unsigned_less_than.rs:16-60 (prefixes [Eq, LessThan], suffixes [One, LessThan], combine),
 * suffixes/less_than.rs + utils/mod.rs:105-125 (uninterleave_bits), poly/signed_identity_poly.rs:213-250
 * (SignedOperandPoly::evaluate).  Prefixes are evaluated as the multilinear extensions they stand for. */
static void uninterleave(uint64_t val, uint32_t *x, uint32_t *y) {           /* utils/mod.rs:105-125 */
    uint64_t xb = (val >> 1) & 0x5555555555555555ull, yb = val & 0x5555555555555555ull;
    xb = (xb | (xb >> 1)) & 0x3333333333333333ull; xb = (xb | (xb >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    xb = (xb | (xb >> 4)) & 0x00FF00FF00FF00FFull; xb = (xb | (xb >> 8)) & 0x0000FFFF0000FFFFull; xb = (xb | (xb >> 16)) & 0xFFFFFFFFull;
    yb = (yb | (yb >> 1)) & 0x3333333333333333ull; yb = (yb | (yb >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    yb = (yb | (yb >> 4)) & 0x00FF00FF00FF00FFull; yb = (yb | (yb >> 8)) & 0x0000FFFF0000FFFFull; yb = (yb | (yb >> 16)) & 0xFFFFFFFFull;
    *x = (uint32_t)xb; *y = (uint32_t)yb;
}

static void ult_init_phase(orc_ps_ult *S, size_t phase) {
    const size_t log_m = 8, m = 256, m_mask = 255, T = S->T;
    if (phase != 0)
        for (size_t t = 0; t < T; t++) {
            const uint64_t k_bound = split_prefix(S->idx[t], (8 - phase) * log_m) & m_mask;
            fr_mul(&S->u[t], &S->v[phase - 1][k_bound], &S->u[t]);
        }
    const size_t suffix_len = (8 - 1 - phase) * log_m;
    for (int s = 0; s < 4; s++) for (size_t y = 0; y < m; y++) fr_zero(&S->Q[s][y]);
    for (size_t t = 0; t < T; t++) {
        const uint64_t y = split_prefix(S->idx[t], suffix_len) & m_mask, sb = split_suffix(S->idx[t], suffix_len);
        uint32_t sx, sy; uninterleave(sb, &sx, &sy);
        fr_add(&S->Q[0][y], &S->u[t], &S->Q[0][y]);                         /* One */
        if (sx < sy) fr_add(&S->Q[1][y], &S->u[t], &S->Q[1][y]);            /* LessThan suffix */
        if (sx) { fr_t w, x; fr_from_u64(sx, &w); fr_mul(&S->u[t], &w, &x); fr_add(&S->Q[2][y], &x, &S->Q[2][y]); }   /* left operand of the suffix */
        if (sy) { fr_t w, x; fr_from_u64(sy, &w); fr_mul(&S->u[t], &w, &x); fr_add(&S->Q[3][y], &x, &S->Q[3][y]); }   /* right operand */
    }
    S->Q_len = m;
    fr_one(&S->v[phase][0]); S->v_len[phase] = 1;
}

void orc_ps_ult_init(orc_ps_ult *S, const uint64_t *idx, size_t log_T, const fr_t *r_node, const fr_t *gamma) {
    memset(S, 0, sizeof *S);
    S->log_T = log_T; S->T = (size_t)1 << log_T; S->idx = idx; S->gamma = *gamma;
    S->u = (fr_t *)malloc(S->T * sizeof(fr_t)); orc_eq_evals(r_node, log_T, 0, S->u);
    for (int s = 0; s < 4; s++) S->Q[s] = (fr_t *)malloc(256 * sizeof(fr_t));
    for (int p = 0; p < 8; p++) S->v[p] = (fr_t *)calloc(256, sizeof(fr_t));
    gse_init(&S->eq, r_node, log_T);
    ult_init_phase(S, 0);
}

void orc_ps_ult_free(orc_ps_ult *S) {
    free(S->u); for (int s = 0; s < 4; s++) free(S->Q[s]);
    for (int p = 0; p < 8; p++) free(S->v[p]);
    if (S->ra) free(S->ra);
    gse_free(&S->eq);
}

/* LT~, EQ~, SignedLeft~, SignedRight~ over the first nv variables x[0..nv) (nv even) of the 64 */
static void ult_prefixes(const fr_t *x, size_t nv, fr_t *lt, fr_t *eq, fr_t *lo, fr_t *ro) {
    fr_t one; fr_one(&one); fr_zero(lt); fr_one(eq); fr_zero(lo); fr_zero(ro);
    for (size_t i = 0; 2 * i + 1 < nv + 1 && 2 * i < nv; i++) {
        const fr_t xi = x[2 * i], yi = x[2 * i + 1];
        fr_t a, b, t, w;
        fr_sub(&one, &xi, &a); fr_mul(&a, &yi, &t); fr_mul(&t, eq, &t); fr_add(lt, &t, lt);       /* (1-x) y eq */
        fr_sub(&one, &yi, &b); fr_mul(&a, &b, &a); fr_mul(&xi, &yi, &t); fr_add(&a, &t, &a); fr_mul(eq, &a, eq);
        fr_pow2((unsigned)(32 - 1 - i), &w);
        fr_mul(&w, &xi, &t); fr_add(lo, &t, lo); fr_mul(&w, &yi, &t); fr_add(ro, &t, ro);
        if (i == 0) { fr_t pen; fr_pow2(32, &pen); fr_mul(&pen, &xi, &t); fr_sub(lo, &t, lo); fr_mul(&pen, &yi, &t); fr_sub(ro, &t, ro); }
    }
}

size_t orc_ps_ult_message(orc_ps_ult *S, size_t round, const fr_t *claim, fr_t *coeffs) {
    if (round < 64) {
        const size_t j = round, half = S->Q_len / 2;
        size_t blen = 0; while (((size_t)1 << blen) < half) blen++;
        fr_t g2; fr_mul(&S->gamma, &S->gamma, &g2);
        fr_t ev[2];
        for (int ci = 0; ci < 2; ci++) {
            fr_t acc; fr_zero(&acc);
            for (size_t i = 0; i < half; i++) {
                fr_t x[64]; const size_t nv = j + 1 + blen;                  /* a whole number of (x, y) pairs: chunks are 8 bits */
                for (size_t q = 0; q < nv; q++) {
                    if (q < j) x[q] = S->r[q];
                    else if (q == j) fr_from_u64(ci ? 2 : 0, &x[q]);
                    else fr_from_u64((i >> (blen - 1 - (q - j - 1))) & 1, &x[q]);
                }
                fr_t lt, eq, lo, ro, q4[4], t, u;
                ult_prefixes(x, nv, &lt, &eq, &lo, &ro);
                for (int s = 0; s < 4; s++) {
                    if (ci == 0) q4[s] = S->Q[s][i];
                    else { fr_add(&S->Q[s][i + half], &S->Q[s][i + half], &q4[s]); fr_sub(&q4[s], &S->Q[s][i], &q4[s]); }
                }
                fr_mul(&lt, &q4[0], &t); fr_mul(&eq, &q4[1], &u); fr_add(&t, &u, &t); fr_add(&acc, &t, &acc);   /* combine: lt * one + eq * lt_suffix */
                fr_mul(&lo, &q4[0], &t); fr_add(&t, &q4[2], &t); fr_mul(&t, &S->gamma, &t); fr_add(&acc, &t, &acc);
                fr_mul(&ro, &q4[0], &t); fr_add(&t, &q4[3], &t); fr_mul(&t, &g2, &t); fr_add(&acc, &t, &acc);
            }
            ev[ci] = acc;
        }
        return orc_unipoly_from_evals_and_hint(claim, ev, 2, coeffs);
    }
    const gse_t *E = &S->eq;
    const fr_t *e_out = E->Eout[E->out_top], *e_in = E->Ein[E->in_top];
    const size_t out_len = (size_t)1 << E->out_top, in_len = (size_t)1 << E->in_top;
    fr_t acc; fr_zero(&acc);
    for (size_t xo = 0; xo < out_len; xo++) {
        fr_t inner; fr_zero(&inner);
        for (size_t xi = 0; xi < in_len; xi++) { const size_t jj = (xo << E->in_top) | xi; fr_t t; fr_mul(&e_in[xi], &S->ra[2 * jj], &t); fr_add(&inner, &t, &inner); }
        fr_mul(&e_out[xo], &inner, &inner); fr_add(&acc, &inner, &acc);
    }
    fr_t q0; fr_mul(&acc, &S->wv, &q0);
    fr_t eq1, eq0, eqm, eq2, c0, c1, l1, l2, inv, ev2[2], hint;
    fr_mul(&E->scalar, &E->w[E->current_index - 1], &eq1); fr_sub(&E->scalar, &eq1, &eq0);
    fr_sub(&eq1, &eq0, &eqm); fr_add(&eq1, &eqm, &eq2);
    fr_mul(&eq0, &q0, &c0); fr_sub(claim, &c0, &c1);
    fr_inv(&eq1, &inv); fr_mul(&c1, &inv, &l1);
    fr_add(&l1, &l1, &l2); fr_sub(&l2, &q0, &l2);
    ev2[0] = c0; fr_mul(&eq2, &l2, &ev2[1]); fr_add(&c0, &c1, &hint);
    return orc_unipoly_from_evals_and_hint(&hint, ev2, 2, coeffs);
}

void orc_ps_ult_ingest(orc_ps_ult *S, size_t round, const fr_t *r) {
    if (round < 64) {
        S->r[S->n_r++] = *r;
        const size_t phase = round / 8;
        size_t ql = S->Q_len;
        for (int s = 0; s < 4; s++) { size_t l = ql; bind_h2l(S->Q[s], &l, r); }
        S->Q_len = ql / 2;
        {
            const size_t n = S->v_len[phase];
            fr_t *nv = (fr_t *)calloc(256, sizeof(fr_t));
            for (size_t i = 0; i < n; i++) { fr_mul(r, &S->v[phase][i], &nv[2 * i + 1]); fr_sub(&S->v[phase][i], &nv[2 * i + 1], &nv[2 * i]); }
            free(S->v[phase]); S->v[phase] = nv; S->v_len[phase] = 2 * n;
        }
        if ((round + 1) % 8 == 0 && phase != 7) ult_init_phase(S, phase + 1);
        if (round + 1 == 64) {
            fr_t lt, eq, lo, ro, t, g2; fr_mul(&S->gamma, &S->gamma, &g2);
            ult_prefixes(S->r, 64, &lt, &eq, &lo, &ro);                      /* val = lt (LessThan suffix of empty bits = 0) */
            S->wv = lt; fr_mul(&S->gamma, &lo, &t); fr_add(&S->wv, &t, &S->wv); fr_mul(&g2, &ro, &t); fr_add(&S->wv, &t, &S->wv);
            S->ra = (fr_t *)malloc(S->T * sizeof(fr_t)); S->ra_len = S->T;
            for (size_t tt = 0; tt < S->T; tt++) {
                fr_t p; fr_one(&p);
                for (size_t ph = 0; ph < 8; ph++) { const uint64_t kb = split_prefix(S->idx[tt], (8 - 1 - ph) * 8) & 255; fr_mul(&p, &S->v[ph][kb], &p); }
                S->ra[tt] = p;
            }
        }
    } else {
        orc_bind(S->ra, S->ra_len, r, ORC_LOW_TO_HIGH); S->ra_len /= 2;
        gse_bind(&S->eq, r);
    }
}
