/* ORACLE — TEST INFRASTRUCTURE ONLY (see bn254.h header note; parity unpinned).
 * Shout lookup argument, prover side (joltworks/src/subprotocols/shout.rs):
 *   ReadRafProver::initialize :193-225   G[idx_j] += E[j]
 *   ReadRafProver::{compute_message, ingest_challenge} :233-262  with val, G and
 *     IdentityPolynomial (poly/identity_poly.rs:21-118) bound separately, HighToLow
 *   compute_ra_evals :550-598, OneHotParams::lookup_index_chunk (config.rs:73-75) */
#include "oracle.h"
#include "shout.h"
#include <stdlib.h>
#include <string.h>

void orc_shout_G(const uint64_t *idx, size_t T, size_t log_K, const fr_t *E, fr_t *G) {
    memset(G, 0, ((size_t)1 << log_K) * sizeof(fr_t));
    for (size_t j = 0; j < T; j++) fr_add(&G[idx[j]], &E[j], &G[idx[j]]);
}

void orc_shout_ra_evals(const uint64_t *idx, size_t T, size_t log_K, size_t log_k_chunk, const fr_t *E, fr_t *G) {
    const size_t d = (log_K + log_k_chunk - 1) / log_k_chunk, kc = (size_t)1 << log_k_chunk;
    memset(G, 0, d * kc * sizeof(fr_t));
    for (size_t j = 0; j < T; j++)
        for (size_t i = 0; i < d; i++) {
            size_t k = (idx[j] >> (log_k_chunk * (d - 1 - i))) & (kc - 1);
            fr_add(&G[i * kc + k], &E[j], &G[i * kc + k]);
        }
}

/* the reference's three-polynomial round loop, literally */
int orc_sumcheck_readraf_prove(fr_t *G, const int32_t *table, size_t log_K, const fr_t *gamma, const fr_t *input_claim,
                               orc_transcript *t, fr_t *proof, u128 *challenges, fr_t *final_G) {
    size_t len = (size_t)1 << log_K;
    fr_t *val = (fr_t *)malloc(len * sizeof(fr_t));
    orc_i32_to_fr(table, len, val);
    size_t num_bound = 0; fr_t int_bound; fr_zero(&int_bound);     /* IdentityPolynomial state */
    orc_transcript_append_scalar(t, input_claim);
    fr_t prev = *input_claim;
    for (size_t rnd = 0; rnd < log_K; rnd++) {
        size_t half = len / 2;
        fr_t e0, e2, m; fr_zero(&e0); fr_zero(&e2);
        { uint64_t c[4] = {0, 0, 0, 0}; c[0] = (uint64_t)1 << (log_K - 1 - num_bound); fp_from_canonical(&ORC_FR, c, &m); }
        for (size_t i = 0; i < half; i++) {
            fr_t v0 = val[i], v2, g0 = G[i], g2, d, i0, i2, t0, t2, fi;
            fr_sub(&val[i + half], &v0, &d); fr_add(&val[i + half], &d, &v2);
            fr_sub(&G[i + half], &g0, &d); fr_add(&G[i + half], &d, &g2);
            /* int: evals[0] = bound*(m+m) + i ; eval(2) = evals[0] + 2m (identity_poly.rs:102-116) */
            fr_add(&m, &m, &d); fr_mul(&int_bound, &d, &i0); fr_from_u64(i, &fi); fr_add(&i0, &fi, &i0);
            fr_add(&i0, &d, &i2);
            fr_mul(gamma, &i0, &t0); fr_add(&v0, &t0, &t0); fr_mul(&g0, &t0, &t0); fr_add(&e0, &t0, &e0);
            fr_mul(gamma, &i2, &t2); fr_add(&v2, &t2, &t2); fr_mul(&g2, &t2, &t2); fr_add(&e2, &t2, &e2);
        }
        fr_t ev[2] = {e0, e2}, coeffs[4], cc[3], r;
        size_t nc = orc_unipoly_from_evals_and_hint(&prev, ev, 2, coeffs);
        size_t ncc = orc_unipoly_compress(coeffs, nc, cc);
        orc_transcript_append_compressed(t, cc, ncc);
        u128 raw; orc_transcript_challenge_optimized(t, &raw, &r);
        challenges[rnd] = raw;
        orc_unipoly_eval(coeffs, nc, &r, &prev);
        proof[rnd * 2] = cc[0]; proof[rnd * 2 + 1] = cc[1];
        orc_bind(val, len, &r, ORC_HIGH_TO_LOW); orc_bind(G, len, &r, ORC_HIGH_TO_LOW);
        fr_add(&int_bound, &int_bound, &int_bound); fr_add(&int_bound, &r, &int_bound);   /* identity_poly.rs:49-52 */
        num_bound++; len = half;
    }
    *final_G = G[0];
    free(val);
    return 0;
}

/* input claim of the read-raf sumcheck: sum_k G[k] (val[k] + gamma k) */
void orc_readraf_claim(const fr_t *G, const int32_t *table, size_t log_K, const fr_t *gamma, fr_t *out) {
    fr_t acc; fr_zero(&acc);
    for (size_t k = 0; k < ((size_t)1 << log_K); k++) {
        fr_t v, kk, t; fr_from_i64(table[k], &v); fr_from_u64(k, &kk); fr_mul(gamma, &kk, &t); fr_add(&v, &t, &t);
        fr_mul(&G[k], &t, &t); fr_add(&acc, &t, &acc);
    }
    *out = acc;
}

/* ---- the same prover as a SumcheckInstanceProver (a member of the softmax stage-3 BatchedSumcheck,
 *      ops/softmax_last_axis/mod.rs:611-671): state between compute_message and ingest_challenge */
void orc_shout_inst_init(orc_shout_inst *S, const fr_t *G, const int32_t *table, size_t log_K, const fr_t *gamma) {
    S->log_K = log_K; S->len = (size_t)1 << log_K; S->num_bound = 0; fr_zero(&S->int_bound); S->gamma = *gamma;
    S->G = (fr_t *)malloc(S->len * sizeof(fr_t)); memcpy(S->G, G, S->len * sizeof(fr_t));
    S->val = (fr_t *)malloc(S->len * sizeof(fr_t)); orc_i32_to_fr(table, S->len, S->val);
}
void orc_shout_inst_free(orc_shout_inst *S) { free(S->G); free(S->val); }

size_t orc_shout_inst_message(orc_shout_inst *S, const fr_t *claim, fr_t *coeffs) {       /* shout.rs:233-255 */
    const size_t half = S->len / 2;
    fr_t e0, e2, m; fr_zero(&e0); fr_zero(&e2);
    { uint64_t c[4] = {0, 0, 0, 0}; c[0] = (uint64_t)1 << (S->log_K - 1 - S->num_bound); fp_from_canonical(&ORC_FR, c, &m); }
    for (size_t i = 0; i < half; i++) {
        fr_t v0 = S->val[i], v2, g0 = S->G[i], g2, d, i0, i2, t0, t2, fi;
        fr_sub(&S->val[i + half], &v0, &d); fr_add(&S->val[i + half], &d, &v2);
        fr_sub(&S->G[i + half], &g0, &d); fr_add(&S->G[i + half], &d, &g2);
        fr_add(&m, &m, &d); fr_mul(&S->int_bound, &d, &i0); fr_from_u64(i, &fi); fr_add(&i0, &fi, &i0);   /* identity_poly.rs:102-116 */
        fr_add(&i0, &d, &i2);
        fr_mul(&S->gamma, &i0, &t0); fr_add(&v0, &t0, &t0); fr_mul(&g0, &t0, &t0); fr_add(&e0, &t0, &e0);
        fr_mul(&S->gamma, &i2, &t2); fr_add(&v2, &t2, &t2); fr_mul(&g2, &t2, &t2); fr_add(&e2, &t2, &e2);
    }
    fr_t ev[2] = {e0, e2};
    return orc_unipoly_from_evals_and_hint(claim, ev, 2, coeffs);
}

void orc_shout_inst_ingest(orc_shout_inst *S, const fr_t *r) {                             /* shout.rs:257-262 */
    orc_bind(S->val, S->len, r, ORC_HIGH_TO_LOW); orc_bind(S->G, S->len, r, ORC_HIGH_TO_LOW);
    fr_add(&S->int_bound, &S->int_bound, &S->int_bound); fr_add(&S->int_bound, r, &S->int_bound);   /* identity_poly.rs:49-52 */
    S->num_bound++; S->len /= 2;
}
void orc_shout_inst_final(const orc_shout_inst *S, fr_t *out) { *out = S->G[0]; }

/* the one-hot read-address polynomial of T lookups at (r_address | r_cycle):  sum_j eq(r_cycle, j) prod_i (bit_i(idx_j) ? r_i : 1 - r_i),
 * r_address[0] <-> the most significant of the log_K index bits (what a read-raf prover's cache_openings appends as its ra claim) */
void orc_ra_claim(const uint64_t *idx, size_t log_T, size_t log_K, const fr_t *r_address, const fr_t *r_cycle, fr_t *out) {
    const size_t T = (size_t)1 << log_T;
    fr_t *E = (fr_t *)malloc(T * sizeof(fr_t));
    orc_eq_evals(r_cycle, log_T, 0, E);
    fr_t one, *nr = (fr_t *)malloc((log_K ? log_K : 1) * sizeof(fr_t));
    fr_from_u64(1, &one);
    for (size_t i = 0; i < log_K; i++) fr_sub(&one, &r_address[i], &nr[i]);
    fr_t acc; fr_zero(&acc);
    for (size_t j = 0; j < T; j++) {
        fr_t w = E[j];
        for (size_t i = 0; i < log_K; i++) {
            const int bit = (int)((idx[j] >> (log_K - 1 - i)) & 1);
            fr_mul(&w, bit ? &r_address[i] : &nr[i], &w);
        }
        fr_add(&acc, &w, &acc);
    }
    *out = acc;
    free(E); free(nr);
}
