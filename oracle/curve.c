/* ORACLE — TEST INFRASTRUCTURE ONLY (see bn254.h header note; parity unpinned).
 * BN254 G1 (y^2 = x^3 + 3 over Fq) and multi-scalar multiplication.
 *
 * The reference calls arkworks for all of this (EXTERNAL, un-vendored):
 *   VariableBaseMSM::msm            joltworks/src/msm/mod.rs:27-181  -> ark_ec Pippenger
 *   batch_g1_additions_multi        joltworks/src/poly/commitment/hyperkzg/mod.rs:551,593
 * A group element has a unique affine representative, so any correct algorithm yields the
 * same (x, y) bytes; this file restates the published algorithms (Jacobian formulas
 * add-2007-bl / dbl-2009-l, bucket-method MSM with arkworks' window heuristic). */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define Q (&ORC_FQ)

static void fq_set_u64(uint64_t v, fq_t *o) { uint64_t c[4] = {v, 0, 0, 0}; fp_from_canonical(Q, c, o); }

void g1_jac_set_inf(g1_jac_t *p) { memset(p, 0, sizeof *p); memcpy(p->x.l, Q->r, 32); memcpy(p->y.l, Q->r, 32); }
int  g1_jac_is_inf(const g1_jac_t *p) { return fp_is_zero(&p->z); }

void g1_generator(g1_aff_t *g) { fq_set_u64(1, &g->x); fq_set_u64(2, &g->y); g->inf = 0; }

int g1_aff_on_curve(const g1_aff_t *p) {
    if (p->inf) return 1;
    fq_t y2, x3, b;
    fp_sqr(Q, &p->y, &y2); fp_sqr(Q, &p->x, &x3); fp_mul(Q, &x3, &p->x, &x3);
    fq_set_u64(3, &b); fp_add(Q, &x3, &b, &x3);
    return fp_eq(&y2, &x3);
}

void g1_jac_from_aff(const g1_aff_t *a, g1_jac_t *o) {
    if (a->inf) { g1_jac_set_inf(o); return; }
    o->x = a->x; o->y = a->y; memcpy(o->z.l, Q->r, 32);
}

void g1_jac_dbl(const g1_jac_t *p, g1_jac_t *o) {       /* dbl-2009-l (a = 0) */
    if (g1_jac_is_inf(p)) { *o = *p; return; }
    fq_t A, B, C, D, E, F, t, X3, Y3, Z3;
    fp_sqr(Q, &p->x, &A); fp_sqr(Q, &p->y, &B); fp_sqr(Q, &B, &C);
    fp_add(Q, &p->x, &B, &t); fp_sqr(Q, &t, &t); fp_sub(Q, &t, &A, &t); fp_sub(Q, &t, &C, &t);
    fp_add(Q, &t, &t, &D);
    fp_add(Q, &A, &A, &E); fp_add(Q, &E, &A, &E);
    fp_sqr(Q, &E, &F);
    fp_sub(Q, &F, &D, &X3); fp_sub(Q, &X3, &D, &X3);
    fp_sub(Q, &D, &X3, &t); fp_mul(Q, &E, &t, &Y3);
    fp_add(Q, &C, &C, &t); fp_add(Q, &t, &t, &t); fp_add(Q, &t, &t, &t);
    fp_sub(Q, &Y3, &t, &Y3);
    fp_mul(Q, &p->y, &p->z, &Z3); fp_add(Q, &Z3, &Z3, &Z3);
    o->x = X3; o->y = Y3; o->z = Z3;
}

void g1_jac_add(const g1_jac_t *p, const g1_jac_t *q, g1_jac_t *o) {   /* add-2007-bl */
    if (g1_jac_is_inf(p)) { *o = *q; return; }
    if (g1_jac_is_inf(q)) { *o = *p; return; }
    fq_t Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, r, V, t, X3, Y3, Z3;
    fp_sqr(Q, &p->z, &Z1Z1); fp_sqr(Q, &q->z, &Z2Z2);
    fp_mul(Q, &p->x, &Z2Z2, &U1); fp_mul(Q, &q->x, &Z1Z1, &U2);
    fp_mul(Q, &p->y, &q->z, &S1); fp_mul(Q, &S1, &Z2Z2, &S1);
    fp_mul(Q, &q->y, &p->z, &S2); fp_mul(Q, &S2, &Z1Z1, &S2);
    if (fp_eq(&U1, &U2)) {
        if (fp_eq(&S1, &S2)) { g1_jac_dbl(p, o); return; }
        g1_jac_set_inf(o); return;
    }
    fp_sub(Q, &U2, &U1, &H);
    fp_add(Q, &H, &H, &I); fp_sqr(Q, &I, &I);
    fp_mul(Q, &H, &I, &J);
    fp_sub(Q, &S2, &S1, &r); fp_add(Q, &r, &r, &r);
    fp_mul(Q, &U1, &I, &V);
    fp_sqr(Q, &r, &X3); fp_sub(Q, &X3, &J, &X3); fp_sub(Q, &X3, &V, &X3); fp_sub(Q, &X3, &V, &X3);
    fp_sub(Q, &V, &X3, &t); fp_mul(Q, &r, &t, &Y3);
    fp_mul(Q, &S1, &J, &t); fp_add(Q, &t, &t, &t); fp_sub(Q, &Y3, &t, &Y3);
    fp_add(Q, &p->z, &q->z, &Z3); fp_sqr(Q, &Z3, &Z3); fp_sub(Q, &Z3, &Z1Z1, &Z3); fp_sub(Q, &Z3, &Z2Z2, &Z3);
    fp_mul(Q, &Z3, &H, &Z3);
    o->x = X3; o->y = Y3; o->z = Z3;
}

void g1_jac_add_aff(const g1_jac_t *p, const g1_aff_t *q, g1_jac_t *o) {
    g1_jac_t qq; g1_jac_from_aff(q, &qq);
    g1_jac_add(p, &qq, o);
}

void g1_jac_neg(const g1_jac_t *p, g1_jac_t *o) { *o = *p; fp_neg(Q, &p->y, &o->y); }

void g1_jac_to_aff(const g1_jac_t *p, g1_aff_t *o) {
    if (g1_jac_is_inf(p)) { memset(o, 0, sizeof *o); o->inf = 1; return; }
    fq_t zi, zi2, zi3;
    fp_inv(Q, &p->z, &zi); fp_sqr(Q, &zi, &zi2); fp_mul(Q, &zi2, &zi, &zi3);
    fp_mul(Q, &p->x, &zi2, &o->x); fp_mul(Q, &p->y, &zi3, &o->y); o->inf = 0;
}

/* scalar given as canonical 4x64 integer */
void g1_mul_canonical(const g1_aff_t *p, const uint64_t k[4], g1_jac_t *o) {
    g1_jac_t acc; g1_jac_set_inf(&acc);
    for (int i = 255; i >= 0; i--) {
        g1_jac_dbl(&acc, &acc);
        if ((k[i >> 6] >> (i & 63)) & 1) g1_jac_add_aff(&acc, p, &acc);
    }
    *o = acc;
}

void g1_mul_fr(const g1_aff_t *p, const fr_t *s, g1_aff_t *o) {
    uint64_t k[4]; fp_to_canonical(&ORC_FR, s, k);
    g1_jac_t r; g1_mul_canonical(p, k, &r); g1_jac_to_aff(&r, o);
}

/* naive MSM: sum_i s_i * P_i by double-and-add (small sizes; the ground truth) */
void orc_msm_naive(const g1_aff_t *bases, const fr_t *scalars, size_t n, g1_aff_t *out) {
    g1_jac_t acc; g1_jac_set_inf(&acc);
    for (size_t i = 0; i < n; i++) {
        uint64_t k[4]; fp_to_canonical(&ORC_FR, &scalars[i], k);
        g1_jac_t t; g1_mul_canonical(&bases[i], k, &t);
        g1_jac_add(&acc, &t, &acc);
    }
    g1_jac_to_aff(&acc, out);
}

/* bucket-method MSM (what arkworks' VariableBaseMSM::msm does): unsigned c-bit windows,
 * c = ln(n) + 2 for n >= 32 else 3 (ark-ec variable_base/mod.rs heuristic), windows in
 * parallel (Rayon there, OpenMP here), running-sum bucket reduction, Horner over windows. */
void orc_msm_pippenger(const g1_aff_t *bases, const fr_t *scalars, size_t n, g1_aff_t *out) {
    if (n == 0) { memset(out, 0, sizeof *out); out->inf = 1; return; }
    unsigned c = n < 32 ? 3 : (unsigned)(log((double)n) * 69.0 / 100.0) + 2;   /* ln_without_floats */
    const unsigned nbits = 254;
    const unsigned nwin = (nbits + c - 1) / c;
    uint64_t *canon = (uint64_t *)malloc(n * 32);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) fp_to_canonical(&ORC_FR, &scalars[i], canon + 4 * i);
    g1_jac_t *wsum = (g1_jac_t *)malloc(nwin * sizeof(g1_jac_t));
#pragma omp parallel for schedule(dynamic, 1)
    for (unsigned w = 0; w < nwin; w++) {
        const size_t nb = ((size_t)1 << c) - 1;
        g1_jac_t *bk = (g1_jac_t *)malloc(nb * sizeof(g1_jac_t));
        for (size_t b = 0; b < nb; b++) g1_jac_set_inf(&bk[b]);
        const unsigned lo = w * c;
        for (size_t i = 0; i < n; i++) {
            const uint64_t *k = canon + 4 * i;
            uint64_t d = k[lo >> 6] >> (lo & 63);
            if ((lo & 63) + c > 64 && (lo >> 6) < 3) d |= k[(lo >> 6) + 1] << (64 - (lo & 63));
            d &= ((uint64_t)1 << c) - 1;
            if (d) g1_jac_add_aff(&bk[d - 1], &bases[i], &bk[d - 1]);
        }
        g1_jac_t run, res; g1_jac_set_inf(&run); g1_jac_set_inf(&res);
        for (size_t b = nb; b-- > 0;) { g1_jac_add(&run, &bk[b], &run); g1_jac_add(&res, &run, &res); }
        wsum[w] = res;
        free(bk);
    }
    g1_jac_t acc = wsum[nwin - 1];
    for (unsigned w = nwin - 1; w-- > 0;) {
        for (unsigned k = 0; k < c; k++) g1_jac_dbl(&acc, &acc);
        g1_jac_add(&acc, &wsum[w], &acc);
    }
    g1_jac_to_aff(&acc, out);
    free(wsum); free(canon);
}

/* one-hot commit: sum of bases[idx[i]] (pure additions) — HyperKZG::commit_one_hot,
 * hyperkzg/mod.rs:520-554 */
void orc_g1_sum_indexed(const g1_aff_t *bases, const uint64_t *idx, size_t n, g1_aff_t *out) {
    g1_jac_t acc; g1_jac_set_inf(&acc);
    for (size_t i = 0; i < n; i++) g1_jac_add_aff(&acc, &bases[idx[i]], &acc);
    g1_jac_to_aff(&acc, out);
}

/* SRS of this build: bases[i] = tau^(i+1) * G, G = (1, 2)  — the reference's
 * layout (powers start at beta^1, kzg.rs:47-53) with tau supplied by the caller.
 * Computed incrementally by scalar multiplication of the previous point (exact). */
void orc_srs_powers(const fr_t *tau, size_t n, g1_aff_t *out) {
    g1_aff_t g; g1_generator(&g);
    fr_t pw = *tau;
    /* independent scalar muls: out[i] = tau^(i+1) * G */
    fr_t *pws = (fr_t *)malloc(n * sizeof(fr_t));
    for (size_t i = 0; i < n; i++) { pws[i] = pw; fr_mul(&pw, tau, &pw); }
#pragma omp parallel for schedule(dynamic, 64)
    for (size_t i = 0; i < n; i++) g1_mul_fr(&g, &pws[i], &out[i]);
    free(pws);
}

/* UniPoly::eval_as_univariate for LargeScalars (joltworks/src/poly/unipoly.rs:247-259):
 * sum_i f[i] * r^i, serial power accumulation as in the reference. */
void orc_eval_as_univariate(const fr_t *f, size_t n, const fr_t *r, fr_t *out) {
    fr_t ev = f[0], pw = *r;
    for (size_t i = 1; i < n; i++) {
        fr_t t; fr_mul(&pw, &f[i], &t); fr_add(&ev, &t, &ev);
        fr_mul(&pw, r, &pw);
    }
    *out = ev;
}

/* Transcript::append_point / append_points (joltworks/src/transcripts/blake2b.rs:166-195):
 * identity -> 64 zero bytes; else one hash over x_be32 || y_be32 (affine, canonical). */
void orc_transcript_append_point(orc_transcript *t, const g1_aff_t *p) {
    uint8_t buf[64];
    if (p->inf) { memset(buf, 0, 64); orc_transcript_append_bytes(t, buf, 64); return; }
    uint64_t c[4];
    fp_to_canonical(Q, &p->x, c);
    for (int i = 0; i < 32; i++) buf[31 - i] = (uint8_t)(c[i >> 3] >> (8 * (i & 7)));
    fp_to_canonical(Q, &p->y, c);
    for (int i = 0; i < 32; i++) buf[63 - i] = (uint8_t)(c[i >> 3] >> (8 * (i & 7)));
    orc_transcript_append_bytes(t, buf, 64);
}

void orc_transcript_append_points(orc_transcript *t, const g1_aff_t *p, size_t n) {
    orc_transcript_append_message(t, "begin_append_vector");
    for (size_t i = 0; i < n; i++) orc_transcript_append_point(t, &p[i]);
    orc_transcript_append_message(t, "end_append_vector");
}

/* ---------------------------------------------------------------- HyperKZG
 * HyperKZG::open (joltworks/src/poly/commitment/hyperkzg/mod.rs:400-447), kzg_open_batch
 * (:231-280), compute_witness_polynomial (:213-229), DensePolynomial::linear_combination
 * (dense_mlpoly.rs:444-499).  srs[i] = g1_powers[i].  point = the ell MontU128Challenge
 * values as Fr.  Outputs: com[ell-1], w[3], v[3*ell] (v[i*ell + j] = f_j(u_i)). */
int orc_hyperkzg_open(const g1_aff_t *srs, const fr_t *poly, size_t ell, const fr_t *point,
                      orc_transcript *t, g1_aff_t *com, g1_aff_t *w, fr_t *v) {
    const size_t n = (size_t)1 << ell;
    /* Phase 1: polys[i+1][j] = point[ell-i-1]*(p[2j+1]-p[2j]) + p[2j]   (:413-428) */
    fr_t **polys = (fr_t **)malloc(ell * sizeof(fr_t *));
    polys[0] = (fr_t *)malloc(n * sizeof(fr_t)); memcpy(polys[0], poly, n * sizeof(fr_t));
    for (size_t i = 0; i + 1 < ell; i++) {
        size_t len = n >> (i + 1);
        polys[i + 1] = (fr_t *)malloc(len * sizeof(fr_t));
        for (size_t j = 0; j < len; j++) {
            fr_t d; fr_sub(&polys[i][2 * j + 1], &polys[i][2 * j], &d);
            fr_mul(&point[ell - i - 1], &d, &d);
            fr_add(&d, &polys[i][2 * j], &polys[i + 1][j]);
        }
    }
    for (size_t i = 1; i < ell; i++) orc_msm_pippenger(srs, polys[i], n >> i, &com[i - 1]);   /* :434 */
    /* Phase 2 (:439-441) */
    orc_transcript_append_points(t, com, ell - 1);
    fr_t r, u[3]; orc_transcript_challenge_scalar(t, &r);
    u[0] = r; fr_neg(&r, &u[1]); fr_mul(&r, &r, &u[2]);
    /* Phase 3: kzg_open_batch */
    for (size_t i = 0; i < 3; i++)
        for (size_t j = 0; j < ell; j++) orc_eval_as_univariate(polys[j], n >> j, &u[i], &v[i * ell + j]);
    orc_transcript_append_scalars(t, v, 3 * ell);
    fr_t *q = (fr_t *)malloc(ell * sizeof(fr_t));            /* challenge_scalar_powers (blake2b.rs:224-231) */
    { fr_t q1; orc_transcript_challenge_scalar(t, &q1); fr_one(&q[0]); for (size_t j = 1; j < ell; j++) fr_mul(&q[j - 1], &q1, &q[j]); }
    fr_t *B = (fr_t *)calloc(n, sizeof(fr_t));
    for (size_t j = 0; j < ell; j++)
        for (size_t i = 0; i < (n >> j); i++) { fr_t m; fr_mul(&polys[j][i], &q[j], &m); fr_add(&B[i], &m, &B[i]); }
    fr_t *h = (fr_t *)malloc(n * sizeof(fr_t));
    for (size_t k = 0; k < 3; k++) {
        fr_zero(&h[n - 1]);
        for (size_t i = n - 1; i >= 1; i--) { fr_t m; fr_mul(&h[i], &u[k], &m); fr_add(&B[i], &m, &h[i - 1]); }
        orc_msm_pippenger(srs, h, n, &w[k]);
    }
    orc_transcript_append_points(t, w, 3);
    fr_t d0; orc_transcript_challenge_scalar(t, &d0);
    for (size_t i = 0; i < ell; i++) free(polys[i]);
    free(polys); free(q); free(B); free(h);
    return 0;
}

/* verify_inner (:451-509) + kzg_verify_batch (:283-366) with the pairing check
 * e(L, H) == e(R, tau*H) replaced by L == tau*R (known-trapdoor SRS).  Returns 1 = accept. */
int orc_hyperkzg_verify_trapdoor(const g1_aff_t *srs, const fr_t *tau, const g1_aff_t *C, size_t ell,
                                 const fr_t *point, const fr_t *y, const g1_aff_t *com, const g1_aff_t *w,
                                 const fr_t *v, orc_transcript *t) {
    orc_transcript_append_points(t, com, ell - 1);
    fr_t r; orc_transcript_challenge_scalar(t, &r);
    if (fp_is_zero(&r) || C->inf) return 0;
    fr_t u[3]; u[0] = r; fr_neg(&r, &u[1]); fr_mul(&r, &r, &u[2]);
    const fr_t *ypos = v, *yneg = v + ell, *Yv = v + 2 * ell;
    fr_t one, two; fr_one(&one); fr_from_u64(2, &two);
    for (size_t i = 0; i < ell; i++) {
        fr_t Ynext = (i + 1 < ell) ? Yv[i + 1] : *y;
        fr_t lhs, rhs, a, b, x = point[ell - i - 1];
        fr_mul(&two, &r, &lhs); fr_mul(&lhs, &Ynext, &lhs);
        fr_sub(&one, &x, &a); fr_mul(&r, &a, &a); fr_add(&ypos[i], &yneg[i], &b); fr_mul(&a, &b, &a);
        fr_sub(&ypos[i], &yneg[i], &b); fr_mul(&x, &b, &b); fr_add(&a, &b, &rhs);
        if (!fp_eq(&lhs, &rhs)) return 0;
    }
    orc_transcript_append_scalars(t, v, 3 * ell);
    fr_t *q = (fr_t *)malloc(ell * sizeof(fr_t));
    { fr_t q1; orc_transcript_challenge_scalar(t, &q1); fr_one(&q[0]); for (size_t j = 1; j < ell; j++) fr_mul(&q[j - 1], &q1, &q[j]); }
    orc_transcript_append_points(t, w, 3);
    fr_t d0, d1, mult; orc_transcript_challenge_scalar(t, &d0); fr_mul(&d0, &d0, &d1);
    fr_add(&one, &d0, &mult); fr_add(&mult, &d1, &mult);
    fr_t Bu[3];
    for (int k = 0; k < 3; k++) { fr_zero(&Bu[k]); for (size_t j = 0; j < ell; j++) { fr_t m; fr_mul(&v[k * ell + j], &q[j], &m); fr_add(&Bu[k], &m, &Bu[k]); } }
    size_t nb = ell + 4;
    g1_aff_t *bases = (g1_aff_t *)malloc(nb * sizeof(g1_aff_t)); fr_t *sc = (fr_t *)malloc(nb * sizeof(fr_t));
    bases[0] = *C; for (size_t j = 1; j < ell; j++) bases[j] = com[j - 1];
    for (size_t j = 0; j < ell; j++) fr_mul(&q[j], &mult, &sc[j]);
    bases[ell] = w[0]; bases[ell + 1] = w[1]; bases[ell + 2] = w[2]; bases[ell + 3] = srs[0];
    sc[ell] = u[0]; fr_mul(&u[1], &d0, &sc[ell + 1]); fr_mul(&u[2], &d1, &sc[ell + 2]);
    { fr_t a, b; fr_mul(&d0, &Bu[1], &a); fr_mul(&d1, &Bu[2], &b); fr_add(&Bu[0], &a, &a); fr_add(&a, &b, &a); fr_neg(&a, &sc[ell + 3]); }
    g1_aff_t L; orc_msm_naive(bases, sc, nb, &L);
    g1_aff_t Rb[3] = {w[0], w[1], w[2]}; fr_t Rs[3]; Rs[0] = one; Rs[1] = d0; Rs[2] = d1;
    g1_aff_t R, tR; orc_msm_naive(Rb, Rs, 3, &R);
    g1_mul_fr(&R, tau, &tR);
    free(q); free(bases); free(sc);
    if (R.inf) return L.inf ? 1 : 0;
    if (L.inf || tR.inf) return (L.inf && tR.inf) ? 1 : 0;
    return fp_eq(&L.x, &tR.x) && fp_eq(&L.y, &tR.y);
}

/* VariableBaseMSM::msm for the narrow MultilinearPolynomial variants (joltworks/src/msm/mod.rs:
 * 38-181): unsigned kinds go to msm_uN; I32/I64 split into positive and negative magnitudes and
 * return msm_u64(pos) - msm_u64(neg).  kind: 0 u8, 1 u16, 2 u32, 3 u64, 4 i32, 5 i64. */
void orc_msm_small(const g1_aff_t *bases, const void *scalars, size_t n, int kind, g1_aff_t *out) {
    g1_aff_t *pb = (g1_aff_t *)malloc((n + 1) * sizeof(g1_aff_t)), *nb = (g1_aff_t *)malloc((n + 1) * sizeof(g1_aff_t));
    fr_t *ps = (fr_t *)malloc((n + 1) * sizeof(fr_t)), *ns = (fr_t *)malloc((n + 1) * sizeof(fr_t));
    size_t np = 0, nn = 0;
    for (size_t i = 0; i < n; i++) {
        uint64_t mag = 0; int neg = 0;
        switch (kind) {
            case 0: mag = ((const uint8_t *)scalars)[i]; break;
            case 1: mag = ((const uint16_t *)scalars)[i]; break;
            case 2: mag = ((const uint32_t *)scalars)[i]; break;
            case 3: mag = ((const uint64_t *)scalars)[i]; break;
            case 4: { int32_t v = ((const int32_t *)scalars)[i]; neg = v < 0; mag = neg ? (uint64_t)(-(int64_t)v) : (uint64_t)v; break; }
            default: { int64_t v = ((const int64_t *)scalars)[i]; neg = v < 0; mag = neg ? (uint64_t)0 - (uint64_t)v : (uint64_t)v; break; }
        }
        if (!mag) continue;
        if (neg) { nb[nn] = bases[i]; fr_from_u64(mag, &ns[nn]); nn++; }
        else { pb[np] = bases[i]; fr_from_u64(mag, &ps[np]); np++; }
    }
    g1_aff_t P, N;
    orc_msm_pippenger(pb, ps, np, &P);
    orc_msm_pippenger(nb, ns, nn, &N);
    g1_jac_t jp, jn, r;
    g1_jac_from_aff(&P, &jp); g1_jac_from_aff(&N, &jn); g1_jac_neg(&jn, &jn);
    g1_jac_add(&jp, &jn, &r);
    g1_jac_to_aff(&r, out);
    free(pb); free(nb); free(ps); free(ns);
}
