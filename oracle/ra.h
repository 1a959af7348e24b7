/* TEST INFRASTRUCTURE ONLY (oracle): state of the one-hot "ra" sumcheck instances (ra.c). */
#ifndef ORC_RA_H
#define ORC_RA_H
#include "oracle.h"

typedef struct {
    fr_t *w; size_t n, out_top, in_top, k_out, k_in, current_index;
    fr_t **Eout, **Ein; fr_t scalar;
} gse_t;                                   /* GruenSplitEqPolynomial, LowToHigh */
void gse_init(gse_t *S, const fr_t *w, size_t n);
void gse_free(gse_t *S);
void gse_bind(gse_t *S, const fr_t *r);
void gse_gruen_deg3(const gse_t *S, const fr_t *q0, const fr_t *qinf, const fr_t *claim, fr_t coeffs[4]);
void orc_gauss_solve(fr_t *m, size_t n, fr_t *c);   /* rows of n+1 Fr: n x n system | rhs */
void orc_unipoly_from_evals_toom(const fr_t *evals, size_t n, fr_t *coeffs);
/* finish_mles_product_sum_from_evals (mles_product_sum.rs:330-376); returns the coefficient count */
size_t orc_finish_product_sum(const fr_t *sum_evals, size_t d, const fr_t *claim, const gse_t *eq, fr_t *coeffs);

enum { ORC_INST_RA_VIRTUAL = 2, ORC_INST_BOOLEANITY = 3, ORC_INST_HAMMING = 4 };

typedef struct { size_t d, log_T, len; fr_t **ra; gse_t eq; } orc_ra_virtual;
/* r_address_chunks: d * log_k Fr (OneHotParams::compute_r_address_chunks output, row i = chunk i);
 * r_cycle: log_T Fr, BIG_ENDIAN */
void   orc_ra_virtual_init(orc_ra_virtual *S, const int32_t *const *H_indices, size_t d, size_t log_k, size_t log_T,
                           const fr_t *r_address_chunks, const fr_t *r_cycle);
void   orc_ra_virtual_free(orc_ra_virtual *S);
size_t orc_ra_virtual_message(orc_ra_virtual *S, const fr_t *claim, fr_t *coeffs);   /* up to d + 2 coefficients */
void   orc_ra_virtual_ingest(orc_ra_virtual *S, const fr_t *r);

typedef struct {
    size_t d, log_k, log_T, len, F_len;
    fr_t *G, *gammas, *F; const int32_t *const *H_indices; fr_t **H;
    gse_t B, D; fr_t eq_r_r;
} orc_booleanity;
/* G: d * 2^log_k Fr; gammas: d Fr (field values of the F::Challenge batching scalars) */
void   orc_booleanity_init(orc_booleanity *S, const fr_t *G, const int32_t *const *H_indices, size_t d, size_t log_k,
                           size_t log_T, const fr_t *gammas, const fr_t *r_address, const fr_t *r_cycle);
void   orc_booleanity_free(orc_booleanity *S);
size_t orc_booleanity_message(orc_booleanity *S, size_t round, const fr_t *claim, fr_t *coeffs);
void   orc_booleanity_ingest(orc_booleanity *S, size_t round, const fr_t *r);

typedef struct { size_t d, log_k, len, stride; fr_t *ra, *gamma_powers; } orc_hamming;
void   orc_hamming_init(orc_hamming *S, const fr_t *G, size_t d, size_t log_k, const fr_t *gamma_powers);
void   orc_hamming_free(orc_hamming *S);
size_t orc_hamming_message(orc_hamming *S, const fr_t *claim, fr_t *coeffs);
void   orc_hamming_ingest(orc_hamming *S, const fr_t *r);

size_t orc_ra_inst_message(int kind, void *st, size_t round, const fr_t *claim, fr_t *c);
void   orc_ra_inst_ingest(int kind, void *st, size_t round, const fr_t *r);
int    orc_ra_inst_prove(int kind, void *st, size_t n_rounds, const fr_t *input_claim, orc_transcript *t, fr_t *compressed,
                         size_t stride, uint32_t *n_coeffs, u128 *challenges);
#endif
