/* TEST INFRASTRUCTURE ONLY (oracle): state of the opening-reduction provers (opening.c). */
#ifndef ORC_OPENING_H
#define ORC_OPENING_H
#include "oracle.h"

typedef struct {
    fr_t *w; size_t n, k_in, k_out, in_top, out_top, current_index;
    fr_t **Ein, **Eout; fr_t scalar;
} gseh_t;                                   /* GruenSplitEqPolynomial, HighToLow */
void gseh_init(gseh_t *S, const fr_t *w, size_t n);
void gseh_free(gseh_t *S);
void gseh_bind(gseh_t *S, const fr_t *r);

enum { ORC_INST_DENSE_OPENING = 5, ORC_INST_ONEHOT_OPENING = 6 };

typedef struct { size_t n, len; fr_t *P; gseh_t D; } orc_dense_opening;
void   orc_dense_opening_init(orc_dense_opening *S, const fr_t *poly, size_t n, const fr_t *point);
void   orc_dense_opening_free(orc_dense_opening *S);
size_t orc_dense_opening_message(orc_dense_opening *S, const fr_t *claim, fr_t *coeffs);
void   orc_dense_opening_ingest(orc_dense_opening *S, const fr_t *r);

typedef struct {
    size_t log_K, log_T, B_len, F_len, H_len; const int32_t *idx;
    fr_t *B, *F, *G, *H; gseh_t D;
} orc_onehot_opening;
void   orc_onehot_opening_init(orc_onehot_opening *S, const int32_t *idx, size_t log_K, size_t log_T,
                               const fr_t *r_address, const fr_t *r_cycle);
void   orc_onehot_opening_free(orc_onehot_opening *S);
size_t orc_onehot_opening_message(orc_onehot_opening *S, size_t round, const fr_t *claim, fr_t *coeffs);
void   orc_onehot_opening_ingest(orc_onehot_opening *S, size_t round, const fr_t *r);
#endif
