/* TEST INFRASTRUCTURE ONLY (oracle): softmax_last_axis two-phase provers and SumAxis (softmax.c). */
#ifndef ORC_SOFTMAX_H
#define ORC_SOFTMAX_H
#include "oracle.h"
#include "ra.h"

enum { ORC_INST_SOFTMAX = 12 };
enum { ORC_SM_EXP_SUM = 0, ORC_SM_MAX_INDICATOR = 1, ORC_SM_RECIP_MULT = 2, ORC_SM_SUM_AXIS = 3 };

typedef struct {
    int kind; size_t log_K, log_N, len, len2, round;
    fr_t *a, *b;               /* ExpSum: a = exp_q; MaxIndicator: a = X, b = e; RecipMult: a = exp_q, b = inv_sum (2^log_K) */
    fr_t *eq_k;                /* EqPolynomial::evals(r_k), 2^log_K (ExpSum, MaxIndicator) */
    fr_t *r_k;
    gse_t gs; int gs_live;
} orc_softmax;
/* a, b: 2^(log_K + log_N) Fr laid out [k][j] (RecipMult's b: 2^log_K); r: log_K Fr (ExpSum r0_k, MaxIndicator r1_k) or
 * log_K + log_N Fr (RecipMult's opening point); SumAxis: a only, HighToLow, log_K = 0 */
void   orc_softmax_init(orc_softmax *S, int kind, const fr_t *a, const fr_t *b, size_t log_K, size_t log_N, const fr_t *r);
void   orc_softmax_free(orc_softmax *S);
size_t orc_softmax_message(orc_softmax *S, size_t round, const fr_t *claim, fr_t *coeffs);
void   orc_softmax_ingest(orc_softmax *S, size_t round, const fr_t *r);
void   orc_softmax_finals(const orc_softmax *S, fr_t *out);     /* a(r) [, b(r)] */
#endif
