/* TEST INFRASTRUCTURE ONLY (oracle): element-wise operator sumchecks (elementwise.c). */
#ifndef ORC_ELEMENTWISE_H
#define ORC_ELEMENTWISE_H
#include "oracle.h"
#include "ra.h"

enum { ORC_INST_ELEMENTWISE = 11 };
enum { ORC_EW_ADD = 0, ORC_EW_SUB = 1, ORC_EW_NEG = 2, ORC_EW_SQUARE = 3, ORC_EW_IFF = 4, ORC_EW_MUL = 5, ORC_EW_CUBE = 6, ORC_EW_DIV = 7, ORC_EW_RSQRT = 8,
       ORC_EW_DOT = 9, ORC_EW_GATHER = 10, ORC_EW_HAMMING_BOOL = 11, ORC_EW_TELEPORT_DIV = 12 };   /* no eq factor: sum of products, UniPoly::from_evals_and_hint */

typedef struct {
    int op; size_t n_ops, n_vars, len;
    fr_t *x[16];                /* operands in cache_openings order (Iff: mask, a, b; Div: left, right, q, R;
                                 * Rsqrt: input, quotient, output, div_remainder, sqrt_remainder) */
    fr_t k[16];                 /* Rsqrt: S^3, gamma; Gather: gamma; HammingBooleanity: gamma_powers */
    gse_t eq;
} orc_elementwise;
/* operands: n_ops arrays of 2^n_vars Fr; r_node_output: n_vars Fr (big-endian) */
void   orc_elementwise_init(orc_elementwise *S, int op, const fr_t *const *operands, size_t n_ops, size_t n_vars, const fr_t *r_node_output,
                            const fr_t *constants, size_t n_constants);
void   orc_elementwise_free(orc_elementwise *S);
size_t orc_elementwise_message(orc_elementwise *S, const fr_t *claim, fr_t *coeffs);
void   orc_elementwise_ingest(orc_elementwise *S, const fr_t *r);
void   orc_elementwise_finals(const orc_elementwise *S, fr_t *out);   /* n_ops Fr */
#endif
