/* TEST INFRASTRUCTURE ONLY (oracle): dense Shout ReadRafProver as a batch member (shout.c). */
#ifndef ORC_SHOUT_H
#define ORC_SHOUT_H
#include "oracle.h"

enum { ORC_INST_SHOUT = 13 };
/* ReadRafProver (joltworks/src/subprotocols/shout.rs:193-277): G, val and the IdentityPolynomial bound HighToLow */
typedef struct { size_t log_K, len, num_bound; fr_t *G, *val; fr_t int_bound, gamma; } orc_shout_inst;
/* G: 2^log_K Fr (orc_shout_G), table: 2^log_K i32 */
void   orc_shout_inst_init(orc_shout_inst *S, const fr_t *G, const int32_t *table, size_t log_K, const fr_t *gamma);
void   orc_shout_inst_free(orc_shout_inst *S);
size_t orc_shout_inst_message(orc_shout_inst *S, const fr_t *claim, fr_t *coeffs);
void   orc_shout_inst_ingest(orc_shout_inst *S, const fr_t *r);
void   orc_shout_inst_final(const orc_shout_inst *S, fr_t *out);    /* G.final_claim() */
#endif
