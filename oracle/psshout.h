/* TEST INFRASTRUCTURE ONLY (oracle): state of the prefix-suffix Shout read-raf prover (psshout.c). */
#ifndef ORC_PSSHOUT_H
#define ORC_PSSHOUT_H
#include "oracle.h"
#include "ra.h"

enum { ORC_INST_PS_RELU = 7 };

typedef struct {
    size_t N, log_T, log_m, m, T, Q_len, RP_len, n_r, ra_len;
    const uint64_t *idx;
    fr_t gamma, *u, *Q[2], *RQ[2], *RP, *v[8], *ra;
    size_t v_len[8];
    fr_t cp_notmsb, cp_word, sid_cp, val, raf_val;
    int has_notmsb, has_word, has_sid;
    fr_t r[160];
    gse_t eq;
    int table;            /* 0 = ReluTable<N>, 1 = RightShiftTable<N> by `shift` bits (lookup_tables/right_shift.rs) */
    size_t shift;
    fr_t cp_rs; int has_rs;
} orc_ps_relu;
/* lookup indices: T = 2^log_T N-bit values; N = 16, 32 or 64; r_node: log_T Fr (big-endian) */
void   orc_ps_relu_init(orc_ps_relu *S, const uint64_t *idx, size_t N, size_t log_T, const fr_t *r_node, const fr_t *gamma);
/* same prover over RightShiftTable<N>: Val(k) = k >> shift (unsigned), RAF = SignedIdentity */
void   orc_ps_rshift_init(orc_ps_relu *S, const uint64_t *idx, size_t N, size_t shift, size_t log_T, const fr_t *r_node, const fr_t *gamma);
void   orc_ps_relu_free(orc_ps_relu *S);
size_t orc_ps_relu_message(orc_ps_relu *S, size_t round, const fr_t *claim, fr_t *coeffs);
void   orc_ps_relu_ingest(orc_ps_relu *S, size_t round, const fr_t *r);

enum { ORC_INST_PS_IDENTITY = 8 };
typedef struct {
    size_t log_K, phases, log_T, log_m, m, T, Q_len, P_len, ra_len;
    const uint64_t *idx;
    fr_t *u, *Q[2], *P, *v[64], *ra;
    size_t v_len[64];
    fr_t cp, raf_val; int has_cp;
    gse_t eq;
} orc_ps_identity;
/* IdentityRCProver (subprotocols/identity_range_check.rs:196-420): sum eq(r_node,t) ra(k,t) Id(k) */
void   orc_ps_identity_init(orc_ps_identity *S, const uint64_t *idx, size_t log_K, size_t phases, size_t log_T, const fr_t *r_node);
void   orc_ps_identity_free(orc_ps_identity *S);
size_t orc_ps_identity_message(orc_ps_identity *S, size_t round, const fr_t *claim, fr_t *coeffs);
void   orc_ps_identity_ingest(orc_ps_identity *S, size_t round, const fr_t *r);

enum { ORC_INST_PS_CLAMP = 9 };
typedef struct {
    size_t N, bound, log_T, log_m, m, T, Q_len, n_r, ra_len;
    int symmetric;
    const uint64_t *idx;
    fr_t gamma, *u, *Q[6], *v[8], *ra, wv;
    size_t v_len[8];
    fr_t r[160];
    gse_t eq;
} orc_ps_clamp;
/* ClampBoundedTable<N, bound, symmetric> read-raf (lookup_tables/clamp.rs), unary, with SignedIdentity RAF */
void   orc_ps_clamp_init(orc_ps_clamp *S, const uint64_t *idx, size_t N, size_t bound, int symmetric, size_t log_T,
                         const fr_t *r_node, const fr_t *gamma);
void   orc_ps_clamp_free(orc_ps_clamp *S);
size_t orc_ps_clamp_message(orc_ps_clamp *S, size_t round, const fr_t *claim, fr_t *coeffs);
void   orc_ps_clamp_ingest(orc_ps_clamp *S, size_t round, const fr_t *r);

enum { ORC_INST_PS_ULT = 10 };
typedef struct {
    size_t log_T, T, Q_len, n_r, ra_len;
    const uint64_t *idx;
    fr_t gamma, *u, *Q[4], *v[8], *ra, wv;
    size_t v_len[8];
    fr_t r[160];
    gse_t eq;
} orc_ps_ult;
/* binary read-raf with UnsignedLessThanTable<32>: lookup index = interleave(x, y), 64 address rounds */
void   orc_ps_ult_init(orc_ps_ult *S, const uint64_t *idx, size_t log_T, const fr_t *r_node, const fr_t *gamma);
void   orc_ps_ult_free(orc_ps_ult *S);
size_t orc_ps_ult_message(orc_ps_ult *S, size_t round, const fr_t *claim, fr_t *coeffs);
void   orc_ps_ult_ingest(orc_ps_ult *S, size_t round, const fr_t *r);
#endif
