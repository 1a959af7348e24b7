/* ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the jolt-atlas
 * ONNXProof::prove hot path (sumcheck rounds, EQ tables, lookup evaluation loops,
 * HyperKZG MSMs).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load liboracle.so.  PARITY UNPINNED against a reference run — see
 * bn254.h for what pins it instead. */
#ifndef ORACLE_H
#define ORACLE_H
#include "bn254.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- BLAKE2b-256 + Blake2bTranscript (joltworks/src/transcripts/blake2b.rs) ---- */
typedef struct {
    uint8_t  state[32];
    uint32_t n_rounds;
    uint8_t *history;      /* optional state-history recorder (blake2b.rs:17-25) */
    size_t   history_cap;  /* in 32-byte entries */
    size_t   history_len;
} orc_transcript;

void orc_blake2b256(const uint8_t *in, size_t len, uint8_t out[32]);
void orc_transcript_new(orc_transcript *t, const char *label);
void orc_transcript_record(orc_transcript *t, uint8_t *buf, size_t cap);
void orc_transcript_append_message(orc_transcript *t, const char *msg);
void orc_transcript_append_bytes(orc_transcript *t, const uint8_t *b, size_t n);
void orc_transcript_append_u64(orc_transcript *t, uint64_t x);
void orc_transcript_append_scalar(orc_transcript *t, const fr_t *a);
void orc_transcript_append_scalars(orc_transcript *t, const fr_t *a, size_t n);
void orc_transcript_challenge_bytes32(orc_transcript *t, uint8_t out[32]);
u128 orc_transcript_challenge_u128(orc_transcript *t);
void orc_transcript_challenge_scalar(orc_transcript *t, fr_t *o);
void orc_transcript_challenge_optimized(orc_transcript *t, u128 *raw, fr_t *o);

/* ---- polynomials (joltworks/src/poly) ---- */
enum { ORC_HIGH_TO_LOW = 0, ORC_LOW_TO_HIGH = 1 };  /* BindingOrder, multilinear_polynomial.rs:69-73 */

void orc_eq_evals(const fr_t *r, size_t n, const fr_t *scaling, fr_t *out); /* eq_poly.rs:92-101 */
void orc_bind(fr_t *z, size_t len, const fr_t *r, int order);               /* dense_mlpoly.rs:84-89 */
void orc_bind_i32(const int32_t *z, size_t len, const fr_t *r, int order, fr_t *out); /* compact_polynomial.rs:272-353 */
void orc_evaluate(const fr_t *z, size_t n_vars, const fr_t *r, fr_t *out);  /* dense_mlpoly.rs:265-305 */
void orc_i32_to_fr(const int32_t *z, size_t len, fr_t *out);

/* UniPoly (joltworks/src/poly/unipoly.rs). coeffs buffers hold >= n entries. */
size_t orc_unipoly_from_evals_and_hint(const fr_t *hint, const fr_t *evals, size_t n_evals, fr_t *coeffs);
void   orc_unipoly_eval(const fr_t *coeffs, size_t n, const fr_t *x, fr_t *out);
size_t orc_unipoly_compress(const fr_t *coeffs, size_t n, fr_t *out);
void   orc_compressed_eval_from_hint(const fr_t *cc, size_t n, const fr_t *hint, const fr_t *x, fr_t *out);
void   orc_transcript_append_compressed(orc_transcript *t, const fr_t *cc, size_t n);

/* ---- sumcheck (joltworks/src/subprotocols/sumcheck.rs:565-599) with the einsum
 * dot-product instance (jolt-atlas-core/src/onnx_proof/ops/einsum/dot.rs:290-375).
 * schedule: 0 none (deg 2), 1 High{a=log_eq,b=low_bits}, 2 Low{a=log_k,b=log_b}.
 * left/right/eq are consumed (bound in place).  proof_coeffs receives, per round,
 * `degree` compressed coefficients (c0, c2[, c3]); challenges the raw u128 draws.
 * final_claims = left(r), right(r), eq(r)|1.  Returns 0 on success. */
int orc_sumcheck_dot_prove(fr_t *left, fr_t *right, fr_t *eq, size_t n_vars, int schedule,
                           size_t sched_a, size_t sched_b, const fr_t *input_claim,
                           orc_transcript *t, fr_t *proof_coeffs, u128 *challenges,
                           fr_t *final_claims);
/* same instance but operands given as i32 (CompactPolynomial first round) */
int orc_sumcheck_dot_prove_i32(const int32_t *left, const int32_t *right, size_t n_vars,
                               const fr_t *input_claim, orc_transcript *t, fr_t *proof_coeffs,
                               u128 *challenges, fr_t *final_claims);
/* SumcheckInstanceProof::verify (sumcheck.rs:653-686): returns 0 ok, fills e and r. */
int orc_sumcheck_verify(const fr_t *proof_coeffs, size_t n_rounds, size_t degree,
                        const fr_t *claim, orc_transcript *t, fr_t *e_out, u128 *challenges);

/* MulProver over GruenSplitEqPolynomial (mul.rs:125-185; split_eq_poly.rs): proof rows of 3 */
int  orc_sumcheck_mul_prove(fr_t *left, fr_t *right, const fr_t *w, size_t n, const fr_t *input_claim,
                            orc_transcript *t, fr_t *proof, u128 *challenges, fr_t *final_claims);
void orc_mul_claim(const fr_t *l, const fr_t *r, const fr_t *w, size_t n, fr_t *out);
void orc_dot_claim(const fr_t *l, const fr_t *r, const fr_t *eq, size_t len, int schedule,
                   size_t sched_a, size_t sched_b, fr_t *out);
int  orc_num_threads(void);

/* BatchedSumcheck::prove (sumcheck.rs:30-184).  kind 0 = dot instance (schedule/sa/sb/eq as
 * above), 1 = Mul instance (w = n_vars Fr).  Operands are bound in place; final_claims as the
 * single-instance provers.  compressed: rows of `stride` Fr, n_coeffs[i] of them valid. */
typedef struct {
    int kind; int schedule; size_t n_vars, sa, sb;
    fr_t *left, *right, *eq; const fr_t *w;
    fr_t input_claim; fr_t final_claims[3];
    void *state;          /* kinds >= 2 (ra.h ORC_INST_*): initialised instance state, n_vars = rounds */
} orc_batched_inst;
int orc_batched_prove(orc_batched_inst *inst, size_t n_inst, orc_transcript *t, fr_t *compressed, size_t stride,
                      uint32_t *n_coeffs, u128 *challenges, size_t *max_rounds_out);
/* BatchedSumcheck::verify (sumcheck.rs:186-262) up to the expected-output check: returns the
 * final batched claim e and the coefficients; claim_i and rounds_i per instance. */
int orc_batched_verify(const fr_t *compressed, size_t stride, const uint32_t *n_coeffs, size_t max_rounds, const fr_t *input_claims,
                       const size_t *rounds, size_t n_inst, orc_transcript *t, fr_t *e_out, fr_t *coeffs_out,
                       u128 *challenges);

/* ---- BN254 G1 + MSM (arkworks, EXTERNAL to the reference tree; call sites
 *      joltworks/src/msm/mod.rs:27-181, hyperkzg/mod.rs:520-596, kzg.rs:195-298) ---- */
typedef struct { fq_t x, y; uint64_t inf; } g1_aff_t;   /* 72 bytes: the arkworks G1Affine image */
typedef struct { fq_t x, y, z; } g1_jac_t;
void g1_generator(g1_aff_t *g);
int  g1_aff_on_curve(const g1_aff_t *p);
void g1_jac_set_inf(g1_jac_t *p);
int  g1_jac_is_inf(const g1_jac_t *p);
void g1_jac_from_aff(const g1_aff_t *a, g1_jac_t *o);
void g1_jac_dbl(const g1_jac_t *p, g1_jac_t *o);
void g1_jac_add(const g1_jac_t *p, const g1_jac_t *q, g1_jac_t *o);
void g1_jac_add_aff(const g1_jac_t *p, const g1_aff_t *q, g1_jac_t *o);
void g1_jac_neg(const g1_jac_t *p, g1_jac_t *o);
void g1_jac_to_aff(const g1_jac_t *p, g1_aff_t *o);
void g1_mul_canonical(const g1_aff_t *p, const uint64_t k[4], g1_jac_t *o);
void g1_mul_fr(const g1_aff_t *p, const fr_t *s, g1_aff_t *o);
void orc_msm_naive(const g1_aff_t *bases, const fr_t *scalars, size_t n, g1_aff_t *out);
void orc_msm_pippenger(const g1_aff_t *bases, const fr_t *scalars, size_t n, g1_aff_t *out);
void orc_g1_sum_indexed(const g1_aff_t *bases, const uint64_t *idx, size_t n, g1_aff_t *out);
/* EvalReductionInstance::prove (subprotocols/evaluation_reduction.rs:91-147): points = N rows of n Fr,
 * h_out capacity n*(N-1)+1; returns h, the reduced point l(x') and claim h(x') */
int orc_eval_reduction_prove(const fr_t *mle, size_t n, const fr_t *points, const fr_t *claims, size_t N,
                             orc_transcript *t, fr_t *h_out, size_t *h_len, fr_t *r_out, fr_t *claim_out);
/* EinsumLayout::fold of the batched layouts (ops/einsum/bmk_rhs_mbn.rs, mbk_rhs_bmn.rs, k_nk_n.rs) */
void orc_einsum_fold_layout(int layout, const int32_t *left, const int32_t *right, size_t b, size_t m, size_t k, size_t n,
                            const fr_t *eq_r_m, const fr_t *eq_r_n, fr_t *left_out, fr_t *right_out);
void orc_einsum_fold_rbmk(int variant, const int32_t *left, const int32_t *right, size_t a, size_t c, size_t b, size_t m,
                          size_t n, size_t k, const fr_t *eq_r_m, const fr_t *eq_r_n, fr_t *left_out, fr_t *right_out);
/* build_materialized_rlc (poly/rlc_polynomial.rs:13-78) */
void orc_rlc_build(const fr_t *const *dense_fr, const int32_t *const *dense_i32, const size_t *dense_len,
                   const fr_t *dense_coeff, size_t n_dense, const int32_t *const *oh_k, const size_t *oh_T,
                   const fr_t *oh_coeff, size_t n_oh, fr_t *joint, size_t joint_len);
/* msm/mod.rs:38-181 narrow-scalar variants; kind 0 u8, 1 u16, 2 u32, 3 u64, 4 i32, 5 i64 */
void orc_msm_small(const g1_aff_t *bases, const void *scalars, size_t n, int kind, g1_aff_t *out);
void orc_srs_powers(const fr_t *tau, size_t n, g1_aff_t *out);
void orc_transcript_append_point(orc_transcript *t, const g1_aff_t *p);   /* blake2b.rs:166-187 */
void orc_transcript_append_points(orc_transcript *t, const g1_aff_t *p, size_t n);
int  orc_hyperkzg_open(const g1_aff_t *srs, const fr_t *poly, size_t ell, const fr_t *point,
                       orc_transcript *t, g1_aff_t *com, g1_aff_t *w, fr_t *v);   /* hyperkzg/mod.rs:400-447 */
int  orc_hyperkzg_verify_trapdoor(const g1_aff_t *srs, const fr_t *tau, const g1_aff_t *C, size_t ell,
                                  const fr_t *point, const fr_t *y, const g1_aff_t *com, const g1_aff_t *w,
                                  const fr_t *v, orc_transcript *t);                /* :451-509, 283-366 */
void orc_eval_as_univariate(const fr_t *f, size_t n, const fr_t *r, fr_t *out);   /* unipoly.rs:247-259 */

#ifdef __cplusplus
}
#endif
#endif
