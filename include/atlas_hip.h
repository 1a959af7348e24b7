/* atlas_hip.h — C-ABI of libatlas_hip.so, the MI355X (gfx950) backend for the
 * jolt-atlas `ONNXProof::prove` hot path.
 *
 * Every entry point is what a Rust `extern "C"` block on the reference side would bind
 * (see INTEGRATION.md); each one cites the reference interface it replaces, as
 * path:line under the jolt-atlas tree.  Plain pointers and sizes only.  All functions
 * return 0 on success and a negative ATLAS_E* code on failure (the Rust shim maps
 * non-zero to the panic / ProofVerifyError::InternalError the reference raises today,
 * joltworks/src/poly/commitment/hyperkzg/commitment_scheme.rs:58-63,118).  No function
 * falls back to the CPU: without a usable HIP device every call fails with
 * ATLAS_ENODEV.
 *
 * Value types are the reference's memory images:
 *   atlas_fr_t   = ark_bn254::Fr   : 4 x u64 LE limbs of the Montgomery residue
 *                                    (joltworks/src/field/ark.rs:16-29)
 *   atlas_u128_t = the u128 drawn by Transcript::challenge_u128; the field value is
 *                  MontU128Challenge::from(it) (field/challenge/mont_ark_u128.rs:51-62)
 *   atlas_transcript_t = Blake2bTranscript {state, n_rounds} (transcripts/blake2b.rs:12-16)
 */
#ifndef ATLAS_HIP_H
#define ATLAS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ATLAS_OK        0
#define ATLAS_ENODEV   (-1)   /* no HIP device / runtime error (see atlas_last_error) */
#define ATLAS_EINVAL   (-2)   /* bad argument (length not a power of two, null, ...) */
#define ATLAS_ENOMEM   (-3)
#define ATLAS_ESTATE   (-4)   /* handle used in the wrong state */
#define ATLAS_EVERIFY  (-5)   /* a proof was rejected (ProofVerifyError) */

typedef struct { uint64_t l[4]; } atlas_fr_t;
typedef struct { uint64_t lo, hi; } atlas_u128_t;
typedef struct { uint8_t state[32]; uint32_t n_rounds; uint32_t pad_[3]; } atlas_transcript_t;

/* BindingOrder (joltworks/src/poly/multilinear_polynomial.rs:69-73) */
#define ATLAS_HIGH_TO_LOW 0
#define ATLAS_LOW_TO_HIGH 1

/* EqSchedule (jolt-atlas-core/src/onnx_proof/ops/einsum/dot.rs:70-95) */
#define ATLAS_EQ_NONE 0   /* degree-2 dot product                       */
#define ATLAS_EQ_HIGH 1   /* High { log_eq = a, low_bits = b }          */
#define ATLAS_EQ_LOW  2   /* Low  { log_k  = a, log_b    = b }          */

/* ---- runtime ------------------------------------------------------------------ */
int  atlas_init(int device_ordinal);            /* hipSetDevice + stream; idempotent */
int  atlas_shutdown(void);
/* One process, N devices: the reference is one Rust process (jolt-atlas-core/src/onnx_proof/mod.rs:153-156), so a drop-in `prove` over the
 * GPUs of a node is N threads of it.  atlas_init_thread gives the CALLING THREAD a runtime of its own on `device_ordinal` (stream set, round
 * channel, device allocator, MSM workspace); every later call of that thread uses it, threads that never call it share the process runtime
 * of atlas_init.  Handles (polynomials, SRS, graphs) are plain device memory and may be created under one runtime and read under another on
 * the same device; an object is proved with by ONE thread at a time.  With atlas_prove_graph_sharded: thread r calls atlas_init_thread(r),
 * builds its graph + SRS handles and joins the shard group as rank r (tests/test_gpu_sharded.py::test_sharded_prove_graph_one_process). */
int  atlas_init_thread(int device_ordinal);
int  atlas_shutdown_thread(void);
const char *atlas_last_error(void);
int  atlas_device_count(void);
int  atlas_sync(void);                           /* drain the library stream */
/* Field value of a challenge: 0 = limbs [0,0,lo,hi] are the Montgomery residue (default,
 * the `mul_hi_bigint_u128` reading), 1 = canonical integer c<<128. */
int  atlas_set_challenge_mode(int mode);
int  atlas_get_challenge_mode(void);
/* Where the Blake2b transcript of the whole-instance provers (atlas_sumcheck_prove_dot, ...) runs.
 * ATLAS_FS_HOST (default): on the calling host thread, over the round channel — kernels of all rounds are
 * enqueued up front, mail their partial sums into pinned memory and poll the round's challenge slot
 * (the reference keeps the transcript on the host too, sumcheck.rs:578-590).  ATLAS_FS_DEVICE: on one
 * wavefront between the data passes, nothing crosses PCIe until the proof is complete.  Same bytes. */
#define ATLAS_FS_HOST   0
#define ATLAS_FS_DEVICE 1
int  atlas_set_fs_mode(int mode);
int  atlas_get_fs_mode(void);

/* ---- transcript (host side; joltworks/src/transcripts/transcript.rs:6-28,
 *      blake2b.rs:81-238) ---------------------------------------------------------- */
int atlas_transcript_new(atlas_transcript_t *t, const uint8_t *label, size_t label_len);
int atlas_transcript_append_message(atlas_transcript_t *t, const uint8_t *msg, size_t len);
int atlas_transcript_append_bytes(atlas_transcript_t *t, const uint8_t *bytes, size_t len);
int atlas_transcript_append_u64(atlas_transcript_t *t, uint64_t x);
int atlas_transcript_append_scalar(atlas_transcript_t *t, const atlas_fr_t *s);
int atlas_transcript_append_scalars(atlas_transcript_t *t, const atlas_fr_t *s, size_t n);
int atlas_transcript_challenge_u128(atlas_transcript_t *t, atlas_u128_t *out);
int atlas_transcript_challenge_scalar(atlas_transcript_t *t, atlas_fr_t *out);
int atlas_challenge_to_fr(const atlas_u128_t *c, atlas_fr_t *out);  /* Into<Fr> for MontU128Challenge */

/* ---- device polynomials (MultilinearPolynomial<F>, multilinear_polynomial.rs:22-35) -- */
typedef struct atlas_poly *atlas_poly_t;          /* opaque; owns HBM */
/* LargeScalars: copy `len` Fr (power of two) host -> HBM */
int atlas_poly_upload_fr(const atlas_fr_t *host, size_t len, atlas_poly_t *out);
/* I32Scalars (CompactPolynomial<i32>, compact_polynomial.rs:20-40): 4 B/coeff in HBM,
 * promoted to Fr by the first bind */
int atlas_poly_upload_i32(const int32_t *host, size_t len, atlas_poly_t *out);
/* wrap memory that is already resident (device pointer), no copy, not owned */
int atlas_poly_wrap_device_fr(void *dptr, size_t len, atlas_poly_t *out);
int atlas_poly_wrap_device_i32(void *dptr, size_t len, atlas_poly_t *out);   /* I32Scalars view of resident int32, not owned */
int atlas_poly_len(atlas_poly_t p, size_t *len);   /* current (bound) length */
int atlas_poly_download(atlas_poly_t p, atlas_fr_t *host, size_t cap);  /* current coeffs */
int atlas_poly_clone(atlas_poly_t p, atlas_poly_t *out);
int atlas_poly_free(atlas_poly_t p);
/* PolynomialBinding::bind_parallel (multilinear_polynomial.rs:657-667;
 * dense_mlpoly.rs:84-89; compact_polynomial.rs:272-353) */
int atlas_poly_bind(atlas_poly_t p, const atlas_u128_t *r, int order);
/* PolynomialBinding::final_claim (len must be 1) */
int atlas_poly_final_claim(atlas_poly_t p, atlas_fr_t *out);

/* ---- sumcheck: EinsumDotProver behind SumcheckInstanceProver
 *      (joltworks/src/subprotocols/sumcheck_prover.rs:10-68;
 *       jolt-atlas-core/src/onnx_proof/ops/einsum/dot.rs:255-375) ---------------------- */
typedef struct atlas_dot_prover *atlas_dot_prover_t;
/* Takes ownership of left/right/eq (they are consumed by binding, like the reference's
 * `left`, `right`, `eq` fields).  eq may be NULL iff schedule == ATLAS_EQ_NONE.
 * MeanOfSquaresReductionProver (jolt-atlas-core/src/onnx_proof/ops/mean_of_squares.rs:363-398) is this prover with
 * left = right = the operand (two handles: atlas_poly_clone), eq = EqPolynomial::evals(r_node_output) and
 * ATLAS_EQ_HIGH { sched_a = log_retained, sched_b = log_reduced }. */
int atlas_dot_prover_new(atlas_poly_t left, atlas_poly_t right, atlas_poly_t eq, int schedule,
                         size_t sched_a, size_t sched_b, atlas_dot_prover_t *out);
int atlas_dot_prover_free(atlas_dot_prover_t p);
/* input_claim(): sum_h L(h) R(h) [EQ(h)] over the unbound operands.  The reference reads
 * it from the opening accumulator (dot.rs:204-207 -> fused_rebase::fused_input_claim);
 * synthetic instances have no accumulator, so the device computes it. */
int atlas_dot_input_claim(atlas_dot_prover_t p, atlas_fr_t *out);
/* compute_message(round, previous_claim) -> UniPoly coefficients c0..c_deg (deg+1 Fr),
 * dot.rs:290-350 + UniPoly::from_evals_and_hint */
int atlas_dot_compute_message(atlas_dot_prover_t p, size_t round, const atlas_fr_t *previous_claim,
                              atlas_fr_t *coeffs_out, size_t *n_coeffs);
/* ingest_challenge(r_j, round), dot.rs:352-375 */
int atlas_dot_ingest_challenge(atlas_dot_prover_t p, const atlas_u128_t *r_j, size_t round);
/* left/right/eq final_claim()s read by cache_openings, dot.rs:377-400 */
int atlas_dot_final_claims(atlas_dot_prover_t p, atlas_fr_t out[3]);

/* Sumcheck::prove for that instance with the transcript resident on the device
 * (joltworks/src/subprotocols/sumcheck.rs:565-599): the whole round loop runs as a chain
 * of launches without a host round-trip.  transcript is read and updated.
 *   compressed_polys : n_rounds * degree Fr, row i = coeffs_except_linear_term of round i
 *   challenges       : n_rounds raw u128 draws (r_sumcheck before masking)
 *   final_claims     : left(r), right(r), eq(r) (or 1)
 * Consumes the prover's polynomials. */
int atlas_sumcheck_prove_dot(atlas_dot_prover_t p, const atlas_fr_t *input_claim,
                             atlas_transcript_t *transcript, atlas_fr_t *compressed_polys,
                             atlas_u128_t *challenges, atlas_fr_t final_claims[3]);

/* ---- EQ tables and MLE evaluation ---------------------------------------------------- */
/* EqPolynomial::evals / evals_with_scaling (joltworks/src/poly/eq_poly.rs:77-101): the 2^n
 * table { scaling * eq(r, x) }, big-endian index (r[0] = MSB); r as Fr (a challenge is
 * converted with atlas_challenge_to_fr).  scaling may be NULL (= 1). */
int atlas_eq_evals(const atlas_fr_t *r, size_t n, const atlas_fr_t *scaling, atlas_poly_t *out);
/* PolynomialEvaluation::evaluate (multilinear_polynomial.rs:766-862; dense_mlpoly.rs:265-305):
 * P(r), r[0] = MSB variable; p is not modified */
int atlas_poly_evaluate(atlas_poly_t p, const atlas_fr_t *r, size_t n, atlas_fr_t *out);
/* MultilinearPolynomial::batch_evaluate (multilinear_polynomial.rs:682): `count` polynomials of
 * 2^n coefficients at one point; out[i] = polys[i](r).  The eq tables are built once and the call
 * synchronises once, whatever `count` (<= 64) is. */
int atlas_poly_evaluate_many(const atlas_poly_t *polys, size_t count, const atlas_fr_t *r, size_t n, atlas_fr_t *out);

/* ---- einsum operand folds: i32 matrix x Fr vector
 *      (EinsumLayout::fold, jolt-atlas-core/src/onnx_proof/ops/einsum/mk_kn_mn.rs:47-79;
 *       the operands are Tensor<i32>, row-major, resident in HBM) ------------------------- */
int atlas_i32_upload(const int32_t *host, size_t n, int32_t **d_out);   /* Tensor<i32> -> HBM */
int atlas_i32_free(int32_t *d);
/* out[j] = sum_h M[j*cols + h] * eq[h]   (`right` of mk,kn->mn with M = B (k x n), eq = eq_r_n);
 * rows must be a power of two; eq = device table of `cols` Fr (atlas_eq_evals) */
int atlas_fold_i32_rows(const int32_t *d_matrix, size_t rows, size_t cols, atlas_poly_t eq, atlas_poly_t *out);
/* out[j] = sum_i M[i*cols + j] * eq[i]   (`left` of mk,kn->mn with M = A (m x k), eq = eq_r_m) */
int atlas_fold_i32_cols(const int32_t *d_matrix, size_t rows, size_t cols, atlas_poly_t eq, atlas_poly_t *out);

/* Strided forms covering the batched layouts (ops/einsum/bmk_rhs_mbn.rs:78-110, mbk_rhs_bmn.rs:78-119,
 * k_nk_n.rs:46-68; transposes of utils/dims.rs:658-690 folded into the output strides):
 *   rows_batched: out[o0*t0 + o1*t1] = sum_{h<R} M[o0*s0 + o1*s1 + h] * eq[h]   (reduce axis contiguous)
 *   cols_batched: out[z*tB + j*tC]   = sum_{i<R} M[z*sB + i*sR + j] * eq[i]     (output axis j contiguous)
 * e.g. bmk,?->mbn: left = cols_batched(B=b, sB=k*m, R=m, sR=k, C=k, tB=1, tC=b); right (kbn) =
 * rows_batched(n0=k, n1=b, s0=b*n, s1=n, R=n, t0=b, t1=1), (bkn) = rows_batched(n0=b, n1=k, s0=k*n,
 * s1=n, R=n, t0=1, t1=b).  Output length n0*n1 (B*C) must be a power of two. */
int atlas_fold_i32_rows_batched(const int32_t *d_matrix, size_t n0, size_t n1, size_t s0, size_t s1, size_t R,
                                size_t t0, size_t t1, atlas_poly_t eq, atlas_poly_t *out);
int atlas_fold_i32_cols_batched(const int32_t *d_matrix, size_t B, size_t sB, size_t R, size_t sR, size_t C,
                                size_t tB, size_t tC, atlas_poly_t eq, atlas_poly_t *out);

/* two batch axes (ops/einsum/rbmk_rbnk_bmn.rs:219-289, the acbmk,kcn->cbmn left operand):
 *   out[z0*tB0 + z1*tB1 + j*tC] = sum_{i<R} M[z0*sB0 + z1*sB1 + i*sR + j] * eq[i],  z0 < B0, z1 < B1, j < C */
int atlas_fold_i32_cols_batched2(const int32_t *d_matrix, size_t B0, size_t sB0, size_t tB0, size_t B1, size_t sB1,
                                 size_t tB1, size_t R, size_t sR, size_t C, size_t tC, atlas_poly_t eq,
                                 atlas_poly_t *out);
/* out[(r*repeat + q)*row_len + j] = base[r*row_len + j]: broadcast of a folded operand over batch axes it does not
 * depend on (rbmk_rbnk_bmn.rs:270-289) */
int atlas_poly_repeat_rows(atlas_poly_t base, size_t rows, size_t row_len, size_t repeat, atlas_poly_t *out);

/* EinsumLayout::fold (jolt-atlas-core/src/onnx_proof/ops/einsum/mod.rs:71-115 registers the layouts; folds in
 * mk_kn_mn.rs:47-79, bmk_rhs_mbn.rs:78-110, mbk_rhs_bmn.rs:78-119, k_nk_n.rs:46-68, rbmk_rbnk_bmn.rs:163-338): the
 * left / right operand polynomials the dot-product sumcheck runs over, from the two Tensor<i32> operands resident in
 * HBM and the eq tables of the output point (eq_r_m over the m axis, eq_r_n over the n axis; atlas_eq_evals).
 * dims: MK_KN_MN (m,k,n); BMK_* / MBK_* (b,m,k,n); K_NK_N (k,n: no left fold, d_left / eq_r_m / left_out may be
 * null); ABMK_ABNK_ABMN (a,b,m,n,k); ACBMK_KCN_CBMN (a,c,b,m,n,k); CBMK_CBKN_AMN (c*b,m,n,k). */
#define ATLAS_EINSUM_MK_KN_MN        0
#define ATLAS_EINSUM_BMK_BKN_MBN     1
#define ATLAS_EINSUM_BMK_KBN_MBN     2
#define ATLAS_EINSUM_MBK_BNK_BMN     3
#define ATLAS_EINSUM_MBK_NBK_BMN     4
#define ATLAS_EINSUM_K_NK_N          5
#define ATLAS_EINSUM_ABMK_ABNK_ABMN  6
#define ATLAS_EINSUM_ACBMK_KCN_CBMN  7
#define ATLAS_EINSUM_CBMK_CBKN_AMN   8
int atlas_einsum_fold(int layout, const size_t *dims, size_t n_dims, const int32_t *d_left, const int32_t *d_right,
                      atlas_poly_t eq_r_m, atlas_poly_t eq_r_n, atlas_poly_t *left_out, atlas_poly_t *right_out);

/* ---- Shout lookup argument: prover-side table builds
 *      (joltworks/src/subprotocols/shout.rs:193-262, 550-598) ---------------------------- */
/* ReadRafProver::initialize: G[k] = sum_{j : lookup_indices[j] = k} E[j], E = eq_r (device
 * table of >= T Fr); out = 2^log_K Fr */
int atlas_shout_read_raf_G(const uint64_t *lookup_indices, size_t T, size_t log_K, atlas_poly_t eq_r,
                           atlas_poly_t *out);
/* compute_ra_evals: the same histogram for each of the d = ceil(log_K / log_k_chunk) chunks of
 * the lookup index (chunk 0 = most significant, config.rs:73-75); out = d * 2^log_k_chunk Fr,
 * row i = chunk i (shared by the RaVirtual / HammingWeight / Booleanity provers) */
int atlas_shout_ra_evals(const uint64_t *lookup_indices, size_t T, size_t log_K, size_t log_k_chunk,
                         atlas_poly_t eq_r_cycle, atlas_poly_t *out);
/* ReadRafProver as a sumcheck instance: sum_k G[k] (val[k] + gamma int[k]) over the 2^log_K
 * table, HighToLow, degree 2.  Takes ownership of G; the returned handle is driven with the
 * atlas_dot_* / atlas_sumcheck_prove_dot entry points (final_claims[0] = G(r), what
 * cache_openings appends, shout.rs:264-277). */
int atlas_shout_read_raf_prover_new(atlas_poly_t G, const int32_t *table, size_t log_K,
                                    const atlas_fr_t *gamma, atlas_dot_prover_t *out);

/* ---- sumcheck: MulProver over GruenSplitEqPolynomial (element-wise ops)
 *      (jolt-atlas-core/src/onnx_proof/ops/mul.rs:125-185;
 *       joltworks/src/poly/split_eq_poly.rs:86-145,331-429,526-597) ----------------------- */
typedef struct atlas_mul_prover *atlas_mul_prover_t;
/* MulProver::initialize: takes ownership of the two operand polynomials (length 2^n);
 * w = r_node_output as n Fr (big-endian point), bound LowToHigh */
int atlas_mul_prover_new(atlas_poly_t left, atlas_poly_t right, const atlas_fr_t *w, size_t n,
                         atlas_mul_prover_t *out);
int atlas_mul_prover_free(atlas_mul_prover_t p);
/* sum_x eq(w,x) L(x) R(x) (the node-output opening the reference reads, mul.rs:60-66) */
int atlas_mul_input_claim(atlas_mul_prover_t p, atlas_fr_t *out);
/* Sumcheck::prove for that instance, transcript on the device; rows of 3 coefficients
 * (c0, c2, c3); final_claims = left(r), right(r), eq(w, r) */
int atlas_sumcheck_prove_mul(atlas_mul_prover_t p, const atlas_fr_t *input_claim,
                             atlas_transcript_t *transcript, atlas_fr_t *compressed_polys,
                             atlas_u128_t *challenges, atlas_fr_t final_claims[3]);
/* The same instance behind the trait's per-round methods (mul.rs:160-185), host-stepped so
 * that it can take part in a BatchedSumcheck: compute_message -> 4 coefficients of
 * gruen_poly_deg_3 (split_eq_poly.rs:331-429) interpolated with the claim hint;
 * ingest_challenge binds both operands LowToHigh and the split-eq scalar. */
int atlas_mul_compute_message(atlas_mul_prover_t p, size_t round, const atlas_fr_t *previous_claim,
                              atlas_fr_t *coeffs_out, size_t *n_coeffs);
int atlas_mul_ingest_challenge(atlas_mul_prover_t p, const atlas_u128_t *r_j, size_t round);
int atlas_mul_final_claims(atlas_mul_prover_t p, atlas_fr_t out[3]);
size_t atlas_mul_num_rounds(atlas_mul_prover_t p);   /* SumcheckInstanceProver::num_rounds */
size_t atlas_dot_num_rounds(atlas_dot_prover_t p);
int atlas_dot_degree(atlas_dot_prover_t p);          /* SumcheckInstanceProver::degree */

/* ---- BatchedSumcheck::prove (joltworks/src/subprotocols/sumcheck.rs:30-184): several
 *      instances of differing round counts under one transcript, front-loaded batching ---- */
typedef struct atlas_batched *atlas_batched_t;
int atlas_batched_new(atlas_batched_t *out);
int atlas_batched_free(atlas_batched_t b);           /* does not free the instances */
/* Vec<&mut dyn SumcheckInstanceProver> entries, in order; input_claim = what
 * SumcheckInstanceProver::input_claim(accumulator) returns for that instance */
int atlas_batched_add_dot(atlas_batched_t b, atlas_dot_prover_t p, const atlas_fr_t *input_claim);
int atlas_batched_add_mul(atlas_batched_t b, atlas_mul_prover_t p, const atlas_fr_t *input_claim);
/* Runs the protocol: appends the input claims, draws the batching coefficients, then per
 * round combines the instances' round polynomials, compresses, appends, draws r_j and has
 * the active instances ingest it.
 *   compressed : max_rounds rows of row_stride Fr; row i holds n_coeffs[i] coefficients
 *                (coeffs_except_linear_term of the batched round polynomial; row_stride must
 *                be at least the largest instance degree)
 *   challenges : max_rounds raw u128 draws
 * Instances are left fully bound; their final claims are read with *_final_claims. */
int atlas_batched_prove(atlas_batched_t b, atlas_transcript_t *transcript, atlas_fr_t *compressed,
                        size_t row_stride, uint32_t *n_coeffs, atlas_u128_t *challenges,
                        size_t *max_rounds_out);

/* ---- generic SumcheckInstanceProver handle (joltworks/src/subprotocols/sumcheck_prover.rs:
 *      10-68): what the constructors below return; host-stepped, so an instance can sit in a
 *      BatchedSumcheck next to the dot / Mul provers ---------------------------------------- */
typedef struct atlas_instance *atlas_instance_t;
size_t atlas_instance_num_rounds(atlas_instance_t i);
size_t atlas_instance_degree(atlas_instance_t i);
int atlas_instance_compute_message(atlas_instance_t i, size_t round, const atlas_fr_t *previous_claim,
                                   atlas_fr_t *coeffs_out, size_t cap, size_t *n_coeffs);
int atlas_instance_ingest_challenge(atlas_instance_t i, const atlas_u128_t *r_j, size_t round);
/* the polynomials' final_claim()s in the order cache_openings appends them */
int atlas_instance_final_claims(atlas_instance_t i, atlas_fr_t *out, size_t cap, size_t *n);
int atlas_instance_free(atlas_instance_t i);
/* Sumcheck::prove (sumcheck.rs:565-599) for one instance; rows as atlas_batched_prove */
int atlas_instance_prove(atlas_instance_t i, const atlas_fr_t *input_claim, atlas_transcript_t *transcript,
                         atlas_fr_t *compressed, size_t row_stride, uint32_t *n_coeffs,
                         atlas_u128_t *challenges);
int atlas_batched_add_instance(atlas_batched_t b, atlas_instance_t i, const atlas_fr_t *input_claim);

/* ---- one-hot "ra" instances of the lookup arguments.  H_indices = d host arrays of T = 2^log_T
 *      int32 (the reference's Vec<Vec<Option<u8>>> / Option<u16>, negative = None).
 *      log_T = 0 (ONE cycle: a lookup operator over a one-element tensor) is accepted by the ra / booleanity /
 *      prefix-suffix read-raf constructors below: the instances then have their address rounds only (RaVirtual:
 *      no round at all) and r_cycle / r_node_output may be NULL -------------------------------- */
/* RaSumcheckProver::gen (subprotocols/ra_virtual.rs:97-125): sum_j eq(r_cycle, j) prod_i ra_i(j),
 * degree d + 1, log_T rounds LowToHigh.  r_address_chunks = d * log_k_chunk Fr, row i = chunk i
 * (OneHotParams::compute_r_address_chunks, config.rs:77-100); r_cycle = log_T Fr, big-endian.
 * final claims: ra_0(r) .. ra_{d-1}(r).  d <= 16. */
int atlas_ra_virtual_new(const int32_t *const *H_indices, size_t d, size_t log_k_chunk, size_t log_T,
                         const atlas_fr_t *r_address_chunks, const atlas_fr_t *r_cycle,
                         atlas_instance_t *out);
/* The same provers straight from the T lookup indices: the d = ceil(log_K / log_k_chunk) chunk rows
 * (compute_instruction_h_indices, shout.rs:532-547; OneHotParams::lookup_index_chunk, config.rs:73-75) are cut
 * on the device and r_address (log_K Fr) is split by compute_r_address_chunks (config.rs:77-100).
 * LIFETIME: lookup_indices may be a host or a device pointer.  A host vector is copied by the call.  A DEVICE vector is
 * BORROWED — the instance reads it in place during its proof (the row gather of Booleanity's first cycle round, the lazy
 * rounds of both provers), not at construction — so it must stay allocated and unchanged until atlas_instance_free of
 * every instance built over it (the graph prover passes the node's witness vectors, which outlive the node's flow). */
int atlas_ra_virtual_from_lookups_new(const uint64_t *lookup_indices, size_t log_T, size_t log_K, size_t log_k_chunk,
                                      const atlas_fr_t *r_address, const atlas_fr_t *r_cycle, atlas_instance_t *out);
int atlas_booleanity_from_lookups_new(const atlas_fr_t *G, const uint64_t *lookup_indices, size_t log_T, size_t log_K,
                                      size_t log_k_chunk, const atlas_fr_t *gammas, const atlas_fr_t *r_address,
                                      const atlas_fr_t *r_cycle, atlas_instance_t *out);
/* BooleanitySumcheckProver::gen (subprotocols/booleanity.rs:169-190): log_k_chunk address rounds
 * over G (d * 2^log_k_chunk Fr, from atlas_shout_ra_evals) then log_T cycle rounds over the
 * gathered H_i; degree 3; input claim 0.  gammas = field values of the d batching challenges. */
int atlas_booleanity_new(const atlas_fr_t *G, const int32_t *const *H_indices, size_t d,
                         size_t log_k_chunk, size_t log_T, const atlas_fr_t *gammas,
                         const atlas_fr_t *r_address, const atlas_fr_t *r_cycle, atlas_instance_t *out);
/* ---- prefix-suffix Shout read-raf sumcheck, unary lookups (joltworks/src/subprotocols/ps_shout/
 *      mod.rs:227-560, unary.rs:110-148 ps_read_raf_prover) with ReluTable<X_LEN>
 *      (lookup_tables/relu.rs) and the SignedIdentity RAF term:
 *        sum_{k,t} eq(r_node_output, t) ra(k, t) (Relu(k) + gamma * SignedIdentity(k)),
 *      X_LEN address rounds (8 phases of X_LEN/8 bits) then log_T cycle rounds, degree 2.
 *      lookup_indices = T = 2^log_T values of X_LEN bits (LookupBits); gamma = the challenge_scalar the
 *      caller drew; input claim = rv_claim + gamma * operand_claim.  Final claim = ra(r).
 *      X_LEN = 16 or 32 (the reference's WordNoMSB suffix is a u32, exact only up to 32 bits). ---- */
int atlas_ps_shout_relu_new(const uint64_t *lookup_indices, size_t log_T, size_t xlen,
                            const atlas_fr_t *r_node_output, const atlas_fr_t *gamma,
                            atlas_instance_t *out);

/* The same read-raf prover for the clamp family ClampBoundedTable<X_LEN, BOUND, SYMMETRIC>
 * (joltworks/src/lookup_tables/clamp.rs:44-121): ClampTable = (X_LEN, 9, true), SaturationTable =
 * (64, 31, true), ActivationClampTable = (X_LEN, MODEL_SCALE + 3, true), SoftmaxClampTable =
 * (X_LEN, SOFTMAX_CLAMP_BOUND, false).  X_LEN = 16, 32 or 64; BOUND <= 31. */
int atlas_ps_shout_clamp_new(const uint64_t *lookup_indices, size_t log_T, size_t xlen, size_t bound,
                             int symmetric, const atlas_fr_t *r_node_output, const atlas_fr_t *gamma,
                             atlas_instance_t *out);
/* Element-wise operator provers (jolt-atlas-core/src/onnx_proof/ops/add.rs:283-304, sub.rs:267-288,
 * neg.rs:206-222, square.rs:163-183, mul.rs:160-199, iff.rs:189-224, cube.rs:159-171): sum_x eq(r_node_output, x)
 * f(operands(x)) with a LowToHigh GruenSplitEqPolynomial; n_vars rounds; degree 2 (Add, Sub, Neg), 3 (Square, Mul,
 * Iff) or 4 (Cube, through compute_mle_product_sum).  `operands` in the order the reference caches their openings
 * (Iff: mask, a, b), each an atlas_poly_t of 2^n_vars coefficients (LargeScalars or I32Scalars; not consumed).
 * final claims = the operands' final claims in that order. */
enum { ATLAS_EW_ADD = 0, ATLAS_EW_SUB = 1, ATLAS_EW_NEG = 2, ATLAS_EW_SQUARE = 3, ATLAS_EW_IFF = 4, ATLAS_EW_MUL = 5,
       ATLAS_EW_CUBE = 6, ATLAS_EW_DIV = 7, ATLAS_EW_RSQRT = 8, ATLAS_EW_DOT = 9, ATLAS_EW_GATHER = 10,
       ATLAS_EW_HAMMING_BOOL = 11, ATLAS_EW_TELEPORT_DIV = 12 };
/* ATLAS_EW_DIV (ops/div.rs:329-362): operands (left, right, q, R), summand right q + R - left, degree 3.
 * ATLAS_EW_RSQRT (ops/rsqrt.rs:390-433): operands (input, quotient, output, div_remainder, sqrt_remainder),
 * constants (S^3, gamma), summand input quotient + div_remainder - S^3 + gamma (output^2 + sqrt_remainder -
 * quotient), degree 3.  ScalarConstDiv (ops/scalar_const_div.rs:227-248) is ATLAS_EW_SUB over (left, R).
 * ATLAS_EW_DOT (ops/reshape.rs:286-315, slice.rs:254-281, concat.rs:290-325): NO eq factor (r_node_output is ignored and
 * may be NULL); operands (input_0, selector_0, input_1, selector_1, ...), summand sum_t input_t selector_t, degree 2.
 * ATLAS_EW_GATHER (ops/gather/mod.rs:232-268): no eq factor; operands (index_onehot, dictionary, identity), constant
 * gamma, summand ra (dictionary + gamma identity), degree 2.
 * ATLAS_EW_HAMMING_BOOL (joltworks/src/subprotocols/hamming_booleanity.rs:131-165): operands hw_0 .. hw_{d-1} (d <= 16),
 * constants gamma_powers (d of them), summand sum_d gamma_d (hw_d^2 - hw_d), degree 3.
 * ATLAS_EW_TELEPORT_DIV (jolt-atlas-core/src/onnx_proof/neural_teleport/division.rs:231-256): operands (input, quotient,
 * remainder), constant tau, summand tau quotient + remainder - input, degree 2.
 * ATLAS_EW_GATHER also serves the small-table activation / Sin / Cos execution provers (ops/activation_clamped/mod.rs:295-332,
 * ops/sin.rs:456-492, cos.rs): ra (table + gamma identity); ATLAS_EW_DOT with one pair is GammaFoldProver
 * (joltworks/src/subprotocols/gamma_fold.rs:131-158).
 * `constants`: n_constants Fr, 2 for Rsqrt, 1 for Gather / TeleportDivision, d for HammingBooleanity, 0 (may be NULL) otherwise. */
int atlas_elementwise_new(int op, const atlas_poly_t *operands, size_t n_operands, const atlas_fr_t *r_node_output,
                          size_t n_vars, const atlas_fr_t *constants, size_t n_constants, atlas_instance_t *out);
/* Lookup indices resident in HBM.  Every constructor above and below that takes `const uint64_t *lookup_indices`
 * accepts host memory or device memory (the copy is made with hipMemcpyDefault), so one upload — or none, when the
 * indices are cut from operand tensors that already live on the device — serves the read-raf prover and the one-hot
 * checks of an operator.
 *   atlas_lookup_indices_from_operands: compute_lookup_indices_from_operands (jolt-atlas-core/src/utils/mod.rs:43-122):
 *   d_right == NULL: `value as u32 as u64`; otherwise interleave_bits(left as u32, right as u32)
 *   (joltworks/src/utils/mod.rs:146-164).  d_left / d_right: device Tensor<i32> (atlas_i32_upload). */
int atlas_u64_upload(const uint64_t *host, size_t n, uint64_t **d_out);
int atlas_u64_free(uint64_t *d);
int atlas_lookup_indices_from_operands(const int32_t *d_left, const int32_t *d_right, size_t n, uint64_t **d_out);
/* The two-phase provers of softmax_last_axis (jolt-atlas-core/src/onnx_proof/ops/softmax_last_axis/exp_sum.rs:146-197,
 * max.rs:185-261, recip_mult.rs:196-268) and SumAxisProver (ops/sum/axis.rs:220-232).  Tensors are [k][j] with
 * K = 2^log_K rows and N = 2^log_N last-axis entries, bound LowToHigh (last axis first); log_K + log_N rounds.
 *   ATLAS_SM_EXP_SUM        a = exp_q;            r = r0_k (log_K Fr);  degree 2 (phase-1 messages have degree 1)
 *   ATLAS_SM_MAX_INDICATOR  a = X, b = e;         r = r1_k (log_K Fr);  degree 3
 *   ATLAS_SM_RECIP_MULT     a = exp_q, b = inv_sum (2^log_K);  r = the opening point (log_K + log_N Fr);  degree 3
 *   ATLAS_SM_SUM_AXIS       a = operand (2^log_N, log_K = 0, HighToLow, r unused);  degree 1
 * Operands are LargeScalars or I32Scalars handles (not consumed); final claims = a(r) [, b(r)]. */
enum { ATLAS_SM_EXP_SUM = 0, ATLAS_SM_MAX_INDICATOR = 1, ATLAS_SM_RECIP_MULT = 2, ATLAS_SM_SUM_AXIS = 3 };
int atlas_softmax_instance_new(int kind, atlas_poly_t a, atlas_poly_t b, size_t log_K, size_t log_N,
                               const atlas_fr_t *r, atlas_instance_t *out);
/* The same unary prover over RightShiftTable<XLEN> (joltworks/src/lookup_tables/right_shift.rs:17-60; used by
 * the Sin / Cos trig downscale, jolt-atlas-core/src/onnx_proof/ops/sin.rs:108, cos.rs:113): Val(k) = k >> shift
 * (unsigned; shift = TRIG_DOWNSCALE_BITS in the reference), prefix TrigRightShift, suffixes [One, TrigRightShift],
 * RAF = SignedIdentity.  xlen = 16 or 32, shift < xlen. */
int atlas_ps_shout_rshift_new(const uint64_t *lookup_indices, size_t log_T, size_t xlen, size_t shift,
                              const atlas_fr_t *r_node_output, const atlas_fr_t *gamma, atlas_instance_t *out);
/* The binary flavour (joltworks/src/subprotocols/ps_shout/binary.rs:148-200 ps_read_raf_prover) with
 * UnsignedLessThanTable<32> (lookup_tables/unsigned_less_than.rs): lookup index = interleave_bits(x, y)
 * (utils/mod.rs:146-164), 64 address rounds, summand ra * (LT(x,y) + gamma * SignedLeft + gamma^2 *
 * SignedRight); input claim = rv_claim + gamma * (left_operand_claim + gamma * right_operand_claim). */
int atlas_ps_shout_ult_new(const uint64_t *lookup_indices, size_t log_T, const atlas_fr_t *r_node_output,
                           const atlas_fr_t *gamma, atlas_instance_t *out);
/* IdentityRCProver::gen (joltworks/src/subprotocols/identity_range_check.rs:196-420): the range check
 * sum_{k,t} eq(r_node_output, t) ra(k, t) Identity(k) over log_K address bits in `phases` phases
 * (IdentityRCProvider::{log_K, phases}), then log_T cycle rounds; degree 2; final claim = ra(r). */
int atlas_identity_range_check_new(const uint64_t *lookup_indices, size_t log_T, size_t log_K, size_t phases,
                                   const atlas_fr_t *r_node_output, atlas_instance_t *out);

/* ---- N-to-1 evaluation reduction: EvalReductionInstance::prove with compute_h and eval_on_l
 *      (joltworks/src/subprotocols/evaluation_reduction.rs:91-147, 213-249).  points = N rows of n
 *      Fr (the opening points of one polynomial), claims = N Fr.  h(t) = P(l(t)) is returned as
 *      trimmed monomial coefficients (at most n (N-1) + 1), appended to the transcript uncompressed,
 *      then r' = l(x'), v' = h(x') for the drawn x'.  mle is not consumed.  N = 1 is the short path
 *      (h = [claim], transcript untouched). ------------------------------------------------------ */
int atlas_eval_reduction_prove(atlas_poly_t mle, const atlas_fr_t *points, const atlas_fr_t *claims,
                               size_t N, size_t n, atlas_transcript_t *transcript, atlas_fr_t *h_out,
                               size_t h_cap, size_t *h_len, atlas_fr_t *r_out, atlas_fr_t *claim_out);

/* ---- opening-reduction provers (joltworks/src/subprotocols/opening_reduction.rs), the instances
 *      ProverOpeningAccumulator::prove_batch_opening_sumcheck batches (poly/opening_proof.rs:447-532).
 *      Both bind HighToLow over the HighToLow GruenSplitEqPolynomial, degree 2. ------------------ */
/* DensePolynomialProverOpening (:355-425): sum_j eq(opening_point, j) P(j); takes ownership of poly
 * (LargeScalars or I32Scalars, length 2^n); final claim = P(r_sumcheck). */
int atlas_dense_opening_new(atlas_poly_t poly, const atlas_fr_t *opening_point, size_t n,
                            atlas_instance_t *out);
/* OneHotPolynomialProverOpening::{new,initialize} (:509-575): sum_{k,j} eq(r_address,k)
 * eq(r_cycle,j) ra(k,j) for the one-hot polynomial given by nonzero_indices (T = 2^log_T host int32,
 * negative = None; OneHotPolynomial::nonzero_indices, one_hot_polynomial.rs:21-28), K = 2^log_K.
 * log_K address rounds then log_T cycle rounds; final claim = H(r). */
int atlas_onehot_opening_new(const int32_t *nonzero_indices, size_t log_K, size_t log_T,
                             const atlas_fr_t *r_address, const atlas_fr_t *r_cycle,
                             atlas_instance_t *out);

/* R one-hot openings that share r_cycle (one EqCycleState in the reference, opening_proof.rs:339-343):
 * out receives R handles, one per polynomial, usable like any other instance but advancing round by
 * round together (as inside one BatchedSumcheck); their cycle-phase folds and binds run as one launch
 * each.  r_addresses = R rows of log_K Fr. */
int atlas_onehot_opening_group_new(const int32_t *const *nonzero_indices, size_t R, size_t log_K, size_t log_T,
                                   const atlas_fr_t *r_addresses, const atlas_fr_t *r_cycle,
                                   atlas_instance_t *out);

/* HammingWeightSumcheckProver::gen (subprotocols/hamming_weight.rs:106-116): sum_k sum_i
 * gamma^i G_i[k], degree 1, log_k_chunk rounds */
int atlas_hamming_weight_new(const atlas_fr_t *G, size_t d, size_t log_k_chunk,
                             const atlas_fr_t *gamma_powers, atlas_instance_t *out);

/* ---- build_materialized_rlc (joltworks/src/poly/rlc_polynomial.rs:13-78): the joint polynomial
 *      joint[i] = sum_j coeff_j * dense_j[i];  joint[k_j[t] * T_j + t] += coeff_j (one-hot) ---- */
typedef struct { atlas_poly_t poly; atlas_fr_t coeff; } atlas_rlc_dense_t;   /* LargeScalars / I32Scalars */
typedef struct {
    const int32_t *k;     /* OneHotPolynomial::nonzero_indices as i32, negative = None; T entries */
    size_t T, K;          /* cycles, address space (one_hot_polynomial.rs:21-40) */
    atlas_fr_t coeff;
    int k_on_device;      /* non-zero: k is a device pointer */
} atlas_rlc_onehot_t;
/* joint length = max(max dense len, max K*T) and must be a power of two (DensePolynomial::new,
 * dense_mlpoly.rs:33-45).  The inputs are not consumed. */
int atlas_rlc_build(const atlas_rlc_dense_t *dense, size_t n_dense, const atlas_rlc_onehot_t *onehot,
                    size_t n_onehot, atlas_poly_t *out);

/* ---- SRS + multi-scalar multiplication: the arithmetic behind the CommitmentScheme
 *      plug-in (joltworks/src/poly/commitment/commitment_scheme.rs:11-131) for HyperKZG --- */
typedef struct { uint64_t l[4]; } atlas_fq_t;            /* ark_bn254::Fq, Montgomery limbs */
/* ark_bn254::G1Affine image: x, y, infinity flag (72 bytes with padding) */
typedef struct { atlas_fq_t x, y; uint64_t infinity; } atlas_g1_affine_t;
typedef struct atlas_srs *atlas_srs_t;                   /* KZGProverKey::g1_powers resident in HBM */
/* Upload `n` affine bases.  `stride_bytes` = distance between consecutive points in the
 * caller's slice (72 for &[ark_bn254::G1Affine]; 64 for packed x,y); x at +0, y at +32,
 * infinity flag byte at +64 when the stride allows.  (SRS / KZGProverKey, kzg.rs:18-143;
 * done once, reused for every commit/open.) */
int atlas_srs_upload(const void *bases, size_t n, size_t stride_bytes, atlas_srs_t *out);
/* SRS::setup (kzg.rs:26-93) for a caller-supplied trapdoor: bases[i] = tau^(i+1) * G1,
 * G1 = (1,2) — the reference's power layout (starts at beta^1).  Test/bench SRS. */
int atlas_srs_generate(const atlas_fr_t *tau, size_t n, atlas_srs_t *out);
int atlas_srs_len(atlas_srs_t s, size_t *len);
int atlas_srs_download(atlas_srs_t s, size_t offset, size_t n, atlas_g1_affine_t *out);
int atlas_srs_free(atlas_srs_t s);
/* Fixed-base table for the first n_points powers (0 = all): 2^(c j) * g1_powers[i] for j < ceil(255 / c), kept in HBM
 * next to the SRS (n_points * ceil(255 / c) * 64 bytes).  The prover key is fixed after setup
 * (KZGProverKey::g1_powers, kzg.rs:107-143; built once in ONNXProof::setup_prover), so this is setup work: one call
 * per key, ~0.2 s at 2^22 points.  window_bits = 0 lets the library choose (20 at 2^22 points).  Every MSM over Fr
 * scalars whose points lie inside the table then needs ceil(255 / c) instead of 20 point additions per scalar;
 * outputs are unchanged (same group elements).  ATLAS_MSM_TAB=0 in the environment ignores the table. */
int atlas_srs_precompute(atlas_srs_t s, size_t n_points, uint32_t window_bits);
/* The same table over the powers [first_point, first_point + n_points) only (n_points = 0: to the end).  Rank r of a sharded proof commits
 * the r-th point range of every long vector (atlas_prove_graph_sharded, atlas_hyperkzg_open_sharded: MSM by point range), so
 * it needs — and with this call holds — 1 / world of the table: first_point = r * len / world, n_points = len / world.  An MSM whose
 * points are not inside the range takes the variable-base path: the same group element, so the proof bytes do not depend on the range. */
int atlas_srs_precompute_range(atlas_srs_t s, size_t first_point, size_t n_points, uint32_t window_bits);
/* bytes of device memory the calling thread's runtime holds through the library's allocator: now, and at most since the last reset */
int atlas_device_memory_stats(size_t *in_use_bytes, size_t *peak_bytes, int reset_peak);
int atlas_srs_table_info(atlas_srs_t s, size_t *n_points, uint32_t *window_bits, uint32_t *levels);
/* VariableBaseMSM::msm / msm_field_elements over bases[offset .. offset+n)
 * (joltworks/src/msm/mod.rs:27-38,184-190; replaces the arkworks Pippenger call).
 * Fails with ATLAS_EINVAL ("KeyLengthError") when the SRS slice is shorter than n. */
int atlas_msm_fr(atlas_srs_t srs, size_t offset, const atlas_fr_t *scalars, size_t n,
                 atlas_g1_affine_t *out);
/* same, scalars = a device-resident polynomial: UnivariateKZG::commit_as_univariate
 * (kzg.rs:285-298) without a host copy; I32Scalars polynomials take the narrow-scalar plan */
int atlas_msm_poly(atlas_srs_t srs, size_t offset, atlas_poly_t poly, atlas_g1_affine_t *out);
/* VariableBaseMSM::msm for the narrow MultilinearPolynomial variants (joltworks/src/msm/mod.rs:
 * 38-181, 193-307: msm_u8 / msm_u16 / msm_u32 / msm_u64 and the sign-split I32 / I64 cases).
 * scalars = n host integers of the given kind.  The window plan follows the data's bit length;
 * negative scalars contribute negated digits (same group element as pos-MSM minus neg-MSM). */
enum { ATLAS_SCALAR_U8 = 0, ATLAS_SCALAR_U16 = 1, ATLAS_SCALAR_U32 = 2, ATLAS_SCALAR_U64 = 3,
       ATLAS_SCALAR_I32 = 4, ATLAS_SCALAR_I64 = 5 };
int atlas_msm_small(atlas_srs_t srs, size_t offset, const void *scalars, size_t n, int kind,
                    atlas_g1_affine_t *out);
/* sum of bases[indices[i]] — HyperKZG::commit_one_hot (hyperkzg/mod.rs:520-554): the caller
 * passes the flat indices k*T + t of the non-zero coefficients; replaces
 * jolt_optimizations::batch_g1_additions_multi */
int atlas_g1_sum_indexed(atlas_srs_t srs, const uint32_t *indices, size_t n, atlas_g1_affine_t *out);
/* HyperKZG::batch_commit_one_hot (hyperkzg/mod.rs:558-596): R one-hot polynomials, polynomial r over K[r] addresses
 * and T[r] cycles, nonzero_indices[r][t] = address of cycle t or negative for None (OneHotPolynomial::nonzero_indices,
 * poly/one_hot_polynomial.rs:21-40).  One upload of all index vectors, one launch for all R sums.  out[r] = commitment. */
int atlas_commit_one_hot_batch(atlas_srs_t srs, const int32_t *const *nonzero_indices, const size_t *K, const size_t *T,
                               size_t R, atlas_g1_affine_t *out);
/* The d RaD witness commitments of one lookup from its 2^log_T lookup indices on the device (generate_node_witnesses,
 * jolt-atlas-core/src/onnx_proof/witness.rs:136-200, + HyperKZG::batch_commit_one_hot, hyperkzg/mod.rs:558-596):
 * d = ceil(log_K / log_k_chunk) one-hot polynomials of 2^log_k_chunk addresses, chunk 0 most significant; out[d]. */
int atlas_commit_lookup_chunks(atlas_srs_t srs, const uint64_t *d_lookups, size_t log_T, size_t log_K, size_t log_k_chunk,
                               atlas_g1_affine_t *out);
/* The same for the lookups of a whole model graph (commit_witness_polynomials, prover.rs:71-87): the chunk polynomials of n_families
 * lookups in one launch and one synchronisation; out receives ceil(log_K_f / log_k_chunk) commitments per family, family after family. */
typedef struct { const uint64_t *d_lookups; size_t log_T, log_K; } atlas_lookup_family_t;
int atlas_commit_lookup_chunks_multi(atlas_srs_t srs, const atlas_lookup_family_t *families, size_t n_families, size_t log_k_chunk,
                                     atlas_g1_affine_t *out);
/* CommitmentScheme::batch_commit (commitment_scheme.rs:76-90 -> UnivariateKZG::commit_batch, kzg.rs:195-243):
 * commitments of n device-resident polynomials (LargeScalars or I32Scalars) against prefixes of the SRS. */
int atlas_commit_batch(atlas_srs_t srs, const atlas_poly_t *polys, size_t n, atlas_g1_affine_t *out);
/* HyperKZG::open (joltworks/src/poly/commitment/hyperkzg/mod.rs:400-447, the body of
 * CommitmentScheme::prove, commitment_scheme.rs:93-108): ell-1 folds, their commitments,
 * the 3*ell univariate evaluations, the batched witness polynomials and their 3
 * commitments, with the transcript interaction of the reference.  `poly` = LargeScalars of
 * length 2^ell (not consumed); `point` = the ell opening-point challenges (raw u128).
 *   com : ell-1 points    w : 3 points    v : 3*ell Fr, v[i*ell + j] = Pi_j(u_i)
 * (HyperKZGProof { com, w, v }, hyperkzg/mod.rs:168-173). */
int atlas_hyperkzg_open(atlas_srs_t srs, atlas_poly_t poly, const atlas_u128_t *point, size_t ell,
                        atlas_transcript_t *transcript, atlas_g1_affine_t *com, atlas_g1_affine_t *w,
                        atlas_fr_t *v);
/* ONNXProof::prove_reduced_openings (jolt-atlas-core/src/onnx_proof/prover.rs:141-176):
 * prepare_for_sumcheck + prove_batch_opening_sumcheck + finalize_batch_opening_sumcheck
 * (joltworks/src/poly/opening_proof.rs:447-532, 611-643) + build_materialized_rlc + PCS::prove.
 * `openings` = the accumulator's sumchecks in BTreeMap<CommittedPoly> order (their claims were
 * appended to the transcript when they were registered, opening_proof.rs:281, 336).  Outputs: the
 * batched sumcheck (rows of 3 Fr, n_coeffs, challenges = r_sumcheck, max_rounds), sumcheck_claims
 * (P_i(r_sumcheck) per opening), and the HyperKZG proof of the joint polynomial at r_sumcheck
 * (com: max_rounds-1 points, w: 3 points, v: 3*max_rounds Fr).  The polynomials are not consumed. */
typedef struct {
    int kind;                 /* 0 = dense (LargeScalars / I32Scalars), 1 = one-hot */
    atlas_poly_t poly;        /* dense: the committed polynomial, length 2^n */
    size_t n;                 /* dense: number of variables */
    const int32_t *k;         /* one-hot: nonzero_indices (T host int32, negative = None) */
    size_t log_K, log_T;      /* one-hot */
    const atlas_fr_t *point;  /* dense: n Fr; one-hot: r_address (log_K) followed by r_cycle (log_T) */
    atlas_fr_t claim;         /* the opening claim P(point) */
    /* one-hot, instead of `k` (k == NULL): the T device-resident lookup indices the chunk polynomial was committed from
     * (atlas_commit_lookup_chunks) and its shift: nonzero index = (d_lookups[t] >> chunk_shift) & (2^log_K - 1)
     * (OneHotParams::lookup_index_chunk, config.rs:73-75).  When EVERY one-hot opening of the call is given this way (log_K <= 4)
     * they are stepped together: one fold / bind launch and one synchronisation per round for all of them. */
    const uint64_t *d_lookups;
    size_t chunk_shift;
} atlas_opening_t;
int atlas_prove_reduced_openings(const atlas_opening_t *openings, size_t n_openings, atlas_srs_t srs,
                                 atlas_transcript_t *transcript, atlas_fr_t *sumcheck_rows, uint32_t *n_coeffs,
                                 atlas_u128_t *challenges, size_t *max_rounds_out, atlas_fr_t *sumcheck_claims,
                                 atlas_g1_affine_t *com, atlas_g1_affine_t *w, atlas_fr_t *v);

/* ProverOpeningAccumulator (joltworks/src/poly/opening_proof.rs:195-643): claim bookkeeping between the operator provers
 * and prove_reduced_openings.  Keys are caller-chosen u64 whose numeric order is the Ord of the Rust key
 * (opening_key <-> OpeningId, poly_key <-> CommittedPoly; types.rs:27-129), so iteration order equals the BTreeMaps'.
 *   append_dense   (:265-315): transcript.append_scalar(claim); records the opening and the polynomial's reduction instance
 *   append_sparse  (:317-372): the R one-hot polynomials of one lookup: one claim appended each, point = r_address || r_cycle
 *   append_virtual (:374-409): transcript.append_scalar(claim); records the opening
 *   get            (:199-243): point and claim recorded under an OpeningId (ATLAS_ESTATE if absent: the reference panics)
 *   prove_reduced_openings: atlas_prove_reduced_openings over the committed openings in CommittedPoly order
 * Dense polynomials are borrowed (alive until the proof), one-hot index rows are copied. */
typedef struct atlas_accumulator *atlas_accumulator_t;
int atlas_accumulator_new(atlas_accumulator_t *out);
int atlas_accumulator_free(atlas_accumulator_t a);
int atlas_accumulator_append_dense(atlas_accumulator_t a, atlas_transcript_t *t, uint64_t opening_key, uint64_t poly_key,
                                   atlas_poly_t poly, const atlas_fr_t *point, size_t n, const atlas_fr_t *claim);
int atlas_accumulator_append_sparse(atlas_accumulator_t a, atlas_transcript_t *t, const uint64_t *opening_keys,
                                    const uint64_t *poly_keys, const int32_t *const *nonzero_indices, size_t R,
                                    size_t log_K, size_t log_T, const atlas_fr_t *r_address, const atlas_fr_t *r_cycle,
                                    const atlas_fr_t *claims);
int atlas_accumulator_append_virtual(atlas_accumulator_t a, atlas_transcript_t *t, uint64_t opening_key,
                                     const atlas_fr_t *point, size_t n, const atlas_fr_t *claim);
int atlas_accumulator_get(atlas_accumulator_t a, uint64_t opening_key, atlas_fr_t *point_out, size_t cap, size_t *n_out,
                          atlas_fr_t *claim_out);
size_t atlas_accumulator_num_committed(atlas_accumulator_t a);
size_t atlas_accumulator_max_rounds(atlas_accumulator_t a);
int atlas_accumulator_prove_reduced_openings(atlas_accumulator_t a, atlas_srs_t srs, atlas_transcript_t *transcript,
                                             atlas_fr_t *sumcheck_rows, uint32_t *n_coeffs, atlas_u128_t *challenges,
                                             size_t *max_rounds_out, atlas_fr_t *sumcheck_claims, atlas_g1_affine_t *com,
                                             atlas_g1_affine_t *w, atlas_fr_t *v);

/* ---- one operator node, composed as the reference composes it: the fused-rescale Einsum mk,kn->mn
 *      (jolt-atlas-core/src/onnx_proof/ops/einsum/mod.rs:71-115 + fused_rebase.rs:215-285 + clamp_lookups/mod.rs:264-309):
 * witness on the device (acc = A B in i64, rescaled = acc >> S, R = acc mod 2^S, output = SatClamp_i32(rescaled):
 * try_rebase_intermediates, fused_rebase.rs:110-128), then remainder advice, clamp PS-Shout (Execution), one-hot checks
 * (RaOneHotChecks), the contraction sumcheck (EinsumMatmul), the remainder range check (RangeCheck) and its one-hot checks
 * (RescaleRemainderRaChecks), every accumulator append mirrored on the transcript in the reference's order.
 *   d_A (m x k), d_B (k x n): Tensor<i32> in HBM; r_node_output: the log2(m n) entries of the node's reduced output opening;
 *   output_claim: output(r_node_output), or NULL to evaluate it here; proofs: the five SumcheckInstanceProofs ark-serialized
 *   back to back, proof_lens[i] bytes each; claims: the scalars appended to the accumulator, in order; d_output (optional):
 *   receives the node output; stage_ms (optional, 6 doubles): witness, Execution, RaOneHotChecks, EinsumMatmul, RangeCheck,
 *   RescaleRemainderRaChecks wall clock. */
typedef struct { size_t m, k, n; uint32_t scale_bits; } atlas_einsum_node_t;
int atlas_prove_einsum_node(const atlas_einsum_node_t *node, const int32_t *d_A, const int32_t *d_B,
                            const atlas_fr_t *r_node_output, const atlas_fr_t *output_claim, atlas_transcript_t *t,
                            uint8_t *proofs, size_t cap, size_t *proofs_len, size_t proof_lens[5], atlas_fr_t *claims,
                            size_t claims_cap, size_t *n_claims, int32_t *d_output, double *stage_ms);

/* Transcript::append_point / append_points (blake2b.rs:166-195), host side */
int atlas_transcript_append_point(atlas_transcript_t *t, const atlas_g1_affine_t *p);
int atlas_transcript_append_points(atlas_transcript_t *t, const atlas_g1_affine_t *p, size_t n);

/* ---- wire formats: ark-serialize 0.5 `serialize_compressed` images, so that proofs and SRS files can be
 *      exchanged with the reference (jolt-atlas-core/src/onnx_proof/proof_serialization.rs:29-224 composes these) -------
 * Fr: 32 B LE canonical.  G1Affine compressed: x 32 B LE, last byte bit 7 = y is the larger of (y, -y), bit 6 = infinity;
 * uncompressed: x || y with the flags on y's last byte.  Vec<T>: u64 LE length + items.  Deserializers validate like
 * Validate::Yes (range, on-curve; BN254 G1 has cofactor 1) and return ATLAS_EINVAL on malformed input. */
int atlas_fr_to_bytes(const atlas_fr_t *a, uint8_t out[32]);
int atlas_fr_from_bytes(const uint8_t in[32], atlas_fr_t *out);
int atlas_g1_to_bytes_compressed(const atlas_g1_affine_t *p, uint8_t out[32]);
int atlas_g1_from_bytes_compressed(const uint8_t in[32], atlas_g1_affine_t *out);
int atlas_g1_to_bytes_uncompressed(const atlas_g1_affine_t *p, uint8_t out[64]);
/* SumcheckInstanceProof { compressed_polys: Vec<CompressedUniPoly> } (sumcheck.rs:624-640; unipoly.rs:27-30):
 * rows[i*row_stride ..] = the n_coeffs[i] coefficients-except-linear of round i.  out == NULL: size query into *len. */
int atlas_sumcheck_proof_serialize(const atlas_fr_t *rows, size_t row_stride, const uint32_t *n_coeffs, size_t n_rounds,
                                   uint8_t *out, size_t cap, size_t *len);
int atlas_sumcheck_proof_deserialize(const uint8_t *in, size_t len, atlas_fr_t *rows, size_t row_stride,
                                     uint32_t *n_coeffs, size_t max_rounds, size_t *n_rounds, size_t *consumed);
/* HyperKZGProof { com, w, v } (hyperkzg/mod.rs:172-177); 368 bytes at ell = 2 (hyperkzg/tests.rs:107-109) */
int atlas_hyperkzg_proof_serialize(const atlas_g1_affine_t *com, size_t n_com, const atlas_g1_affine_t *w,
                                   const atlas_fr_t *v, size_t ell, uint8_t *out, size_t cap, size_t *len);
int atlas_hyperkzg_proof_deserialize(const uint8_t *in, size_t len, atlas_g1_affine_t *com, size_t cap_com, size_t *n_com,
                                     atlas_g1_affine_t *w, atlas_fr_t *v, size_t cap_ell, size_t *ell, size_t *consumed);
/* HyperKZGSRS::load_from_file / save_to_file (hyperkzg/mod.rs:60-95): SRS { g1_powers, g2_powers, g_products }
 * compressed (kzg.rs:18-23).  load keeps the first max_points (0 = all) g1_powers, decompressed on the device;
 * save writes g1_powers with empty g2_powers / g_products (the prover half). */
int atlas_srs_load_file(const char *path, size_t max_points, atlas_srs_t *out);
int atlas_srs_save_file(atlas_srs_t srs, const char *path);

/* ---- one instance / one MSM sharded over the GPUs of a node (one process per GPU) -------
 * Rank g of `world` holds the strided shard L_g[k] = L[k*world + g] of each operand (the
 * HighToLow pairs (i, i + len/2) stay on one rank).  Per round: local partial message ->
 * all-gather of world*2 Fr (RCCL over xGMI, done by the caller with torch.distributed) ->
 * the same transcript step on every rank.  Degree-2 LargeScalars instances. */
int atlas_dot_shard_begin(atlas_dot_prover_t p, const atlas_fr_t *input_claim,
                          const atlas_transcript_t *transcript);
int atlas_dot_shard_local_message(atlas_dot_prover_t p, atlas_fr_t *out2);        /* this rank's 2 partial evals */
int atlas_dot_shard_round(atlas_dot_prover_t p, const atlas_fr_t *gathered, size_t world); /* world*2, rank-major */
int atlas_dot_shard_local_final(atlas_dot_prover_t p, atlas_fr_t *out2);          /* (L_g, R_g) after the local rounds */
int atlas_dot_shard_finish(atlas_dot_prover_t p, const atlas_fr_t *gathered_lr, size_t world,
                           atlas_transcript_t *transcript, atlas_fr_t *compressed_polys,
                           atlas_u128_t *challenges, atlas_fr_t final_claims[3]);
int atlas_fr_sum(const atlas_fr_t *v, size_t n, atlas_fr_t *out);                 /* host: sum of per-rank claims */
/* The same instance in ONE call per rank, transcript on every rank's host thread: the launches of all local rounds are
 * enqueued up front (round channel), the per-round exchange of the ranks' partial sums (64 B each) goes through a POSIX
 * shared-memory board instead of a collective — the round is latency-bound and the sums are in host memory already —
 * and the last log2(world) rounds are host arithmetic.  Every rank returns the same proof.
 *   atlas_shard_group_open: `name` = a shm name unique to the job ("/atlas_<port>_<run>"), the same on every rank;
 *   rank 0 creates the board, the call returns when all `world` ranks have joined.
 *   atlas_shard_allgather: all[r*n_bytes ..] = rank r's record (<= 496 B); used for the claim and for MSM partials.
 *   atlas_sumcheck_prove_dot_sharded: p = this rank's strided shard (len / world coefficients), input_claim = the claim
 *   of the whole instance; outputs sized for num_rounds(p) + log2(world) rounds. */
typedef struct atlas_shard_group *atlas_shard_group_t;
int atlas_shard_group_open(const char *name, int world, int rank, atlas_shard_group_t *out);
int atlas_shard_group_close(atlas_shard_group_t grp);
/* The board's failure handshake.  A rank whose part of a sharded call fails calls atlas_shard_fail_exchange(grp, code) INSTEAD of the
 * exchange the other ranks are about to make (the sharded entry points of this library do so themselves): the others' exchange returns
 * ATLAS_ENODEV at once instead of after the board's timeout, atlas_shard_remote_failed tells them which rank gave up and why (rank -1:
 * nobody did — the exchange timed out), and every rank has made the same number of exchanges, so the group stays usable. */
int atlas_shard_fail_exchange(atlas_shard_group_t grp, int code);
int atlas_shard_remote_failed(atlas_shard_group_t grp, int *rank, int *code);
/* How long a wait may last before a proof is given up, in seconds (<= 0 keeps a setting; defaults 2 / 10 / 30, or ATLAS_DEVICE_WAIT_S /
 * ATLAS_HOST_WAIT_S / ATLAS_BOARD_WAIT_S at atlas_init / atlas_shard_group_open):
 *   device_wait_s  a launch's wait for a round's challenge (a RUNTIME word next to the channel's abort flag, read by every polling launch);
 *   host_wait_s    the proving thread's wait for the sums / tables a launch mails;
 *   board_wait_s   a rank's wait for the other ranks' records (groups opened afterwards).
 * The first two belong to the calling thread's runtime.  A device shared by several ranks' processes (time-sliced) needs more than the
 * defaults: atlas_prove_graph_sharded multiplies the first two by its world size for the length of the call. */
int atlas_set_timeouts(double device_wait_s, double host_wait_s, double board_wait_s);
int atlas_get_timeouts(double *device_wait_s, double *host_wait_s, double *board_wait_s);
/* HyperKZG::open (hyperkzg/mod.rs:400-447) with its commitments split by point range over the group's ranks: every rank
 * passes the whole polynomial and its copy of the SRS, runs the same transcript and returns the same HyperKZGProof as atlas_hyperkzg_open
 * (same bytes); the MSMs — Pi_1.. and the three witness polynomials, ~95 % of the open — are 1/world each, two exchanges of partial points. */
/* One element-wise operator sumcheck (Mul, Add, Sub, Square, Iff, Div, Rsqrt, Cube: LowToHigh over the Gruen split-eq, mul.rs:160-199)
 * sharded by contiguous blocks: `inst` = atlas_elementwise_new over this rank's block of 2^(n - log2 world) coefficients per
 * operand with the LOW coordinates r_node_output[log2 world ..]; r_high = the leading log2 world coordinates.  One 32..96-byte record per
 * rank and round through the board, then the operands' world final values once; every rank gets the whole proof (n rows), the challenges
 * and the operands' final claims — the same bytes as the unsharded instance. */
int atlas_elementwise_prove_sharded(atlas_instance_t inst, atlas_shard_group_t grp, const atlas_fr_t *r_high, const atlas_fr_t *input_claim,
                                    atlas_transcript_t *transcript, atlas_fr_t *compressed, size_t row_stride, uint32_t *n_coeffs,
                                    atlas_u128_t *challenges, atlas_fr_t *finals, size_t finals_cap, size_t *n_finals);
int atlas_hyperkzg_open_sharded(atlas_srs_t srs, atlas_shard_group_t group, atlas_poly_t poly, const atlas_u128_t *point, size_t ell,
                                atlas_transcript_t *transcript, atlas_g1_affine_t *com, atlas_g1_affine_t *w, atlas_fr_t *v);
int atlas_shard_allgather(atlas_shard_group_t grp, const void *mine, size_t n_bytes, void *all);
int atlas_sumcheck_prove_dot_sharded(atlas_dot_prover_t p, atlas_shard_group_t grp, const atlas_fr_t *input_claim,
                                     atlas_transcript_t *transcript, atlas_fr_t *compressed_polys,
                                     atlas_u128_t *challenges, atlas_fr_t final_claims[3]);
/* host: sum of the per-rank partial MSM results of a point-range sharded commitment */
int atlas_g1_sum_affine(const atlas_g1_affine_t *pts, size_t n, atlas_g1_affine_t *out);

/* ---- measurement: HIP-event time of the launches issued by the last
 *      atlas_sumcheck_prove_dot / atlas_msm_* call, on the library stream ------------- */
typedef struct {
    double total_ms;          /* first launch -> last launch complete */
    double pass_ms;           /* sum over the data passes (eval + fused bind/eval) */
    double fs_ms;             /* sum over transcript (fs) launches + tail */
    uint64_t pass_bytes;      /* algorithmic bytes moved by the data passes */
    uint32_t n_pass, n_fs;
} atlas_timing_t;
int atlas_set_timing(int enabled);   /* per-launch events; off by default */
/* chip-wide 32x32->64 multiply-add rate (v_mad_u64_u32), measured now: the ceiling of the MSM bucket accumulation */
int atlas_measure_mad_peak(double *mads_per_s);
int atlas_last_timing(atlas_timing_t *out);

/* Mul::prove with fused rescaling (ops/mul.rs through impl_fused_rescale_proof_api, ops/mod.rs:569-612) for one node:
 * out = (left * right) >> scale_bits element-wise over 2^log_T i32 values on the device.  Same five proofs and claim order as
 * atlas_prove_einsum_node, with MulProver (Gruen split-eq of r_node_output, degree 3) as the operator's sumcheck. */
int atlas_prove_mul_node(const int32_t *d_left, const int32_t *d_right, size_t log_T, uint32_t scale_bits,
                         const atlas_fr_t *r_node_output, const atlas_fr_t *output_claim, atlas_transcript_t *transcript,
                         uint8_t *proofs, size_t cap, size_t *proofs_len, size_t proof_lens[5], atlas_fr_t *claims,
                         size_t claims_cap, size_t *n_claims, int32_t *d_output, double *stage_ms);

/* The operators proved by one sumcheck (impl_standard_sumcheck_proof_api, ops/mod.rs:505-560): op = ATLAS_EW_MUL for And
 * (ops/and.rs: MulProver without rescaling, 2 operands) or ATLAS_EW_IFF (ops/iff.rs, 3 operands: mask, a, b).  output_claim =
 * the node-output opening claim (the sumcheck's input claim).  One proof; claims: the operand claims in input order. */
int atlas_prove_standard_node(int op, const int32_t *const *d_operands, size_t n_operands, size_t log_T, const atlas_fr_t *r_node_output,
                              const atlas_fr_t *output_claim, atlas_transcript_t *transcript, uint8_t *proofs, size_t cap,
                              size_t *proofs_len, size_t proof_lens[1], atlas_fr_t *claims, size_t claims_cap, size_t *n_claims);
/* Square::prove with fused rescaling (ops/square.rs): out = (x * x) >> scale_bits; as atlas_prove_mul_node with one operand. */
int atlas_prove_square_node(const int32_t *d_input, size_t log_T, uint32_t scale_bits, const atlas_fr_t *r_node_output,
                            const atlas_fr_t *output_claim, atlas_transcript_t *transcript, uint8_t *proofs, size_t cap,
                            size_t *proofs_len, size_t proof_lens[5], atlas_fr_t *claims, size_t claims_cap, size_t *n_claims,
                            int32_t *d_output, double *stage_ms);
/* Add::prove / Sub::prove (ops/add.rs:70-105, ops/sub.rs) for one node: prove_clamp_lookup over the i64 accumulation
 * left +- right (clamp_lookups/mod.rs:264-309) and the operand tie left(r), right(r).  Two proofs (Execution,
 * RaOneHotChecks); claims: acc, ra, the 3 d one-hot claims, left, right; stage_ms[3] optional. */
int atlas_prove_addsub_node(const int32_t *d_left, const int32_t *d_right, size_t log_T, int subtract, const atlas_fr_t *r_node_output,
                            const atlas_fr_t *output_claim, atlas_transcript_t *transcript, uint8_t *proofs, size_t cap,
                            size_t *proofs_len, size_t proof_lens[2], atlas_fr_t *claims, size_t claims_cap, size_t *n_claims,
                            int32_t *d_output, double *stage_ms);

/* ReLU::prove (jolt-atlas-core/src/onnx_proof/ops/relu.rs:22-70) for one node: OpLookupProvider::read_raf_prove over
 * ReluTable<32> (op_lookups/mod.rs:250-267), Sumcheck::prove with its ra opening, ra_onehot_provers + BatchedSumcheck::prove
 * (shout.rs:399-466).  d_input: the operand, 2^log_T i32 on the device; r_node_output: the node-output opening point;
 * output_claim: relu(x)~(r) or NULL (evaluated here); d_output: optional device buffer for the output tensor.  Two proofs
 * (Execution, RaOneHotChecks) ark-serialized back to back, claims in accumulator order, stage_ms[3] optional. */
int atlas_prove_relu_node(const int32_t *d_input, size_t log_T, const atlas_fr_t *r_node_output, const atlas_fr_t *output_claim,
                          atlas_transcript_t *transcript, uint8_t *proofs, size_t cap, size_t *proofs_len, size_t proof_lens[2],
                          atlas_fr_t *claims, size_t claims_cap, size_t *n_claims, int32_t *d_output, double *stage_ms);

/* ---- verification (host arithmetic; the reference's verifier is CPU code as well) ----------------------------------
 * SumcheckInstanceProof::verify (subprotocols/sumcheck.rs:653-686): replays the transcript over the compressed round
 * polynomials (row i: n_coeffs[i] coefficients, linear term omitted) and returns e = g_v(r_v) and the raw challenges.
 * ATLAS_EVERIFY when a round polynomial exceeds degree_bound (ProofVerifyError::InvalidInputLength).  As in the reference
 * the last check, e == oracle(r), is the caller's. */
int atlas_sumcheck_proof_verify(const atlas_fr_t *compressed, size_t row_stride, const uint32_t *n_coeffs, size_t n_rounds,
                                size_t degree_bound, const atlas_fr_t *claim, atlas_transcript_t *transcript,
                                atlas_fr_t *final_claim, atlas_u128_t *challenges);
/* BatchedSumcheck::verify (sumcheck.rs:187-259) up to the output claim: input claims appended, one batching coefficient
 * per instance drawn, claims combined with mul_pow_2(max_rounds - num_rounds[i]), then proof.verify with the largest degree.
 * Instance i uses the last num_rounds[i] challenges.  The caller runs each instance's cache_openings (transcript!) and
 * expected_output_claim, then atlas_batched_sumcheck_check compares (ATLAS_EVERIFY = SumcheckVerificationError). */
int atlas_batched_sumcheck_verify(const atlas_fr_t *compressed, size_t row_stride, const uint32_t *n_coeffs, size_t max_rounds,
                                  const atlas_fr_t *input_claims, const size_t *num_rounds, const size_t *degrees,
                                  size_t n_instances, atlas_transcript_t *transcript, atlas_fr_t *batching_coeffs,
                                  atlas_fr_t *output_claim, atlas_u128_t *challenges);
int atlas_batched_sumcheck_check(const atlas_fr_t *batching_coeffs, const atlas_fr_t *expected_output_claims, size_t n_instances,
                                 const atlas_fr_t *output_claim);

/* ---- HyperKZG::verify (joltworks/src/poly/commitment/hyperkzg/mod.rs:451-509 verify_inner, :283-366 kzg_verify_batch) on the
 *      host: the transcript replay of atlas_hyperkzg_open, the consistency of (Y, ypos, yneg), the batched KZG check
 *      e(L, vk.g2) = e(R, vk.beta_g2) through a BN254 pairing (the arithmetic lives in arkworks, external to the reference;
 *      csrc/host_pairing.hpp states it).  G2 points: x = x[0] + x[1] u, y likewise, Montgomery Fq limbs (the ark_bn254 G2Affine image).
 *      ATLAS_OK = accept, ATLAS_EVERIFY = ProofVerifyError::InternalError. ---- */
typedef struct { atlas_fq_t x[2], y[2]; uint64_t infinity; } atlas_g2_affine_t;
typedef struct { atlas_g1_affine_t g1; atlas_g2_affine_t g2, beta_g2; } atlas_hyperkzg_vk_t;      /* HyperKZGVerifierKey { kzg_vk } (kzg.rs:146-166) */
int atlas_g2_generator(atlas_g2_affine_t *out);
int atlas_g2_mul(const atlas_g2_affine_t *p, const atlas_fr_t *k, atlas_g2_affine_t *out);
/* Pairing::multi_pairing(p, q).is_zero(): is prod_i e(p_i, q_i) the identity of G_T? */
int atlas_pairing_check(const atlas_g1_affine_t *p, const atlas_g2_affine_t *q, size_t n, int *is_one);
/* the verifier key of an SRS generated from a known trapdoor (atlas_srs_generate: tests and benches): g1 = the SRS's first power,
 * g2 = the G2 generator, beta_g2 = tau g2.  A key from a ceremony is filled in by the caller from its own g2 powers. */
int atlas_hyperkzg_vk_from_trapdoor(const atlas_fr_t *tau, const atlas_g1_affine_t *g1, atlas_hyperkzg_vk_t *out);
/* commitment = C, point = the ell opening-point challenges (raw u128, as atlas_hyperkzg_open takes them), y = the claimed P(point),
 * com / w / v = HyperKZGProof { com (ell - 1), w (3), v (3 x ell) }; the transcript advances exactly as the prover's did */
int atlas_hyperkzg_verify(const atlas_hyperkzg_vk_t *vk, const atlas_g1_affine_t *commitment, const atlas_u128_t *point, size_t ell,
                          const atlas_fr_t *y, const atlas_g1_affine_t *com, const atlas_g1_affine_t *w, const atlas_fr_t *v,
                          atlas_transcript_t *transcript);

/* ---- the whole proof: ONNXProof::prove over a model graph resident in the library ---------
 * The graph is the tracer's ComputationGraph (atlas-onnx-tracer/src/node/mod.rs:12-24: idx, operator, inputs, output_dims) in the
 * tracer's operator vocabulary (atlas-onnx-tracer/src/ops/mod.rs:117-155).  Every dimension must be a power of two (the reference
 * pads each dimension to the next power of two before it builds MLEs; a graph description pads its shapes up front).
 * params / shape per operator:
 *   INPUT, CONSTANT (constant = the tensor), IDENTITY, ADD, SUB, AND, IFF (mask, a, b), RELU, RESHAPE: none
 *   MUL, SQUARE, CUBE: params[0] = scale (the fused rebase is by `scale` bits, 2 * scale for CUBE; fused_rebase.rs:71-79)
 *   EINSUM: params[0] = ATLAS_EINSUM_* layout, params[1] = scale; shape = the layout's dims as atlas_einsum_fold takes them
 *   MOVEAXIS: params[0] = source, params[1] = destination;  SLICE: params = axis, start, end;  BROADCAST: none (dims = target)
 * Nodes are added in increasing index order; inputs name earlier nodes (the tracer's topological numbering). */
enum { ATLAS_OP_INPUT = 0, ATLAS_OP_CONSTANT, ATLAS_OP_IDENTITY, ATLAS_OP_ADD, ATLAS_OP_SUB, ATLAS_OP_MUL, ATLAS_OP_SQUARE, ATLAS_OP_CUBE,
       ATLAS_OP_AND, ATLAS_OP_IFF, ATLAS_OP_RELU, ATLAS_OP_EINSUM, ATLAS_OP_RESHAPE, ATLAS_OP_MOVEAXIS, ATLAS_OP_BROADCAST, ATLAS_OP_SLICE,
       ATLAS_OP_CONCAT, ATLAS_OP_SUM, ATLAS_OP_SCALAR_CONST_DIV, ATLAS_OP_DIV, ATLAS_OP_MEAN_OF_SQUARES, ATLAS_OP_RSQRT, ATLAS_OP_SOFTMAX,
       ATLAS_OP_TANH, ATLAS_OP_GATHER_LARGE, ATLAS_OP_GATHER_SMALL, ATLAS_OP_ERF, ATLAS_OP_SIGMOID,
       ATLAS_OP_NEG, ATLAS_OP_IS_NAN, ATLAS_OP_CLAMP, ATLAS_OP_SIN, ATLAS_OP_COS };
typedef struct atlas_graph *atlas_graph_t;
int atlas_graph_new(atlas_graph_t *out);
int atlas_graph_free(atlas_graph_t g);
int atlas_graph_add_node(atlas_graph_t g, size_t idx, int op, const size_t *inputs, size_t n_inputs, const size_t *dims, size_t n_dims,
                         const int64_t *params, size_t n_params, const size_t *shape, size_t n_shape, const int32_t *constant);
int atlas_graph_set_outputs(atlas_graph_t g, const size_t *idx, size_t n);
size_t atlas_graph_num_nodes(atlas_graph_t g);
/* Model::trace (atlas-onnx-tracer/src/model/trace.rs:8-21) on the device: one host tensor per Input node (ascending index);
 * every node output stays in HBM together with the witness the prover derives from it (i64 accumulations, remainders,
 * lookup indices: generate_node_witnesses, jolt-atlas-core/src/onnx_proof/witness.rs:136-200). */
int atlas_graph_trace(atlas_graph_t g, const int32_t *const *inputs, size_t n_inputs);
/* a traced node's output tensor (host_out == NULL: length query) */
int atlas_graph_node_output(atlas_graph_t g, size_t idx, int32_t *host_out, size_t cap, size_t *len);
/* ONNXProof::prove (jolt-atlas-core/src/onnx_proof/mod.rs:153-200): trace, append_inputs_to_transcript, commit_witness_polynomials,
 * output_claim, iop (NodeEvalReduction + the operator provers in reverse node order), prove_reduced_openings, finalize.
 * proof receives the ark-serialized ONNXProof (proof_serialization.rs:200-224: opening claims, proofs by ProofId, commitments,
 * eval-reduction proofs, the reduced opening proof); proof == NULL: size query.  final_transcript (optional): the prover's
 * transcript after the last append.  timing (optional): the stage split of the reference's tracing spans. */
typedef struct {
    double trace_ms, commit_ms, iop_ms, reduction_ms, hyperkzg_ms, total_ms;   /* reduction_ms excludes hyperkzg_ms */
    size_t n_nodes, n_committed, n_sumchecks;
} atlas_graph_timing_t;
int atlas_prove_graph(atlas_graph_t g, atlas_srs_t srs, const int32_t *const *inputs, size_t n_inputs, uint8_t *proof, size_t cap,
                      size_t *proof_len, atlas_transcript_t *final_transcript, atlas_graph_timing_t *timing);
/* ONNXProof::prove over the GPUs of a node, one process per GPU (BASELINE config 4): every rank of `grp` holds the model, the inputs
 * and the SRS and makes this call with the same arguments.  The trace and the IOP run on every rank (a serial Fiat-Shamir chain);
 * commit_witness_polynomials is split by polynomial range (jolt-atlas-core/src/onnx_proof/prover.rs:71-87) and the commitment groups of
 * HyperKZG::open over the joint polynomial by point range (prover.rs:141-176; joltworks/src/poly/commitment/hyperkzg/mod.rs:400-447);
 * partial results cross the shared-memory board of the group.  Every rank returns the same bytes as atlas_prove_graph on one GPU. */
int atlas_prove_graph_sharded(atlas_graph_t g, atlas_srs_t srs, atlas_shard_group_t grp, const int32_t *const *inputs, size_t n_inputs,
                              uint8_t *proof, size_t cap, size_t *proof_len, atlas_transcript_t *final_transcript,
                              atlas_graph_timing_t *timing);

/* EvalReductionInstance::verify (joltworks/src/subprotocols/evaluation_reduction.rs:150-210): the verifier's half of
 * atlas_eval_reduction_prove.  h = the proof's polynomial (ignored for N = 1).  ATLAS_EVERIFY = InvalidOpeningProof. */
int atlas_eval_reduction_verify(const atlas_fr_t *points, const atlas_fr_t *claims, size_t N, size_t n, const atlas_fr_t *h, size_t h_len,
                                atlas_transcript_t *transcript, atlas_fr_t *r_out, atlas_fr_t *claim_out);
/* ONNXProof::verify (jolt-atlas-core/src/onnx_proof/mod.rs:207-241, verifier.rs) for a graph built with atlas_graph_add_node: the
 * serialized ONNXProof of atlas_prove_graph against the model (its constants), the inputs and the claimed output tensor
 * (ModelExecutionIO) under a HyperKZG verifier key.  Host arithmetic; the device evaluates the public tensors and adds up the joint
 * commitment.  Nothing of a prover's trace is read.  ATLAS_OK = accept; ATLAS_EVERIFY = ProofVerifyError; ATLAS_EINVAL = the graph holds
 * an operator without a verifier composition (composed: every operator atlas_prove_graph composes). */
int atlas_verify_graph(atlas_graph_t g, const atlas_hyperkzg_vk_t *vk, const int32_t *const *inputs, size_t n_inputs, const int32_t *output,
                       size_t output_len, const uint8_t *proof, size_t proof_len, atlas_transcript_t *final_transcript);

#ifdef __cplusplus
}
#endif
#endif
