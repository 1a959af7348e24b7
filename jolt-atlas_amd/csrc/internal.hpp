// Entry points shared between translation units of the library that are not part of the C-ABI.
#pragma once
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_field.hpp"

// compute_ra_evals (shout.rs:550-598) with the d x 2^log_k_chunk tables returned on the host (shout.hip)
int atlas_rt_shout_ra_evals_host(const uint64_t* lookup_indices, size_t T, size_t log_K, size_t log_k_chunk, atlas_poly_t eq_r_cycle,
                                 std::vector<atlas_host::Fr>& G);
// wall clock of the HyperKZG::open inside the last atlas_prove_reduced_openings call (reduced_openings.hip)
double atlas_rt_last_hyperkzg_ms();
// atlas_prove_reduced_openings with the ranks of a sharded whole proof (NULL: one GPU); reduced_openings.hip
int atlas_rt_prove_reduced_openings(const atlas_opening_t* openings, size_t n_open, atlas_srs_t srs, atlas_transcript_t* transcript, atlas_fr_t* sumcheck_rows,
                                    uint32_t* n_coeffs, atlas_u128_t* challenges, size_t* max_rounds_out, atlas_fr_t* sumcheck_claims, atlas_g1_affine_t* com,
                                    atlas_g1_affine_t* w, atlas_fr_t* v, atlas_shard_group_t sh);

// compute_ra_evals in two halves (shout.hip): the launch — word sums into a pinned box, no wait — and the reduction on the host once the caller
// has waited for the library stream anyway (a node starts with such a wait: its witness openings).  finish(wait = true) waits itself.
struct atlas_rt_ra_ticket;
int atlas_rt_shout_ra_evals_launch(const uint64_t* lookup_indices, size_t T, size_t log_K, size_t log_k_chunk, atlas_poly_t eq_r_cycle, atlas_rt_ra_ticket** out);
int atlas_rt_shout_ra_evals_finish(atlas_rt_ra_ticket* t, bool wait, std::vector<atlas_host::Fr>& G);
void atlas_rt_shout_ra_evals_drop(atlas_rt_ra_ticket* t);
// the prefix-suffix constructors over an eq table the caller holds (EqPolynomial::evals(r_node_output), T Fr on the device; it must outlive
// the instance): psshout.hip
int atlas_rt_ps_shout_clamp_new(const uint64_t* lookup_indices, size_t log_T, size_t xlen, size_t bound, int symmetric, const atlas_fr_t* r_node_output,
                                const atlas_fr_t* gamma, atlas_poly_t eq_shared, atlas_instance_t* out, bool defer = false);
int atlas_rt_identity_range_check_new(const uint64_t* lookup_indices, size_t log_T, size_t log_K, size_t phases, const atlas_fr_t* r_node_output,
                                      atlas_poly_t eq_shared, atlas_instance_t* out, bool defer = false);
// defer = true: the launches of the constructor only (no wait for the device) — for an instance built AHEAD of its proof, under a wait its node
// makes anyway (node_flow.hip.h: NodePre); gamma may be null then and is handed over with atlas_rt_ps_set_gamma once it has been drawn
int atlas_rt_ps_set_gamma(atlas_instance_t inst, const atlas_fr_t* gamma);
// MultilinearPolynomial::evaluate of <= 3 polynomials against a full eq table on the device (spliteq.hip); waits for the library stream
int atlas_rt_evaluate_with_eq(const atlas_poly_t* polys, size_t count, atlas_poly_t eq_full, atlas_fr_t* out);
int atlas_rt_eval_event_record(atlas_poly_t eq_full);      // spliteq.hip: the point of the library stream an evaluation against this eq table waits for
void atlas_rt_eval_event_drop(atlas_poly_t eq_full);
bool atlas_rt_eval_event_live(atlas_poly_t eq_full);
// ReadRafProver over a table of at most 2^12 entries as a host-arithmetic instance (shout.hip): G = the device histogram (downloaded)
int atlas_rt_shout_read_raf_host_new(atlas_poly_t G, const int32_t* table, size_t log_K, const atlas_fr_t* gamma, atlas_instance_t* out);
// a BatchedSumcheck member over zero variables: no rounds, the given final claims (elementwise.hip)
int atlas_rt_const_member_new(const atlas_fr_t* finals, size_t n, size_t degree, atlas_instance_t* out);
int atlas_rt_ps_shout_relu_new(const uint64_t* lookup_indices, size_t log_T, size_t xlen, const atlas_fr_t* r_node_output, const atlas_fr_t* gamma,
                               atlas_poly_t eq_shared, atlas_instance_t* out, bool defer = false);
