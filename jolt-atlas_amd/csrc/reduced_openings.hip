// ONNXProof::prove_reduced_openings (jolt-atlas-core/src/onnx_proof/prover.rs:141-176): the last
// stage of the prover — the batched opening-reduction sumcheck over every committed polynomial,
// the claim/gamma exchange of ProverOpeningAccumulator::finalize_batch_opening_sumcheck
// (joltworks/src/poly/opening_proof.rs:611-643), the joint polynomial of build_materialized_rlc and
// the HyperKZG opening at r_sumcheck.  Everything here is a composition of entry points that exist
// on their own; the accumulator's map bookkeeping (a22) stays with the caller, who passes the
// openings in BTreeMap<CommittedPoly> order.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_field.hpp"
#include "internal.hpp"
#include "runtime.hpp"
#include "shard_group.hpp"

namespace H = atlas_host;
using atlas_rt::fail;

struct atlas_rt_pool_row { const uint64_t* d_lookups; size_t shift, log_T; const atlas_fr_t* point; };
int atlas_rt_onehot_pool_new(const atlas_rt_pool_row* in, size_t n, size_t log_K, size_t batch_max_rounds, atlas_instance_t* out, const int32_t** d_idx_rows);   // opening.hip

static thread_local double g_last_open_ms = 0;
int atlas_rt_chunk_index_rows(const atlas_rt_pool_row* in, size_t n, size_t log_K, int32_t** d_buf, const int32_t** rows);      // opening.hip
int atlas_rt_dense_pool_new(const atlas_poly_t* polys, const atlas_fr_t* const* points, const size_t* ns, size_t count, size_t batch_max_rounds, atlas_instance_t* out);   // opening.hip
int atlas_rt_batched_set_shard(atlas_batched_t b, atlas_shard_group_t sh);                                                        // batched.hip
int atlas_rt_batched_add_remote(atlas_batched_t b, size_t rounds, const atlas_fr_t* input_claim);
double atlas_rt_last_hyperkzg_ms() { return g_last_open_ms; }

// `sh` (may be NULL): the ranks of a sharded whole proof (atlas_prove_graph_sharded).  Every rank runs the reduction sumcheck and builds the
// joint polynomial; the opening's commitment groups are split by point range over the ranks (atlas_hyperkzg_open_sharded).
int atlas_rt_prove_reduced_openings(const atlas_opening_t* openings, size_t n_open, atlas_srs_t srs,
                                    atlas_transcript_t* transcript, atlas_fr_t* sumcheck_rows, uint32_t* n_coeffs,
                                    atlas_u128_t* challenges, size_t* max_rounds_out, atlas_fr_t* sumcheck_claims,
                                    atlas_g1_affine_t* com, atlas_g1_affine_t* w, atlas_fr_t* v, atlas_shard_group_t sh) {
    NEED_INIT();
    if (!openings || n_open == 0 || !srs || !transcript || !sumcheck_rows || !n_coeffs || !challenges || !max_rounds_out ||
        !sumcheck_claims || !com || !w || !v)
        return fail(ATLAS_EINVAL, "prove_reduced_openings: null argument");
    const bool trace = getenv("ATLAS_TRACE") != nullptr;            // wall clock of the stages on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!trace) return;
        atlas_sync();
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[atlas trace] prove_reduced_openings %-26s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t_prev).count());
        t_prev = t1;
    };
    std::vector<atlas_instance_t> inst(n_open, nullptr);
    atlas_batched_t b = nullptr;
    atlas_poly_t joint = nullptr;
    int rc = ATLAS_OK;
    int32_t* d_remote_idx = nullptr;              // sharded: the index rows of the pool rows other ranks step (for the joint polynomial)
    auto cleanup = [&]() {
        for (auto i : inst) if (i) atlas_instance_free(i);
        if (b) atlas_batched_free(b);
        if (joint) atlas_poly_free(joint);
        if (d_remote_idx) { std::lock_guard<atlas_rt::Mutex> lkg(atlas_rt::rt().mu); hipFree(d_remote_idx); d_remote_idx = nullptr; }
    };
    // prepare_for_sumcheck: one opening-reduction instance per committed polynomial (the dense ones work on a
    // copy: the joint polynomial needs the originals)
    // one-hot openings with the same (log_K, log_T, r_cycle) share their cycle-phase launches (EqCycleState sharing,
    // opening_proof.rs:339-343)
    std::vector<char> done(n_open, 0);
    // A sharded whole proof (review item 4a): the members of the reduction sumcheck are split over the ranks — the pooled one-hot rows (all but
    // a handful of the members of a transformer: 8772 of 8823 for the GPT-2-shaped graph) in contiguous shares, everything else on rank 0.
    // owner[i] = the rank that steps member i; the others carry it as a remote member of the batch (batched.hip: atlas_rt_batched_add_remote).
    // ATLAS_REDUCTION_REPLICATED=1: every rank steps every member, as before (A/B).
    const int world = sh ? sh->world : 1, my_rank = sh ? sh->rank : 0;
    const bool split = world > 1 && getenv("ATLAS_REDUCTION_REPLICATED") == nullptr;
    std::vector<int> owner(n_open, split ? 0 : my_rank);
    // the one-hot openings over device-resident lookups with log_K <= 4 (and equal: the RaD chunk polynomials): one pool, stepped together
    // (opening.hip OneHotPool); the others (GatherRa of GatherSmall: all the dictionary's address bits in one polynomial) take the grouped path below
    std::vector<const int32_t*> pool_idx(n_open, nullptr);
    {
        const bool pool = getenv("ATLAS_NO_OPENING_POOL") == nullptr;
        size_t lk = 0, batch_rounds = 0;
        for (size_t i = 0; i < n_open; i++) {
            const atlas_opening_t& O = openings[i];
            const size_t nr = O.kind ? O.log_K + O.log_T : O.n;
            batch_rounds = nr > batch_rounds ? nr : batch_rounds;
        }
        auto eligible = [&](const atlas_opening_t& O) { return O.kind == 1 && O.d_lookups && !O.k && O.point && O.log_K <= 4 && (lk == 0 || O.log_K == lk); };
        std::vector<atlas_rt_pool_row> rows; std::vector<size_t> where;
        for (size_t i = 0; i < n_open && pool; i++)
            if (eligible(openings[i])) { lk = openings[i].log_K; rows.push_back(atlas_rt_pool_row{openings[i].d_lookups, openings[i].chunk_shift, openings[i].log_T, openings[i].point}); where.push_back(i); }
        if (!rows.empty() && split) {
            // this rank's share of the pool: instances for it, bare index rows for the rest (the joint polynomial needs every row)
            const size_t np = rows.size(), lo = np * (size_t)my_rank / (size_t)world, hi = np * (size_t)(my_rank + 1) / (size_t)world;
            for (size_t q = 0; q < np; q++) { int r = 0; while (!(np * (size_t)r / (size_t)world <= q && q < np * (size_t)(r + 1) / (size_t)world)) r++; owner[where[q]] = r; }
            std::vector<atlas_rt_pool_row> mine(rows.begin() + (ptrdiff_t)lo, rows.begin() + (ptrdiff_t)hi), rest;
            std::vector<size_t> rest_where;
            for (size_t q = 0; q < np; q++) if (q < lo || q >= hi) { rest.push_back(rows[q]); rest_where.push_back(where[q]); }
            if (!mine.empty()) {
                std::vector<atlas_instance_t> pi(mine.size(), nullptr); std::vector<const int32_t*> px(mine.size(), nullptr);
                { PROF("reduced: onehot pool new"); rc = atlas_rt_onehot_pool_new(mine.data(), mine.size(), lk, batch_rounds, pi.data(), px.data()); }
                for (size_t q = 0; q < mine.size() && !rc; q++) { inst[where[lo + q]] = pi[q]; pool_idx[where[lo + q]] = px[q]; }
            }
            if (!rc && !rest.empty()) {
                std::vector<const int32_t*> px(rest.size(), nullptr);
                rc = atlas_rt_chunk_index_rows(rest.data(), rest.size(), lk, &d_remote_idx, px.data());
                for (size_t q = 0; q < rest.size() && !rc; q++) pool_idx[rest_where[q]] = px[q];
            }
            for (size_t q = 0; q < np; q++) done[where[q]] = 1;
        } else if (!rows.empty()) {
            std::vector<atlas_instance_t> pi(rows.size(), nullptr); std::vector<const int32_t*> px(rows.size(), nullptr);
            { PROF("reduced: onehot pool new"); rc = atlas_rt_onehot_pool_new(rows.data(), rows.size(), lk, batch_rounds, pi.data(), px.data()); }
            for (size_t q = 0; q < rows.size() && !rc; q++) { inst[where[q]] = pi[q]; pool_idx[where[q]] = px[q]; done[where[q]] = 1; }
        }
    }
    // openings given by device lookups outside the pool (mixed log_K, or ATLAS_NO_OPENING_POOL): their index rows on the host
    std::vector<atlas_opening_t> local(openings, openings + n_open);
    std::vector<std::vector<int32_t>> host_rows;
    host_rows.reserve(n_open);
    for (size_t i = 0; i < n_open && !rc; i++) {
        atlas_opening_t& O = local[i];
        if (O.kind != 1 || done[i] || O.k || !O.d_lookups) continue;
        const size_t T = (size_t)1 << O.log_T;
        std::vector<uint64_t> lk(T);
        {
            std::lock_guard<atlas_rt::Mutex> lkg(atlas_rt::rt().mu);
            hipError_t e = hipMemcpyAsync(lk.data(), O.d_lookups, T * 8, hipMemcpyDeviceToHost, atlas_rt::rt().stream);
            if (e == hipSuccess) e = hipStreamSynchronize(atlas_rt::rt().stream);
            if (e != hipSuccess) { rc = atlas_rt::fail(ATLAS_ENODEV, "prove_reduced_openings: lookup indices to the host", e); break; }   // falls through to cleanup()
        }
        host_rows.emplace_back(T);
        for (size_t j = 0; j < T; j++) host_rows.back()[j] = (int32_t)(O.chunk_shift >= 64 ? 0 : ((lk[j] >> O.chunk_shift) & (((uint64_t)1 << O.log_K) - 1)));
        O.k = host_rows.back().data();
    }
    openings = local.data();
    // the dense members this rank steps, one pool for all of them (opening.hip DensePool; ATLAS_NO_DENSE_POOL=1: an instance each, as before)
    if (!rc && getenv("ATLAS_NO_DENSE_POOL") == nullptr) {
        size_t batch_rounds = 0;
        for (size_t i = 0; i < n_open; i++) { const size_t nr = openings[i].kind ? openings[i].log_K + openings[i].log_T : openings[i].n; batch_rounds = nr > batch_rounds ? nr : batch_rounds; }
        std::vector<atlas_poly_t> dp; std::vector<const atlas_fr_t*> dpt; std::vector<size_t> dn, where;
        for (size_t i = 0; i < n_open; i++) {
            const atlas_opening_t& O = openings[i];
            if (O.kind != 0 || owner[i] != my_rank || !O.poly || !O.point || O.n == 0 || O.n > 26) continue;      // (constants and malformed members: the path below, with its checks)
            size_t plen = 0;
            if (atlas_poly_len(O.poly, &plen) || plen != ((size_t)1 << O.n)) continue;
            dp.push_back(O.poly); dpt.push_back(O.point); dn.push_back(O.n); where.push_back(i);
        }
        if (dp.size() >= 2) {
            std::vector<atlas_instance_t> di(dp.size(), nullptr);
            { PROF("reduced: dense pool new"); rc = atlas_rt_dense_pool_new(dp.data(), dpt.data(), dn.data(), dp.size(), batch_rounds, di.data()); }
            for (size_t q = 0; q < dp.size() && !rc; q++) { inst[where[q]] = di[q]; done[where[q]] = 1; }
        }
    }
    for (size_t i = 0; i < n_open && !rc; i++) {
        const atlas_opening_t& O = openings[i];
        if (owner[i] != my_rank) continue;                 // (another rank steps it; the members outside the pool all belong to rank 0)
        if (O.kind == 0) {
            if (done[i]) continue;
            atlas_poly_t c = nullptr;
            if (!O.poly || (!O.point && O.n)) { rc = fail(ATLAS_EINVAL, "prove_reduced_openings: dense opening without polynomial/point"); break; }
            PROF("reduced: dense clone + new");
            rc = atlas_poly_clone(O.poly, &c);
            if (!rc) { rc = atlas_dense_opening_new(c, O.point, O.n, &inst[i]); if (rc) atlas_poly_free(c); }
            continue;
        }
        if (done[i]) continue;
        if (!O.k || !O.point) { rc = fail(ATLAS_EINVAL, "prove_reduced_openings: one-hot opening without indices/point"); break; }
        std::vector<size_t> members;
        for (size_t q = i; q < n_open; q++) {
            const atlas_opening_t& Q = openings[q];
            if (Q.kind == 1 && !done[q] && owner[q] == my_rank && Q.k && Q.point && Q.log_K == O.log_K && Q.log_T == O.log_T &&
                std::memcmp(Q.point + Q.log_K, O.point + O.log_K, O.log_T * sizeof(atlas_fr_t)) == 0)
                members.push_back(q);
        }
        std::vector<const int32_t*> idx(members.size());
        std::vector<atlas_fr_t> ra(members.size() * O.log_K);
        std::vector<atlas_instance_t> rows(members.size(), nullptr);
        for (size_t q = 0; q < members.size(); q++) {
            idx[q] = openings[members[q]].k;
            std::memcpy(&ra[q * O.log_K], openings[members[q]].point, O.log_K * sizeof(atlas_fr_t));
        }
        { PROF("reduced: onehot group new (outside the pool)"); rc = atlas_onehot_opening_group_new(idx.data(), members.size(), O.log_K, O.log_T, ra.data(), O.point + O.log_K, rows.data()); }
        for (size_t q = 0; q < members.size() && !rc; q++) { inst[members[q]] = rows[q]; done[members[q]] = 1; }
    }
    mark("instances (prepare)");
    // prove_batch_opening_sumcheck: BatchedSumcheck::prove over the instances (degree 2: rows of 3)
    if (!rc) rc = atlas_batched_new(&b);
    if (!rc && split) rc = atlas_rt_batched_set_shard(b, sh);
    for (size_t i = 0; i < n_open && !rc; i++) {
        const atlas_opening_t& O = openings[i];
        rc = owner[i] == my_rank ? atlas_batched_add_instance(b, inst[i], &O.claim) : atlas_rt_batched_add_remote(b, O.kind ? O.log_K + O.log_T : O.n, &O.claim);
    }
    if (!rc) rc = atlas_batched_prove(b, transcript, sumcheck_rows, 3, n_coeffs, challenges, max_rounds_out);
    mark("batched sumcheck");
    // cache_openings -> sumcheck_claims (opening_reduction.rs:238-246), then finalize (:611-643)
    for (size_t i = 0; i < n_open && !rc; i++) {
        size_t nf = 0;
        if (owner[i] == my_rank) rc = atlas_instance_final_claims(inst[i], &sumcheck_claims[i], 1, &nf);
        else std::memset(&sumcheck_claims[i], 0, sizeof(atlas_fr_t));
    }
    if (!rc && split) {                                             // every rank needs every member's final claim: its owner's record
        std::vector<atlas_fr_t> all((size_t)world * n_open);
        if (!sh->allgather_bulk(sumcheck_claims, n_open * sizeof(atlas_fr_t), all.data())) rc = fail(ATLAS_ENODEV, "prove_reduced_openings (sharded): a rank did not answer");
        for (size_t i = 0; i < n_open && !rc; i++) sumcheck_claims[i] = all[(size_t)owner[i] * n_open + i];
    }
    if (rc) { cleanup(); return rc; }
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(transcript);
    H::tr_append_scalars(T, reinterpret_cast<const H::Fr*>(sumcheck_claims), n_open);
    std::vector<H::Fr> gamma(n_open);                               // challenge_scalar_powers (blake2b.rs:224-231)
    {
        const H::Fr q = H::tr_challenge_scalar(T);
        gamma[0] = H::one();
        for (size_t i = 1; i < n_open; i++) gamma[i] = H::mul(gamma[i - 1], q);
    }
    // build_materialized_rlc(gamma_powers, poly_map)
    std::vector<atlas_rlc_dense_t> dense;
    std::vector<atlas_rlc_onehot_t> onehot;
    for (size_t i = 0; i < n_open; i++) {
        const atlas_opening_t& O = openings[i];
        if (O.kind == 0) { atlas_rlc_dense_t d; d.poly = O.poly; std::memcpy(&d.coeff, &gamma[i], 32); dense.push_back(d); }
        else {
            atlas_rlc_onehot_t o; o.k = pool_idx[i] ? pool_idx[i] : O.k; o.T = (size_t)1 << O.log_T; o.K = (size_t)1 << O.log_K; o.k_on_device = pool_idx[i] ? 1 : 0;
            std::memcpy(&o.coeff, &gamma[i], 32); onehot.push_back(o);
        }
    }
    rc = atlas_rlc_build(dense.data(), dense.size(), onehot.data(), onehot.size(), &joint);
    // the members go BEFORE the opening allocates its arena (6 n Fr: 3.2 GB at n = 2^24): the pools' vectors (9 GB + the index rows on one GPU, 1 / world
    // of the vectors per rank) and the opening's arena are then never resident together — the peak of the stage is the larger, not the sum
    for (auto& i : inst) if (i) { atlas_instance_free(i); i = nullptr; }
    if (b) { atlas_batched_free(b); b = nullptr; }
    if (d_remote_idx) { std::lock_guard<atlas_rt::Mutex> lkg(atlas_rt::rt().mu); hipFree(d_remote_idx); d_remote_idx = nullptr; }
    mark("claims + joint polynomial");
    size_t jlen = 0;
    if (!rc) atlas_poly_len(joint, &jlen);
    if (!rc && jlen != ((size_t)1 << *max_rounds_out)) rc = fail(ATLAS_EINVAL, "prove_reduced_openings: joint polynomial length != 2^max_rounds");
    // PCS::prove(generators, &rlc, &r_sumcheck, None, transcript) = HyperKZG::open
    atlas_sync();
    const auto t_open = std::chrono::steady_clock::now();
    if (!rc) rc = sh ? atlas_hyperkzg_open_sharded(srs, sh, joint, challenges, *max_rounds_out, transcript, com, w, v)
                     : atlas_hyperkzg_open(srs, joint, challenges, *max_rounds_out, transcript, com, w, v);
    atlas_sync();
    g_last_open_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_open).count();
    mark("HyperKZG::open");
    cleanup();
    return rc;
}

extern "C" int atlas_prove_reduced_openings(const atlas_opening_t* openings, size_t n_open, atlas_srs_t srs,
                                            atlas_transcript_t* transcript, atlas_fr_t* sumcheck_rows, uint32_t* n_coeffs,
                                            atlas_u128_t* challenges, size_t* max_rounds_out, atlas_fr_t* sumcheck_claims,
                                            atlas_g1_affine_t* com, atlas_g1_affine_t* w, atlas_fr_t* v) {
    return atlas_rt_prove_reduced_openings(openings, n_open, srs, transcript, sumcheck_rows, n_coeffs, challenges, max_rounds_out, sumcheck_claims, com, w, v, nullptr);
}
