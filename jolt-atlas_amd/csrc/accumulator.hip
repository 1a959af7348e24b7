// ProverOpeningAccumulator (joltworks/src/poly/opening_proof.rs:195-643) behind the C-ABI: the claim bookkeeping that
// sits between the operator provers and the last stage of ONNXProof::prove.
//   append_dense  (:265-315)  transcript.append_scalar(claim); openings[opening_id] = (point, claim); one
//                             opening-reduction instance per committed polynomial (sumchecks keyed by CommittedPoly:
//                             a later append for the same polynomial replaces the earlier one)
//   append_sparse (:317-372)  the same for the d one-hot polynomials of a lookup: one claim appended per polynomial,
//                             point = r_address || r_cycle
//   append_virtual (:374-409) transcript.append_scalar(claim); openings[opening_id] = (point, claim)
//   get (get_virtual_polynomial_opening / get_committed_polynomial_opening, :199-243)
//   prove_reduced_openings    prepare_for_sumcheck + prove_batch_opening_sumcheck + finalize (:447-532, :611-643) + the
//                             joint opening, over the committed openings in BTreeMap<CommittedPoly> order
// Keys are 64-bit integers chosen by the caller so that their numeric order is the Ord of the Rust key types
// (OpeningId / CommittedPoly, jolt-atlas-core/src/onnx_proof/types.rs:27-129): the maps here are ordered like the
// reference's BTreeMaps.  Host code only; the arithmetic is in the entry points it calls.
#include <cstring>
#include <map>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_field.hpp"
#include "runtime.hpp"

namespace H = atlas_host;
using atlas_rt::fail;

namespace {
struct Opening { std::vector<atlas_fr_t> point; atlas_fr_t claim; };
struct Committed {
    int kind = 0;                       // 0 dense, 1 one-hot
    atlas_poly_t poly = nullptr;        // dense (borrowed)
    std::vector<int32_t> k;             // one-hot indices (copied)
    size_t log_K = 0, log_T = 0, n = 0;
    std::vector<atlas_fr_t> point;
    atlas_fr_t claim;
};
}  // namespace

struct atlas_accumulator {
    std::map<uint64_t, Opening> openings;        // OpeningId -> (point, claim)
    std::map<uint64_t, Committed> sumchecks;     // CommittedPoly -> its opening-reduction instance
};

extern "C" {

int atlas_accumulator_new(atlas_accumulator_t* out) {
    if (!out) return fail(ATLAS_EINVAL, "accumulator_new");
    *out = new atlas_accumulator();
    return ATLAS_OK;
}
int atlas_accumulator_free(atlas_accumulator_t a) { delete a; return ATLAS_OK; }

int atlas_accumulator_append_dense(atlas_accumulator_t a, atlas_transcript_t* t, uint64_t opening_key, uint64_t poly_key,
                                   atlas_poly_t poly, const atlas_fr_t* point, size_t n, const atlas_fr_t* claim) {
    if (!a || !t || !poly || (!point && n) || !claim) return fail(ATLAS_EINVAL, "accumulator_append_dense: null argument");
    size_t len = 0;
    atlas_poly_len(poly, &len);
    if (len != ((size_t)1 << n)) return fail(ATLAS_EINVAL, "accumulator_append_dense: polynomial length != 2^n");
    H::tr_append_scalar(*reinterpret_cast<H::Transcript*>(t), *reinterpret_cast<const H::Fr*>(claim));
    Opening O; O.point.assign(point, point + n); O.claim = *claim;
    a->openings[opening_key] = O;
    Committed Cm; Cm.kind = 0; Cm.poly = poly; Cm.n = n; Cm.point = O.point; Cm.claim = *claim;
    a->sumchecks[poly_key] = std::move(Cm);
    return ATLAS_OK;
}

int atlas_accumulator_append_sparse(atlas_accumulator_t a, atlas_transcript_t* t, const uint64_t* opening_keys, const uint64_t* poly_keys,
                                    const int32_t* const* nonzero_indices, size_t R, size_t log_K, size_t log_T,
                                    const atlas_fr_t* r_address, const atlas_fr_t* r_cycle, const atlas_fr_t* claims) {
    if (!a || !t || !opening_keys || !poly_keys || !nonzero_indices || !r_address || !r_cycle || !claims || R == 0)
        return fail(ATLAS_EINVAL, "accumulator_append_sparse: null argument");
    if (log_K == 0 || log_K > 16 || log_T == 0 || log_T > 26) return fail(ATLAS_EINVAL, "accumulator_append_sparse: 1 <= log_K <= 16, 1 <= log_T <= 26");
    const size_t T = (size_t)1 << log_T, K = (size_t)1 << log_K;
    for (size_t r = 0; r < R; r++) {
        if (!nonzero_indices[r]) return fail(ATLAS_EINVAL, "accumulator_append_sparse: null index row");
        for (size_t j = 0; j < T; j++) if (nonzero_indices[r][j] >= (int32_t)K) return fail(ATLAS_EINVAL, "accumulator_append_sparse: index >= K");
    }
    H::Transcript& T_ = *reinterpret_cast<H::Transcript*>(t);
    for (size_t r = 0; r < R; r++) H::tr_append_scalar(T_, *reinterpret_cast<const H::Fr*>(&claims[r]));      // :335-339
    std::vector<atlas_fr_t> concat(r_address, r_address + log_K);
    concat.insert(concat.end(), r_cycle, r_cycle + log_T);
    for (size_t r = 0; r < R; r++) {
        Opening O; O.point = concat; O.claim = claims[r];
        a->openings[opening_keys[r]] = O;
        Committed Cm; Cm.kind = 1; Cm.log_K = log_K; Cm.log_T = log_T; Cm.point = concat; Cm.claim = claims[r];
        Cm.k.assign(nonzero_indices[r], nonzero_indices[r] + T);
        a->sumchecks[poly_keys[r]] = std::move(Cm);
    }
    return ATLAS_OK;
}

int atlas_accumulator_append_virtual(atlas_accumulator_t a, atlas_transcript_t* t, uint64_t opening_key, const atlas_fr_t* point, size_t n,
                                     const atlas_fr_t* claim) {
    if (!a || !t || (!point && n) || !claim) return fail(ATLAS_EINVAL, "accumulator_append_virtual: null argument");
    H::tr_append_scalar(*reinterpret_cast<H::Transcript*>(t), *reinterpret_cast<const H::Fr*>(claim));
    Opening O; O.point.assign(point, point + n); O.claim = *claim;
    a->openings[opening_key] = O;
    return ATLAS_OK;
}

int atlas_accumulator_get(atlas_accumulator_t a, uint64_t opening_key, atlas_fr_t* point_out, size_t cap, size_t* n_out, atlas_fr_t* claim_out) {
    if (!a || !n_out || !claim_out) return fail(ATLAS_EINVAL, "accumulator_get: null argument");
    auto it = a->openings.find(opening_key);
    if (it == a->openings.end()) return fail(ATLAS_ESTATE, "accumulator_get: no opening under this key");     // the reference panics
    *n_out = it->second.point.size();
    *claim_out = it->second.claim;
    if (point_out) {
        if (cap < it->second.point.size()) return fail(ATLAS_EINVAL, "accumulator_get: point buffer too small");
        std::memcpy(point_out, it->second.point.data(), it->second.point.size() * sizeof(atlas_fr_t));
    }
    return ATLAS_OK;
}

size_t atlas_accumulator_num_committed(atlas_accumulator_t a) { return a ? a->sumchecks.size() : 0; }
size_t atlas_accumulator_max_rounds(atlas_accumulator_t a) {
    size_t m = 0;
    if (a) for (auto& kv : a->sumchecks) { const size_t n = kv.second.kind == 0 ? kv.second.n : kv.second.log_K + kv.second.log_T; m = n > m ? n : m; }
    return m;
}

int atlas_accumulator_prove_reduced_openings(atlas_accumulator_t a, atlas_srs_t srs, atlas_transcript_t* transcript, atlas_fr_t* sumcheck_rows,
                                             uint32_t* n_coeffs, atlas_u128_t* challenges, size_t* max_rounds_out, atlas_fr_t* sumcheck_claims,
                                             atlas_g1_affine_t* com, atlas_g1_affine_t* w, atlas_fr_t* v) {
    if (!a || a->sumchecks.empty()) return fail(ATLAS_EINVAL, "accumulator_prove_reduced_openings: no committed openings");
    std::vector<atlas_opening_t> ops;
    ops.reserve(a->sumchecks.size());
    for (auto& kv : a->sumchecks) {                      // BTreeMap<CommittedPoly> order
        const Committed& Cm = kv.second;
        atlas_opening_t O;
        std::memset(&O, 0, sizeof(O));
        O.kind = Cm.kind; O.poly = Cm.poly; O.n = Cm.n; O.k = Cm.kind ? Cm.k.data() : nullptr;
        O.log_K = Cm.log_K; O.log_T = Cm.log_T; O.point = Cm.point.data(); O.claim = Cm.claim;
        ops.push_back(O);
    }
    return atlas_prove_reduced_openings(ops.data(), ops.size(), srs, transcript, sumcheck_rows, n_coeffs, challenges, max_rounds_out,
                                        sumcheck_claims, com, w, v);
}

}  // extern "C"
