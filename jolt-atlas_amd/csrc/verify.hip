// Sumcheck verification, product side (first piece).  Host arithmetic only, as in the reference: the
// verifier replays the transcript over the compressed round polynomials and returns the claim the instances' oracles
// must meet.  Restates (paths under the jolt-atlas tree, joltworks/src/):
//   SumcheckInstanceProof::verify        subprotocols/sumcheck.rs:653-686
//   BatchedSumcheck::verify              subprotocols/sumcheck.rs:187-259
//   CompressedUniPoly::{degree, eval_from_hint, append_to_transcript}   poly/unipoly.rs:519-557
// The final comparison against sum_i coeff_i * expected_output_claim_i(r) is split off (atlas_batched_sumcheck_check): the
// expected claims come from the instances' opening claims, which live with the caller's accumulator (cache_openings runs
// between the two calls, exactly where the reference runs it, so the transcript order is the reference's).
// HyperKZG::verify needs the BN254 pairing and is not built (the GPU tests verify openings with the known trapdoor).
#include <cstring>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_field.hpp"
#include "runtime.hpp"

namespace H = atlas_host;
using atlas_rt::fail;

namespace {

// one round of SumcheckInstanceProof::verify: degree check, append, challenge, eval_from_hint
int verify_round(H::Transcript& T, const H::Fr* cc, uint32_t n, size_t degree_bound, H::Fr& e, atlas_u128_t* chal, int mode) {
    if (n == 0) return fail(ATLAS_EINVAL, "sumcheck_verify: empty round polynomial");
    if (n > degree_bound) return fail(ATLAS_EVERIFY, "sumcheck_verify: InvalidInputLength (round polynomial above the degree bound)");
    H::tr_append_message(T, "UniPoly_begin");
    for (uint32_t k = 0; k < n; k++) H::tr_append_scalar(T, cc[k]);
    H::tr_append_message(T, "UniPoly_end");
    uint64_t lo, hi;
    H::tr_challenge_u128(T, lo, hi);
    chal->lo = lo; chal->hi = hi;
    const H::Fr x = H::challenge_to_fr(lo, hi, mode);
    // eval_from_hint (unipoly.rs:519-533): linear term from the hint, then the powers of x
    H::Fr lin = H::sub(H::sub(e, cc[0]), cc[0]);
    for (uint32_t k = 1; k < n; k++) lin = H::sub(lin, cc[k]);
    H::Fr pt = x, sum = H::add(cc[0], H::mul(x, lin));
    for (uint32_t k = 1; k < n; k++) { pt = H::mul(pt, x); sum = H::add(sum, H::mul(cc[k], pt)); }
    e = sum;
    return ATLAS_OK;
}

}  // namespace

extern "C" {

int atlas_sumcheck_proof_verify(const atlas_fr_t* compressed, size_t row_stride, const uint32_t* n_coeffs, size_t n_rounds, size_t degree_bound,
                                const atlas_fr_t* claim, atlas_transcript_t* transcript, atlas_fr_t* final_claim, atlas_u128_t* challenges) {
    if ((n_rounds && (!compressed || !n_coeffs || !challenges)) || !claim || !transcript || !final_claim)
        return fail(ATLAS_EINVAL, "sumcheck_proof_verify: null argument");
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(transcript);
    H::Fr e;
    std::memcpy(&e, claim, 32);
    for (size_t i = 0; i < n_rounds; i++) {
        if (n_coeffs[i] > row_stride) return fail(ATLAS_EINVAL, "sumcheck_proof_verify: n_coeffs beyond row_stride");
        int rc = verify_round(T, reinterpret_cast<const H::Fr*>(compressed) + i * row_stride, n_coeffs[i], degree_bound, e, &challenges[i],
                              atlas_rt::g.challenge_mode);
        if (rc) return rc;
    }
    std::memcpy(final_claim, &e, 32);
    return ATLAS_OK;
}

int atlas_batched_sumcheck_verify(const atlas_fr_t* compressed, size_t row_stride, const uint32_t* n_coeffs, size_t max_rounds,
                                  const atlas_fr_t* input_claims, const size_t* num_rounds, const size_t* degrees, size_t n_instances,
                                  atlas_transcript_t* transcript, atlas_fr_t* batching_coeffs, atlas_fr_t* output_claim, atlas_u128_t* challenges) {
    if (!input_claims || !num_rounds || !degrees || !transcript || !batching_coeffs || !output_claim || n_instances == 0 ||
        (max_rounds && (!compressed || !n_coeffs || !challenges)))
        return fail(ATLAS_EINVAL, "batched_sumcheck_verify: null argument");
    size_t mr = 0, md = 0;
    for (size_t i = 0; i < n_instances; i++) { mr = num_rounds[i] > mr ? num_rounds[i] : mr; md = degrees[i] > md ? degrees[i] : md; }
    if (mr != max_rounds) return fail(ATLAS_EVERIFY, "batched_sumcheck_verify: the proof has a different number of rounds than the instances");
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(transcript);
    const H::Fr* ic = reinterpret_cast<const H::Fr*>(input_claims);
    for (size_t i = 0; i < n_instances; i++) H::tr_append_scalar(T, ic[i]);                     // :205-208
    H::Fr* coeff = reinterpret_cast<H::Fr*>(batching_coeffs);
    for (size_t i = 0; i < n_instances; i++) coeff[i] = H::tr_challenge_scalar(T);               // challenge_vector :210
    H::Fr e = H::zero();
    const H::Fr two = H::from_u64(2);
    for (size_t i = 0; i < n_instances; i++) {                                                   // :221-229
        H::Fr c = ic[i];
        for (size_t k = 0; k < max_rounds - num_rounds[i]; k++) c = H::mul(c, two);              // mul_pow_2
        e = H::add(e, H::mul(c, coeff[i]));
    }
    for (size_t i = 0; i < max_rounds; i++) {                                                    // proof.verify :231-232
        if (n_coeffs[i] > row_stride) return fail(ATLAS_EINVAL, "batched_sumcheck_verify: n_coeffs beyond row_stride");
        int rc = verify_round(T, reinterpret_cast<const H::Fr*>(compressed) + i * row_stride, n_coeffs[i], md, e, &challenges[i], atlas_rt::g.challenge_mode);
        if (rc) return rc;
    }
    std::memcpy(output_claim, &e, 32);
    return ATLAS_OK;
}

int atlas_batched_sumcheck_check(const atlas_fr_t* batching_coeffs, const atlas_fr_t* expected_output_claims, size_t n_instances,
                                 const atlas_fr_t* output_claim) {
    if (!batching_coeffs || !expected_output_claims || !output_claim || n_instances == 0) return fail(ATLAS_EINVAL, "batched_sumcheck_check: null argument");
    const H::Fr* c = reinterpret_cast<const H::Fr*>(batching_coeffs);
    const H::Fr* x = reinterpret_cast<const H::Fr*>(expected_output_claims);
    H::Fr s = H::zero();
    for (size_t i = 0; i < n_instances; i++) s = H::add(s, H::mul(x[i], c[i]));                  // :234-253
    H::Fr o;
    std::memcpy(&o, output_claim, 32);
    if (!(s == o)) return fail(ATLAS_EVERIFY, "SumcheckVerificationError: output claim != sum coeff_i * expected_output_claim_i");
    return ATLAS_OK;
}

}  // extern "C"
