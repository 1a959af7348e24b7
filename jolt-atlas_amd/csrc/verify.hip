// Sumcheck verification, product side (first piece).  Host arithmetic only, as in the reference: the
// verifier replays the transcript over the compressed round polynomials and returns the claim the instances' oracles
// must meet.  Restates (paths under the jolt-atlas tree, joltworks/src/):
//   SumcheckInstanceProof::verify        subprotocols/sumcheck.rs:653-686
//   BatchedSumcheck::verify              subprotocols/sumcheck.rs:187-259
//   CompressedUniPoly::{degree, eval_from_hint, append_to_transcript}   poly/unipoly.rs:519-557
// The final comparison against sum_i coeff_i * expected_output_claim_i(r) is split off (atlas_batched_sumcheck_check): the
// expected claims come from the instances' opening claims, which live with the caller's accumulator (cache_openings runs
// between the two calls, exactly where the reference runs it, so the transcript order is the reference's).
// HyperKZG::verify (poly/commitment/hyperkzg/mod.rs:451-509 verify_inner, :283-366 kzg_verify_batch) on the host with the BN254
// pairing of host_pairing.hpp: atlas_hyperkzg_verify; atlas_pairing_check = P::multi_pairing(..).is_zero().
#include <cstring>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_field.hpp"
#include "host_pairing.hpp"
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
                              atlas_rt::rt().challenge_mode);
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
        int rc = verify_round(T, reinterpret_cast<const H::Fr*>(compressed) + i * row_stride, n_coeffs[i], md, e, &challenges[i], atlas_rt::rt().challenge_mode);
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

// ---------------------------------------------------------------- HyperKZG::verify
namespace {

H::G1Aff g1_in(const atlas_g1_affine_t& p) {
    H::G1Aff a;
    if (p.infinity) return H::G1Aff{H::q_zero(), H::q_zero()};
    std::memcpy(a.x.l, p.x.l, 32); std::memcpy(a.y.l, p.y.l, 32);
    return a;
}
H::G2Aff g2_in(const atlas_g2_affine_t& p) {
    H::G2Aff a; a.inf = p.infinity != 0;
    std::memcpy(a.x.c0.l, p.x[0].l, 32); std::memcpy(a.x.c1.l, p.x[1].l, 32);
    std::memcpy(a.y.c0.l, p.y[0].l, 32); std::memcpy(a.y.c1.l, p.y[1].l, 32);
    if (a.inf) a = H::g2_inf();
    return a;
}
void g2_out(const H::G2Aff& a, atlas_g2_affine_t* o) {
    std::memset(o, 0, sizeof(*o));
    o->infinity = a.inf ? 1 : 0;
    if (a.inf) return;
    std::memcpy(o->x[0].l, a.x.c0.l, 32); std::memcpy(o->x[1].l, a.x.c1.l, 32);
    std::memcpy(o->y[0].l, a.y.c0.l, 32); std::memcpy(o->y[1].l, a.y.c1.l, 32);
}
// k * P on the host (XYZZ double-and-add), k a Montgomery Fr
H::G1X g1_mul_fr(const H::G1Aff& P, const H::Fr& k) {
    uint64_t c[4];
    H::to_canonical(k, c);
    H::G1X acc = H::gx_inf();
    const H::G1X base = H::gx_from_aff(P);
    for (int i = 255; i >= 0; i--) {
        acc = H::gx_dbl(acc);
        if ((c[i >> 6] >> (i & 63)) & 1) acc = H::gx_add(acc, base);
    }
    return acc;
}
void append_points(H::Transcript& T, const atlas_g1_affine_t* p, size_t n) { atlas_transcript_append_points(reinterpret_cast<atlas_transcript_t*>(&T), p, n); }

}  // namespace

extern "C" {

static_assert(sizeof(atlas_g2_affine_t) == 136, "G2Affine image: x = c0 + c1 u, y, infinity flag");

int atlas_g2_generator(atlas_g2_affine_t* out) {
    if (!out) return fail(ATLAS_EINVAL, "g2_generator");
    g2_out(H::g2_generator(), out);
    return ATLAS_OK;
}

int atlas_g2_mul(const atlas_g2_affine_t* p, const atlas_fr_t* k, atlas_g2_affine_t* out) {
    if (!p || !k || !out) return fail(ATLAS_EINVAL, "g2_mul");
    const H::G2Aff P = g2_in(*p);
    if (!H::g2_on_curve(P)) return fail(ATLAS_EINVAL, "g2_mul: point not on the twist");
    uint64_t c[4];
    H::to_canonical(*reinterpret_cast<const H::Fr*>(k), c);
    g2_out(H::g2_mul(P, c), out);
    return ATLAS_OK;
}

int atlas_pairing_check(const atlas_g1_affine_t* p, const atlas_g2_affine_t* q, size_t n, int* is_one) {
    if ((!p || !q) && n) return fail(ATLAS_EINVAL, "pairing_check");
    if (!is_one) return fail(ATLAS_EINVAL, "pairing_check");
    std::vector<H::G1Aff> P(n); std::vector<H::G2Aff> Q(n);
    for (size_t i = 0; i < n; i++) {
        P[i] = g1_in(p[i]); Q[i] = g2_in(q[i]);
        if (!H::g2_on_curve(Q[i])) return fail(ATLAS_EINVAL, "pairing_check: G2 point not on the twist");
    }
    *is_one = H::multi_pairing_is_one(P.data(), Q.data(), n) ? 1 : 0;
    return ATLAS_OK;
}

// HyperKZGVerifierKey for an SRS whose trapdoor is known (tests, benches: atlas_srs_generate): g1 = the SRS's first power,
// g2 = the generator of G2, beta_g2 = tau * g2 — the relation e(L, g2) = e(R, beta_g2) kzg_verify_batch checks
int atlas_hyperkzg_vk_from_trapdoor(const atlas_fr_t* tau, const atlas_g1_affine_t* g1, atlas_hyperkzg_vk_t* out) {
    if (!tau || !g1 || !out) return fail(ATLAS_EINVAL, "hyperkzg_vk_from_trapdoor");
    out->g1 = *g1;
    const H::G2Aff G = H::g2_generator();
    uint64_t c[4];
    H::to_canonical(*reinterpret_cast<const H::Fr*>(tau), c);
    g2_out(G, &out->g2);
    g2_out(H::g2_mul(G, c), &out->beta_g2);
    return ATLAS_OK;
}

int atlas_hyperkzg_verify(const atlas_hyperkzg_vk_t* vk, const atlas_g1_affine_t* C, const atlas_u128_t* point, size_t ell, const atlas_fr_t* y_,
                          const atlas_g1_affine_t* com, const atlas_g1_affine_t* w, const atlas_fr_t* v_, atlas_transcript_t* transcript) {
    if (!vk || !C || (!point && ell) || !y_ || (!com && ell > 1) || !w || !v_ || !transcript || ell == 0) return fail(ATLAS_EINVAL, "hyperkzg_verify: null argument");
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(transcript);
    const H::Fr* v = reinterpret_cast<const H::Fr*>(v_);
    const H::Fr y = *reinterpret_cast<const H::Fr*>(y_);
    const int mode = atlas_rt::rt().challenge_mode;
    // verify_inner (:451-509)
    append_points(T, com, ell - 1);
    const H::Fr r = H::tr_challenge_scalar(T);
    const H::Fr zero = H::zero(), one = H::one();
    if (std::memcmp(&r, &zero, 32) == 0 || C->infinity) return fail(ATLAS_EVERIFY, "hyperkzg_verify: InternalError (r = 0 or C = 0)");
    const H::Fr u[3] = {r, H::neg(r), H::mul(r, r)};
    const H::Fr *ypos = v, *yneg = v + ell, *Yv = v + 2 * ell;
    const H::Fr two_r = H::add(r, r);
    for (size_t i = 0; i < ell; i++) {                                 // consistency of (Y, ypos, yneg)
        const H::Fr Ynext = i + 1 < ell ? Yv[i + 1] : y;
        const H::Fr x = H::challenge_to_fr(point[ell - i - 1].lo, point[ell - i - 1].hi, mode);
        const H::Fr lhs = H::mul(two_r, Ynext);
        const H::Fr rhs = H::add(H::mul(H::mul(r, H::sub(one, x)), H::add(ypos[i], yneg[i])), H::mul(x, H::sub(ypos[i], yneg[i])));
        if (std::memcmp(&lhs, &rhs, 32) != 0) return fail(ATLAS_EVERIFY, "hyperkzg_verify: InternalError (Y / ypos / yneg inconsistent)");
    }
    // kzg_verify_batch (:283-366)
    H::tr_append_scalars(T, v, 3 * ell);
    std::vector<H::Fr> q(ell);
    { const H::Fr q1 = H::tr_challenge_scalar(T); q[0] = one; for (size_t j = 1; j < ell; j++) q[j] = H::mul(q[j - 1], q1); }
    append_points(T, w, 3);
    const H::Fr d0 = H::tr_challenge_scalar(T), d1 = H::mul(d0, d0);
    const H::Fr mult = H::add(H::add(one, d0), d1);
    H::Fr Bu[3];
    for (int k = 0; k < 3; k++) { Bu[k] = zero; for (size_t j = 0; j < ell; j++) Bu[k] = H::add(Bu[k], H::mul(v[k * ell + j], q[j])); }
    // L = sum_j q_j (1 + d0 + d1) com_j + u0 W0 + u1 d0 W1 + u2 d1 W2 - (B(u0) + d0 B(u1) + d1 B(u2)) g1,  com_0 = C
    H::G1X L = H::gx_inf();
    for (size_t j = 0; j < ell; j++) L = H::gx_add(L, g1_mul_fr(g1_in(j == 0 ? *C : com[j - 1]), H::mul(q[j], mult)));
    L = H::gx_add(L, g1_mul_fr(g1_in(w[0]), u[0]));
    L = H::gx_add(L, g1_mul_fr(g1_in(w[1]), H::mul(u[1], d0)));
    L = H::gx_add(L, g1_mul_fr(g1_in(w[2]), H::mul(u[2], d1)));
    L = H::gx_add(L, g1_mul_fr(g1_in(vk->g1), H::neg(H::add(H::add(Bu[0], H::mul(d0, Bu[1])), H::mul(d1, Bu[2])))));
    H::G1X Rp = H::gx_from_aff(g1_in(w[0]));
    Rp = H::gx_add(Rp, g1_mul_fr(g1_in(w[1]), d0));
    Rp = H::gx_add(Rp, g1_mul_fr(g1_in(w[2]), d1));
    H::G1Aff pts[2] = {H::gx_to_aff(L), H::gx_to_aff(Rp)};
    if (!(H::q_is_zero(pts[1].x) && H::q_is_zero(pts[1].y))) pts[1].y = H::q_neg(pts[1].y);      // -R
    const H::G2Aff qs[2] = {g2_in(vk->g2), g2_in(vk->beta_g2)};
    if (!H::g2_on_curve(qs[0]) || !H::g2_on_curve(qs[1])) return fail(ATLAS_EINVAL, "hyperkzg_verify: verifier key not on the twist");
    // e(L, g2) e(-R, beta_g2) == 1
    if (!H::multi_pairing_is_one(pts, qs, 2)) return fail(ATLAS_EVERIFY, "hyperkzg_verify: InternalError (pairing check)");
    return ATLAS_OK;
}

}  // extern "C"
