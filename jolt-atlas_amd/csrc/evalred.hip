// N-to-1 evaluation reduction: EvalReductionInstance::prove, compute_h, eval_on_l
// (joltworks/src/subprotocols/evaluation_reduction.rs:91-147, 213-249).
//
// The reference builds h(t) = P(l(t)) by folding 2^n univariate polynomials on one thread
// (O(2^n * n * N) coefficient operations).  Here h is recovered from its values: deg h <= n (N-1),
// so D + 1 = n (N-1) + 1 multilinear evaluations P(l(0)), ..., P(l(D)) determine it — each one is
// the split-eq evaluation kernel over the device-resident coefficients (atlas_poly_evaluate), and the
// O(D^2) Newton interpolation back to monomial coefficients is host arithmetic.  The coefficient
// vector is the exact polynomial with trailing zeros trimmed, which is what the reference's
// Add / Mul / from_coeff chain yields (unipoly.rs:415-476).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_poly.hpp"
#include "runtime.hpp"

namespace H = atlas_host;
using atlas_rt::fail;
using atlas_rt::g;

namespace {

// coefficients of the polynomial through (0, e0), ..., (D, eD): Newton forward differences, then
// expansion of the Newton basis prod_{j<k} (x - j)
std::vector<H::Fr> interpolate_consecutive(const std::vector<H::Fr>& ev) {
    const size_t n = ev.size();
    std::vector<H::Fr> dd = ev;                       // dd[k] -> Delta^k f(0) / k!
    for (size_t k = 1; k < n; k++) {
        for (size_t i = n - 1; i >= k; i--) dd[i] = H::sub(dd[i], dd[i - 1]);
    }
    H::Fr fact = H::one();
    for (size_t k = 1; k < n; k++) { fact = H::mul(fact, H::from_u64(k)); dd[k] = H::mul(dd[k], H::inv(fact)); }
    std::vector<H::Fr> c = {dd[n - 1]};
    for (size_t k = n - 1; k-- > 0;) {                // c <- c * (x - k) + dd[k]
        const H::Fr mk = H::from_u64(k);
        std::vector<H::Fr> nx(c.size() + 1);
        for (size_t i = 0; i <= c.size(); i++) {
            const H::Fr up = i > 0 ? c[i - 1] : H::zero();
            const H::Fr dn = i < c.size() ? H::mul(mk, c[i]) : H::zero();
            nx[i] = H::sub(up, dn);
        }
        nx[0] = H::add(nx[0], dd[k]);
        c.swap(nx);
    }
    return c;
}

H::Fr horner(const std::vector<H::Fr>& c, const H::Fr& x) {
    H::Fr acc = H::zero();
    for (size_t i = c.size(); i-- > 0;) acc = H::add(H::mul(acc, x), c[i]);
    return acc;
}

}  // namespace

extern "C" int atlas_eval_reduction_prove(atlas_poly_t mle, const atlas_fr_t* points, const atlas_fr_t* claims, size_t N,
                                          size_t n, atlas_transcript_t* transcript, atlas_fr_t* h_out, size_t h_cap,
                                          size_t* h_len, atlas_fr_t* r_out, atlas_fr_t* claim_out) {
    NEED_INIT();
    if (!mle || !points || !claims || !transcript || !h_out || !h_len || !r_out || !claim_out) return fail(ATLAS_EINVAL, "eval_reduction: null argument");
    if (N == 0) return fail(ATLAS_EINVAL, "eval_reduction: EmptyInput");
    size_t len = 0;
    atlas_poly_len(mle, &len);
    if (len != ((size_t)1 << n)) return fail(ATLAS_EINVAL, "eval_reduction: InvalidInputLength (point length != num_vars)");
    const H::Fr* pts = reinterpret_cast<const H::Fr*>(points);
    if (N == 1) {                                                   // short path, evaluation_reduction.rs:113-127
        if (h_cap < 1) return fail(ATLAS_EINVAL, "eval_reduction: h buffer too small");
        std::memcpy(h_out, claims, 32); *h_len = 1;
        std::memcpy(r_out, points, n * 32); std::memcpy(claim_out, claims, 32);
        return ATLAS_OK;
    }
    const size_t D = n * (N - 1);
    if (h_cap < D + 1) return fail(ATLAS_EINVAL, "eval_reduction: h buffer too small");
    // l: per-variable interpolant through (j, points[j][i])  (group_by_variable + UniPoly::from_evals)
    std::vector<std::vector<H::Fr>> var(n);
    for (size_t i = 0; i < n; i++) {
        std::vector<H::Fr> ev(N);
        for (size_t j = 0; j < N; j++) ev[j] = pts[j * n + i];
        var[i] = interpolate_consecutive(ev);
    }
    // h on 0..D by multilinear evaluation on the device
    std::vector<H::Fr> hev(D + 1), lt(n ? n : 1);
    for (size_t t = 0; t <= D; t++) {
        const H::Fr x = H::from_u64(t);
        for (size_t i = 0; i < n; i++) lt[i] = horner(var[i], x);
        int rc = atlas_poly_evaluate(mle, (const atlas_fr_t*)lt.data(), n, (atlas_fr_t*)&hev[t]);
        if (rc) return rc;
    }
    std::vector<H::Fr> h = interpolate_consecutive(hev);
    H::trim(h);
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(transcript);
    H::tr_append_message(T, "UncompressedUniPoly_begin");           // unipoly.rs:540-548
    for (auto& c : h) H::tr_append_scalar(T, c);
    H::tr_append_message(T, "UncompressedUniPoly_end");
    uint64_t lo, hi;
    H::tr_challenge_u128(T, lo, hi);
    const H::Fr xp = H::challenge_to_fr(lo, hi, g.challenge_mode);
    for (size_t i = 0; i < n; i++) { const H::Fr v = horner(var[i], xp); std::memcpy(&r_out[i], &v, 32); }   // eval_on_l
    const H::Fr v = horner(h, xp);
    std::memcpy(claim_out, &v, 32);
    std::memcpy(h_out, h.data(), h.size() * 32);
    *h_len = h.size();
    return ATLAS_OK;
}
