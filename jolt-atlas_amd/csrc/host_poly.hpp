// Host-side univariate helpers shared by the host-stepped sumcheck instances: the O(degree)
// arithmetic that turns device-reduced sums into the round polynomial.  Restates (paths under
// the jolt-atlas tree, joltworks/src/):
//   UniPoly::from_coeff            poly/unipoly.rs:39-52
//   UniPoly::from_evals_toom       poly/unipoly.rs:103-134 (+ utils/gaussian_elimination.rs)
//   gruen_poly_deg_3               poly/split_eq_poly.rs:379-429
//   finish_mles_product_sum_from_evals   subprotocols/mles_product_sum.rs:330-376
//   GruenSplitEqPolynomial::{new,bind} (LowToHigh) poly/split_eq_poly.rs:97-121,331-348
#pragma once
#include <array>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "host_field.hpp"

namespace atlas_host {

inline void trim(std::vector<Fr>& c) {
    const Fr z = zero();
    while (!c.empty() && c.back() == z) c.pop_back();
    if (c.empty()) c.push_back(z);
}

// unique solution of the n x n system m[i][0..n) * c = m[i][n]
inline std::vector<Fr> gauss_solve(std::vector<std::vector<Fr>>& m) {
    const size_t n = m.size();
    const Fr z = zero();
    for (size_t col = 0; col < n; col++) {
        size_t piv = col;
        while (piv < n && m[piv][col] == z) piv++;
        if (piv != col) std::swap(m[piv], m[col]);
        const Fr iv = inv(m[col][col]);
        for (size_t k = 0; k <= n; k++) m[col][k] = mul(m[col][k], iv);
        for (size_t r = 0; r < n; r++) {
            if (r == col || m[r][col] == z) continue;
            const Fr f = m[r][col];
            for (size_t k = 0; k <= n; k++) m[r][k] = sub(m[r][k], mul(f, m[col][k]));
        }
    }
    std::vector<Fr> c(n);
    for (size_t i = 0; i < n; i++) c[i] = m[i][n];
    return c;
}

// evals on [0, 1, ..., n-2, inf] -> n coefficients.  The system matrix depends on n only, so its
// inverse is computed once per n (Gauss-Jordan on [A | I]) and a call is an n x n matrix-vector
// product; the solution of a non-singular system does not depend on how it is solved.
inline const std::vector<std::vector<Fr>>& toom_inverse(size_t n) {
    static std::mutex mu;
    static std::map<size_t, std::vector<std::vector<Fr>>> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(n);
    if (it != cache.end()) return it->second;
    const Fr z = zero();
    std::vector<std::vector<Fr>> m(n, std::vector<Fr>(2 * n, z));
    for (size_t i = 0; i + 1 < n; i++) {
        const Fr x = from_u64(i);
        m[i][0] = one();
        for (size_t j = 1; j < n; j++) m[i][j] = mul(m[i][j - 1], x);
    }
    m[n - 1][n - 1] = one();
    for (size_t i = 0; i < n; i++) m[i][n + i] = one();
    for (size_t col = 0; col < n; col++) {
        size_t piv = col;
        while (m[piv][col] == z) piv++;
        if (piv != col) std::swap(m[piv], m[col]);
        const Fr iv = inv(m[col][col]);
        for (size_t k = 0; k < 2 * n; k++) m[col][k] = mul(m[col][k], iv);
        for (size_t r = 0; r < n; r++) {
            if (r == col || m[r][col] == z) continue;
            const Fr f = m[r][col];
            for (size_t k = 0; k < 2 * n; k++) m[r][k] = sub(m[r][k], mul(f, m[col][k]));
        }
    }
    std::vector<std::vector<Fr>> invm(n, std::vector<Fr>(n));
    for (size_t i = 0; i < n; i++) for (size_t j = 0; j < n; j++) invm[i][j] = m[i][n + j];
    return cache.emplace(n, std::move(invm)).first->second;
}

inline std::vector<Fr> from_evals_toom_matrix(const std::vector<Fr>& evals) {       // n^2 multiplications: kept as the cross-check of tools/check_host_poly.cpp
    const size_t n = evals.size();
    const auto& M = toom_inverse(n);
    std::vector<Fr> c(n, zero());
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) c[i] = add(c[i], mul(M[i][j], evals[j]));
    return c;
}

// a * k for a small integer k (k < 64): the five-limb product minus j p, j estimated from the bits above 2^244 (p / 2^244 = 774.3: the
// estimate is low by at most one), then one conditional subtraction.  ~1/3 of a Montgomery multiplication; same residue.
inline Fr mul_small(const Fr& a, uint32_t k) {
    static const std::vector<std::array<uint64_t, 5>> KP = [] {
        std::vector<std::array<uint64_t, 5>> t(65);
        t[0] = {0, 0, 0, 0, 0};
        for (size_t j = 1; j <= 64; j++) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)t[j - 1][i] + FR_P[i]; t[j][i] = (uint64_t)c; c >>= 64; } t[j][4] = t[j - 1][4] + (uint64_t)c; }
        return t;
    }();
    uint64_t t[5]; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] * k; t[i] = (uint64_t)c; c >>= 64; }
    t[4] = (uint64_t)c;
    const uint64_t top = (t[4] << 12) | (t[3] >> 52);
    const uint64_t* kp = KP[top / 775].data();
    u128 br = 0;
    for (int i = 0; i < 5; i++) { const u128 d = (u128)t[i] - kp[i] - br; t[i] = (uint64_t)d; br = (d >> 64) & 1; }
    Fr o{{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_p(o.l)) sub_p(o.l);
    if (geq_p(o.l)) sub_p(o.l);
    return o;
}

// The same coefficients in O(n) multiplications: with c = the value at infinity (the leading coefficient, degree D = n - 1),
// q(x) = p(x) - c x^D has degree < D and is known at 0 .. D - 1; its forward differences at 0 divided by k! are its coefficients in the
// falling-factorial basis, and Horner over (x - k) with the small integers k turns that basis into monomials.  Against the n x n
// matrix-vector product (289 multiplications for the degree-17 round polynomial of a 64-bit one-hot check, once per round on the host's
// critical path): 2 (n - 3) multiplications, n^2 / 2 small ones and n^2 additions.  The interpolant is unique: same coefficients.
inline std::vector<Fr> from_evals_toom(const std::vector<Fr>& evals) {
    const size_t n = evals.size();
    if (n < 3 || n > 64) return from_evals_toom_matrix(evals);
    const size_t D = n - 1;
    struct Consts { std::vector<Fr> powD, invfact; };
    static std::mutex mu;
    static std::map<size_t, Consts> cache;
    const Consts* K;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(n);
        if (it == cache.end()) {
            Consts c;
            c.powD.resize(D); c.invfact.resize(D);
            Fr f = one();
            for (size_t i = 0; i < D; i++) {
                Fr pw = one(); const Fr x = from_u64(i);
                for (size_t j = 0; j < D; j++) pw = mul(pw, x);
                c.powD[i] = pw;
                if (i) f = mul(f, x);
                c.invfact[i] = inv(f);
            }
            it = cache.emplace(n, std::move(c)).first;
        }
        K = &it->second;
    }
    const Fr& lead = evals[D];
    Fr d[64];
    d[0] = evals[0]; d[1] = sub(evals[1], lead);
    for (size_t i = 2; i < D; i++) d[i] = sub(evals[i], mul(lead, K->powD[i]));
    for (size_t k = 1; k < D; k++)                                   // after pass k: d[i] = Delta^k q(i - k) for i >= k; d[k] = Delta^k q(0) stays
        for (size_t i = D - 1; i >= k; i--) d[i] = sub(d[i], d[i - 1]);
    for (size_t k = 2; k < D; k++) d[k] = mul(d[k], K->invfact[k]);
    std::vector<Fr> c(n, zero());
    size_t len = 1;                                                   // c[0 .. len) = the polynomial built so far, coefficients in ASCENDING order (c[0] = the constant term)
    c[0] = d[D - 1];
    for (size_t k = D - 1; k-- > 0;) {                                // poly = poly * (x - k) + d[k]
        c[len] = c[len - 1];
        for (size_t j = len - 1; j >= 1; j--) c[j] = sub(c[j - 1], mul_small(c[j], (uint32_t)k));
        c[0] = sub(d[k], mul_small(c[0], (uint32_t)k));
        len++;
    }
    c[D] = lead;
    return c;
}

// LowToHigh GruenSplitEqPolynomial bookkeeping that lives on the host: w, current_scalar, the
// table tops.  The prefix tables themselves are wherever the instance keeps them.
struct GseState {
    std::vector<Fr> w;
    size_t n = 0, m = 0, k_out = 0, k_in = 0, out_top = 0, in_top = 0, current_index = 0;
    Fr scalar = one();
    void init(const Fr* w_, size_t n_) {
        w.assign(w_, w_ + n_);
        n = n_; m = n / 2; k_out = m; k_in = n ? n - 1 - m : 0;
        out_top = k_out; in_top = k_in; current_index = n; scalar = one();
    }
    const Fr& w_cur() const { return w[current_index - 1]; }
    void bind(const Fr& r) {
        const Fr& wc = w_cur();
        const Fr wr = mul(wc, r);
        scalar = mul(scalar, add(add(sub(sub(one(), wc), r), wr), wr));
        current_index -= 1;
        if (n / 2 < current_index && in_top > 0) in_top--;
        else if (0 < current_index && out_top > 0) out_top--;
        have_inv_eq1 = have_inv_eq0w = false;
    }
    // table tops after k binds (a function of k only: the launches of every round can be enqueued up front)
    void tops_after(size_t k, size_t& ot, size_t& it) const {
        ot = this->k_out; it = this->k_in;
        size_t ci = n;
        for (size_t j = 0; j < k; j++) {
            ci -= 1;
            if (n / 2 < ci && it > 0) it--;
            else if (0 < ci && ot > 0) ot--;
        }
    }
    // Gruen's division (split_eq_poly.rs:410-413) needs 1 / (scalar w_cur) [and the product-sum finish 1 / (1 - w_cur)];
    // the operand is known as soon as the previous challenge is: computing it before the round's sums arrive takes the
    // 8 us inversion off the critical path
    mutable Fr inv_eq1, inv_eq0w;
    mutable bool have_inv_eq1 = false, have_inv_eq0w = false;
    void prepare_inverses(bool one_minus_w) const {
        if (current_index == 0) return;
        if (!have_inv_eq1) { inv_eq1 = inv(mul(scalar, w_cur())); have_inv_eq1 = true; }
        if (one_minus_w && !have_inv_eq0w) { inv_eq0w = inv(sub(one(), w_cur())); have_inv_eq0w = true; }
    }
    const Fr& get_inv_eq1() const { if (!have_inv_eq1) { inv_eq1 = inv(mul(scalar, w_cur())); have_inv_eq1 = true; } return inv_eq1; }
    const Fr& get_inv_eq0w() const { if (!have_inv_eq0w) { inv_eq0w = inv(sub(one(), w_cur())); have_inv_eq0w = true; } return inv_eq0w; }
};

// 4 coefficients (fixed length, UniPoly::from_evals of 4 points)
inline void gruen_deg3(const GseState& S, const Fr& q0, const Fr& qinf, const Fr& claim, Fr coeffs[4]) {
    const Fr eq1 = mul(S.scalar, S.w_cur()), eq0 = sub(S.scalar, eq1), eqm = sub(eq1, eq0);
    const Fr eq2 = add(eq1, eqm), eq3 = add(eq2, eqm);
    const Fr c0 = mul(eq0, q0), c1 = sub(claim, c0);
    const Fr q1 = mul(c1, S.get_inv_eq1());
    const Fr e2 = add(qinf, qinf);
    const Fr q2 = add(sub(add(q1, q1), q0), e2);
    const Fr q3 = add(add(sub(add(q2, q1), q0), e2), e2);
    const Fr ev[3] = {c0, mul(eq2, q2), mul(eq3, q3)};
    unipoly_from_evals_and_hint(add(c0, c1), ev, 3, coeffs);
}

// sum_evals = h on [1, ..., d-1, inf] (already scaled by current_scalar) -> g = eq(X, w_cur) * h
inline std::vector<Fr> finish_product_sum(const std::vector<Fr>& sum_evals, const Fr& claim, const GseState& S) {
    const size_t d = sum_evals.size();
    const Fr r = S.w_cur();
    const Fr eq0 = sub(one(), r);
    Fr e0 = sub(claim, mul(r, sum_evals[0]));
    if (d > 1) e0 = mul(e0, S.get_inv_eq0w());
    std::vector<Fr> toom(d + 1);
    toom[0] = e0;
    for (size_t k = 0; k < d; k++) toom[k + 1] = sum_evals[k];
    const std::vector<Fr> tmp = from_evals_toom(toom);
    const Fr xc = sub(add(r, r), one());
    std::vector<Fr> c(d + 2, zero());
    for (size_t i = 0; i < d + 1; i++) {
        c[i] = add(c[i], mul(tmp[i], eq0));
        c[i + 1] = add(c[i + 1], mul(tmp[i], xc));
    }
    trim(c);
    return c;
}

// host prefix tables of w[0..k) (EqPolynomial::evals_cached, eq_poly.rs:174-192)
inline std::vector<std::vector<Fr>> eq_cached(const Fr* w, size_t k) {
    std::vector<std::vector<Fr>> t(k + 1);
    t[0] = {one()};
    for (size_t j = 0; j < k; j++) {
        t[j + 1].resize((size_t)2 << j);
        for (size_t i = 0; i < ((size_t)1 << j); i++) {
            const Fr hi = mul(t[j][i], w[j]);
            t[j + 1][2 * i + 1] = hi;
            t[j + 1][2 * i] = sub(t[j][i], hi);
        }
    }
    return t;
}

// EqPolynomial::evals, big-endian index (eq_poly.rs:77-101)
inline std::vector<Fr> eq_evals(const Fr* r, size_t n) { return eq_cached(r, n)[n]; }

// gruen_poly_deg_2 over a LowToHigh state (cached inverse)
inline void gruen_deg2(const GseState& S, const Fr& q0, const Fr& claim, Fr coeffs[3]) {
    const Fr eq1 = mul(S.scalar, S.w_cur()), eq0 = sub(S.scalar, eq1), eqm = sub(eq1, eq0), eq2 = add(eq1, eqm);
    const Fr c0 = mul(eq0, q0), c1 = sub(claim, c0);
    const Fr l1 = mul(c1, S.get_inv_eq1());
    const Fr l2 = sub(add(l1, l1), q0);
    const Fr ev[2] = {c0, mul(eq2, l2)};
    unipoly_from_evals_and_hint(add(c0, c1), ev, 2, coeffs);
}

}  // namespace atlas_host

namespace atlas_host {

// HighToLow GruenSplitEqPolynomial bookkeeping (split_eq_poly.rs:121-145, 349-372):
// w = [w_first | w_in (n/2) | w_out (rest)], current_index counts bound variables.
struct GseStateH {
    std::vector<Fr> w;
    size_t n = 0, k_in = 0, k_out = 0, in_top = 0, out_top = 0, current_index = 0;
    Fr scalar = one();
    void init(const Fr* w_, size_t n_) {
        w.assign(w_, w_ + n_);
        n = n_;
        const size_t m = n / 2;
        k_in = n ? (m < n - 1 ? m : n - 1) : 0;
        k_out = n ? n - 1 - k_in : 0;
        in_top = k_in; out_top = k_out; current_index = 0; scalar = one();
    }
    const Fr& w_cur() const { return w[current_index]; }
    void bind(const Fr& r) {
        const Fr& wc = w_cur();
        const Fr wr = mul(wc, r);
        scalar = mul(scalar, add(add(sub(sub(one(), wc), r), wr), wr));
        current_index += 1;
        if (current_index <= n / 2 && in_top > 0) in_top--;
        else if (current_index <= n && out_top > 0) out_top--;
    }
};

// gruen_poly_deg_2 (split_eq_poly.rs:379-428): 3 coefficients, fixed length
inline void gruen_deg2(const Fr& scalar, const Fr& w_cur, const Fr& q0, const Fr& claim, Fr coeffs[3]) {
    const Fr eq1 = mul(scalar, w_cur), eq0 = sub(scalar, eq1), eqm = sub(eq1, eq0), eq2 = add(eq1, eqm);
    const Fr c0 = mul(eq0, q0), c1 = sub(claim, c0);
    const Fr l1 = mul(c1, inv(eq1));
    const Fr l2 = sub(add(l1, l1), q0);
    const Fr ev[2] = {c0, mul(eq2, l2)};
    unipoly_from_evals_and_hint(add(c0, c1), ev, 2, coeffs);
}

}  // namespace atlas_host
