// Host-side BN254 pairing for the verifier: HyperKZG::verify ends in
//     P::multi_pairing([L, -R], [vk.g2, vk.beta_g2]).is_zero()        (joltworks/src/poly/commitment/hyperkzg/mod.rs:365)
// whose arithmetic lives in arkworks (external to the reference tree).  The check only asks whether a product of pairings is the
// identity of G_T, so any bilinear non-degenerate pairing on (G1, G2) decides it identically; this is the ate pairing with Miller
// loop length t - 1 = 6 x^2 over the sextic D-type twist E'(Fq2): y^2 = x^3 + 3 / (9 + u), affine line functions (a verifier runs two
// Miller loops per opening proof: clarity over speed, ~10 ms), final exponentiation by square-and-multiply.
// Tower: Fq2 = Fq[u]/(u^2 + 1), Fq6 = Fq2[v]/(v^3 - xi), xi = 9 + u, Fq12 = Fq6[w]/(w^2 - v).  Checked against the big-integer model
// oracle/pymodel/pairing.py (bilinearity, non-degeneracy, identical G_T elements) in tests/test_verify.py.
#pragma once
#include <vector>

#include "host_curve.hpp"

namespace atlas_host {

struct Fq2 { Fq c0, c1; };
inline Fq2 f2_zero() { return Fq2{q_zero(), q_zero()}; }
inline Fq2 f2_one() { return Fq2{q_one(), q_zero()}; }
inline bool f2_is_zero(const Fq2& a) { return q_is_zero(a.c0) && q_is_zero(a.c1); }
inline bool f2_eq(const Fq2& a, const Fq2& b) { return q_eq(a.c0, b.c0) && q_eq(a.c1, b.c1); }
inline Fq2 f2_add(const Fq2& a, const Fq2& b) { return Fq2{q_add(a.c0, b.c0), q_add(a.c1, b.c1)}; }
inline Fq2 f2_sub(const Fq2& a, const Fq2& b) { return Fq2{q_sub(a.c0, b.c0), q_sub(a.c1, b.c1)}; }
inline Fq2 f2_neg(const Fq2& a) { return Fq2{q_neg(a.c0), q_neg(a.c1)}; }
inline Fq2 f2_mul(const Fq2& a, const Fq2& b) {          // (a0 + a1 u)(b0 + b1 u), u^2 = -1 (Karatsuba)
    const Fq t0 = q_mul(a.c0, b.c0), t1 = q_mul(a.c1, b.c1);
    return Fq2{q_sub(t0, t1), q_sub(q_sub(q_mul(q_add(a.c0, a.c1), q_add(b.c0, b.c1)), t0), t1)};
}
inline Fq2 f2_sqr(const Fq2& a) { return f2_mul(a, a); }
inline Fq2 f2_scal(const Fq2& a, const Fq& k) { return Fq2{q_mul(a.c0, k), q_mul(a.c1, k)}; }
inline Fq2 f2_dbl(const Fq2& a) { return f2_add(a, a); }
inline Fq2 f2_inv(const Fq2& a) {
    const Fq n = q_inv(q_add(q_sqr(a.c0), q_sqr(a.c1)));
    return Fq2{q_mul(a.c0, n), q_neg(q_mul(a.c1, n))};
}
inline Fq2 f2_mul_xi(const Fq2& a) {                      // (9 + u) a
    const Fq nine = q_from_u64(9);
    return Fq2{q_sub(q_mul(a.c0, nine), a.c1), q_add(q_mul(a.c1, nine), a.c0)};
}

struct Fq6 { Fq2 c0, c1, c2; };
inline Fq6 f6_zero() { return Fq6{f2_zero(), f2_zero(), f2_zero()}; }
inline Fq6 f6_one() { return Fq6{f2_one(), f2_zero(), f2_zero()}; }
inline Fq6 f6_add(const Fq6& a, const Fq6& b) { return Fq6{f2_add(a.c0, b.c0), f2_add(a.c1, b.c1), f2_add(a.c2, b.c2)}; }
inline Fq6 f6_sub(const Fq6& a, const Fq6& b) { return Fq6{f2_sub(a.c0, b.c0), f2_sub(a.c1, b.c1), f2_sub(a.c2, b.c2)}; }
inline Fq6 f6_neg(const Fq6& a) { return Fq6{f2_neg(a.c0), f2_neg(a.c1), f2_neg(a.c2)}; }
inline Fq6 f6_mul(const Fq6& a, const Fq6& b) {
    Fq6 o;
    o.c0 = f2_add(f2_mul(a.c0, b.c0), f2_mul_xi(f2_add(f2_mul(a.c1, b.c2), f2_mul(a.c2, b.c1))));
    o.c1 = f2_add(f2_add(f2_mul(a.c0, b.c1), f2_mul(a.c1, b.c0)), f2_mul_xi(f2_mul(a.c2, b.c2)));
    o.c2 = f2_add(f2_add(f2_mul(a.c0, b.c2), f2_mul(a.c1, b.c1)), f2_mul(a.c2, b.c0));
    return o;
}
inline Fq6 f6_mul_v(const Fq6& a) { return Fq6{f2_mul_xi(a.c2), a.c0, a.c1}; }
inline Fq6 f6_inv(const Fq6& a) {
    const Fq2 c0 = f2_sub(f2_sqr(a.c0), f2_mul_xi(f2_mul(a.c1, a.c2)));
    const Fq2 c1 = f2_sub(f2_mul_xi(f2_sqr(a.c2)), f2_mul(a.c0, a.c1));
    const Fq2 c2 = f2_sub(f2_sqr(a.c1), f2_mul(a.c0, a.c2));
    const Fq2 t = f2_add(f2_mul(a.c0, c0), f2_mul_xi(f2_add(f2_mul(a.c2, c1), f2_mul(a.c1, c2))));
    const Fq2 ti = f2_inv(t);
    return Fq6{f2_mul(c0, ti), f2_mul(c1, ti), f2_mul(c2, ti)};
}

struct Fq12 { Fq6 c0, c1; };
inline Fq12 f12_one() { return Fq12{f6_one(), f6_zero()}; }
inline bool f12_is_one(const Fq12& a) {
    const Fq12 o = f12_one();
    return f2_eq(a.c0.c0, o.c0.c0) && f2_is_zero(a.c0.c1) && f2_is_zero(a.c0.c2) && f2_is_zero(a.c1.c0) && f2_is_zero(a.c1.c1) && f2_is_zero(a.c1.c2);
}
inline Fq12 f12_mul(const Fq12& a, const Fq12& b) {
    const Fq6 t0 = f6_mul(a.c0, b.c0), t1 = f6_mul(a.c1, b.c1);
    return Fq12{f6_add(t0, f6_mul_v(t1)), f6_add(f6_mul(a.c0, b.c1), f6_mul(a.c1, b.c0))};
}
inline Fq12 f12_conj(const Fq12& a) { return Fq12{a.c0, f6_neg(a.c1)}; }
inline Fq12 f12_inv(const Fq12& a) {
    const Fq6 t = f6_inv(f6_sub(f6_mul(a.c0, a.c0), f6_mul_v(f6_mul(a.c1, a.c1))));
    return Fq12{f6_mul(a.c0, t), f6_neg(f6_mul(a.c1, t))};
}
// a^e, e little-endian 64-bit words
inline Fq12 f12_pow(const Fq12& a, const uint64_t* e, int words) {
    Fq12 r = f12_one();
    bool started = false;
    for (int i = words * 64 - 1; i >= 0; i--) {
        if (started) r = f12_mul(r, r);
        if ((e[i >> 6] >> (i & 63)) & 1) { r = started ? f12_mul(r, a) : a; started = true; }
    }
    return r;
}

// ---- G2 on the twist, affine; inf flag
struct G2Aff { Fq2 x, y; bool inf; };
inline G2Aff g2_inf() { return G2Aff{f2_zero(), f2_zero(), true}; }
inline Fq2 g2_b() { return f2_mul(Fq2{q_from_u64(3), q_zero()}, f2_inv(Fq2{q_from_u64(9), q_one()})); }
inline bool g2_on_curve(const G2Aff& Q) { return Q.inf || f2_eq(f2_sqr(Q.y), f2_add(f2_mul(f2_sqr(Q.x), Q.x), g2_b())); }
inline G2Aff g2_add(const G2Aff& A, const G2Aff& B) {
    if (A.inf) return B;
    if (B.inf) return A;
    Fq2 lam;
    if (f2_eq(A.x, B.x)) {
        if (f2_is_zero(f2_add(A.y, B.y))) return g2_inf();
        const Fq2 xx = f2_sqr(A.x);
        lam = f2_mul(f2_add(f2_dbl(xx), xx), f2_inv(f2_dbl(A.y)));
    } else lam = f2_mul(f2_sub(B.y, A.y), f2_inv(f2_sub(B.x, A.x)));
    G2Aff o; o.inf = false;
    o.x = f2_sub(f2_sub(f2_sqr(lam), A.x), B.x);
    o.y = f2_sub(f2_mul(lam, f2_sub(A.x, o.x)), A.y);
    return o;
}
// k * Q, k = 4 little-endian words (canonical integer)
inline G2Aff g2_mul(const G2Aff& Q, const uint64_t k[4]) {
    G2Aff acc = g2_inf();
    for (int i = 255; i >= 0; i--) {
        acc = g2_add(acc, acc);
        if ((k[i >> 6] >> (i & 63)) & 1) acc = g2_add(acc, Q);
    }
    return acc;
}
inline Fq q_from_words(const uint64_t w[4]) { Fq t{{w[0], w[1], w[2], w[3]}}; Fq r2{{FQ_R2[0], FQ_R2[1], FQ_R2[2], FQ_R2[3]}}; return q_mul(t, r2); }
inline G2Aff g2_generator() {           // the arkworks / EIP-197 generator of the order-r subgroup of E'(Fq2)
    static const uint64_t X0[4] = {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL};
    static const uint64_t X1[4] = {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL};
    static const uint64_t Y0[4] = {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL};
    static const uint64_t Y1[4] = {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL};
    return G2Aff{Fq2{q_from_words(X0), q_from_words(X1)}, Fq2{q_from_words(Y0), q_from_words(Y1)}, false};
}

// the line through T with twist slope lam, at P = (xP, yP) of E(Fq):  yP - lam xP w + (lam x_T - y_T) w^3,  w^3 = v w
inline Fq12 pairing_line(const G2Aff& T, const Fq2& lam, const G1Aff& P) {
    Fq12 l;
    l.c0 = Fq6{Fq2{P.y, q_zero()}, f2_zero(), f2_zero()};
    l.c1 = Fq6{f2_neg(f2_scal(lam, P.x)), f2_sub(f2_mul(lam, T.x), T.y), f2_zero()};
    return l;
}
// f_{t-1, Q}(P); a point at infinity on either side gives 1
inline Fq12 miller_loop(const G1Aff& P, const G2Aff& Q) {
    if (Q.inf || (q_is_zero(P.x) && q_is_zero(P.y))) return f12_one();
    // t - 1 = 6 x^2, x = 4965661367192848881
    static const uint64_t LOOP[2] = {0xf83e9682e87cfd46ULL, 0x6f4d8248eeb859fbULL};   // 6 * 0x44E992B44A6909F1^2, little-endian words
    Fq12 f = f12_one();
    G2Aff T = Q;
    bool started = false;
    for (int i = 127; i >= 0; i--) {
        const bool bit = (LOOP[i >> 6] >> (i & 63)) & 1;
        if (!started) { started = bit; continue; }
        const Fq2 xx = f2_sqr(T.x);
        Fq2 lam = f2_mul(f2_add(f2_dbl(xx), xx), f2_inv(f2_dbl(T.y)));
        f = f12_mul(f12_mul(f, f), pairing_line(T, lam, P));
        T = g2_add(T, T);
        if (bit) {
            lam = f2_mul(f2_sub(Q.y, T.y), f2_inv(f2_sub(Q.x, T.x)));
            f = f12_mul(f, pairing_line(T, lam, P));
            T = g2_add(T, Q);
        }
    }
    return f;
}
// ^((p^12 - 1) / r) = ^(p^6 - 1) ^(p^2 + 1) ^((p^4 - p^2 + 1) / r)
inline Fq12 final_exponentiation(const Fq12& f0) {
    static const uint64_t P2[8] = {0x3b5458a2275d69b1ULL, 0xa602072d09eac101ULL, 0x4a50189c6d96cadcULL, 0x04689e957a1242c8ULL,
                                   0x26edfa5c34c6b38dULL, 0xb00b855116375606ULL, 0x599a6f7c0348d21cULL, 0x0925c4b8763cbf9cULL};          // p^2
    static const uint64_t HARD[12] = {0xe81bb482ccdf42b1ULL, 0x5abf5cc4f49c36d4ULL, 0xf1154e7e1da014fdULL, 0xdcc7b44c87cdbacfULL,
                                      0xaaa441e3954bcf8aULL, 0x6b887d56d5095f23ULL, 0x79581e16f3fd90c6ULL, 0x3b1b1355d189227dULL,
                                      0x4e529a5861876f6bULL, 0x6c0eb522d5b12278ULL, 0x331ec15183177fafULL, 0x01baaa710b0759adULL};      // (p^4 - p^2 + 1) / r
    Fq12 f = f12_mul(f12_conj(f0), f12_inv(f0));
    f = f12_mul(f12_pow(f, P2, 8), f);
    return f12_pow(f, HARD, 12);
}
// prod e(P_i, Q_i) == 1 ?
inline bool multi_pairing_is_one(const G1Aff* P, const G2Aff* Q, size_t n) {
    Fq12 f = f12_one();
    for (size_t i = 0; i < n; i++) f = f12_mul(f, miller_loop(P[i], Q[i]));
    return f12_is_one(final_exponentiation(f));
}

}  // namespace atlas_host
