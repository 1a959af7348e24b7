// Host-side BN254 Fq and G1 (XYZZ) — only for the O(windows) tail of an MSM (Horner over
// the per-window sums, one inversion to affine) and for small fixed tables.  Product code;
// shares nothing with oracle/.
#pragma once
#include <cstdint>
#include <cstring>

namespace atlas_host {

typedef unsigned __int128 u128;

struct Fq {
    uint64_t l[4];
};

static const uint64_t FQ_P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL,
                                 0x30644e72e131a029ULL};
static const uint64_t FQ_R[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL,
                                 0x0e0a77c19a07df2fULL};
static const uint64_t FQ_R2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL,
                                  0x06d89f71cab8351fULL};
static const uint64_t FQ_INV = 0x87d20782e4866389ULL;

inline bool q_geq_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) if (a[i] != FQ_P[i]) return a[i] > FQ_P[i];
    return true;
}
inline void q_sub_p(uint64_t a[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - FQ_P[i] - br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
inline Fq q_zero() { return Fq{{0, 0, 0, 0}}; }
inline Fq q_one() { return Fq{{FQ_R[0], FQ_R[1], FQ_R[2], FQ_R[3]}}; }
inline bool q_is_zero(const Fq& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
inline bool q_eq(const Fq& a, const Fq& b) { return std::memcmp(a.l, b.l, 32) == 0; }
inline Fq q_add(const Fq& a, const Fq& b) {
    Fq o; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; o.l[i] = (uint64_t)c; c >>= 64; }
    if (q_geq_p(o.l)) q_sub_p(o.l);
    return o;
}
inline Fq q_sub(const Fq& a, const Fq& b) {
    Fq o; u128 br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a.l[i] - b.l[i] - br; o.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)o.l[i] + FQ_P[i]; o.l[i] = (uint64_t)c; c >>= 64; } }
    return o;
}
inline Fq q_neg(const Fq& a) { return q_is_zero(a) ? a : q_sub(q_zero(), a); }
inline Fq q_mul(const Fq& a, const Fq& b) {
    uint64_t t[9] = {0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.l[i] * b.l[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
        t[i + 4] = (uint64_t)c;
    }
    for (int i = 0; i < 4; i++) {
        uint64_t m = t[i] * FQ_INV; u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)m * FQ_P[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
        for (int k = i + 4; c && k < 9; k++) { c += t[k]; t[k] = (uint64_t)c; c >>= 64; }
    }
    Fq o{{t[4], t[5], t[6], t[7]}};
    if (t[8] || q_geq_p(o.l)) q_sub_p(o.l);
    return o;
}
inline Fq q_sqr(const Fq& a) { return q_mul(a, a); }
inline Fq q_dbl(const Fq& a) { return q_add(a, a); }
inline Fq q_inv(const Fq& a) {
    uint64_t e[4] = {FQ_P[0] - 2, FQ_P[1], FQ_P[2], FQ_P[3]};
    Fq acc = q_one(), base = a;
    for (int i = 0; i < 256; i++) { if ((e[i >> 6] >> (i & 63)) & 1) acc = q_mul(acc, base); base = q_sqr(base); }
    return acc;
}
inline Fq q_from_u64(uint64_t v) { Fq t{{v, 0, 0, 0}}; Fq r2{{FQ_R2[0], FQ_R2[1], FQ_R2[2], FQ_R2[3]}}; return q_mul(t, r2); }

struct G1Aff { Fq x, y; };                 // (0,0) = infinity (same image as the device)
struct G1X { Fq x, y, zz, zzz; };          // XYZZ, zz == 0 = infinity

inline G1X gx_inf() { return G1X{q_zero(), q_zero(), q_zero(), q_zero()}; }
inline bool gx_is_inf(const G1X& p) { return q_is_zero(p.zz); }
inline G1X gx_dbl(const G1X& p) {
    if (gx_is_inf(p)) return p;
    Fq U = q_dbl(p.y), V = q_sqr(U), W = q_mul(U, V), S = q_mul(p.x, V), xx = q_sqr(p.x);
    Fq M = q_add(q_dbl(xx), xx);
    G1X o;
    o.x = q_sub(q_sqr(M), q_dbl(S));
    o.y = q_sub(q_mul(M, q_sub(S, o.x)), q_mul(W, p.y));
    o.zz = q_mul(V, p.zz); o.zzz = q_mul(W, p.zzz);
    return o;
}
inline G1X gx_add(const G1X& p, const G1X& q) {
    if (gx_is_inf(p)) return q;
    if (gx_is_inf(q)) return p;
    Fq U1 = q_mul(p.x, q.zz), U2 = q_mul(q.x, p.zz), S1 = q_mul(p.y, q.zzz), S2 = q_mul(q.y, p.zzz);
    Fq P = q_sub(U2, U1), R = q_sub(S2, S1);
    if (q_is_zero(P)) return q_is_zero(R) ? gx_dbl(p) : gx_inf();
    Fq PP = q_sqr(P), PPP = q_mul(P, PP), Q = q_mul(U1, PP);
    G1X o;
    o.x = q_sub(q_sub(q_sqr(R), PPP), q_dbl(Q));
    o.y = q_sub(q_mul(R, q_sub(Q, o.x)), q_mul(S1, PPP));
    o.zz = q_mul(q_mul(p.zz, q.zz), PP); o.zzz = q_mul(q_mul(p.zzz, q.zzz), PPP);
    return o;
}
inline G1X gx_from_aff(const G1Aff& a) {
    if (q_is_zero(a.x) && q_is_zero(a.y)) return gx_inf();
    return G1X{a.x, a.y, q_one(), q_one()};
}
inline G1Aff gx_to_aff(const G1X& p) {
    if (gx_is_inf(p)) return G1Aff{q_zero(), q_zero()};
    Fq izzz = q_inv(p.zzz);
    Fq izz = q_sqr(q_mul(izzz, p.zz));
    return G1Aff{q_mul(p.x, izz), q_mul(p.y, izzz)};
}

// many XYZZ points -> affine with ONE field inversion (Montgomery's trick over the zzz coordinates): q_inv is a 254-bit
// exponentiation (~23 us), which is what a graph's thousands of witness commitments used to pay one by one
inline void gx_batch_to_aff(const G1X* p, size_t n, G1Aff* out) {
    if (n == 0) return;
    Fq* pre = new Fq[n];
    Fq acc = q_one();
    for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!gx_is_inf(p[i])) acc = q_mul(acc, p[i].zzz); }
    Fq inv = q_inv(acc);
    for (size_t i = n; i-- > 0;) {
        if (gx_is_inf(p[i])) { out[i] = G1Aff{q_zero(), q_zero()}; continue; }
        const Fq izzz = q_mul(inv, pre[i]);
        inv = q_mul(inv, p[i].zzz);
        const Fq izz = q_sqr(q_mul(izzz, p[i].zz));
        out[i] = G1Aff{q_mul(p[i].x, izz), q_mul(p[i].y, izzz)};
    }
    delete[] pre;
}

}  // namespace atlas_host
