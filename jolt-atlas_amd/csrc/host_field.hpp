// Host-side BN254 Fr (4 x u64 Montgomery limbs) and the Blake2b transcript, for the thin
// C++ layer above the kernels: UniPoly interpolation per round (a handful of field ops),
// transcript absorb/squeeze when the caller drives the rounds itself, HyperKZG glue.
// Mirrors joltworks/src/field/ark.rs (value semantics of ark_bn254::Fr) and
// joltworks/src/transcripts/blake2b.rs.  This is product code: it shares nothing with
// oracle/.
#pragma once
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <cstdint>
#include <cstring>
#include <vector>

namespace atlas_host {

typedef unsigned __int128 u128;

struct Fr {
    uint64_t l[4];
    bool operator==(const Fr& o) const { return std::memcmp(l, o.l, 32) == 0; }
};

static const uint64_t FR_P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
                                 0x30644e72e131a029ULL};
static const uint64_t FR_R[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL,
                                 0x0e0a77c19a07df2fULL};
static const uint64_t FR_R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL,
                                  0x0216d0b17f4e44a5ULL};
static const uint64_t FR_INV = 0xc2e1f593efffffffULL;

inline bool geq_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] != FR_P[i]) return a[i] > FR_P[i];
    }
    return true;
}
inline void sub_p(uint64_t a[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - FR_P[i] - br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
inline Fr zero() { return Fr{{0, 0, 0, 0}}; }
inline Fr one() { return Fr{{FR_R[0], FR_R[1], FR_R[2], FR_R[3]}}; }
inline Fr add(const Fr& a, const Fr& b) {
    Fr o; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; o.l[i] = (uint64_t)c; c >>= 64; }
    if (geq_p(o.l)) sub_p(o.l);
    return o;
}
inline Fr sub(const Fr& a, const Fr& b) {
    Fr o; u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.l[i] - b.l[i] - br; o.l[i] = (uint64_t)d; br = (d >> 64) & 1;
    }
    if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)o.l[i] + FR_P[i]; o.l[i] = (uint64_t)c; c >>= 64; } }
    return o;
}
inline Fr neg(const Fr& a) { return sub(zero(), a); }
// CIOS Montgomery multiplication, fully unrolled (p < 2^254: the running value stays below 2p, five words suffice).
// 14 ns with g++ -O3 against 28 ns for the product-then-reduce loop it replaces (39 ns with hipcc's host clang); the
// host finishes every round polynomial with a few hundred of these.
inline Fr mul_sos(const Fr& a, const Fr& b) {          // the old form, kept as the cross-check of tests/test_host_field.py
    uint64_t t[9] = {0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.l[i] * b.l[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
        t[i + 4] = (uint64_t)c;
    }
    for (int i = 0; i < 4; i++) {
        uint64_t m = t[i] * FR_INV; u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)m * FR_P[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
        for (int k = i + 4; c && k < 9; k++) { c += t[k]; t[k] = (uint64_t)c; c >>= 64; }
    }
    Fr o{{t[4], t[5], t[6], t[7]}};
    if (t[8] || geq_p(o.l)) sub_p(o.l);
    return o;
}
template <int FIRST>
inline Fr mul_from(const Fr& a, const Fr& b) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    const uint64_t b0 = b.l[0], b1 = b.l[1], b2 = b.l[2], b3 = b.l[3];
    const uint64_t p0 = FR_P[0], p1 = FR_P[1], p2 = FR_P[2], p3 = FR_P[3];
#define ATLAS_CIOS_STEP(ai)                                                           \
    do {                                                                              \
        u128 c = (u128)(ai) * b0 + t0; t0 = (uint64_t)c; c >>= 64;                     \
        c += (u128)(ai) * b1 + t1; t1 = (uint64_t)c; c >>= 64;                         \
        c += (u128)(ai) * b2 + t2; t2 = (uint64_t)c; c >>= 64;                         \
        c += (u128)(ai) * b3 + t3; t3 = (uint64_t)c; c >>= 64;                         \
        const uint64_t t4n = t4 + (uint64_t)c;                                         \
        const uint64_t m = t0 * FR_INV;                                                \
        c = (u128)m * p0 + t0; c >>= 64;                                               \
        c += (u128)m * p1 + t1; t0 = (uint64_t)c; c >>= 64;                            \
        c += (u128)m * p2 + t2; t1 = (uint64_t)c; c >>= 64;                            \
        c += (u128)m * p3 + t3; t2 = (uint64_t)c; c >>= 64;                            \
        c += t4n; t3 = (uint64_t)c; t4 = (uint64_t)(c >> 64);                          \
    } while (0)
    if (!(FIRST == 2)) { ATLAS_CIOS_STEP(a.l[0]); ATLAS_CIOS_STEP(a.l[1]); }
    ATLAS_CIOS_STEP(a.l[2]); ATLAS_CIOS_STEP(a.l[3]);
#undef ATLAS_CIOS_STEP
    Fr o{{t0, t1, t2, t3}};
    if (t4 || geq_p(o.l)) sub_p(o.l);
    return o;
}
inline Fr mul(const Fr& a, const Fr& b) { return mul_from<0>(a, b); }
// a * b for a left operand whose two low limbs are zero — a MontU128Challenge as challenge_to_fr(.., 0) returns it: the
// first two CIOS steps leave the accumulator at zero (a_i = 0, so m = 0), skipping them is exact and halves the work
inline Fr mul_hi2(const Fr& a, const Fr& b) { return mul_from<2>(a, b); }
inline Fr mul_challenge(const Fr& r, const Fr& x) { return (r.l[0] | r.l[1]) == 0 ? mul_hi2(r, x) : mul(r, x); }
inline Fr from_canonical(const uint64_t c[4]) { Fr t{{c[0], c[1], c[2], c[3]}}; Fr r2{{FR_R2[0], FR_R2[1], FR_R2[2], FR_R2[3]}}; return mul(t, r2); }
inline Fr from_u64(uint64_t v) { uint64_t c[4] = {v, 0, 0, 0}; return from_canonical(c); }
inline void to_canonical(const Fr& a, uint64_t c[4]) { Fr o = mul(a, Fr{{1, 0, 0, 0}}); std::memcpy(c, o.l, 32); }
inline Fr pow_p_minus_2(const Fr& a) {
    uint64_t e[4] = {FR_P[0] - 2, FR_P[1], FR_P[2], FR_P[3]};
    Fr acc = one(), base = a;
    for (int i = 0; i < 256; i++) { if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base); base = mul(base, base); }
    return acc;
}
// Inverse by the binary extended Euclidean algorithm on the 256-bit residue (8 us against 19 us for the Fermat
// ladder on the build host; the host takes one or two inversions per sumcheck round for Gruen's division).
// a is a Montgomery residue aR: the integer inverse t = (aR)^-1 is turned into a^-1 R by one multiplication
// with R^3.  inv(0) = 0 like the ladder.
namespace detail {
inline bool is_zero4(const uint64_t a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
inline bool is_one4(const uint64_t a[4]) { return a[0] == 1 && (a[1] | a[2] | a[3]) == 0; }
inline bool geq4(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i];
    return true;
}
inline void sub4(uint64_t a[4], const uint64_t b[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
inline uint64_t add4(uint64_t a[4], const uint64_t b[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; a[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
inline void shr1(uint64_t a[4], uint64_t top) {
    for (int i = 0; i < 3; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 63);
    a[3] = (a[3] >> 1) | (top << 63);
}
inline void halve_mod_p(uint64_t x[4]) {            // x / 2 mod p for x < p
    uint64_t top = 0;
    if (x[0] & 1) top = add4(x, FR_P);
    shr1(x, top);
}
}  // namespace detail

inline Fr inv_binary_eea(const Fr& a) {
    using namespace detail;
    if (is_zero4(a.l)) return zero();
    uint64_t u[4], v[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
    std::memcpy(u, a.l, 32); std::memcpy(v, FR_P, 32);
    while (!is_one4(u) && !is_one4(v)) {
        while (!(u[0] & 1)) { shr1(u, 0); halve_mod_p(x1); }
        while (!(v[0] & 1)) { shr1(v, 0); halve_mod_p(x2); }
        if (geq4(u, v)) { sub4(u, v); if (geq4(x1, x2)) sub4(x1, x2); else { add4(x1, FR_P); sub4(x1, x2); } }
        else { sub4(v, u); if (geq4(x2, x1)) sub4(x2, x1); else { add4(x2, FR_P); sub4(x2, x1); } }
    }
    Fr t; std::memcpy(t.l, is_one4(u) ? x1 : x2, 32);
    static const Fr R3 = mul(Fr{{FR_R2[0], FR_R2[1], FR_R2[2], FR_R2[3]}}, Fr{{FR_R2[0], FR_R2[1], FR_R2[2], FR_R2[3]}});
    return mul(t, R3);
}


// The same inverse by Bernstein-Yang divsteps in batches of 62 (the layout of libsecp256k1's modinv64, variable
// time): the decisions of 62 steps need only the low words of f and g, so a batch is a 64-bit loop followed by one
// 2x2 matrix applied to the 256-bit values.  ~1.3 us against 8 us for the bit-at-a-time loop above — the host
// takes one inversion per sumcheck round for Gruen's division (split_eq_poly.rs:410-413), which made the
// transcript thread the slowest stage of a pipelined round.
namespace detail {
struct S62 { int64_t v[5]; };
struct T2x2 { int64_t u, v, q, r; };
typedef __int128 i128;
constexpr uint64_t M62 = ~(uint64_t)0 >> 2;
inline S62 to_s62(const uint64_t a[4]) {
    S62 o;
    o.v[0] = (int64_t)(a[0] & M62);
    o.v[1] = (int64_t)(((a[0] >> 62) | (a[1] << 2)) & M62);
    o.v[2] = (int64_t)(((a[1] >> 60) | (a[2] << 4)) & M62);
    o.v[3] = (int64_t)(((a[2] >> 58) | (a[3] << 6)) & M62);
    o.v[4] = (int64_t)(a[3] >> 56);
    return o;
}
inline int64_t divsteps_62_var(int64_t eta, uint64_t f0, uint64_t g0, T2x2& t) {
    uint64_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0, m, w;
    int i = 62, limit, zeros;
    for (;;) {
        zeros = __builtin_ctzll(g | (~(uint64_t)0 << i));
        g >>= zeros; u <<= zeros; v <<= zeros; eta -= zeros; i -= zeros;
        if (i == 0) break;
        if (eta < 0) {
            uint64_t tmp;
            eta = -eta;
            tmp = f; f = g; g = (uint64_t)0 - tmp;
            tmp = u; u = q; q = (uint64_t)0 - tmp;
            tmp = v; v = r; r = (uint64_t)0 - tmp;
            limit = ((int)eta + 1) > i ? i : ((int)eta + 1);
            m = (~(uint64_t)0 >> (64 - limit)) & 63u;
            w = (f * g * (f * f - 2)) & m;
        } else {
            limit = ((int)eta + 1) > i ? i : ((int)eta + 1);
            m = (~(uint64_t)0 >> (64 - limit)) & 15u;
            w = f + (((f + 1) & 4) << 1);
            w = ((uint64_t)0 - w) * g & m;
        }
        g += f * w; q += u * w; r += v * w;
    }
    t.u = (int64_t)u; t.v = (int64_t)v; t.q = (int64_t)q; t.r = (int64_t)r;
    return eta;
}
inline void update_de_62(S62& d, S62& e, const T2x2& t, const S62& mod, uint64_t mod_inv62) {
    const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
    const int64_t sd = d.v[4] >> 63, se = e.v[4] >> 63;
    int64_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    i128 cd = (i128)u * d.v[0] + (i128)v * e.v[0], ce = (i128)q * d.v[0] + (i128)r * e.v[0];
    md -= (int64_t)((mod_inv62 * (uint64_t)cd + (uint64_t)md) & M62);
    me -= (int64_t)((mod_inv62 * (uint64_t)ce + (uint64_t)me) & M62);
    cd += (i128)mod.v[0] * md; ce += (i128)mod.v[0] * me;
    cd >>= 62; ce >>= 62;
    for (int k = 1; k < 5; k++) {
        cd += (i128)u * d.v[k] + (i128)v * e.v[k] + (i128)mod.v[k] * md;
        ce += (i128)q * d.v[k] + (i128)r * e.v[k] + (i128)mod.v[k] * me;
        if (k < 4) { d.v[k - 1] = (int64_t)((uint64_t)cd & M62); cd >>= 62; e.v[k - 1] = (int64_t)((uint64_t)ce & M62); ce >>= 62; }
        else { d.v[3] = (int64_t)((uint64_t)cd & M62); cd >>= 62; e.v[3] = (int64_t)((uint64_t)ce & M62); ce >>= 62; }
    }
    d.v[4] = (int64_t)cd; e.v[4] = (int64_t)ce;
}
inline void update_fg_62_var(int len, S62& f, S62& g, const T2x2& t) {
    const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
    i128 cf = (i128)u * f.v[0] + (i128)v * g.v[0], cg = (i128)q * f.v[0] + (i128)r * g.v[0];
    cf >>= 62; cg >>= 62;
    for (int i = 1; i < len; i++) {
        cf += (i128)u * f.v[i] + (i128)v * g.v[i]; cg += (i128)q * f.v[i] + (i128)r * g.v[i];
        f.v[i - 1] = (int64_t)((uint64_t)cf & M62); cf >>= 62;
        g.v[i - 1] = (int64_t)((uint64_t)cg & M62); cg >>= 62;
    }
    f.v[len - 1] = (int64_t)cf; g.v[len - 1] = (int64_t)cg;
}
}  // namespace detail

inline Fr inv(const Fr& a) {
    using namespace detail;
    if (is_zero4(a.l)) return zero();
    static const S62 MOD = to_s62(FR_P);
    static const uint64_t MOD_INV62 = [] { uint64_t x = FR_P[0]; for (int i = 0; i < 6; i++) x *= 2 - FR_P[0] * x; return x & M62; }();
    S62 d{{0, 0, 0, 0, 0}}, e{{1, 0, 0, 0, 0}}, f = MOD, g = to_s62(a.l);
    int len = 5;
    int64_t eta = -1;
    for (int it = 0; it < 16; it++) {          // 12 batches suffice for 256 bits (741 divsteps); the bound is a guard
        T2x2 t;
        eta = divsteps_62_var(eta, (uint64_t)f.v[0], (uint64_t)g.v[0], t);
        update_de_62(d, e, t, MOD, MOD_INV62);
        update_fg_62_var(len, f, g, t);
        if (g.v[0] == 0) {
            int64_t c = 0;
            for (int j = 1; j < len; j++) c |= g.v[j];
            if (c == 0) break;
        }
        const int64_t fn = f.v[len - 1], gn = g.v[len - 1];
        int64_t c = ((int64_t)len - 2) >> 63;
        c |= fn ^ (fn >> 63); c |= gn ^ (gn >> 63);
        if (c == 0) { f.v[len - 2] |= (int64_t)((uint64_t)fn << 62); g.v[len - 2] |= (int64_t)((uint64_t)gn << 62); --len; }
    }
    // f = +-1 and d x = f (mod p) with d in (-2p, p): t = f d mod p.  Signed 62-bit limbs -> two's complement 320 bits
    uint64_t w[5];
    {
        uint64_t lim[5];
        for (int k = 0; k < 5; k++) lim[k] = (uint64_t)d.v[k];
        w[0] = (lim[0] & M62) | (lim[1] << 62);
        w[1] = ((lim[1] & M62) >> 2) | (lim[2] << 60);
        w[2] = ((lim[2] & M62) >> 4) | (lim[3] << 58);
        w[3] = ((lim[3] & M62) >> 6) | (lim[4] << 56);
        w[4] = (uint64_t)(d.v[4] >> 8);
    }
    auto add_p = [&] { u128 c = 0; for (int k = 0; k < 5; k++) { c += (u128)w[k] + (k < 4 ? FR_P[k] : 0); w[k] = (uint64_t)c; c >>= 64; } };
    while ((int64_t)w[4] < 0) add_p();
    uint64_t t4[4] = {w[0], w[1], w[2], w[3]};
    while (w[4] != 0 || geq_p(t4)) {           // not reached for d < p; kept as a guard
        u128 br = 0;
        for (int k = 0; k < 5; k++) { u128 dd = (u128)w[k] - (k < 4 ? FR_P[k] : 0) - br; w[k] = (uint64_t)dd; br = (dd >> 64) & 1; }
        t4[0] = w[0]; t4[1] = w[1]; t4[2] = w[2]; t4[3] = w[3];
    }
    Fr t{{t4[0], t4[1], t4[2], t4[3]}};
    bool f_neg = false;                                    // f = -1: the sign sits in its top non-zero limb
    for (int k = len - 1; k >= 0; k--) if (f.v[k] != 0) { f_neg = f.v[k] < 0; break; }
    if (f_neg) t = sub(zero(), t);
    static const Fr R3 = mul(Fr{{FR_R2[0], FR_R2[1], FR_R2[2], FR_R2[3]}}, Fr{{FR_R2[0], FR_R2[1], FR_R2[2], FR_R2[3]}});
    return mul(t, R3);
}

// MontU128Challenge::from(c) as an Fr (mont_ark_u128.rs:51-62,79-84)
inline Fr challenge_to_fr(uint64_t lo, uint64_t hi, int mode) {
    hi &= 0x1fffffffffffffffULL;
    if (mode == 0) return Fr{{0, 0, lo, hi}};
    uint64_t c[4] = {0, 0, lo, hi};
    return from_canonical(c);
}

// ---------------------------------------------------------------- BLAKE2b-256 (RFC 7693)
struct Blake2b256 {
    uint64_t h[8]; uint8_t buf[128]; size_t buflen; uint64_t t;
    Blake2b256() {
        static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                       0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                       0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        for (int i = 0; i < 8; i++) h[i] = IV[i];
        h[0] ^= 0x01010020ULL; buflen = 0; t = 0;
    }
    static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    void compress_portable(const uint8_t* block, bool last) {
        static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                       0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                       0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        static const uint8_t S[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        uint64_t m[16], v[16];
        for (int i = 0; i < 16; i++) std::memcpy(&m[i], block + 8 * i, 8);
        for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
        v[12] ^= t; if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32); v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; r++) {
            const uint8_t* s = S[r % 10];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    }
    // The same compression on AVX2 (rows of the 4 x 4 state as four 256-bit vectors, the column and diagonal steps of a round as two vector
    // G's with lane rotations in between — the layout of the BLAKE2 reference's blake2b-round.h).  Measured and NOT adopted (round 6): 1.3x
    // faster per call in a loop of its own, and 2-3 % SLOWER over whole proofs on the GPU box's host in three alternations
    // (profiles/r06m_blake_ab.txt: nanoGPT iop 303-310 ms against 287-296) — a compression every few microseconds between scalar field
    // arithmetic pays for the vector unit's state more than it saves.  ATLAS_BLAKE_AVX2=1 selects it; same function (checked against the loop
    // on 20 000 random blocks and hashlib's vectors).  What IS kept from the experiment: a transcript step is ONE block built in place (tr_absorb).
#if defined(__x86_64__)
    __attribute__((target("avx2"))) void compress_avx2(const uint8_t* block, bool last) {
        static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                       0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                       0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        static const uint8_t S[12][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        uint64_t m[16];
        std::memcpy(m, block, 128);
        __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(h)), b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(h + 4));
        __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(IV));
        __m256i d = _mm256_xor_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(IV + 4)), _mm256_set_epi64x(0, last ? -1LL : 0, 0, (long long)t));
        const __m256i a0 = a, b0 = b;
        const __m256i r24 = _mm256_setr_epi8(3, 4, 5, 6, 7, 0, 1, 2, 11, 12, 13, 14, 15, 8, 9, 10, 3, 4, 5, 6, 7, 0, 1, 2, 11, 12, 13, 14, 15, 8, 9, 10);
        const __m256i r16 = _mm256_setr_epi8(2, 3, 4, 5, 6, 7, 0, 1, 10, 11, 12, 13, 14, 15, 8, 9, 2, 3, 4, 5, 6, 7, 0, 1, 10, 11, 12, 13, 14, 15, 8, 9);
#define ATLAS_B2_G(x, y)                                                                                                   \
        a = _mm256_add_epi64(_mm256_add_epi64(a, b), x); d = _mm256_shuffle_epi32(_mm256_xor_si256(d, a), 0xB1);          \
        c = _mm256_add_epi64(c, d); b = _mm256_shuffle_epi8(_mm256_xor_si256(b, c), r24);                                 \
        a = _mm256_add_epi64(_mm256_add_epi64(a, b), y); d = _mm256_shuffle_epi8(_mm256_xor_si256(d, a), r16);            \
        c = _mm256_add_epi64(c, d); { const __m256i bc = _mm256_xor_si256(b, c); b = _mm256_xor_si256(_mm256_srli_epi64(bc, 63), _mm256_add_epi64(bc, bc)); }
        for (int r = 0; r < 12; r++) {
            const uint8_t* s = S[r];
            {   // columns
                const __m256i x = _mm256_set_epi64x((long long)m[s[6]], (long long)m[s[4]], (long long)m[s[2]], (long long)m[s[0]]);
                const __m256i y = _mm256_set_epi64x((long long)m[s[7]], (long long)m[s[5]], (long long)m[s[3]], (long long)m[s[1]]);
                ATLAS_B2_G(x, y)
            }
            b = _mm256_permute4x64_epi64(b, 0x39); c = _mm256_permute4x64_epi64(c, 0x4E); d = _mm256_permute4x64_epi64(d, 0x93);
            {   // diagonals
                const __m256i x = _mm256_set_epi64x((long long)m[s[14]], (long long)m[s[12]], (long long)m[s[10]], (long long)m[s[8]]);
                const __m256i y = _mm256_set_epi64x((long long)m[s[15]], (long long)m[s[13]], (long long)m[s[11]], (long long)m[s[9]]);
                ATLAS_B2_G(x, y)
            }
            b = _mm256_permute4x64_epi64(b, 0x93); c = _mm256_permute4x64_epi64(c, 0x4E); d = _mm256_permute4x64_epi64(d, 0x39);
        }
#undef ATLAS_B2_G
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(h), _mm256_xor_si256(a0, _mm256_xor_si256(a, c)));
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(h + 4), _mm256_xor_si256(b0, _mm256_xor_si256(b, d)));
    }
    static bool use_avx2() { static const bool v = __builtin_cpu_supports("avx2") && getenv("ATLAS_BLAKE_AVX2") != nullptr; return v; }
#else
    void compress_avx2(const uint8_t* block, bool last) { compress_portable(block, last); }
    static bool use_avx2() { return false; }
#endif
    void compress(const uint8_t* block, bool last) { if (use_avx2()) compress_avx2(block, last); else compress_portable(block, last); }
    void update(const uint8_t* in, size_t n) {
        while (n) {
            if (buflen == 128) { t += 128; compress(buf, false); buflen = 0; }
            size_t k = 128 - buflen; if (k > n) k = n;
            std::memcpy(buf + buflen, in, k); buflen += k; in += k; n -= k;
        }
    }
    void finalize(uint8_t out[32]) {
        t += buflen; std::memset(buf + buflen, 0, 128 - buflen); compress(buf, true);
        std::memcpy(out, h, 32);
    }
};

// ---------------------------------------------------------------- Blake2bTranscript
struct Transcript {   // same image as atlas_transcript_t
    uint8_t state[32]; uint32_t n_rounds; uint32_t pad_[3];
};
inline void tr_absorb(Transcript& T, const uint8_t* p1, size_t n1, const uint8_t* p2 = nullptr, size_t n2 = 0) {
    if (n1 + n2 <= 64) {                                       // state | round | <= 64 bytes: ONE block, built in place (every scalar, label and challenge step)
        Blake2b256 H;
        uint8_t* blk = H.buf;
        std::memcpy(blk, T.state, 32);
        std::memset(blk + 32, 0, 96);
        blk[60] = (uint8_t)(T.n_rounds >> 24); blk[61] = (uint8_t)(T.n_rounds >> 16); blk[62] = (uint8_t)(T.n_rounds >> 8); blk[63] = (uint8_t)T.n_rounds;
        if (n1) std::memcpy(blk + 64, p1, n1);
        if (n2) std::memcpy(blk + 64 + n1, p2, n2);
        H.t = 64 + n1 + n2;
        H.compress(blk, true);
        std::memcpy(T.state, H.h, 32); T.n_rounds += 1;
        return;
    }
    Blake2b256 H; uint8_t pre[32] = {0};
    pre[28] = (uint8_t)(T.n_rounds >> 24); pre[29] = (uint8_t)(T.n_rounds >> 16);
    pre[30] = (uint8_t)(T.n_rounds >> 8); pre[31] = (uint8_t)T.n_rounds;
    H.update(T.state, 32); H.update(pre, 32);
    if (n1) H.update(p1, n1);
    if (n2) H.update(p2, n2);
    H.finalize(T.state); T.n_rounds += 1;
}
inline void tr_new(Transcript& T, const uint8_t* label, size_t n) {
    uint8_t b[32] = {0}; std::memcpy(b, label, n > 32 ? 32 : n);
    Blake2b256 H; H.update(b, 32); H.finalize(T.state); T.n_rounds = 0; T.pad_[0] = T.pad_[1] = T.pad_[2] = 0;
}
inline void tr_append_message(Transcript& T, const uint8_t* msg, size_t n) {
    uint8_t b[32] = {0}; std::memcpy(b, msg, n > 32 ? 32 : n); tr_absorb(T, b, 32);
}
inline void tr_append_message(Transcript& T, const char* msg) { tr_append_message(T, (const uint8_t*)msg, std::strlen(msg)); }
inline void tr_append_bytes(Transcript& T, const uint8_t* p, size_t n) { tr_absorb(T, p, n); }
inline void tr_append_u64(Transcript& T, uint64_t x) {
    uint8_t b[32] = {0}; for (int i = 0; i < 8; i++) b[24 + i] = (uint8_t)(x >> (8 * (7 - i))); tr_absorb(T, b, 32);
}
inline void fr_be_bytes(const Fr& a, uint8_t out[32]) {
    uint64_t c[4]; to_canonical(a, c);
    for (int i = 0; i < 32; i++) out[31 - i] = (uint8_t)(c[i >> 3] >> (8 * (i & 7)));
}
inline void tr_append_scalar(Transcript& T, const Fr& a) { uint8_t b[32]; fr_be_bytes(a, b); tr_absorb(T, b, 32); }
inline void tr_append_scalars(Transcript& T, const Fr* a, size_t n) {
    tr_append_message(T, "begin_append_vector");
    for (size_t i = 0; i < n; i++) tr_append_scalar(T, a[i]);
    tr_append_message(T, "end_append_vector");
}
inline void tr_challenge_u128(Transcript& T, uint64_t& lo, uint64_t& hi) {
    tr_absorb(T, nullptr, 0); std::memcpy(&lo, T.state, 8); std::memcpy(&hi, T.state + 8, 8);
}
inline Fr tr_challenge_scalar(Transcript& T) {   // 16 digest bytes as a big-endian integer
    tr_absorb(T, nullptr, 0);
    uint64_t hi = 0, lo = 0;
    for (int i = 0; i < 8; i++) { hi = (hi << 8) | T.state[i]; lo = (lo << 8) | T.state[8 + i]; }
    uint64_t c[4] = {lo, hi, 0, 0};
    return from_canonical(c);
}

// UniPoly::from_evals_and_hint for degree 2 / 3 (unipoly.rs:66-98); returns #coeffs
inline int unipoly_from_evals_and_hint(const Fr& hint, const Fr* ev, int deg, Fr* c) {
    static const Fr two_inv = inv(from_u64(2)), six_inv = inv(from_u64(6));
    Fr e0 = ev[0], e1 = sub(hint, e0), e2 = ev[1];
    c[0] = e0;
    if (deg == 2) {
        c[2] = mul(add(sub(sub(e0, e1), e1), e2), two_inv);
        c[1] = sub(sub(e1, e0), c[2]);
        return 3;
    }
    Fr e3 = ev[2];
    Fr u = sub(e1, e2); u = add(add(u, u), u);
    c[3] = mul(add(sub(e3, e0), u), six_inv);
    Fr t = mul(add(sub(sub(e0, e1), e1), e2), two_inv);
    c[2] = sub(sub(sub(t, c[3]), c[3]), c[3]);
    c[1] = sub(sub(sub(e1, e0), c[2]), c[3]);
    return 4;
}

}  // namespace atlas_host
