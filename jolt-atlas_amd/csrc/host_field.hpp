// Host-side BN254 Fr (4 x u64 Montgomery limbs) and the Blake2b transcript, for the thin
// C++ layer above the kernels: UniPoly interpolation per round (a handful of field ops),
// transcript absorb/squeeze when the caller drives the rounds itself, HyperKZG glue.
// Mirrors joltworks/src/field/ark.rs (value semantics of ark_bn254::Fr) and
// joltworks/src/transcripts/blake2b.rs.  This is product code: it shares nothing with
// oracle/.
#pragma once
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
inline Fr mul(const Fr& a, const Fr& b) {
    // SOS: full 512-bit product, then 4 Montgomery reduction sweeps
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

inline Fr inv(const Fr& a) {
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
    void compress(const uint8_t* block, bool last) {
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
