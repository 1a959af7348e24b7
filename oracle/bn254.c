/* ORACLE — TEST INFRASTRUCTURE ONLY (see bn254.h header note; parity unpinned).
 * 4x64-bit Montgomery arithmetic for BN254 Fr and Fq (CIOS with unsigned __int128).
 * Results are canonical reduced residues, hence identical to arkworks' Fp<MontBackend>
 * for every operation (reference call sites: joltworks/src/field/ark.rs:241-297). */
#include "bn254.h"
#include <string.h>

const fp_params_t ORC_FR = {
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
    {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL},
    0xc2e1f593efffffffULL};

const fp_params_t ORC_FQ = {
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
    {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL},
    0x87d20782e4866389ULL};

static inline int geq(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}

static inline uint64_t add4(uint64_t o[4], const uint64_t a[4], const uint64_t b[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; o[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}

static inline uint64_t sub4(uint64_t o[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - br;
        o[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}

void fp_add(const fp_params_t *P, const fp_t *a, const fp_t *b, fp_t *o) {
    uint64_t t[4];
    uint64_t c = add4(t, a->l, b->l);
    if (c || geq(t, P->p)) sub4(t, t, P->p);
    memcpy(o->l, t, 32);
}

void fp_sub(const fp_params_t *P, const fp_t *a, const fp_t *b, fp_t *o) {
    uint64_t t[4];
    if (sub4(t, a->l, b->l)) add4(t, t, P->p);
    memcpy(o->l, t, 32);
}

int fp_is_zero(const fp_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
int fp_eq(const fp_t *a, const fp_t *b) { return memcmp(a->l, b->l, 32) == 0; }

void fp_neg(const fp_params_t *P, const fp_t *a, fp_t *o) {
    if (fp_is_zero(a)) { memset(o, 0, 32); return; }
    uint64_t t[4]; sub4(t, P->p, a->l); memcpy(o->l, t, 32);
}

void fp_mul(const fp_params_t *P, const fp_t *a, const fp_t *b, fp_t *o) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c; c >>= 64;
        }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * P->inv;
        c = (u128)m * P->p[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * P->p[j] + t[j];
            t[j - 1] = (uint64_t)c; c >>= 64;
        }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || geq(t, P->p)) sub4(t, t, P->p);
    memcpy(o->l, t, 32);
}

void fp_sqr(const fp_params_t *P, const fp_t *a, fp_t *o) { fp_mul(P, a, a, o); }

void fp_inv(const fp_params_t *P, const fp_t *a, fp_t *o) {
    /* Fermat: a^(p-2).  (arkworks uses binary EEA; the inverse is unique.) */
    uint64_t e[4]; const uint64_t two[4] = {2, 0, 0, 0};
    sub4(e, P->p, two);
    fp_t acc; memcpy(acc.l, P->r, 32);
    fp_t base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(P, &acc, &base, &acc);
        fp_sqr(P, &base, &base);
    }
    *o = acc;
}

void fp_from_canonical(const fp_params_t *P, const uint64_t c[4], fp_t *o) {
    fp_t t, r2; memcpy(t.l, c, 32); memcpy(r2.l, P->r2, 32);
    fp_mul(P, &t, &r2, o);
}

void fp_to_canonical(const fp_params_t *P, const fp_t *a, uint64_t c[4]) {
    fp_t one = {{1, 0, 0, 0}}, t;
    fp_mul(P, a, &one, &t);
    memcpy(c, t.l, 32);
}

/* ---------------------------------------------------------------- Fr */
void fr_add(const fr_t *a, const fr_t *b, fr_t *o) { fp_add(&ORC_FR, a, b, o); }
void fr_sub(const fr_t *a, const fr_t *b, fr_t *o) { fp_sub(&ORC_FR, a, b, o); }
void fr_neg(const fr_t *a, fr_t *o) { fp_neg(&ORC_FR, a, o); }
void fr_mul(const fr_t *a, const fr_t *b, fr_t *o) { fp_mul(&ORC_FR, a, b, o); }
void fr_inv(const fr_t *a, fr_t *o) { fp_inv(&ORC_FR, a, o); }
void fr_zero(fr_t *o) { memset(o, 0, 32); }
void fr_one(fr_t *o) { memcpy(o->l, ORC_FR.r, 32); }

void fr_from_u64(uint64_t v, fr_t *o) {
    uint64_t c[4] = {v, 0, 0, 0};
    fp_from_canonical(&ORC_FR, c, o);
}

void fr_from_u128(u128 v, fr_t *o) {
    uint64_t c[4] = {(uint64_t)v, (uint64_t)(v >> 64), 0, 0};
    fp_from_canonical(&ORC_FR, c, o);
}

void fr_from_i64(int64_t v, fr_t *o) {
    if (v < 0) {
        fr_from_u64((uint64_t)0 - (uint64_t)v, o);
        fr_neg(o, o);
    } else {
        fr_from_u64((uint64_t)v, o);
    }
}

void fr_to_bytes_le(const fr_t *a, uint8_t out[32]) {
    uint64_t c[4]; fp_to_canonical(&ORC_FR, a, c);
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(c[i >> 3] >> (8 * (i & 7)));
}

void fr_to_bytes_be(const fr_t *a, uint8_t out[32]) {
    uint8_t le[32]; fr_to_bytes_le(a, le);
    for (int i = 0; i < 32; i++) out[i] = le[31 - i];
}

void fr_from_bytes_le_mod_order(const uint8_t *b, size_t n, fr_t *o) {
    /* Horner over bytes from the most significant end: acc = acc*256 + byte. */
    fr_t acc, k256; fr_zero(&acc); fr_from_u64(256, &k256);
    for (size_t i = n; i-- > 0;) {
        fr_t d; fr_mul(&acc, &k256, &acc); fr_from_u64(b[i], &d); fr_add(&acc, &d, &acc);
    }
    *o = acc;
}

static int g_challenge_mode = 0;
void orc_set_challenge_mode(int mode) { g_challenge_mode = mode; }
int orc_get_challenge_mode(void) { return g_challenge_mode; }

void fr_from_challenge(u128 c, fr_t *o) {
    u128 m = c & (~(u128)0 >> 3);
    if (g_challenge_mode == 0) {
        o->l[0] = 0; o->l[1] = 0; o->l[2] = (uint64_t)m; o->l[3] = (uint64_t)(m >> 64);
    } else {
        uint64_t cc[4] = {0, 0, (uint64_t)m, (uint64_t)(m >> 64)};
        fp_from_canonical(&ORC_FR, cc, o);
    }
}
