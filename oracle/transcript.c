/* ORACLE — TEST INFRASTRUCTURE ONLY (see bn254.h header note; parity unpinned).
 * BLAKE2b-256 (RFC 7693; the reference uses crate `blake2` 0.10.6, Cargo.lock:352-354)
 * and the Blake2bTranscript of joltworks/src/transcripts/blake2b.rs. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static const uint64_t B2B_IV[8] = {
    0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
    0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};

static const uint8_t B2B_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
    {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4},
    {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13},
    {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11},
    {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5},
    {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
    {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

static inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

static void b2b_compress(uint64_t h[8], const uint8_t block[128], uint64_t t, int last) {
    uint64_t v[16], m[16];
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2B_IV[i]; }
    v[12] ^= t;            /* t fits in 64 bits for every transcript input */
    if (last) v[14] = ~v[14];
    for (int i = 0; i < 16; i++) {
        uint64_t w = 0;
        for (int k = 7; k >= 0; k--) w = (w << 8) | block[8 * i + k];
        m[i] = w;
    }
#define B2B_G(a, b, c, d, x, y)                                   \
    do {                                                          \
        v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); \
        v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 24); \
        v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); \
        v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 63); \
    } while (0)
    for (int r = 0; r < 12; r++) {
        const uint8_t *s = B2B_SIGMA[r];
        B2B_G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        B2B_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        B2B_G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        B2B_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        B2B_G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        B2B_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        B2B_G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        B2B_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef B2B_G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

void orc_blake2b256(const uint8_t *in, size_t len, uint8_t out[32]) {
    uint64_t h[8];
    for (int i = 0; i < 8; i++) h[i] = B2B_IV[i];
    h[0] ^= 0x01010000ULL ^ 32; /* digest 32, no key, fanout 1, depth 1 */
    uint8_t block[128];
    size_t off = 0;
    while (len - off > 128) {
        b2b_compress(h, in + off, off + 128, 0);
        off += 128;
    }
    memset(block, 0, 128);
    if (len > off) memcpy(block, in + off, len - off);
    b2b_compress(h, block, len, 1);
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(h[i >> 3] >> (8 * (i & 7)));
}

/* ------------------------------------------------------------ transcript */

/* blake2b.rs:31-37 + update_state :64-78 — hash(state || 0^28 || n_rounds_be || payload) */
static void tr_absorb(orc_transcript *t, const uint8_t *payload, size_t n) {
    uint8_t stack[64 + 256];
    uint8_t dig[32];
    uint8_t *buf = n <= 256 ? stack : (uint8_t *)malloc(64 + n);   /* append_bytes of a model input: 4 bytes per element, any length */
    memcpy(buf, t->state, 32);
    memset(buf + 32, 0, 28);
    buf[60] = (uint8_t)(t->n_rounds >> 24); buf[61] = (uint8_t)(t->n_rounds >> 16);
    buf[62] = (uint8_t)(t->n_rounds >> 8);  buf[63] = (uint8_t)t->n_rounds;
    memcpy(buf + 64, payload, n);
    orc_blake2b256(buf, 64 + n, dig);
    if (buf != stack) free(buf);
    memcpy(t->state, dig, 32);
    t->n_rounds += 1;
    if (t->history && t->history_len < t->history_cap)
        memcpy(t->history + 32 * t->history_len, dig, 32);
    t->history_len += 1;
}

void orc_transcript_new(orc_transcript *t, const char *label) { /* blake2b.rs:81-100 */
    uint8_t buf[32]; size_t n = strlen(label);
    memset(buf, 0, 32); memcpy(buf, label, n > 32 ? 32 : n);
    orc_blake2b256(buf, 32, t->state);
    t->n_rounds = 0; t->history = 0; t->history_cap = 0; t->history_len = 0;
}

void orc_transcript_record(orc_transcript *t, uint8_t *buf, size_t cap) {
    t->history = buf; t->history_cap = cap; t->history_len = 0;
}

void orc_transcript_append_message(orc_transcript *t, const char *msg) { /* :109-122 */
    uint8_t buf[32]; size_t n = strlen(msg);
    memset(buf, 0, 32); memcpy(buf, msg, n > 32 ? 32 : n);
    tr_absorb(t, buf, 32);
}

void orc_transcript_append_bytes(orc_transcript *t, const uint8_t *b, size_t n) { /* :124-128 */
    tr_absorb(t, b, n);
}

void orc_transcript_append_u64(orc_transcript *t, uint64_t x) { /* :130-136 */
    uint8_t buf[32]; memset(buf, 0, 24);
    for (int i = 0; i < 8; i++) buf[24 + i] = (uint8_t)(x >> (8 * (7 - i)));
    tr_absorb(t, buf, 32);
}

void orc_transcript_append_scalar(orc_transcript *t, const fr_t *a) { /* :138-146 */
    uint8_t be[32]; fr_to_bytes_be(a, be);
    tr_absorb(t, be, 32);
}

void orc_transcript_append_scalars(orc_transcript *t, const fr_t *a, size_t n) { /* :158-164 */
    orc_transcript_append_message(t, "begin_append_vector");
    for (size_t i = 0; i < n; i++) orc_transcript_append_scalar(t, &a[i]);
    orc_transcript_append_message(t, "end_append_vector");
}

void orc_transcript_challenge_bytes32(orc_transcript *t, uint8_t out[32]) { /* :57-62 */
    tr_absorb(t, 0, 0);
    memcpy(out, t->state, 32);
}

u128 orc_transcript_challenge_u128(orc_transcript *t) { /* :197-202 */
    uint8_t d[32]; orc_transcript_challenge_bytes32(t, d);
    u128 v = 0;
    for (int i = 15; i >= 0; i--) v = (v << 8) | d[i];
    return v;
}

void orc_transcript_challenge_scalar(orc_transcript *t, fr_t *o) { /* :204-215 */
    uint8_t d[32], rev[16]; orc_transcript_challenge_bytes32(t, d);
    for (int i = 0; i < 16; i++) rev[i] = d[15 - i];
    fr_from_bytes_le_mod_order(rev, 16, o);
}

void orc_transcript_challenge_optimized(orc_transcript *t, u128 *raw, fr_t *o) { /* :233-238 */
    u128 c = orc_transcript_challenge_u128(t);
    if (raw) *raw = c;
    fr_from_challenge(c, o);
}
