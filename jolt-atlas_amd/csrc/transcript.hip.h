// Device-resident Fiat-Shamir transcript.
//
// The reference keeps the transcript on the host and crosses it once per sumcheck round
// (joltworks/src/subprotocols/sumcheck.rs:578-590).  Here the Blake2bTranscript
// (joltworks/src/transcripts/blake2b.rs:12-256) lives in HBM next to the polynomials so a
// whole sumcheck is a chain of launches with no host round-trip: the round kernel writes
// its partial sums, one small workgroup folds them, interpolates the round polynomial,
// absorbs it, squeezes the 128-bit challenge and leaves it in device memory for the next
// bind.  BLAKE2b-256 per RFC 7693 (the reference uses crate blake2 0.10.6).
#pragma once
#include "field.hip.h"

namespace atlas {

struct DevTranscript {   // image of atlas_transcript_t (include/atlas_hip.h)
    uint8_t state[32];
    uint32_t n_rounds;
    uint32_t pad_[3];
};

__device__ __forceinline__ uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

// One BLAKE2b compression of a single final block. m = 16 message words, t = byte count.
__device__ inline void blake2b_final_block(uint64_t h[8], const uint64_t m[16], uint64_t t) {
    const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                            0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                            0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    const uint8_t SIGMA[10][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
        {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4},
        {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13},
        {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11},
        {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5},
        {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    uint64_t v[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
    v[12] ^= t;
    v[14] = ~v[14];
#define ATLAS_G(a, b, c, d, x, y)                                   \
    do {                                                            \
        v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32);   \
        v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 24);   \
        v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16);   \
        v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 63);   \
    } while (0)
#pragma unroll
    for (int r = 0; r < 12; r++) {
        const uint8_t* s = SIGMA[r % 10];
        ATLAS_G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        ATLAS_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        ATLAS_G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        ATLAS_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        ATLAS_G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        ATLAS_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        ATLAS_G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        ATLAS_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef ATLAS_G
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

// Transcript kept in registers as 4 little-endian u64 words of the 32-byte state.
struct RegTranscript {
    uint64_t s[4];
    uint32_t n_rounds;
};

__device__ __forceinline__ uint64_t bswap64(uint64_t x) {
    return ((uint64_t)__builtin_bswap32((uint32_t)x) << 32) | __builtin_bswap32((uint32_t)(x >> 32));
}

// state' = BLAKE2b-256(state || 0^28 || n_rounds_be32 || payload[0..n_words*8))
// (blake2b.rs:31-37,64-78).  payload is given as little-endian u64 words, n_words <= 8.
__device__ inline void tr_absorb_words(RegTranscript& T, const uint64_t* payload, int n_words) {
    uint64_t m[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = 0;
    m[0] = T.s[0]; m[1] = T.s[1]; m[2] = T.s[2]; m[3] = T.s[3];
    // bytes 60..63 = n_rounds big-endian -> top half of word 7
    m[7] = (uint64_t)__builtin_bswap32(T.n_rounds) << 32;
    for (int i = 0; i < n_words; i++) m[8 + i] = payload[i];
    uint64_t h[8] = {0x6a09e667f3bcc908ULL ^ 0x01010020ULL, 0xbb67ae8584caa73bULL,
                     0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL,
                     0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    blake2b_final_block(h, m, 64 + 8 * (uint64_t)n_words);
    T.s[0] = h[0]; T.s[1] = h[1]; T.s[2] = h[2]; T.s[3] = h[3];
    T.n_rounds += 1;
}

// append_message with a label packed by the host into 4 LE words (blake2b.rs:109-122)
__device__ __forceinline__ void tr_append_label(RegTranscript& T, const uint64_t label[4]) {
    tr_absorb_words(T, label, 4);
}

// append_scalar: 32-byte big-endian canonical integer (blake2b.rs:138-146)
__device__ inline void tr_append_scalar(RegTranscript& T, const Fr& a_mont) {
    Fr c = fe_from_mont<FrParams>(a_mont);
    uint64_t w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // BE byte string: word i holds canonical limbs (7-2i, 6-2i) byte-swapped
        uint64_t limb = ((uint64_t)c.v[7 - 2 * i] << 32) | c.v[6 - 2 * i];
        w[i] = bswap64(limb);
    }
    tr_absorb_words(T, w, 4);
}

// challenge_u128 (blake2b.rs:57-62,197-202): squeeze, first 16 digest bytes as LE u128
__device__ inline void tr_challenge_u128(RegTranscript& T, uint64_t& lo, uint64_t& hi) {
    tr_absorb_words(T, nullptr, 0);
    lo = T.s[0];
    hi = T.s[1];
}

// MontU128Challenge::new (mont_ark_u128.rs:51-62) -> Montgomery limbs used by the binds.
// mode 0: limbs [0,0,lo,hi&mask] ARE the residue; mode 1: canonical c<<128 converted.
__device__ inline Fr challenge_to_mont(uint64_t lo, uint64_t hi, int mode) {
    hi &= 0x1fffffffffffffffULL;  // u128::MAX >> 3
    Fr c = fe_zero();
    c.v[4] = (uint32_t)lo; c.v[5] = (uint32_t)(lo >> 32);
    c.v[6] = (uint32_t)hi; c.v[7] = (uint32_t)(hi >> 32);
    if (mode == 0) return c;
    return fe_to_mont<FrParams>(c);
}

__device__ __forceinline__ RegTranscript tr_load(const DevTranscript* d) {
    RegTranscript T;
    const uint64_t* p = reinterpret_cast<const uint64_t*>(d->state);
    T.s[0] = p[0]; T.s[1] = p[1]; T.s[2] = p[2]; T.s[3] = p[3];
    T.n_rounds = d->n_rounds;
    return T;
}

__device__ __forceinline__ void tr_store(DevTranscript* d, const RegTranscript& T) {
    uint64_t* p = reinterpret_cast<uint64_t*>(d->state);
    p[0] = T.s[0]; p[1] = T.s[1]; p[2] = T.s[2]; p[3] = T.s[3];
    d->n_rounds = T.n_rounds;
}

}  // namespace atlas
