// Device-resident Fiat-Shamir transcript.
//
// The reference keeps the transcript on the host and crosses it once per sumcheck round
// (joltworks/src/subprotocols/sumcheck.rs:578-590).  Here the Blake2bTranscript
// (joltworks/src/transcripts/blake2b.rs:12-256) lives in HBM next to the polynomials so a
// whole sumcheck is a chain of launches with no host round-trip: the round kernel writes
// its partial sums, one wavefront folds them, interpolates the round polynomial, absorbs
// it, squeezes the 128-bit challenge and leaves it in device memory for the next bind.
//
// BLAKE2b-256 (RFC 7693; the reference uses crate blake2 0.10.6) is the serial spine of
// every round, so it is laid out for latency: the 4x4 state matrix lives in one quad of
// lanes (lane l holds row elements a=v[l], b=v[4+l], c=v[8+l], d=v[12+l]); the column
// step is one G per lane, the diagonal step is the same G after a DPP quad_perm rotation
// of b, c, d.  Everything runs on the VALU of one wavefront (all quads compute the same
// thing); a uniform single-lane formulation gets scalarised onto the SALU and runs 4-5x
// longer.
#pragma once
#include "field.hip.h"

namespace atlas {

struct DevTranscript {   // image of atlas_transcript_t (include/atlas_hip.h)
    uint8_t state[32];
    uint32_t n_rounds;
    uint32_t pad_[3];
};

__device__ __forceinline__ uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

__device__ __forceinline__ uint64_t bswap64(uint64_t x) {
    return ((uint64_t)__builtin_bswap32((uint32_t)x) << 32) | __builtin_bswap32((uint32_t)(x >> 32));
}

// an opaque VGPR zero: makes everything derived from it "divergent" for the compiler so
// wave-uniform transcript/field code stays on the VALU instead of being scalarised
__device__ __forceinline__ uint32_t vgpr_zero() {
    uint32_t z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

template <int CTRL>
__device__ __forceinline__ uint64_t quad_perm64(uint64_t x) {
    int lo = (int)(uint32_t)x, hi = (int)(uint32_t)(x >> 32);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

// 4-way select by quad lane (q = lane & 3)
__device__ __forceinline__ uint64_t sel4(uint32_t q, uint64_t x0, uint64_t x1, uint64_t x2, uint64_t x3) {
    uint64_t lo = (q & 1) ? x1 : x0;
    uint64_t hi = (q & 1) ? x3 : x2;
    return (q & 2) ? hi : lo;
}

// Wave-cooperative transcript: lane l holds state word (l & 3) in `sw`.
struct WaveTranscript {
    uint64_t sw;
    uint32_t n_rounds;
};

// per-lane byte offsets into the 16-word message block for round r, slot k:
//   k=0,1 column step  (words sigma[r][2q], sigma[r][2q+1])
//   k=2,3 diagonal step (words sigma[r][8+2q], sigma[r][9+2q])
struct WaveBlakeSched {
    uint32_t off[48];
};

__device__ __forceinline__ WaveBlakeSched wave_blake_sched(uint32_t q) {
    // SIGMA rows packed per quad lane: 16 bits per round = nibbles (2q, 2q+1, 8+2q, 9+2q)
    constexpr uint8_t S[12][16] = {
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
    WaveBlakeSched W;
#pragma unroll
    for (int r = 0; r < 12; r++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int base = (k < 2) ? k : 8 + (k - 2);
            const uint32_t o0 = 8u * S[r][base + 0], o1 = 8u * S[r][base + 2], o2 = 8u * S[r][base + 4],
                           o3 = 8u * S[r][base + 6];
            uint32_t lo = (q & 1) ? o1 : o0;
            uint32_t hi = (q & 1) ? o3 : o2;
            W.off[r * 4 + k] = (q & 2) ? hi : lo;
        }
    }
    return W;
}

#define ATLAS_G(a, b, c, d, x, y)                          \
    do {                                                   \
        a = a + b + (x); d = rotr64(d ^ a, 32);            \
        c = c + d;       b = rotr64(b ^ c, 24);            \
        a = a + b + (y); d = rotr64(d ^ a, 16);            \
        c = c + d;       b = rotr64(b ^ c, 63);            \
    } while (0)

// One final-block compression from the fixed initial chaining value (every transcript
// hash is a single block: 64-byte prefix + <= 64-byte payload).  msg = 16 words in LDS,
// already written.  Returns, in lane l, digest word (l & 3).
__device__ __forceinline__ uint64_t wave_blake2b_block(const uint64_t* msg, const WaveBlakeSched& W,
                                                       uint32_t q, uint64_t t_bytes) {
    const uint64_t iv_lo = sel4(q, 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                0xa54ff53a5f1d36f1ULL);
    const uint64_t iv_hi = sel4(q, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL,
                                0x5be0cd19137e2179ULL);
    const uint64_t h_lo = iv_lo ^ (q == 0 ? 0x01010020ULL : 0ULL);   // digest 32, fanout 1, depth 1
    uint64_t a = h_lo, b = iv_hi, c = iv_lo;
    uint64_t d = iv_hi ^ (q == 0 ? t_bytes : 0ULL) ^ (q == 2 ? ~0ULL : 0ULL);
    const unsigned char* mb = reinterpret_cast<const unsigned char*>(msg);
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t x = *reinterpret_cast<const uint64_t*>(mb + W.off[r * 4 + 0]);
        uint64_t y = *reinterpret_cast<const uint64_t*>(mb + W.off[r * 4 + 1]);
        ATLAS_G(a, b, c, d, x, y);
        b = quad_perm64<0x39>(b);   // lane i <- lane (i+1)&3
        c = quad_perm64<0x4E>(c);   // lane i <- lane (i+2)&3
        d = quad_perm64<0x93>(d);   // lane i <- lane (i+3)&3
        x = *reinterpret_cast<const uint64_t*>(mb + W.off[r * 4 + 2]);
        y = *reinterpret_cast<const uint64_t*>(mb + W.off[r * 4 + 3]);
        ATLAS_G(a, b, c, d, x, y);
        b = quad_perm64<0x93>(b);
        c = quad_perm64<0x4E>(c);
        d = quad_perm64<0x39>(d);
    }
    return h_lo ^ a ^ c;
}
#undef ATLAS_G

// LDS scratch of the wave-cooperative transcript: one 128-byte message block.
struct WaveTranscriptLds {
    uint64_t msg[16];
};

// must be called once (by the whole wave) before the first absorb: zero the fixed words
__device__ __forceinline__ void wt_init_lds(WaveTranscriptLds* S, uint32_t lane) {
    if (lane < 16) S->msg[lane] = 0;
}

// state' = BLAKE2b-256(state || 0^28 || n_rounds_be32 || payload) (blake2b.rs:31-37,64-78)
// pw = payload word for this lane's quad position (0 when has_payload == false)
__device__ __forceinline__ void wt_absorb(WaveTranscript& T, WaveTranscriptLds* S, const WaveBlakeSched& W,
                                          uint32_t lane, uint64_t pw, bool has_payload) {
    const uint32_t q = lane & 3;
    if (lane < 4) {
        S->msg[lane] = T.sw;
        S->msg[8 + lane] = pw;
    }
    if (lane == 0) S->msg[7] = (uint64_t)__builtin_bswap32(T.n_rounds) << 32;   // bytes 60..63 BE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    T.sw = wave_blake2b_block(S->msg, W, q, has_payload ? 96 : 64);
    T.n_rounds += 1;
    __builtin_amdgcn_wave_barrier();
}

// append_message with a label packed by the host into 4 LE words (blake2b.rs:109-122)
__device__ __forceinline__ void wt_append_label(WaveTranscript& T, WaveTranscriptLds* S, const WaveBlakeSched& W,
                                                uint32_t lane, const uint64_t label[4]) {
    wt_absorb(T, S, W, lane, sel4(lane & 3, label[0], label[1], label[2], label[3]), true);
}

// append_scalar: 32-byte big-endian canonical integer (blake2b.rs:138-146).
// canon = canonical (non-Montgomery) limbs, same value in every lane.
__device__ __forceinline__ void wt_append_canonical(WaveTranscript& T, WaveTranscriptLds* S,
                                                    const WaveBlakeSched& W, uint32_t lane, const Fr& canon) {
    // BE byte string: word i = byte-swapped canonical u64 limb (3 - i)
    const uint64_t w0 = bswap64(((uint64_t)canon.v[7] << 32) | canon.v[6]);
    const uint64_t w1 = bswap64(((uint64_t)canon.v[5] << 32) | canon.v[4]);
    const uint64_t w2 = bswap64(((uint64_t)canon.v[3] << 32) | canon.v[2]);
    const uint64_t w3 = bswap64(((uint64_t)canon.v[1] << 32) | canon.v[0]);
    wt_absorb(T, S, W, lane, sel4(lane & 3, w0, w1, w2, w3), true);
}

// challenge_u128 (blake2b.rs:57-62,197-202): squeeze, first 16 digest bytes as LE u128
__device__ __forceinline__ void wt_challenge_u128(WaveTranscript& T, WaveTranscriptLds* S, const WaveBlakeSched& W,
                                                  uint32_t lane, uint64_t& lo, uint64_t& hi) {
    wt_absorb(T, S, W, lane, 0, false);
    const uint32_t slo = (uint32_t)T.sw, shi = (uint32_t)(T.sw >> 32);
    lo = ((uint64_t)(uint32_t)__shfl((int)shi, 0, 64) << 32) | (uint32_t)__shfl((int)slo, 0, 64);
    hi = ((uint64_t)(uint32_t)__shfl((int)shi, 1, 64) << 32) | (uint32_t)__shfl((int)slo, 1, 64);
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

// load / store: every lane reads its quad's word (vz = opaque zero keeps it on the VALU)
__device__ __forceinline__ WaveTranscript wt_load(const DevTranscript* d, uint32_t lane, uint32_t vz) {
    WaveTranscript T;
    const uint64_t* p = reinterpret_cast<const uint64_t*>(d->state);
    T.sw = p[(lane & 3) + vz];
    T.n_rounds = (&d->n_rounds)[vz];
    return T;
}

__device__ __forceinline__ void wt_store(DevTranscript* d, const WaveTranscript& T, uint32_t lane) {
    uint64_t* p = reinterpret_cast<uint64_t*>(d->state);
    if (lane < 4) p[lane] = T.sw;
    if (lane == 0) d->n_rounds = T.n_rounds;
}

}  // namespace atlas
