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

// (kept for call-site compatibility: the message schedule now lives in registers)
struct WaveBlakeSched {
    uint32_t unused;
};
__device__ __forceinline__ WaveBlakeSched wave_blake_sched(uint32_t) { return WaveBlakeSched{0}; }

// 64-bit lanes of the hash as explicit 32-bit halves: keeps every value in free-standing
// VGPRs (the compiler's u64 ops want aligned register pairs and pay for it in v_mov).
struct U64 {
    uint32_t lo, hi;
};
__device__ __forceinline__ U64 u64_split(uint64_t x) { return U64{(uint32_t)x, (uint32_t)(x >> 32)}; }
__device__ __forceinline__ uint64_t u64_join(U64 x) { return ((uint64_t)x.hi << 32) | x.lo; }
__device__ __forceinline__ U64 u64_add(U64 a, U64 b) {
    U64 o;
    asm("v_add_co_u32 %0, vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc"
        : "=&v"(o.lo), "=v"(o.hi) : "v"(a.lo), "v"(a.hi), "v"(b.lo), "v"(b.hi) : "vcc");
    return o;
}
__device__ __forceinline__ U64 u64_xor(U64 a, U64 b) { return U64{a.lo ^ b.lo, a.hi ^ b.hi}; }
template <int N>
__device__ __forceinline__ U64 u64_rotr(U64 x) {
    if constexpr (N == 32) return U64{x.hi, x.lo};
    else if constexpr (N < 32) return U64{__builtin_amdgcn_alignbit(x.hi, x.lo, N), __builtin_amdgcn_alignbit(x.lo, x.hi, N)};
    else return U64{__builtin_amdgcn_alignbit(x.lo, x.hi, N - 32), __builtin_amdgcn_alignbit(x.hi, x.lo, N - 32)};
}
template <int CTRL>
__device__ __forceinline__ U64 u64_quad_perm(U64 x) {
    return U64{(uint32_t)__builtin_amdgcn_mov_dpp((int)x.lo, CTRL, 0xf, 0xf, true),
               (uint32_t)__builtin_amdgcn_mov_dpp((int)x.hi, CTRL, 0xf, 0xf, true)};
}
__device__ __forceinline__ U64 u64_sel4(uint32_t q, U64 x0, U64 x1, U64 x2, U64 x3) {
    const bool b0 = q & 1, b1 = q & 2;
    U64 lo{b0 ? x1.lo : x0.lo, b0 ? x1.hi : x0.hi}, hi{b0 ? x3.lo : x2.lo, b0 ? x3.hi : x2.hi};
    return U64{b1 ? hi.lo : lo.lo, b1 ? hi.hi : lo.hi};
}

// G with the message words added only when they can be non-zero (ZX / ZY known at compile time)
template <bool ZX, bool ZY>
__device__ __forceinline__ void blake_g(U64& a, U64& b, U64& c, U64& d, U64 x, U64 y) {
    a = u64_add(a, b); if constexpr (!ZX) a = u64_add(a, x);
    d = u64_rotr<32>(u64_xor(d, a));
    c = u64_add(c, d); b = u64_rotr<24>(u64_xor(b, c));
    a = u64_add(a, b); if constexpr (!ZY) a = u64_add(a, y);
    d = u64_rotr<16>(u64_xor(d, a));
    c = u64_add(c, d); b = u64_rotr<63>(u64_xor(b, c));
}

// One final-block compression from the fixed initial chaining value (every transcript hash
// is a single block: 64-byte prefix + <= 64-byte payload).  M = the 16 message words, each
// uniform across the quad, held in registers; words 4-6 and 12-15 are always zero, 8-11 are
// zero without payload.  The per-lane word of a round is a 4-way select among compile-time
// known registers (no LDS, no waits on the hash's critical path).
// Returns, in lane l, digest word (l & 3).
template <bool PAYLOAD>
__device__ __forceinline__ uint64_t wave_blake2b_block(const U64 M[16], uint32_t q, uint32_t t_bytes) {
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
    const uint64_t iv_lo = sel4(q, 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                0xa54ff53a5f1d36f1ULL);
    const uint64_t iv_hi = sel4(q, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL,
                                0x5be0cd19137e2179ULL);
    const uint64_t h_lo = iv_lo ^ (q == 0 ? 0x01010020ULL : 0ULL);   // digest 32, fanout 1, depth 1
    U64 a = u64_split(h_lo), b = u64_split(iv_hi), c = u64_split(iv_lo);
    U64 d = u64_split(iv_hi ^ (q == 0 ? (uint64_t)t_bytes : 0ULL) ^ (q == 2 ? ~0ULL : 0ULL));
    auto zero_word = [](int w) constexpr { return (w >= 4 && w <= 6) || w >= 12 || (!PAYLOAD && w >= 8); };
#define ATLAS_WORD(r, k0) u64_sel4(q, M[S[r][k0]], M[S[r][k0 + 2]], M[S[r][k0 + 4]], M[S[r][k0 + 6]])
#define ATLAS_ZERO(r, k0) (zero_word(S[r][k0]) && zero_word(S[r][k0 + 2]) && zero_word(S[r][k0 + 4]) && zero_word(S[r][k0 + 6]))
#define ATLAS_ROUND(r)                                                                                   \
    do {                                                                                                  \
        blake_g<ATLAS_ZERO(r, 0), ATLAS_ZERO(r, 1)>(a, b, c, d, ATLAS_WORD(r, 0), ATLAS_WORD(r, 1));       \
        b = u64_quad_perm<0x39>(b); c = u64_quad_perm<0x4E>(c); d = u64_quad_perm<0x93>(d);               \
        blake_g<ATLAS_ZERO(r, 8), ATLAS_ZERO(r, 9)>(a, b, c, d, ATLAS_WORD(r, 8), ATLAS_WORD(r, 9));       \
        b = u64_quad_perm<0x93>(b); c = u64_quad_perm<0x4E>(c); d = u64_quad_perm<0x39>(d);               \
    } while (0)
    ATLAS_ROUND(0); ATLAS_ROUND(1); ATLAS_ROUND(2); ATLAS_ROUND(3); ATLAS_ROUND(4); ATLAS_ROUND(5);
    ATLAS_ROUND(6); ATLAS_ROUND(7); ATLAS_ROUND(8); ATLAS_ROUND(9); ATLAS_ROUND(10); ATLAS_ROUND(11);
#undef ATLAS_ROUND
#undef ATLAS_ZERO
#undef ATLAS_WORD
    return h_lo ^ u64_join(u64_xor(a, c));
}

// (kept for call-site compatibility; no LDS is used any more)
struct WaveTranscriptLds {
    uint64_t unused;
};
__device__ __forceinline__ void wt_init_lds(WaveTranscriptLds*, uint32_t) {}

// state' = BLAKE2b-256(state || 0^28 || n_rounds_be32 || payload) (blake2b.rs:31-37,64-78)
// pw = payload word for this lane's quad position (ignored when has_payload == false)
template <bool PAYLOAD>
__device__ __forceinline__ void wt_absorb_t(WaveTranscript& T, uint32_t lane, uint64_t pw) {
    const uint32_t q = lane & 3;
    U64 M[16];
#pragma unroll
    for (int i = 0; i < 16; i++) M[i] = U64{0, 0};
    const U64 sw = u64_split(T.sw);
    M[0] = u64_quad_perm<0x00>(sw); M[1] = u64_quad_perm<0x55>(sw);      // broadcast lane k of the quad
    M[2] = u64_quad_perm<0xAA>(sw); M[3] = u64_quad_perm<0xFF>(sw);
    M[7] = U64{0, __builtin_bswap32(T.n_rounds)};                        // bytes 60..63 big-endian
    if constexpr (PAYLOAD) {
        const U64 p = u64_split(pw);
        M[8] = u64_quad_perm<0x00>(p); M[9] = u64_quad_perm<0x55>(p);
        M[10] = u64_quad_perm<0xAA>(p); M[11] = u64_quad_perm<0xFF>(p);
    }
    T.sw = wave_blake2b_block<PAYLOAD>(M, q, PAYLOAD ? 96u : 64u);
    T.n_rounds += 1;
}
__device__ __forceinline__ void wt_absorb(WaveTranscript& T, WaveTranscriptLds*, const WaveBlakeSched&,
                                          uint32_t lane, uint64_t pw, bool has_payload) {
    if (has_payload) wt_absorb_t<true>(T, lane, pw);
    else wt_absorb_t<false>(T, lane, pw);
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
