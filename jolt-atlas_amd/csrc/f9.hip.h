// 9 x 29-bit lazy-limb arithmetic for BN254 Fr / Fq on gfx950.
//
// Why: the 8 x 32-bit CIOS of field.hip.h spends half its instructions moving carries around
// (v_mad_u64_u32 has a 64-bit addend but no carry-in, so every limb product needs a second
// 64-bit add plus register-pair shuffles).  With 29-bit limbs a column sum of 18 products
// fits a 64-bit accumulator, so a Montgomery multiplication is 2*81 back-to-back
// v_mad_u64_u32 with the accumulator as addend and ~40 cheap instructions: measured
// 134 G mul/s vs 82 G mul/s (tools/exp_f9.hip), i.e. the multiplier pipe is the only limit.
// Additions are 9 independent v_add_u32 (no carry chain); reductions are deferred.
//
// Conventions.  A value X = sum l[k] 2^(29k).  "Normalized": l[k] < 2^29 (k < 8).  HBM always
// holds the arkworks image (8 x u32 Montgomery residue, R = 2^256); values written by F9
// kernels are congruent mod p and < 2.1p (not necessarily < p); anything that leaves the
// device (proof coefficients, claims) goes through f9_canon* which returns the canonical
// residue, so results stay bit-identical to the reference.
//   mont9(A, B) = A*B*2^-261 mod p  (9 radix-2^29 reduction steps), output normalized, < p + A*B/2^261.
//   The extra 2^-5 relative to R = 2^256 is compensated by pre-scaling one operand by 32
//   (the bind challenge, once per kernel) or by a final multiplication with Montgomery(32).
#pragma once
#include "field.hip.h"

namespace atlas {

struct F9 {
    uint32_t l[9];
};

constexpr uint32_t F9_MASK = (1u << 29) - 1;

struct Fr9Params {
    static constexpr uint32_t INV29 = 0x0fffffffu;
    static constexpr uint32_t TOP2P = 0x0060c89cu;   // floor(2p / 2^232)
    __device__ __host__ static constexpr uint32_t p(int i) {
        constexpr uint32_t V[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u,
                                   0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return V[i];
    }
    __device__ __host__ static constexpr uint32_t twop(int i) {
        constexpr uint32_t V[9] = {0x00000002u, 0x1e1f593fu, 0x1cb848a1u, 0x0fa121e6u, 0x0b0ba506u,
                                   0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu};
        return V[i];
    }
    // 4p with every limb >= 2^30 (k < 8): a + sub4p - b never borrows for b.l[k] < 2^30
    __device__ __host__ static constexpr uint32_t sub4p(int i) {
        constexpr uint32_t V[9] = {0x40000004u, 0x5c3eb27cu, 0x59709141u, 0x5f4243cbu, 0x56174a0au,
                                   0x4b6d0300u, 0x429b8502u, 0x597098ceu, 0x00c19137u};
        return V[i];
    }
    // 2^261 - 2p, normalized limbs: adding it and dropping bit 29 of the top limb subtracts 2p
    __device__ __host__ static constexpr uint32_t comp2p(int i) {
        constexpr uint32_t V[9] = {0x1ffffffeu, 0x01e0a6c0u, 0x0347b75eu, 0x105ede19u, 0x14f45af9u,
                                   0x1a497e7eu, 0x1eb23d7du, 0x0347b397u, 0x1f9f3763u};
        return V[i];
    }
    // t * INV29 mod 2^32: INV29 = 2^28 - 1, two full-rate instructions instead of a quarter-rate v_mul_lo_u32
    __device__ static __forceinline__ uint32_t mul_inv29(uint32_t t) {
        uint32_t sh = t << 28;
        asm("" : "+v"(sh));                 // keeps instcombine from folding the pair back into a multiply
        return sh - t;
    }
    using Base = FrParams;
    // Montgomery(32) = 32 * 2^256 mod p, as 8 x u32 (multiplying by it undoes the 2^-5 of mont9)
    __device__ __host__ static constexpr uint32_t mont32(int i) {
        constexpr uint32_t V[8] = {0x8fffff57u, 0x2fd4e156u, 0xa494b01au, 0x75bba827u,
                                   0x819caa80u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u};
        return V[i];
    }
};

struct Fq9Params {
    static constexpr uint32_t INV29 = 0x04866389u;
    static constexpr uint32_t TOP2P = 0x0060c89cu;
    __device__ __host__ static constexpr uint32_t p(int i) {
        constexpr uint32_t V[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                                   0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return V[i];
    }
    __device__ __host__ static constexpr uint32_t twop(int i) {
        constexpr uint32_t V[9] = {0x10f9fa8eu, 0x0208c16du, 0x18e5469eu, 0x05aa45a1u, 0x0b0bb2f0u,
                                   0x05b68181u, 0x014dc282u, 0x1cb84c68u, 0x0060c89cu};
        return V[i];
    }
    __device__ __host__ static constexpr uint32_t sub4p(int i) {
        constexpr uint32_t V[9] = {0x41f3f51cu, 0x441182d9u, 0x51ca8d3au, 0x4b548b41u, 0x561765deu,
                                   0x4b6d0300u, 0x429b8502u, 0x597098ceu, 0x00c19137u};
        return V[i];
    }
    __device__ __host__ static constexpr uint32_t comp2p(int i) {
        constexpr uint32_t V[9] = {0x0f060572u, 0x1df73e92u, 0x071ab961u, 0x1a55ba5eu, 0x14f44d0fu,
                                   0x1a497e7eu, 0x1eb23d7du, 0x0347b397u, 0x1f9f3763u};
        return V[i];
    }
    __device__ static __forceinline__ uint32_t mul_inv29(uint32_t t) { return t * INV29; }
    using Base = FqParams;
    __device__ __host__ static constexpr uint32_t mont32(int i) {
        constexpr uint32_t V[8] = {0x157ccc21u, 0x4e8384ebu, 0x0ce148c3u, 0xfb90a602u,
                                   0x819caa36u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u};
        return V[i];
    }
};

// 8 x 32 (value < 2^256) -> normalized F9
__device__ __forceinline__ F9 f9_from_fe(const Fe& a) {
    F9 o;
    o.l[0] = a.v[0] & F9_MASK;
    o.l[1] = __builtin_amdgcn_alignbit(a.v[1], a.v[0], 29) & F9_MASK;
    o.l[2] = __builtin_amdgcn_alignbit(a.v[2], a.v[1], 26) & F9_MASK;
    o.l[3] = __builtin_amdgcn_alignbit(a.v[3], a.v[2], 23) & F9_MASK;
    o.l[4] = __builtin_amdgcn_alignbit(a.v[4], a.v[3], 20) & F9_MASK;
    o.l[5] = __builtin_amdgcn_alignbit(a.v[5], a.v[4], 17) & F9_MASK;
    o.l[6] = __builtin_amdgcn_alignbit(a.v[6], a.v[5], 14) & F9_MASK;
    o.l[7] = __builtin_amdgcn_alignbit(a.v[7], a.v[6], 11) & F9_MASK;
    o.l[8] = a.v[7] >> 8;
    return o;
}

// normalized F9 with value < 2^256 -> 8 x 32
__device__ __forceinline__ Fe f9_to_fe(const F9& a) {
    Fe o;
    o.v[0] = a.l[0] | (a.l[1] << 29);
    o.v[1] = (a.l[1] >> 3) | (a.l[2] << 26);
    o.v[2] = (a.l[2] >> 6) | (a.l[3] << 23);
    o.v[3] = (a.l[3] >> 9) | (a.l[4] << 20);
    o.v[4] = (a.l[4] >> 12) | (a.l[5] << 17);
    o.v[5] = (a.l[5] >> 15) | (a.l[6] << 14);
    o.v[6] = (a.l[6] >> 18) | (a.l[7] << 11);
    o.v[7] = (a.l[7] >> 21) | (a.l[8] << 8);
    return o;
}

__device__ __forceinline__ F9 f9_load(const Fe* p) { return f9_from_fe(fe_load(p)); }

__device__ __forceinline__ F9 f9_zero() {
    F9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = 0;
    return o;
}

// limb-wise add, no carries: limb bounds add up
__device__ __forceinline__ F9 f9_add(const F9& a, const F9& b) {
    F9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = a.l[i] + b.l[i];
    return o;
}

// a - b + 4p, no borrows; needs b.l[k] < 2^30 (k < 8) and b < 4p
template <class P9>
__device__ __forceinline__ F9 f9_sub(const F9& a, const F9& b) {
    F9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = a.l[i] + P9::sub4p(i) - b.l[i];
    return o;
}

// carry propagation only (value unchanged): limbs -> < 2^29, top limb takes the rest
__device__ __forceinline__ F9 f9_norm(const F9& a) {
    F9 o;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t s = a.l[i] + c;      // a.l[i] < 2^32 - 2^3: no wrap
        o.l[i] = s & F9_MASK;
        c = s >> 29;
    }
    o.l[8] = a.l[8] + c;
    return o;
}

// a + c * b for a small constant c (c * b.l[i] + a.l[i] < 2^63), carries propagated: normalized limbs, the value
// is NOT reduced (a + c b must stay below 2^261; as the lazy operand of f9_mul it may be several p)
__device__ __forceinline__ F9 f9_axpy_small(const F9& a, const F9& b, uint32_t c) {
    F9 o;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t s = (uint64_t)b.l[i] * c + a.l[i] + carry;
        o.l[i] = (uint32_t)s & F9_MASK;
        carry = s >> 29;
    }
    o.l[8] = (uint32_t)((uint64_t)b.l[8] * c + a.l[8] + carry);
    return o;
}

// carry propagation + one conditional subtraction of 2p: for an input < ~4p the result is
// normalized and < 2p(1 + 2^-20).  The decision uses a lower estimate of the top limb, so
// the subtraction never underflows.
template <class P9, int MAXK = 2>
__device__ __forceinline__ F9 f9_norm_red(const F9& a) {
    // k = how many times 2p is subtracted (0..MAXK): any input < (2 MAXK + 2) p comes out
    // < 2p(1+2^-20).  floor(2jp / 2^232) <= j TOP2P + j, so top_est > j TOP2P + j  =>  value > 2jp.
    const uint32_t top_est = a.l[8] + (a.l[7] >> 29);
    uint32_t k = 0;
#pragma unroll
    for (uint32_t j = 1; j <= (uint32_t)MAXK; j++) k = top_est > j * P9::TOP2P + j ? j : k;
    F9 o;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t lo = a.l[i] & F9_MASK, hi = a.l[i] >> 29;      // a.l[i] may be >= 2^31
        const int32_t s = (int32_t)lo - (int32_t)(k * P9::twop(i)) + c;
        o.l[i] = (uint32_t)s & F9_MASK;
        c = (s >> 29) + (int32_t)hi;
    }
    o.l[8] = a.l[8] - k * P9::twop(8) + (uint32_t)c;
    return o;
}

// A*B*2^-261 mod p.  Limb bounds: one operand < 2^29 per limb and the other < 2^31.4, or both
// < 2^30 (column sums stay below 2^64).  LO = first limb of b that may be non-zero.
// Output normalized; value < p + A*B/2^261.
template <class P9, int LO = 0>
__device__ __forceinline__ F9 f9_mul(const F9& a, const F9& b) {
    uint64_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = LO; i < 9; i++) {
        const uint32_t bi = b.l[i];
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)a.l[j] * bi;
        const uint32_t m = P9::mul_inv29((uint32_t)t[0]) & F9_MASK;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)m * P9::p(j);
        const uint64_t carry = t[0] >> 29;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] = t[j + 1];
        t[0] += carry;
        t[9] = 0;
    }
    F9 o;
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { c += t[j]; o.l[j] = (uint32_t)c & F9_MASK; c >>= 29; }
    o.l[8] = (uint32_t)(c + t[8]);
    return o;
}

// (s - [s >= 2p] 2p) + A*B*2^-261: the multiplication with a running sum folded in.  s normalized (limbs < 2^29, value
// < 2^256).  Limb k of the addend rides as the START value of the accumulator that ends up at index k — t[9] of reduction
// step k, which the Montgomery steps never see at index 0 — so the addition costs no instruction beyond the multiply-add it
// seeds; the conditional subtraction is "+ (2^261 - 2p)" on those limbs with bit 29 of the top limb dropped at the end.
// Output normalized; value < 2p(1 + 2^-22) + p + A*B/2^261 (a fixed point for running sums and for a + r(b - a)): it
// replaces f9_norm_red(f9_add(s, f9_mul(a, b))), whose carry chain with a multiply per limb was a tenth of the data pass.
template <class P9, int LO = 0>
__device__ __forceinline__ F9 f9_mul_addred(const F9& a, const F9& b, const F9& s) {
    const uint32_t mask = s.l[8] > P9::TOP2P ? 0xffffffffu : 0u;        // s.l[8] >= TOP2P + 1  =>  s > 2p
    uint32_t x[9];
#pragma unroll
    for (int i = 0; i < 9; i++) x[i] = s.l[i] + (P9::comp2p(i) & mask);
    uint64_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int k = 0; k < LO; k++) t[9 - LO + k] = x[k];                   // skipped steps: the limbs they would have seeded
#pragma unroll
    for (int i = LO; i < 9; i++) {
        const uint32_t bi = b.l[i];
        t[9] = x[i];
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)a.l[j] * bi;
        const uint32_t m = P9::mul_inv29((uint32_t)t[0]) & F9_MASK;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)m * P9::p(j);
        const uint64_t carry = t[0] >> 29;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] = t[j + 1];
        t[0] += carry;
    }
    F9 o;
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { c += t[j]; o.l[j] = (uint32_t)c & F9_MASK; c >>= 29; }
    o.l[8] = (uint32_t)(c + t[8]) & F9_MASK;
    return o;
}

// 32 * x for a normalized x < 2^256 (pre-scaling of the bind challenge): shift by 5 bits
__device__ __forceinline__ F9 f9_shl5(const F9& a) {
    F9 o;
    o.l[0] = (a.l[0] << 5) & F9_MASK;
#pragma unroll
    for (int i = 1; i < 8; i++) o.l[i] = ((a.l[i] << 5) | (a.l[i - 1] >> 24)) & F9_MASK;
    o.l[8] = (a.l[8] << 5) | (a.l[7] >> 24);
    return o;
}

// exact canonical residue (8 x u32, < p) of a normalized F9 with value < 2^256
template <class P9>
__device__ __forceinline__ Fe f9_canon(const F9& a) {
    Fe x = f9_to_fe(a);
    Fe one = fe_one<typename P9::Base>();
    return fe_mul<typename P9::Base, 0>(x, one);        // x * R * R^-1, fully reduced
}

// canonical residue of 32*a: undoes the 2^-5 carried by sums of f9_mul products
template <class P9>
__device__ __forceinline__ Fe f9_canon_x32(const F9& a) {
    Fe x = f9_to_fe(a);
    Fe k;
#pragma unroll
    for (int i = 0; i < 8; i++) k.v[i] = P9::mont32(i);
    return fe_mul<typename P9::Base, 0>(x, k);          // x * 32R * R^-1
}

__device__ __forceinline__ F9 f9_shfl_xor(const F9& a, int mask) {
    F9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = __shfl_xor(a.l[i], mask, 64);
    return o;
}

// wavefront sum with reduction to < ~2p after every step
template <class P9>
__device__ __forceinline__ F9 f9_wave_sum(F9 a) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a = f9_norm_red<P9>(f9_add(a, f9_shfl_xor(a, m)));
    return a;
}

// Wavefront sum WITHOUT reductions: limb-wise adds, a carry pass after every third doubling (8 x (2^29 - 1) fits a word).  Input: normalized
// limbs, value < ~4 p.  Output in EVERY lane: normalized limbs (the top limb unmasked), value = the exact integer sum (< 64 x the input bound:
// top limb < 2^31) — for sums that are mailed as lazy limbs and reduced once on the host (channel.hpp: sum_to_fr), or reduced once by
// f9_reduce_i64.  The steps inside a 16-lane row are DPP moves (no LDS crossbar): 72 of the 108 ds_bpermute of f9_wave_sum go, and all
// twelve reductions with their per-limb multiplies (the wavefront sum of two accumulators was 1 150 of the ~1 400 instructions behind the
// loop of a data pass: 2 us on the critical path of every pass).  EVERY lane of the wavefront must be active (DPP reads disabled lanes as
// whatever they hold): call it outside divergent code, with zero in the lanes that have nothing to add.
template <int CTRL>
__device__ __forceinline__ F9 f9_add_dpp(const F9& a) {
    F9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = a.l[i] + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.l[i], CTRL, 0xf, 0xf, false);
    return o;
}
__device__ __forceinline__ F9 f9_wave_sum_lazy(F9 a) {
    a = f9_add_dpp<0xb1>(a);                 // + lane ^ 1
    a = f9_add_dpp<0x4e>(a);                 // + lane ^ 2: the quad
    a = f9_add_dpp<0x124>(a);                // + the quad 4 lanes round the row: 8 lanes
    a = f9_norm(a);
    a = f9_add_dpp<0x128>(a);                // + 8 lanes round the row: the 16-lane row
    a = f9_add(a, f9_shfl_xor(a, 16));
    a = f9_add(a, f9_shfl_xor(a, 32));
    return f9_norm(a);
}

// ---- small signed linear combinations (the extrapolation steps of the split product, ra.hip)
// V = sum t[i] 2^(29 i) with signed 64-bit columns, |V| < 2^16.5 p, |t[i]| < 2^46  ->  the representative of V mod p in (0.99 p, 2.01 p),
// normalized.  q = floor(V / p) - 1 is estimated from the two top columns in single precision: the top columns give V / 2^232 within
// [-1, +2), their shift by 8 bits is exact in an int32, and the float product is off by less than 0.005 (tools/model_split16.py walks
// the bounds with exact integers); the quotient then rides through the same signed carry pass as the columns.
template <class P9>
__device__ __forceinline__ F9 f9_reduce_i64(int64_t (&t)[9]) {
    const int64_t vt = t[8] + (t[7] >> 29);
    constexpr float INV = 256.0f / 3171406.4487f;                                   // 2^240 / p
    const int32_t q = (int32_t)__builtin_floorf((float)(int32_t)(vt >> 8) * INV) - 1;
    F9 o;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int64_t s = t[i] - (int64_t)q * (int32_t)P9::p(i) + c;
        o.l[i] = (uint32_t)s & F9_MASK;
        c = s >> 29;
    }
    o.l[8] = (uint32_t)(t[8] - (int64_t)q * (int32_t)P9::p(8) + c);
    return o;
}

// the same for a value held as limbs of up to 32 bits (the output of f9_wave_sum_lazy, or a few of them added): (0.99 p, 2.01 p), normalized
template <class P9>
__device__ __forceinline__ F9 f9_reduce_lazy(const F9& a) {
    int64_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = (int64_t)a.l[i];
    return f9_reduce_i64<P9>(t);
}

// sum_j c[j] * v[j] mod p for small signed constants (sum |c[j]| v[j] < 2^16.5 p; v[j] normalized): (0.99 p, 2.01 p), normalized
template <class P9, int N>
__device__ __forceinline__ F9 f9_lincomb(const F9* const (&v)[N], const int (&c)[N]) {
    int64_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        t[i] = 0;
#pragma unroll
        for (int j = 0; j < N; j++) t[i] += (int64_t)(int32_t)v[j]->l[i] * c[j];
    }
    return f9_reduce_i64<P9>(t);
}

// the limbs of another lane by a DPP pattern (quad_perm 0x00-0xff, row_shr 0x110 + n, row_ror 0x120 + n); every lane must be active
template <int CTRL>
__device__ __forceinline__ F9 f9_dpp(const F9& a) {
    F9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.l[i], CTRL, 0xf, 0xf, false);
    return o;
}

}  // namespace atlas
