// BN254 Fr / Fq device arithmetic for gfx950.
//
// Replaces, on the device, the scalar arithmetic the reference forwards to arkworks
// (joltworks/src/field/ark.rs:241-297; MontU128Challenge multiply
// joltworks/src/field/challenge/macros.rs:274-283).  A field element is the arkworks
// memory image — 4 x u64 little-endian limbs of the Montgomery residue a*2^256 mod p —
// viewed as 8 x u32 so that one element is two global_load_dwordx4 and the 32x32->64
// multiplier (v_mad_u64_u32) is used directly.  No MFMA: these are modular reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace atlas {

struct alignas(16) Fe {  // generic 256-bit field element (Fr or Fq), Montgomery form
    uint32_t v[8];
};

struct FrParams {
    static constexpr uint32_t INV = 0xefffffffu;  // -p^-1 mod 2^32
    __device__ __host__ static constexpr uint32_t p(int i) {
        constexpr uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                   0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return P[i];
    }
    __device__ __host__ static constexpr uint32_t one(int i) {  // R mod p
        constexpr uint32_t V[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                   0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return V[i];
    }
    __device__ __host__ static constexpr uint32_t r2(int i) {  // R^2 mod p
        constexpr uint32_t V[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                   0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return V[i];
    }
};

struct FqParams {
    static constexpr uint32_t INV = 0xe4866389u;
    __device__ __host__ static constexpr uint32_t p(int i) {
        constexpr uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                   0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return P[i];
    }
    __device__ __host__ static constexpr uint32_t one(int i) {
        constexpr uint32_t V[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                   0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return V[i];
    }
    __device__ __host__ static constexpr uint32_t r2(int i) {
        constexpr uint32_t V[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                   0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return V[i];
    }
};

// ---- 256-bit helpers ---------------------------------------------------------------
template <class P>
__device__ __forceinline__ bool fe_geq_p(const uint32_t* a) {
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (a[i] > P::p(i)) return true;
        if (a[i] < P::p(i)) return false;
    }
    return true;
}

// o = a - p if a >= p else a   (branch-free: subtract, keep by borrow)
template <class P>
__device__ __forceinline__ void fe_cond_sub_p(uint32_t* a) {
    uint32_t t[8];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)a[i] - P::p(i) - br;
        t[i] = (uint32_t)d;
        br = (d >> 32) & 1;
    }
    if (!br) {
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = t[i];
    }
}

template <class P>
__device__ __forceinline__ Fe fe_add(const Fe& a, const Fe& b) {
    Fe o;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        o.v[i] = (uint32_t)c;
        c >>= 32;
    }
    // p < 2^254 so a+b < 2^255: no carry out of limb 7
    fe_cond_sub_p<P>(o.v);
    return o;
}

template <class P>
__device__ __forceinline__ Fe fe_sub(const Fe& a, const Fe& b) {
    Fe o;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)a.v[i] - b.v[i] - br;
        o.v[i] = (uint32_t)d;
        br = (d >> 32) & 1;
    }
    if (br) {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += (uint64_t)o.v[i] + P::p(i);
            o.v[i] = (uint32_t)c;
            c >>= 32;
        }
    }
    return o;
}

template <class P>
__device__ __forceinline__ Fe fe_dbl(const Fe& a) { return fe_add<P>(a, a); }

template <class P>
__device__ __forceinline__ Fe fe_neg(const Fe& a) {
    Fe z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.v[i] = 0;
    return fe_sub<P>(z, a);
}

__device__ __forceinline__ bool fe_is_zero(const Fe& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}

__device__ __forceinline__ bool fe_eq(const Fe& a, const Fe& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

__device__ __forceinline__ Fe fe_zero() {
    Fe z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.v[i] = 0;
    return z;
}

template <class P>
__device__ __forceinline__ Fe fe_one() {
    Fe z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.v[i] = P::one(i);
    return z;
}

// ---- Montgomery multiplication (CIOS, 32-bit limbs) ----------------------------------
// LO = index of the first limb of b that may be non-zero.  LO = 4 is the
// MontU128Challenge fast path: limbs [0,0,lo,hi] (as u64) have four zero u32 limbs, and
// the CIOS outer iterations for a zero b-limb on a zero accumulator are no-ops, which is
// exactly the saving `mul_hi_bigint_u128` takes on the CPU.
template <class P, int LO = 0>
__device__ __forceinline__ Fe fe_mul(const Fe& a, const Fe& b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = LO; i < 8; i++) {
        uint64_t c = 0;
        const uint32_t bi = b.v[i];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t s = (uint64_t)a.v[j] * bi + t[j] + c;
            t[j] = (uint32_t)s;
            c = s >> 32;
        }
        uint32_t t8 = t[8] + (uint32_t)c;  // p < 2^254: the running value stays < 2^(32*9)
        const uint32_t m = t[0] * P::INV;
        uint64_t s = (uint64_t)m * P::p(0) + t[0];
        c = s >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            s = (uint64_t)m * P::p(j) + t[j] + c;
            t[j - 1] = (uint32_t)s;
            c = s >> 32;
        }
        s = (uint64_t)t8 + c;
        t[7] = (uint32_t)s;
        t[8] = (uint32_t)(s >> 32);
    }
    fe_cond_sub_p<P>(t);
    Fe o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.v[i] = t[i];
    return o;
}

template <class P>
__device__ __forceinline__ Fe fe_sqr(const Fe& a) { return fe_mul<P>(a, a); }

// a * small (u32) : Montgomery form of a*k is (aR)*k mod p -> multiply then reduce by
// repeated conditional subtraction is not enough for large k; use mont_mul with k*R2 path
// only where needed.  For tiny constants use additions.
template <class P>
__device__ __forceinline__ Fe fe_mul3(const Fe& a) { return fe_add<P>(fe_dbl<P>(a), a); }

// Montgomery residue of a canonical integer (< p): a * R^2 * R^-1
template <class P>
__device__ __forceinline__ Fe fe_to_mont(const Fe& canon) {
    Fe r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.v[i] = P::r2(i);
    return fe_mul<P>(canon, r2);
}

// canonical integer of a Montgomery residue: a * 1 * R^-1
template <class P>
__device__ __forceinline__ Fe fe_from_mont(const Fe& a) {
    Fe one = fe_zero();
    one.v[0] = 1;
    return fe_mul<P>(a, one);
}

// ---- memory -------------------------------------------------------------------------
__device__ __forceinline__ Fe fe_load(const Fe* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    Fe o;
    o.v[0] = lo.x; o.v[1] = lo.y; o.v[2] = lo.z; o.v[3] = lo.w;
    o.v[4] = hi.x; o.v[5] = hi.y; o.v[6] = hi.z; o.v[7] = hi.w;
    return o;
}

__device__ __forceinline__ void fe_store(Fe* p, const Fe& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

__device__ __forceinline__ Fe fe_shfl_xor(const Fe& a, int mask) {
    Fe o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.v[i] = __shfl_xor(a.v[i], mask, 64);
    return o;
}

__device__ __forceinline__ Fe fe_shfl(const Fe& a, int lane) {
    Fe o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.v[i] = __shfl(a.v[i], lane, 64);
    return o;
}

// Fr shorthands
using Fr = Fe;
__device__ __forceinline__ Fr fr_add(const Fr& a, const Fr& b) { return fe_add<FrParams>(a, b); }
__device__ __forceinline__ Fr fr_sub(const Fr& a, const Fr& b) { return fe_sub<FrParams>(a, b); }
__device__ __forceinline__ Fr fr_dbl(const Fr& a) { return fe_dbl<FrParams>(a); }
__device__ __forceinline__ Fr fr_neg(const Fr& a) { return fe_neg<FrParams>(a); }
__device__ __forceinline__ Fr fr_mul(const Fr& a, const Fr& b) { return fe_mul<FrParams, 0>(a, b); }
// a * challenge, challenge = Montgomery limbs with v[0..3] == 0 (mode 0)
__device__ __forceinline__ Fr fr_mul_hi(const Fr& a, const Fr& c) { return fe_mul<FrParams, 4>(a, c); }
__device__ __forceinline__ Fr fr_one() { return fe_one<FrParams>(); }

// wavefront (64-lane) sum of one Fr per lane; result valid in every lane
__device__ __forceinline__ Fr fr_wave_sum(Fr a) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a = fr_add(a, fe_shfl_xor(a, m));
    return a;
}

// Fr of a signed 64-bit integer (JoltField::from_i64, ark.rs:127-150)
__device__ __forceinline__ Fr fr_from_i64(int64_t x) {
    uint64_t mag = x < 0 ? (uint64_t)0 - (uint64_t)x : (uint64_t)x;
    Fr c = fe_zero();
    c.v[0] = (uint32_t)mag;
    c.v[1] = (uint32_t)(mag >> 32);
    Fr m = fe_to_mont<FrParams>(c);
    return x < 0 ? fr_neg(m) : m;
}

}  // namespace atlas
