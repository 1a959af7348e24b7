// BN254 G1 on the device: XYZZ coordinates (x = X/ZZ, y = Y/ZZZ) over Fq.
//
// Replaces the group arithmetic the reference gets from arkworks (EXTERNAL): the bucket
// additions inside VariableBaseMSM::msm (joltworks/src/msm/mod.rs:27-181) and the batched
// affine additions of commit_one_hot (hyperkzg/mod.rs:520-554).  A group element has one
// affine representative, so the result bytes do not depend on the coordinate system used
// on the way.  Formulas: EFD shortw/xyzz madd-2008-s, add-2008-s, dbl-2008-s-1 (a = 0).
#pragma once
#include "field.hip.h"

namespace atlas {

using Fq = Fe;
__device__ __forceinline__ Fq fq_add(const Fq& a, const Fq& b) { return fe_add<FqParams>(a, b); }
__device__ __forceinline__ Fq fq_sub(const Fq& a, const Fq& b) { return fe_sub<FqParams>(a, b); }
__device__ __forceinline__ Fq fq_dbl(const Fq& a) { return fe_add<FqParams>(a, a); }
__device__ __forceinline__ Fq fq_neg(const Fq& a) { return fe_neg<FqParams>(a); }
__device__ __forceinline__ Fq fq_mul(const Fq& a, const Fq& b) { return fe_mul<FqParams, 0>(a, b); }
__device__ __forceinline__ Fq fq_sqr(const Fq& a) { return fe_mul<FqParams, 0>(a, a); }
__device__ __forceinline__ Fq fq_one() { return fe_one<FqParams>(); }

struct alignas(16) G1Affine {   // 64 bytes; (0, 0) encodes the point at infinity
    Fq x, y;
};

struct G1Xyzz {                 // ZZ == 0 <=> infinity
    Fq x, y, zz, zzz;
};

__device__ __forceinline__ G1Xyzz g1_inf() {
    G1Xyzz p;
    p.x = fe_zero(); p.y = fe_zero(); p.zz = fe_zero(); p.zzz = fe_zero();
    return p;
}
__device__ __forceinline__ bool g1_is_inf(const G1Xyzz& p) { return fe_is_zero(p.zz); }
__device__ __forceinline__ bool g1_aff_is_inf(const G1Affine& p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }

__device__ __forceinline__ G1Affine g1_aff_load(const G1Affine* p) {
    G1Affine o;
    o.x = fe_load(&p->x);
    o.y = fe_load(&p->y);
    return o;
}

__device__ __forceinline__ G1Xyzz g1_from_aff(const G1Affine& a) {
    G1Xyzz p;
    if (g1_aff_is_inf(a)) return g1_inf();
    p.x = a.x; p.y = a.y; p.zz = fq_one(); p.zzz = fq_one();
    return p;
}

// 2 * (affine point)  [mdbl-2008-s-1]
__device__ __forceinline__ G1Xyzz g1_dbl_aff(const G1Affine& a) {
    G1Xyzz o;
    Fq U = fq_dbl(a.y);
    Fq V = fq_sqr(U);
    Fq W = fq_mul(U, V);
    Fq S = fq_mul(a.x, V);
    Fq xx = fq_sqr(a.x);
    Fq M = fq_add(fq_dbl(xx), xx);
    o.x = fq_sub(fq_sqr(M), fq_dbl(S));
    o.y = fq_sub(fq_mul(M, fq_sub(S, o.x)), fq_mul(W, a.y));
    o.zz = V;
    o.zzz = W;
    return o;
}

// 2 * P  [dbl-2008-s-1]
__device__ __forceinline__ G1Xyzz g1_dbl(const G1Xyzz& p) {
    if (g1_is_inf(p)) return p;
    G1Xyzz o;
    Fq U = fq_dbl(p.y);
    Fq V = fq_sqr(U);
    Fq W = fq_mul(U, V);
    Fq S = fq_mul(p.x, V);
    Fq xx = fq_sqr(p.x);
    Fq M = fq_add(fq_dbl(xx), xx);
    o.x = fq_sub(fq_sqr(M), fq_dbl(S));
    o.y = fq_sub(fq_mul(M, fq_sub(S, o.x)), fq_mul(W, p.y));
    o.zz = fq_mul(V, p.zz);
    o.zzz = fq_mul(W, p.zzz);
    return o;
}

// P + (affine Q), Q != infinity  [madd-2008-s]; negate_q adds -Q
__device__ __forceinline__ G1Xyzz g1_madd(const G1Xyzz& p, const G1Affine& q_in, bool negate_q) {
    G1Affine q = q_in;
    if (negate_q) q.y = fq_neg(q.y);
    if (g1_is_inf(p)) {
        G1Xyzz o;
        o.x = q.x; o.y = q.y; o.zz = fq_one(); o.zzz = fq_one();
        return o;
    }
    Fq U2 = fq_mul(q.x, p.zz);
    Fq S2 = fq_mul(q.y, p.zzz);
    Fq P = fq_sub(U2, p.x);
    Fq R = fq_sub(S2, p.y);
    if (fe_is_zero(P)) {
        if (fe_is_zero(R)) return g1_dbl_aff(q);
        return g1_inf();
    }
    Fq PP = fq_sqr(P);
    Fq PPP = fq_mul(P, PP);
    Fq Q = fq_mul(p.x, PP);
    G1Xyzz o;
    o.x = fq_sub(fq_sub(fq_sqr(R), PPP), fq_dbl(Q));
    o.y = fq_sub(fq_mul(R, fq_sub(Q, o.x)), fq_mul(p.y, PPP));
    o.zz = fq_mul(p.zz, PP);
    o.zzz = fq_mul(p.zzz, PPP);
    return o;
}

// P + Q  [add-2008-s]
__device__ __forceinline__ G1Xyzz g1_add(const G1Xyzz& p, const G1Xyzz& q) {
    if (g1_is_inf(p)) return q;
    if (g1_is_inf(q)) return p;
    Fq U1 = fq_mul(p.x, q.zz);
    Fq U2 = fq_mul(q.x, p.zz);
    Fq S1 = fq_mul(p.y, q.zzz);
    Fq S2 = fq_mul(q.y, p.zzz);
    Fq P = fq_sub(U2, U1);
    Fq R = fq_sub(S2, S1);
    if (fe_is_zero(P)) {
        if (fe_is_zero(R)) return g1_dbl(p);
        return g1_inf();
    }
    Fq PP = fq_sqr(P);
    Fq PPP = fq_mul(P, PP);
    Fq Q = fq_mul(U1, PP);
    G1Xyzz o;
    o.x = fq_sub(fq_sub(fq_sqr(R), PPP), fq_dbl(Q));
    o.y = fq_sub(fq_mul(R, fq_sub(Q, o.x)), fq_mul(S1, PPP));
    o.zz = fq_mul(fq_mul(p.zz, q.zz), PP);
    o.zzz = fq_mul(fq_mul(p.zzz, q.zzz), PPP);
    return o;
}

__device__ __forceinline__ void g1_store(G1Xyzz* dst, const G1Xyzz& p) {
    fe_store(&dst->x, p.x); fe_store(&dst->y, p.y); fe_store(&dst->zz, p.zz); fe_store(&dst->zzz, p.zzz);
}
__device__ __forceinline__ G1Xyzz g1_load(const G1Xyzz* src) {
    G1Xyzz p;
    p.x = fe_load(&src->x); p.y = fe_load(&src->y); p.zz = fe_load(&src->zz); p.zzz = fe_load(&src->zzz);
    return p;
}

// a^(q-2) (Fermat), for XYZZ -> affine on the device (SRS generation only)
__device__ inline Fq fq_inv(const Fq& a) {
    // q - 2, little-endian u32 limbs
    const uint32_t e[8] = {0xd87cfd45u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                           0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    Fq acc = fq_one(), base = a;
    for (int i = 0; i < 254; i++) {
        if ((e[i >> 5] >> (i & 31)) & 1) acc = fq_mul(acc, base);
        base = fq_sqr(base);
    }
    return acc;
}

__device__ inline G1Affine g1_to_aff(const G1Xyzz& p) {
    G1Affine o;
    if (g1_is_inf(p)) { o.x = fe_zero(); o.y = fe_zero(); return o; }
    // ZZ = Z^2, ZZZ = Z^3: ZZ/ZZZ = 1/Z, so 1/ZZ = (ZZ/ZZZ)^2
    Fq izzz = fq_inv(p.zzz);
    Fq izz = fq_sqr(fq_mul(izzz, p.zz));
    o.x = fq_mul(p.x, izz);
    o.y = fq_mul(p.y, izzz);
    return o;
}

}  // namespace atlas
