// BN254 G1 mixed addition on the 29-bit lazy-limb arithmetic (f9.hip.h) — the inner loop of the
// Pippenger bucket accumulation (msm_kernels.hip.h).  Same formula as g1_madd in curve.hip.h
// (EFD xyzz madd-2008-s); coordinates are kept as normalized F9 residues < 2.06 q between
// additions and written back canonical once per bucket, so the result bytes are unchanged.
//
// Lazy-reduction schedule (values in multiples of q, limbs in bits):
//   products            normalized, < 1.2 q
//   P = U2 - X1 + 4q    lazy (< 5.1 q, limbs < 2^31) -> f9_norm before squaring
//   X3 = RR - PPP - 2Q  lazy (< 9.2 q)              -> f9_norm_red<4>  (< 2.06 q)
//   T  = Q - X3 + 4q    lazy, multiplied by normalized R
//   Y3 = R T - Y1 PPP   lazy (< 5.2 q)              -> f9_norm_red<2>
// f9_mul yields A B 2^-261; with Montgomery residues (R = 2^256) that is the Montgomery
// product times 2^-5, so one operand of every product is taken from the "x32" copies below.
#pragma once
#include "curve.hip.h"
#include "f9.hip.h"

namespace atlas {

struct G1Xyzz9 {
    F9 x, y, zz, zzz;
    bool inf;
};

__device__ __forceinline__ G1Xyzz9 g1_inf_f9() {
    G1Xyzz9 a;
    a.x = f9_zero(); a.y = f9_zero(); a.zz = f9_zero(); a.zzz = f9_zero();
    a.inf = true;
    return a;
}

// 32 * a mod-free (value grows 32x): only for canonical inputs (a < q): result < 32 q < 2^259
__device__ __forceinline__ F9 f9_x32(const F9& a) { return f9_shl5(a); }

// k*q (k = 0..5) test of a normalized value known to be in [0, 6q): true iff value == k q
template <class P9>
__device__ __forceinline__ bool f9_is_multiple_of_p(const F9& a) {
    // the lowest limb decides almost always (a match has probability 6 / 2^29): six compares instead of six 9-limb ones
    bool maybe = false;
#pragma unroll
    for (uint32_t k = 0; k <= 5; k++) maybe = maybe || a.l[0] == ((k * P9::p(0)) & F9_MASK);
    if (!maybe) return false;
    bool any = false;
#pragma unroll
    for (uint32_t k = 0; k <= 5; k++) {
        // limbs of k*q, normalized
        uint32_t c = 0, diff = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t s = k * P9::p(i) + c;
            diff |= (s & F9_MASK) ^ a.l[i];
            c = s >> 29;
        }
        diff |= (k * P9::p(8) + c) ^ a.l[8];
        any = any || diff == 0;
    }
    return any;
}

// acc += (x2, y2) [affine, canonical Montgomery residues, not infinity]; negate -> acc -= point
__device__ __forceinline__ void g1_madd_f9(G1Xyzz9& acc, const G1Affine& q_in, bool negate) {
    using P9 = Fq9Params;
    G1Affine q = q_in;
    if (negate) q.y = fq_neg(q.y);
    if (acc.inf) {
        acc.x = f9_from_fe(q.x); acc.y = f9_from_fe(q.y);
        acc.zz = f9_from_fe(fq_one()); acc.zzz = acc.zz;
        acc.inf = false;
        return;
    }
    // x2, y2 pre-scaled by 32 so that f9_mul returns true Montgomery products
    const F9 x2 = f9_x32(f9_from_fe(q.x)), y2 = f9_x32(f9_from_fe(q.y));
    const F9 U2 = f9_mul<P9>(acc.zz, x2);                 // limbs: zz normalized, x2 normalized
    const F9 S2 = f9_mul<P9>(acc.zzz, y2);
    const F9 P = f9_norm(f9_sub<P9>(U2, acc.x));          // < 5.1 q
    const F9 R = f9_norm(f9_sub<P9>(S2, acc.y));
    if (f9_is_multiple_of_p<P9>(P)) {
        // same x: doubling or cancellation — rare; take the exact 8x32 path
        G1Xyzz a;
        a.x = f9_canon<P9>(acc.x); a.y = f9_canon<P9>(acc.y); a.zz = f9_canon<P9>(acc.zz); a.zzz = f9_canon<P9>(acc.zzz);
        const G1Xyzz r = g1_madd(a, q, false);
        acc.inf = g1_is_inf(r);
        acc.x = f9_from_fe(r.x); acc.y = f9_from_fe(r.y); acc.zz = f9_from_fe(r.zz); acc.zzz = f9_from_fe(r.zzz);
        return;
    }
    // the remaining products need one operand scaled by 32: scale P and R once (values < 5.1 q -> < 164 q,
    // still inside the 9-limb range 2^261 = 168 q, and f9_mul's output bound q + A B / 2^261 stays < 2 q
    // because the partner operands are < 5.1 q: 164 * 5.1 / 168 < 5)
    const F9 P32 = f9_norm(f9_shl5(P));                   // normalized limbs, value 32 P
    const F9 PP = f9_mul<P9>(P, P32);                     // P^2  (Montgomery), < q + 5.1*164/168 q < 6 q ... reduce
    const F9 PPn = f9_norm_red<P9, 3>(PP);                 // PP < 6.6 q
    const F9 PP32 = f9_norm(f9_shl5(PPn));                // < 66 q
    const F9 PPP = f9_norm_red<P9, 2>(f9_mul<P9>(P, PP32));
    const F9 Q = f9_norm_red<P9, 2>(f9_mul<P9>(acc.x, PP32));
    const F9 R32 = f9_norm(f9_shl5(R));
    const F9 RR = f9_norm_red<P9, 3>(f9_mul<P9>(R, R32));
    // X3 = RR - PPP - 2Q
    const F9 X3 = f9_norm_red<P9, 5>(f9_sub<P9>(f9_sub<P9>(RR, PPP), f9_add(Q, Q)));
    // Y3 = R (Q - X3) - Y1 PPP
    const F9 T = f9_norm(f9_sub<P9>(Q, X3));              // < 6.1 q
    const F9 M1 = f9_norm_red<P9, 3>(f9_mul<P9>(T, R32));   // < 7.3 q
    const F9 PPP32 = f9_norm(f9_shl5(PPP));
    const F9 M2 = f9_norm_red<P9, 2>(f9_mul<P9>(acc.y, PPP32));
    const F9 Y3 = f9_norm_red<P9, 3>(f9_sub<P9>(M1, M2));
    acc.zz = f9_norm_red<P9, 2>(f9_mul<P9>(acc.zz, PP32));
    acc.zzz = f9_norm_red<P9, 2>(f9_mul<P9>(acc.zzz, PPP32));
    acc.x = X3;
    acc.y = Y3;
}

__device__ __forceinline__ G1Xyzz g1_from_f9(const G1Xyzz9& a) {
    using P9 = Fq9Params;
    if (a.inf) return g1_inf();
    G1Xyzz o;
    o.x = f9_canon<P9>(a.x); o.y = f9_canon<P9>(a.y); o.zz = f9_canon<P9>(a.zz); o.zzz = f9_canon<P9>(a.zzz);
    return o;
}

}  // namespace atlas
