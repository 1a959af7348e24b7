// HIP kernels for the sumcheck hot loop on gfx950.
//
// Device counterpart of (paths under /root/reference):
//   EinsumDotProver::{compute_message,ingest_challenge}  jolt-atlas-core/src/onnx_proof/ops/einsum/dot.rs:290-375
//   MultilinearPolynomial::sumcheck_evals[_array]         joltworks/src/poly/multilinear_polynomial.rs:418-454,873-905
//   DensePolynomial::bind_parallel (HighToLow/LowToHigh)  joltworks/src/poly/dense_mlpoly.rs:84-141,209-239
//   CompactPolynomial::bind_parallel (first bind)         joltworks/src/poly/compact_polynomial.rs:272-353
//   Sumcheck::prove round body                            joltworks/src/subprotocols/sumcheck.rs:578-590
//   UniPoly::{from_evals_and_hint,compress,evaluate}      joltworks/src/poly/unipoly.rs:55-98,219-245,307-318
//
// Data layout: a polynomial is a dense array of 32-byte Montgomery Fr in HBM (the arkworks
// image), bound IN PLACE high-to-low: thread i of a pass owns indices {i, i+q, i+2q, i+3q}
// (q = len/4), so one pass binds variable j with r_j (writes i, i+q) and at the same time
// accumulates the round-(j+1) message from the two values it just produced — each
// coefficient is read once and written once per round.  Per-round sums go
// thread -> wavefront (__shfl_xor) -> LDS -> one partial per workgroup; the `fs` workgroup
// folds the partials and runs the transcript, leaving r_{j+1} in HBM for the next pass.
#pragma once
#include "channel.hip.h"
#include "f9.hip.h"
#include "transcript.hip.h"

namespace atlas {

constexpr int SC_THREADS = 256;      // 4 wavefronts per workgroup
constexpr int SC_MAX_BLOCKS = 2048;  // 8 workgroups per CU, grid-stride beyond
constexpr int SC_TAIL_LOG = 10;      // instances of <= 2^10 coefficients finish in one launch

enum { EQ_NONE = 0, EQ_IDX = 1, EQ_PAIR = 2 };

struct EqView {
    const Fr* p;
    int mode;        // EQ_NONE | EQ_IDX (p[i & mask], same at every point) | EQ_PAIR (p[i>>shift], p[(i>>shift)+half])
    uint32_t shift;
    uint32_t mask;
    uint32_t half;
};

struct ScConsts {          // Montgomery constants + packed transcript labels
    Fr two_inv, six_inv;
    Fr k32;                // 2^32  (Montgomery)  : fe_mul_small<1>(k32, x)  == Fr(x)
    Fr k64;                // 2^64  (Montgomery)  : fe_mul_small<2>(k64, x)  == Fr(x)
    uint64_t lbl_begin[4]; // "UniPoly_begin"
    uint64_t lbl_end[4];   // "UniPoly_end"
};

struct ScCtx {             // per-instance control block in HBM
    DevTranscript tr;
    Fr claim;              // previous_claim
    Fr r;                  // challenge of the round just closed, Montgomery limbs
    Fr r_s64;              // r * 2^64 (Montgomery) for small-scalar binds
};

// A * d / 2^(32*NL) mod p for an NL-limb plain integer d (CIOS over d's limbs only).
// With A = (x * 2^(32*NL)) in Montgomery form this yields Montgomery(x*d): the device form
// of JoltField::mul_u64 (joltworks/src/field/ark.rs:241-250) without a Barrett step.
template <class P, int NL>
__device__ __forceinline__ Fe fe_mul_small(const Fe& a, const uint32_t* d) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint64_t c = 0;
        const uint32_t bi = d[i];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t s = (uint64_t)a.v[j] * bi + t[j] + c;
            t[j] = (uint32_t)s;
            c = s >> 32;
        }
        uint32_t t8 = t[8] + (uint32_t)c;
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

// SmallScalar::to_field for i32 (small_scalar.rs:180-182 -> from_i32, ark.rs:113-125)
__device__ __forceinline__ Fr fr_from_i32(int32_t x, const Fr& k32) {
    uint32_t mag = x < 0 ? 0u - (uint32_t)x : (uint32_t)x;
    Fr m = fe_mul_small<FrParams, 1>(k32, &mag);
    return x < 0 ? fr_neg(m) : m;
}

// ---- operand loaders: Fr (LargeScalars) or i32 (I32Scalars) ---------------------------
template <class T> struct Src;
template <> struct Src<Fr> {
    static __device__ __forceinline__ Fr get(const Fr* p, size_t i, const ScConsts&) { return fe_load(p + i); }
};
template <> struct Src<int32_t> {
    static __device__ __forceinline__ Fr get(const int32_t* p, size_t i, const ScConsts& K) {
        return fr_from_i32(p[i], K.k32);
    }
};

// a + r*(b-a): DensePolynomial bind for Fr operands
__device__ __forceinline__ Fr bind_pair(const Fr& a, const Fr& b, const Fr& r, bool r_hi_only) {
    Fr m = fr_sub(b, a);
    Fr rm = r_hi_only ? fr_mul_hi(m, r) : fr_mul(m, r);
    return fr_add(a, rm);
}

// CompactPolynomial first bind for i32 operands: to_field(a) +/- r*|b-a|
__device__ __forceinline__ Fr bind_pair_i32(int32_t a, int32_t b, const ScCtx* cx, const ScConsts& K,
                                            const Fr& r_s64) {
    Fr fa = fr_from_i32(a, K.k32);
    if (a == b) return fa;
    int64_t d = (int64_t)b - (int64_t)a;
    uint64_t mag = d < 0 ? (uint64_t)(-d) : (uint64_t)d;
    uint32_t dl[2] = {(uint32_t)mag, (uint32_t)(mag >> 32)};
    Fr m = fe_mul_small<FrParams, 2>(r_s64, dl);
    (void)cx;
    return d < 0 ? fr_sub(fa, m) : fr_add(fa, m);
}

// evaluations at 0, 2, 3 of the line through (0 -> a, 1 -> b)
template <int DEG>
__device__ __forceinline__ void line_evals(const Fr& a, const Fr& b, Fr e[DEG]) {
    e[0] = a;
    Fr m = fr_sub(b, a);
    Fr v = b;
#pragma unroll
    for (int k = 1; k < DEG; k++) { v = fr_add(v, m); e[k] = v; }
}

template <int DEG>
__device__ __forceinline__ void accumulate(Fr acc[DEG], const Fr& l0, const Fr& l1, const Fr& r0,
                                           const Fr& r1, const EqView& eq, size_t i) {
    Fr l[DEG], r[DEG];
    line_evals<DEG>(l0, l1, l);
    line_evals<DEG>(r0, r1, r);
    if (eq.mode == EQ_NONE) {
#pragma unroll
        for (int k = 0; k < DEG; k++) acc[k] = fr_add(acc[k], fr_mul(l[k], r[k]));
    } else if (eq.mode == EQ_IDX) {
        Fr e = fe_load(eq.p + (i & eq.mask));
#pragma unroll
        for (int k = 0; k < DEG; k++) acc[k] = fr_add(acc[k], fr_mul(fr_mul(l[k], r[k]), e));
    } else {
        Fr q[DEG];
        size_t x = i >> eq.shift;
        line_evals<DEG>(fe_load(eq.p + x), fe_load(eq.p + x + eq.half), q);
#pragma unroll
        for (int k = 0; k < DEG; k++) acc[k] = fr_add(acc[k], fr_mul(fr_mul(l[k], r[k]), q[k]));
    }
}

// workgroup reduction of DEG running sums; partial[blockIdx.x*DEG + k]
template <int DEG>
__device__ __forceinline__ void block_reduce_store(Fr acc[DEG], Fr* partials) {
    __shared__ Fr red[SC_THREADS / 64][DEG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        Fr s = fr_wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < DEG) {
        Fr s = red[0][threadIdx.x];
        for (int w = 1; w < SC_THREADS / 64; w++) s = fr_add(s, red[w][threadIdx.x]);
        fe_store(partials + (size_t)blockIdx.x * DEG + threadIdx.x, s);
    }
}

// ---- where a pass gets its challenge from and where its partial sums go ------------------
// DevIo: transcript on the device (k_fs_round / k_dot_tail): challenge in the control block, one
// canonical partial per workgroup in HBM.  ChanIo: transcript on the host over the round channel
// (channel.hip.h): challenge from the round's slot, partials mailed to pinned host memory.
struct DevIo {
    const ScCtx* cx;
    Fr* partials;
    __device__ __forceinline__ bool challenge(Fr& r) const { r = fe_load(&cx->r); return true; }
    template <int DEG> __device__ __forceinline__ void emit(Fr acc[DEG]) const { block_reduce_store<DEG>(acc, partials); }
};
struct ChanIo {
    RoundIo io;
    int challenge_mode;
    __device__ __forceinline__ bool challenge(Fr& r) const {
        uint64_t lo, hi;
        if (!ch_wait_r(io, lo, hi)) return false;
        r = challenge_to_mont(lo, hi, challenge_mode);
        return true;
    }
    template <int DEG> __device__ __forceinline__ void emit(Fr acc[DEG]) const {
        __shared__ Fr red[SC_THREADS / 64][DEG];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < DEG; k++) {
            Fr s = fr_wave_sum(acc[k]);
            if (lane == 0) red[wave][k] = s;
        }
        __shared__ uint32_t stage[9 * DEG];
        __syncthreads();
        if (threadIdx.x < 64) {
            Fr s = fe_zero();
            if (threadIdx.x < DEG) {
                s = red[0][threadIdx.x];
                for (int w = 1; w < SC_THREADS / 64; w++) s = fr_add(s, red[w][threadIdx.x]);
            }
            ch_mail_wave_fe(io, blockIdx.x * ch_stride(DEG), DEG, s, stage);
        }
    }
};

// ---- round-0 message (no bind): compute_message over the untouched operands -----------
template <int DEG, class T, class IO>
__global__ __launch_bounds__(SC_THREADS) void k_dot_eval(const T* __restrict__ L, const T* __restrict__ R,
                                                         EqView eq, size_t half, IO out,
                                                         ScConsts K) {
    Fr acc[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) acc[k] = fe_zero();
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < half;
         i += (size_t)gridDim.x * SC_THREADS) {
        Fr l0 = Src<T>::get(L, i, K), l1 = Src<T>::get(L, i + half, K);
        Fr r0 = Src<T>::get(R, i, K), r1 = Src<T>::get(R, i + half, K);
        accumulate<DEG>(acc, l0, l1, r0, r1, eq, i);
    }
    out.template emit<DEG>(acc);
}

// input claim of the instance, sum_h L(h) R(h) [EQ(h)] over the whole cube (what the
// reference reads from the opening accumulator, dot.rs:204-207; needed for synthetic inputs)
template <class T>
__global__ __launch_bounds__(SC_THREADS) void k_dot_claim(const T* __restrict__ L, const T* __restrict__ R,
                                                          EqView eq, size_t len, Fr* partials, ScConsts K) {
    Fr acc[1];
    acc[0] = fe_zero();
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < len;
         i += (size_t)gridDim.x * SC_THREADS) {
        Fr t = fr_mul(Src<T>::get(L, i, K), Src<T>::get(R, i, K));
        if (eq.mode == EQ_IDX) t = fr_mul(t, fe_load(eq.p + (i & eq.mask)));
        else if (eq.mode == EQ_PAIR) t = fr_mul(t, fe_load(eq.p + (i >> eq.shift)));
        acc[0] = fr_add(acc[0], t);
    }
    block_reduce_store<1>(acc, partials);
}

// ---- fused pass: ingest_challenge(r_j) + compute_message(round j+1) --------------------
// len = current length (>= 4); q = len/4. Fr operands are bound in place; i32 operands
// are read from Lsrc/Rsrc and the bound Fr written to Ldst/Rdst (len/2 entries).
// BIND_EQ: the EQ table has the same length/indexing as L,R (EqSchedule::Low, rounds >=
// log_k) and is bound in the same pass.
template <int DEG, class T, bool BIND_EQ, class IO>
__global__ __launch_bounds__(SC_THREADS) void k_dot_bind_eval(const T* Lsrc, const T* Rsrc, Fr* Ldst,
                                                              Fr* Rdst, Fr* eq_rw, EqView eq, size_t q,
                                                              IO io, ScConsts K, int r_hi_only) {
    const ScCtx* cx = nullptr;
    Fr acc[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) acc[k] = fe_zero();
    Fr r;
    if (!io.challenge(r)) return;
    Fr r_s64;
    if constexpr (sizeof(T) == 4) r_s64 = fr_mul(r, K.k64);   // Montgomery(r * 2^64) for small-scalar binds
    const bool hi = r_hi_only != 0;
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < q;
         i += (size_t)gridDim.x * SC_THREADS) {
        Fr l0, l1, r0, r1;
        if constexpr (sizeof(T) == 4) {
            l0 = bind_pair_i32(Lsrc[i], Lsrc[i + 2 * q], cx, K, r_s64);
            l1 = bind_pair_i32(Lsrc[i + q], Lsrc[i + 3 * q], cx, K, r_s64);
            r0 = bind_pair_i32(Rsrc[i], Rsrc[i + 2 * q], cx, K, r_s64);
            r1 = bind_pair_i32(Rsrc[i + q], Rsrc[i + 3 * q], cx, K, r_s64);
        } else {
            const Fr* Lp = reinterpret_cast<const Fr*>(Lsrc);
            const Fr* Rp = reinterpret_cast<const Fr*>(Rsrc);
            Fr a0 = fe_load(Lp + i), a1 = fe_load(Lp + i + q), a2 = fe_load(Lp + i + 2 * q),
               a3 = fe_load(Lp + i + 3 * q);
            l0 = bind_pair(a0, a2, r, hi);
            l1 = bind_pair(a1, a3, r, hi);
            Fr b0 = fe_load(Rp + i), b1 = fe_load(Rp + i + q), b2 = fe_load(Rp + i + 2 * q),
               b3 = fe_load(Rp + i + 3 * q);
            r0 = bind_pair(b0, b2, r, hi);
            r1 = bind_pair(b1, b3, r, hi);
        }
        fe_store(Ldst + i, l0); fe_store(Ldst + i + q, l1);
        fe_store(Rdst + i, r0); fe_store(Rdst + i + q, r1);
        if constexpr (BIND_EQ) {
            Fr e0 = fe_load(eq_rw + i), e1 = fe_load(eq_rw + i + q), e2 = fe_load(eq_rw + i + 2 * q),
               e3 = fe_load(eq_rw + i + 3 * q);
            Fr q0 = bind_pair(e0, e2, r, hi), q1 = bind_pair(e1, e3, r, hi);
            fe_store(eq_rw + i, q0); fe_store(eq_rw + i + q, q1);
            Fr l[DEG], rr[DEG], qq[DEG];
            line_evals<DEG>(l0, l1, l); line_evals<DEG>(r0, r1, rr); line_evals<DEG>(q0, q1, qq);
#pragma unroll
            for (int k = 0; k < DEG; k++) acc[k] = fr_add(acc[k], fr_mul(fr_mul(l[k], rr[k]), qq[k]));
        } else {
            accumulate<DEG>(acc, l0, l1, r0, r1, eq, i);
        }
    }
    io.template emit<DEG>(acc);
}

// the same bind with the challenge taken from the IO policy (EQ tables bound ahead of a fused pass)
template <class IO>
static __global__ __launch_bounds__(SC_THREADS) void k_bind_hi_io(Fr* z, size_t half, IO io, int r_hi_only) {
    Fr r;
    if (!io.challenge(r)) return;
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < half;
         i += (size_t)gridDim.x * SC_THREADS) {
        Fr a = fe_load(z + i), b = fe_load(z + i + half);
        fe_store(z + i, bind_pair(a, b, r, r_hi_only != 0));
    }
}

// plain in-place high-to-low bind of one array (EQ tables bound ahead of a fused pass,
// and the PolynomialBinding::bind_parallel entry point of the C-ABI)
static __global__ __launch_bounds__(SC_THREADS) void k_bind_hi(Fr* z, size_t half, const Fr* r_ptr, int r_hi_only) {
    const Fr r = fe_load(r_ptr);
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < half;
         i += (size_t)gridDim.x * SC_THREADS) {
        Fr a = fe_load(z + i), b = fe_load(z + i + half);
        fe_store(z + i, bind_pair(a, b, r, r_hi_only != 0));
    }
}

// low-to-high bind (pairs 2i, 2i+1) into a separate destination (dense_mlpoly.rs:219-239)
static __global__ __launch_bounds__(SC_THREADS) void k_bind_lo(const Fr* z, Fr* out, size_t half, const Fr* r_ptr,
                                                        int r_hi_only) {
    const Fr r = fe_load(r_ptr);
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < half;
         i += (size_t)gridDim.x * SC_THREADS) {
        Fr a = fe_load(z + 2 * i), b = fe_load(z + 2 * i + 1);
        fe_store(out + i, bind_pair(a, b, r, r_hi_only != 0));
    }
}

// ---- the serial spine of one round (one wavefront, VALU): sumcheck.rs:579-588 ------------
// Every lane holds the same field values; the transcript state is spread over each quad.
__device__ __forceinline__ Fr fr_half(const Fr& a) {      // a/2: (a + (a odd ? p : 0)) >> 1
    uint32_t t[8];
    const uint32_t odd = a.v[0] & 1u;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + (odd ? FrParams::p(i) : 0u);
        t[i] = (uint32_t)c;
        c >>= 32;
    }
    Fr o;
#pragma unroll
    for (int i = 0; i < 7; i++) o.v[i] = (t[i] >> 1) | (t[i + 1] << 31);
    o.v[7] = (t[7] >> 1) | ((uint32_t)c << 31);
    return o;
}

// Montgomery residue -> canonical integer on the 29-bit limbs (exact: see fs_round_wave)
__device__ __forceinline__ Fr fs_from_mont(const Fr& x) {
    using P9 = Fr9Params;
    F9 k = f9_zero();
    k.l[0] = 32u;
    return f9_to_fe(f9_mul<P9>(f9_from_fe(x), k));
}

struct FsScratch {           // LDS used by the wave-cooperative transcript
    WaveTranscriptLds wt;
};

template <int DEG>
__device__ __forceinline__ void fs_round_wave(WaveTranscript& T, FsScratch* S, const WaveBlakeSched& W,
                                              uint32_t lane, const Fr ev[DEG], Fr& claim, Fr& r_out,
                                              Fr* proof_row, uint64_t* chal_row, const ScConsts& K,
                                              int challenge_mode) {
    // UniPoly::from_evals_and_hint (unipoly.rs:91-98) -> degree2/degree3 closed forms (:66-89)
    Fr c[DEG + 1];
    const Fr e0 = ev[0];
    const Fr e1 = fr_sub(claim, e0);
    const Fr e2 = ev[1];
    c[0] = e0;
    if constexpr (DEG == 2) {
        c[2] = fr_half(fr_add(fr_sub(fr_sub(e0, e1), e1), e2));
        c[1] = fr_sub(fr_sub(e1, e0), c[2]);
    } else {
        const Fr e3 = ev[2];
        Fr u = fr_sub(e1, e2);
        u = fr_add(fr_dbl(u), u);
        c[3] = fr_mul(fr_add(fr_sub(e3, e0), u), K.six_inv);
        Fr t = fr_half(fr_add(fr_sub(fr_sub(e0, e1), e1), e2));
        c[2] = fr_sub(fr_sub(fr_sub(t, c[3]), c[3]), c[3]);
        c[1] = fr_sub(fr_sub(fr_sub(e1, e0), c[2]), c[3]);
    }
    // canonical integers for the transcript (independent of the hash chain: overlaps with it).
    // 29-bit-limb path: x * 32 * 2^-261 = x * 2^-256, a product with a one-limb multiplier (90
    // multiply-adds instead of the 8x32 reduction); the result is < p for x != 0, 0 for x = 0.
    Fr canon[DEG];
    canon[0] = fs_from_mont(c[0]);
#pragma unroll
    for (int k = 2; k <= DEG; k++) canon[k - 1] = fs_from_mont(c[k]);
    // compress (drop the linear term) + append_to_transcript (unipoly.rs:307-318,550-558)
    wt_append_label(T, &S->wt, W, lane, K.lbl_begin);
#pragma unroll
    for (int k = 0; k < DEG; k++) wt_append_canonical(T, &S->wt, W, lane, canon[k]);
    wt_append_label(T, &S->wt, W, lane, K.lbl_end);
    // challenge_scalar_optimized (blake2b.rs:233-238)
    uint64_t lo, hi;
    wt_challenge_u128(T, &S->wt, W, lane, lo, hi);
    const Fr r = challenge_to_mont(lo, hi, challenge_mode);
    // previous_claim = poly.evaluate(r_j) (unipoly.rs:229-245), Horner form: every product has
    // the sparse challenge as one factor (same value, exact arithmetic)
    if (challenge_mode == 0) {
        // 29-bit limbs: r pre-scaled by 32 keeps the products in Montgomery form; the running value
        // stays < 2.3p and is reduced exactly at the end
        using P9 = Fr9Params;
        const F9 r32 = f9_shl5(f9_from_fe(r));
        F9 acc = f9_from_fe(c[DEG]);
#pragma unroll
        for (int k = DEG - 1; k >= 0; k--) acc = f9_add(f9_mul<P9, 4>(acc, r32), f9_from_fe(c[k]));
        Fr ev_r = f9_to_fe(f9_norm(acc));
        fe_cond_sub_p<FrParams>(ev_r.v); fe_cond_sub_p<FrParams>(ev_r.v); fe_cond_sub_p<FrParams>(ev_r.v);
        claim = ev_r;
    } else {
        Fr ev_r = c[DEG];
#pragma unroll
        for (int k = DEG - 1; k >= 0; k--) ev_r = fr_add(fr_mul(ev_r, r), c[k]);
        claim = ev_r;
    }
    r_out = r;
    if (lane == 0) {
        fe_store(proof_row + 0, c[0]);
#pragma unroll
        for (int k = 2; k <= DEG; k++) fe_store(proof_row + k - 1, c[k]);
        chal_row[0] = lo; chal_row[1] = hi;
    }
}

// fold the per-workgroup partials, then run the serial spine on wavefront 0.
// first != 0: Sumcheck::prove's `transcript.append_scalar(&input_claim)` (sumcheck.rs:573-574)
template <int DEG>
static __global__ __launch_bounds__(SC_THREADS) void k_fs_round(ScCtx* cx, const Fr* partials, int n_partials,
                                                         Fr* proof_row, uint64_t* chal_row, ScConsts K,
                                                         int first, int challenge_mode) {
    __shared__ Fr red[SC_THREADS / 64][DEG];
    __shared__ FsScratch fs;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Fr acc[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) acc[k] = fe_zero();
    for (int b = threadIdx.x; b < n_partials; b += SC_THREADS) {
#pragma unroll
        for (int k = 0; k < DEG; k++) acc[k] = fr_add(acc[k], fe_load(partials + (size_t)b * DEG + k));
    }
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        Fr s = fr_wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    if (wave == 0) wt_init_lds(&fs.wt, lane);
    __syncthreads();
    if (wave != 0) return;
    const uint32_t vz = vgpr_zero();
    const WaveBlakeSched W = wave_blake_sched(lane & 3);
    Fr ev[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const Fr* rp = &red[0][0] + vz;
        Fr s = rp[k];
        for (int w = 1; w < SC_THREADS / 64; w++) s = fr_add(s, rp[w * DEG + k]);
        ev[k] = s;
    }
    WaveTranscript T = wt_load(&cx->tr, lane, vz);
    Fr claim = fe_load(&cx->claim + vz);
    if (first) wt_append_canonical(T, &fs.wt, W, lane, fe_from_mont<FrParams>(claim));
    Fr r;
    fs_round_wave<DEG>(T, &fs, W, lane, ev, claim, r, proof_row, chal_row, K, challenge_mode);
    wt_store(&cx->tr, T, lane);
    if (lane == 0) {
        fe_store(&cx->claim, claim);
        fe_store(&cx->r, r);
    }
}

// ---- tail: every remaining round of an instance with len <= 2^SC_TAIL_LOG in ONE launch --
// L, R (and the EQ table when it is bound with them) live in LDS; the transcript runs on
// lane 0 between the eval and bind halves of each round.  sched: 0 none, 1 High{a,b},
// 2 Low{a,b} (EqSchedule, dot.rs:70-95); round0 = index of the first round handled here.
struct TailArgs {
    const void* L; const void* R; Fr* eq;     // global sources (eq may be null)
    uint32_t len;                              // current length of L,R (power of two, >= 1)
    uint32_t eq_len;                           // current length of the eq table (0 if none)
    int src_i32;                               // L,R are int32 (only when no round has run yet)
    int sched; uint32_t a, b;
    uint32_t round0, n_rounds;
    int first;                                 // absorb the input claim first
    int pending_bind;                          // cx->r holds a challenge not yet applied to L,R(,eq)
    int challenge_mode;
};

template <int DEG>
static __global__ __launch_bounds__(SC_THREADS) void k_dot_tail(TailArgs A, ScCtx* cx, Fr* proof, uint64_t* chal,
                                                         Fr* finals, ScConsts K) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Fr* sL = reinterpret_cast<Fr*>(smem_raw);
    Fr* sR = sL + (1u << SC_TAIL_LOG);
    Fr* sE = sR + (1u << SC_TAIL_LOG);
    __shared__ Fr red[SC_THREADS / 64][DEG];
    __shared__ Fr s_r;
    __shared__ FsScratch fs;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool hi_only = A.challenge_mode == 0;

    uint32_t len = A.len, eq_len = A.eq_len;
    for (uint32_t i = tid; i < len; i += SC_THREADS) {
        if (A.src_i32) {
            sL[i] = fr_from_i32(reinterpret_cast<const int32_t*>(A.L)[i], K.k32);
            sR[i] = fr_from_i32(reinterpret_cast<const int32_t*>(A.R)[i], K.k32);
        } else {
            sL[i] = fe_load(reinterpret_cast<const Fr*>(A.L) + i);
            sR[i] = fe_load(reinterpret_cast<const Fr*>(A.R) + i);
        }
    }
    for (uint32_t i = tid; i < eq_len; i += SC_THREADS) sE[i] = fe_load(A.eq + i);
    __syncthreads();

    // wavefront 0 carries the transcript and the running claim (same values in every lane)
    WaveTranscript T;
    WaveBlakeSched W;
    Fr claim;
    uint32_t vz = 0;
    if (wave == 0) {
        vz = vgpr_zero();
        W = wave_blake_sched(lane & 3);
        wt_init_lds(&fs.wt, lane);
        T = wt_load(&cx->tr, lane, vz);
        claim = fe_load(&cx->claim + vz);
        if (A.first) wt_append_canonical(T, &fs.wt, W, lane, fe_from_mont<FrParams>(claim));
    }

    // a challenge produced by the last fs launch that the big-pass chain did not apply
    int pending = A.pending_bind;
    uint32_t round = A.round0;
    if (pending) {
        if (tid == 0) s_r = fe_load(&cx->r);
        __syncthreads();
    }

    for (;;) {
        if (pending) {
            // ingest_challenge(r_{round-1}) (dot.rs:352-375)
            const Fr r = s_r;
            const uint32_t prev = round - 1;
            const uint32_t half = len / 2;
            for (uint32_t i = tid; i < half; i += SC_THREADS) {
                sL[i] = bind_pair(sL[i], sL[i + half], r, hi_only);
                sR[i] = bind_pair(sR[i], sR[i + half], r, hi_only);
            }
            bool bind_eq = (A.sched == 1 && prev < A.a) || (A.sched == 2 && prev >= A.a);
            if (bind_eq) {
                const uint32_t eh = eq_len / 2;
                __syncthreads();
                for (uint32_t i = tid; i < eh; i += SC_THREADS) sE[i] = bind_pair(sE[i], sE[i + eh], r, hi_only);
                eq_len = eh;
            }
            len = half;
            __syncthreads();
        }
        if (round == A.n_rounds) break;

        // compute_message(round) (dot.rs:290-350)
        const uint32_t half = len / 2;
        EqView eq;
        eq.p = sE; eq.mode = EQ_NONE; eq.shift = 0; eq.mask = 0; eq.half = 0;
        if (A.sched == 1) {
            if (round < A.a) { eq.mode = EQ_PAIR; eq.shift = A.b; eq.half = eq_len / 2; }
            else { eq.mode = EQ_IDX; eq.mask = 0; }
        } else if (A.sched == 2) {
            if (round < A.a) { eq.mode = EQ_IDX; eq.mask = (1u << A.b) - 1; }
            else { eq.mode = EQ_PAIR; eq.shift = 0; eq.half = eq_len / 2; }
        }
        Fr acc[DEG];
#pragma unroll
        for (int k = 0; k < DEG; k++) acc[k] = fe_zero();
        for (uint32_t i = tid; i < half; i += SC_THREADS)
            accumulate<DEG>(acc, sL[i], sL[i + half], sR[i], sR[i + half], eq, i);
#pragma unroll
        for (int k = 0; k < DEG; k++) {
            Fr s = fr_wave_sum(acc[k]);
            if (lane == 0) red[wave][k] = s;
        }
        __syncthreads();
        if (wave == 0) {
            Fr ev[DEG];
            const Fr* rp = &red[0][0] + vz;
#pragma unroll
            for (int k = 0; k < DEG; k++) {
                Fr s = rp[k];
                for (int w = 1; w < SC_THREADS / 64; w++) s = fr_add(s, rp[w * DEG + k]);
                ev[k] = s;
            }
            Fr r;
            fs_round_wave<DEG>(T, &fs, W, lane, ev, claim, r, proof + (size_t)round * DEG,
                               chal + 2 * (size_t)round, K, A.challenge_mode);
            if (lane == 0) s_r = r;
        }
        __syncthreads();
        pending = 1;
        round += 1;
    }

    if (wave == 0) wt_store(&cx->tr, T, lane);
    if (tid == 0) {
        fe_store(&cx->claim, claim);
        fe_store(&cx->r, s_r);
        // final_claim()s cached by cache_openings (dot.rs:377-400)
        fe_store(finals + 0, sL[0]);
        fe_store(finals + 1, sR[0]);
        fe_store(finals + 2, A.sched ? sE[0] : fr_one());
    }
}

// ---- tail over the round channel: every remaining round of an instance with len <= 2^cap_log in ONE
// resident launch, transcript on the host.  Per round the workgroup mails DEG sums (exact arithmetic,
// canonical residues) and thread 0 polls the round's challenge slot: 2.8 us device -> host -> device
// (tools/exp_channel.hip) instead of five BLAKE2b compressions on one wavefront.  1024 threads: every
// coefficient pair of a 2^11 instance has its own lane.
constexpr int SC_TAIL_THREADS = 1024;     // k_dot_tail2_f9 (sumcheck_f9_kernels.hip.h)
constexpr int SC_TAIL_X_THREADS = 512;    // k_dot_tail_ch (exact 8 x 32 arithmetic: 256 VGPRs per lane, no spills)
constexpr int SC_TAIL_CH_LOG = 11;        // EqSchedule::None: L, R of 2^11 Fr = 128 KB of the 160 KB LDS

struct TailChArgs {
    const void* L; const void* R; Fr* eq;
    uint32_t len, eq_len;
    int src_i32;
    int sched; uint32_t a, b;
    uint32_t round0, n_rounds;
    int pending_bind;                 // the challenge of round0 - 1 is still to be applied
    int challenge_mode;
    uint32_t cap_log;                 // LDS arrays hold 2^cap_log Fr each
    Chunk* mail;                      // round k's DEG records start at record (k - round0) * DEG; finals follow the last round
    const Chunk* r_host;              // slot of round 0 (round k: r_host + k * r_slot_chunks)
    uint32_t r_slot_chunks;
    uint32_t* abort_flag;
    uint32_t tag_mail0, tag_r0;       // tags of round 0's records / challenge; round k adds k; finals use tag_mail0 + n_rounds
};

template <int DEG>
static __global__ __launch_bounds__(SC_TAIL_X_THREADS) void k_dot_tail_ch(TailChArgs A, ScConsts K) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Fr* sL = reinterpret_cast<Fr*>(smem_raw);
    Fr* sR = sL + (1u << A.cap_log);
    Fr* sE = sR + (1u << A.cap_log);
    __shared__ Fr red[SC_TAIL_X_THREADS / 64][DEG];
    __shared__ uint64_t s_ch[3];
    __shared__ uint32_t stage[27];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool hi_only = A.challenge_mode == 0;

    uint32_t len = A.len, eq_len = A.eq_len;
    for (uint32_t i = tid; i < len; i += SC_TAIL_X_THREADS) {
        if (A.src_i32) {
            sL[i] = fr_from_i32(reinterpret_cast<const int32_t*>(A.L)[i], K.k32);
            sR[i] = fr_from_i32(reinterpret_cast<const int32_t*>(A.R)[i], K.k32);
        } else {
            sL[i] = fe_load(reinterpret_cast<const Fr*>(A.L) + i);
            sR[i] = fe_load(reinterpret_cast<const Fr*>(A.R) + i);
        }
    }
    for (uint32_t i = tid; i < eq_len; i += SC_TAIL_X_THREADS) sE[i] = fe_load(A.eq + i);
    __syncthreads();

    int pending = A.pending_bind;
    uint32_t round = A.round0;
    for (;;) {
        if (pending) {
            // ingest_challenge(r_{round-1}) (dot.rs:352-375)
            const uint32_t prev = round - 1;
            if (tid == 0) {
                uint64_t l = 0, h = 0;
                const bool ok = ch_poll_slot(A.r_host + (size_t)prev * A.r_slot_chunks, A.tag_r0 + prev, A.abort_flag, l, h);
                s_ch[0] = l; s_ch[1] = h; s_ch[2] = ok ? 1 : 0;
            }
            __syncthreads();
            if (!s_ch[2]) return;
            const Fr r = challenge_to_mont(s_ch[0], s_ch[1], A.challenge_mode);
            const uint32_t half = len / 2;
            for (uint32_t i = tid; i < half; i += SC_TAIL_X_THREADS) {
                sL[i] = bind_pair(sL[i], sL[i + half], r, hi_only);
                sR[i] = bind_pair(sR[i], sR[i + half], r, hi_only);
            }
            const bool bind_eq = (A.sched == 1 && prev < A.a) || (A.sched == 2 && prev >= A.a);
            if (bind_eq) {
                const uint32_t eh = eq_len / 2;
                for (uint32_t i = tid; i < eh; i += SC_TAIL_X_THREADS) sE[i] = bind_pair(sE[i], sE[i + eh], r, hi_only);
                eq_len = eh;
            }
            len = half;
            __syncthreads();
        }
        if (round == A.n_rounds) break;

        // compute_message(round) (dot.rs:290-350)
        const uint32_t half = len / 2;
        EqView eq;
        eq.p = sE; eq.mode = EQ_NONE; eq.shift = 0; eq.mask = 0; eq.half = 0;
        if (A.sched == 1) {
            if (round < A.a) { eq.mode = EQ_PAIR; eq.shift = A.b; eq.half = eq_len / 2; }
            else { eq.mode = EQ_IDX; eq.mask = 0; }
        } else if (A.sched == 2) {
            if (round < A.a) { eq.mode = EQ_IDX; eq.mask = (1u << A.b) - 1; }
            else { eq.mode = EQ_PAIR; eq.shift = 0; eq.half = eq_len / 2; }
        }
        const uint32_t n_waves = half <= 64 ? 1 : (half + 63) / 64 > SC_TAIL_X_THREADS / 64 ? SC_TAIL_X_THREADS / 64 : (half + 63) / 64;
        if (wave < n_waves) {
            Fr acc[DEG];
#pragma unroll
            for (int k = 0; k < DEG; k++) acc[k] = fe_zero();
            for (uint32_t i = tid; i < half; i += SC_TAIL_X_THREADS)
                accumulate<DEG>(acc, sL[i], sL[i + half], sR[i], sR[i + half], eq, i);
#pragma unroll
            for (int k = 0; k < DEG; k++) {
                Fr s = fr_wave_sum(acc[k]);
                if (lane == 0) red[wave][k] = s;
            }
        }
        __syncthreads();
        if (tid < 64) {
            Fr s = fe_zero();
            if (tid < DEG) {
                s = red[0][tid];
                for (uint32_t w = 1; w < n_waves; w++) s = fr_add(s, red[w][tid]);
            }
            const RoundIo io{A.mail, nullptr, nullptr, 1u, A.abort_flag, A.tag_mail0 + round, 0u};
            ch_mail_wave_fe(io, (round - A.round0) * ch_stride(DEG), DEG, s, stage);
        }
        pending = 1;
        round += 1;
    }
    // final_claim()s cached by cache_openings (dot.rs:377-400)
    if (tid < 64) {
        const RoundIo io{A.mail, nullptr, nullptr, 1u, A.abort_flag, A.tag_mail0 + A.n_rounds, 0u};
        const Fr f = tid == 0 ? sL[0] : tid == 1 ? sR[0] : (A.sched ? sE[0] : fr_one());
        ch_mail_wave_fe(io, (A.n_rounds - A.round0) * ch_stride(DEG), 3, f, stage);
    }
}

}  // namespace atlas
