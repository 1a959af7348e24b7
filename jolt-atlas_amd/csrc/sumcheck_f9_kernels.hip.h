// Degree-2 dot-product sumcheck data passes on the 29-bit lazy-limb arithmetic (f9.hip.h).
//
// Same contract as k_dot_eval / k_dot_bind_eval<2, Fr, false> in sumcheck_kernels.hip.h
// (EinsumDotProver with EqSchedule::None, LargeScalars operands; dot.rs:290-375), with two
// differences that are invisible outside the device:
//   * the bound coefficients written back to HBM are residues < 2.1p, not necessarily < p
//     (CANON_OUT = true writes canonical residues — used for the pass that hands over to the
//     LDS tail kernel, which runs the exact 8x32 arithmetic);
//   * per-workgroup partial sums are canonical (one exact multiplication by Montgomery(32)
//     per workgroup also removes the 2^-5 that the 9-step Montgomery reduction carries).
// Work per index i (q = len/4): 4 sparse multiplications by the challenge (5 of 9 limbs
// non-zero after pre-scaling by 32), 2 full multiplications, no carry chains.
#pragma once
#include "f9.hip.h"
#include "sumcheck_kernels.hip.h"

namespace atlas {

// workgroup reduction of two F9 running sums -> canonical partial[blockIdx.x*2 + k]
__device__ __forceinline__ void f9_block_reduce_store2(F9 acc0, F9 acc2, Fr* partials) {
    using P9 = Fr9Params;
    __shared__ F9 red9[SC_THREADS / 64][2];
    acc0 = f9_wave_sum_lazy(acc0);                            // plain integer sums (f9.hip.h), one reduction per sum below
    acc2 = f9_wave_sum_lazy(acc2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red9[wave][0] = acc0; red9[wave][1] = acc2; }
    __syncthreads();
    if (threadIdx.x < 2) {
        F9 s = red9[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < SC_THREADS / 64; w++) s = f9_add(s, red9[w][threadIdx.x]);
        fe_store(partials + (size_t)blockIdx.x * 2 + threadIdx.x, f9_canon_x32<P9>(f9_reduce_lazy<P9>(f9_norm(s))));
    }
}

// the same sums mailed to the host as lazy 9 x 29-bit limbs (still carrying the 2^-5 of the 9-step reduction): the host sums the
// records limb-wise and reduces once (channel.hpp: sum_to_fr takes any 32-bit limbs), so nothing is reduced here — the wavefront and
// workgroup sums are plain integer sums with carry passes (f9_wave_sum_lazy; inputs < 2.1 p: the workgroup's sum is < 540 p, its top
// limb < 2^31)
__device__ __forceinline__ void f9_block_reduce_mail2(F9 acc0, F9 acc2, const RoundIo& io) {
    __shared__ F9 red9m[SC_THREADS / 64][2];
    acc0 = f9_wave_sum_lazy(acc0);
    acc2 = f9_wave_sum_lazy(acc2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red9m[wave][0] = acc0; red9m[wave][1] = acc2; }
    __shared__ uint32_t stage9m[18];
    __syncthreads();
    if (threadIdx.x < 64) {
        F9 s = f9_zero();
        if (threadIdx.x < 2) {
            s = red9m[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < SC_THREADS / 64; w++) s = f9_add(s, red9m[w][threadIdx.x]);     // four normalized values: limbs < 2^31
            s = f9_norm(s);
        }
        ch_mail_wave_f9(io, blockIdx.x * ch_stride(2), 2, s, stage9m);
    }
}

struct DevIoF9 {
    const ScCtx* cx;
    Fr* partials;
    __device__ __forceinline__ bool challenge(Fr& r) const { r = fe_load(&cx->r); return true; }
    __device__ __forceinline__ void emit2(const F9& a0, const F9& a2) const { f9_block_reduce_store2(a0, a2, partials); }
    __device__ __forceinline__ void stamp(uint32_t) const {}
};
struct ChanIoF9 {
    RoundIo io;
    // diagnosis (ATLAS_DEV_STAMPS=1): workgroup 0's stamp of an event of this launch (its entry and the challenge's arrival are stamped by ch_wait_r)
    __device__ __forceinline__ void stamp(uint32_t ev) const { if (blockIdx.x == 0 && threadIdx.x == 0) ch_stamp(io.abort_flag, ev, io.tag_mail); }
    __device__ __forceinline__ bool challenge(Fr& r) const {
        uint64_t lo, hi;
        if (!ch_wait_r(io, lo, hi)) return false;
        r = challenge_to_mont(lo, hi, 0);
        return true;
    }
    __device__ __forceinline__ void emit2(const F9& a0, const F9& a2) const { f9_block_reduce_mail2(a0, a2, io); }
};

// round-0 message over untouched (canonical) operands
template <class IO>
__global__ __launch_bounds__(SC_THREADS) void k_dot_eval2_f9(const Fr* __restrict__ L, const Fr* __restrict__ R,
                                                             size_t half, IO out) {
    using P9 = Fr9Params;
    F9 acc0 = f9_zero(), acc2 = f9_zero();
    size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * SC_THREADS;
    // two tiles in flight, as in the bind pass below: the loads of iterations i + stride and i + 2 stride are requested before the two
    // products of iteration i are issued (one wavefront per SIMD: nothing else hides the HBM latency; it was 0.49 of peak without)
    Fe x0, x1, y0, y1, u0, u1, v0, v1;
    if (i < half) { x0 = fe_load(L + i); x1 = fe_load(L + i + half); y0 = fe_load(R + i); y1 = fe_load(R + i + half); }
    if (i + stride < half) { u0 = fe_load(L + i + stride); u1 = fe_load(L + i + stride + half); v0 = fe_load(R + i + stride); v1 = fe_load(R + i + stride + half); }
    for (; i < half; i += stride) {
        const F9 l0 = f9_from_fe(x0), l1 = f9_from_fe(x1), r0 = f9_from_fe(y0), r1 = f9_from_fe(y1);
        x0 = u0; x1 = u1; y0 = v0; y1 = v1;
        const size_t nx = i + 2 * stride;
        if (nx < half) { u0 = fe_load(L + nx); u1 = fe_load(L + nx + half); v0 = fe_load(R + nx); v1 = fe_load(R + nx + half); }
        const F9 l2 = f9_norm(f9_add(l1, f9_sub<P9>(l1, l0))), r2 = f9_add(r1, f9_sub<P9>(r1, r0));      // r2: lazy limbs < 2^31.4
        acc0 = f9_mul_addred<P9>(l0, r0, acc0);
        acc2 = f9_mul_addred<P9>(l2, r2, acc2);
    }
    out.emit2(f9_norm_red<P9, 1>(acc0), f9_norm_red<P9, 1>(acc2));
}

// fused ingest_challenge(r_j) + compute_message(j+1); operands bound in place (thread i owns
// i, i+q, i+2q, i+3q).  The next iteration's eight coefficients are requested before the
// current ones are consumed, so HBM latency sits under ~700 multiply-adds.
template <bool CANON_OUT, class IO>
__global__ __launch_bounds__(SC_THREADS) void k_dot_bind_eval2_f9(Fr* L, Fr* R, size_t q, IO io) {
    using P9 = Fr9Params;
    F9 acc0 = f9_zero(), acc2 = f9_zero();
    size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * SC_THREADS;
    Fe x0, x1, x2, x3, y0, y1, y2, y3;
    if (i < q) {          // requested before the challenge is waited for: the first tile's HBM latency hides under the wait
        x0 = fe_load(L + i); x1 = fe_load(L + i + q); x2 = fe_load(L + i + 2 * q); x3 = fe_load(L + i + 3 * q);
        y0 = fe_load(R + i); y1 = fe_load(R + i + q); y2 = fe_load(R + i + 2 * q); y3 = fe_load(R + i + 3 * q);
    }
    Fr r_fe;
    if (!io.challenge(r_fe)) return;
    const F9 r32 = f9_shl5(f9_from_fe(r_fe));     // 32 * r: limbs 0..3 stay zero
    for (; i < q; i += stride) {
        const F9 a0 = f9_from_fe(x0), a1 = f9_from_fe(x1), a2 = f9_from_fe(x2), a3 = f9_from_fe(x3);
        const F9 b0 = f9_from_fe(y0), b1 = f9_from_fe(y1), b2 = f9_from_fe(y2), b3 = f9_from_fe(y3);
        const size_t nx = i + stride;
        if (nx < q) {
            x0 = fe_load(L + nx); x1 = fe_load(L + nx + q); x2 = fe_load(L + nx + 2 * q); x3 = fe_load(L + nx + 3 * q);
            y0 = fe_load(R + nx); y1 = fe_load(R + nx + q); y2 = fe_load(R + nx + 2 * q); y3 = fe_load(R + nx + 3 * q);
        }
        // a + r*(b - a): the difference carries +4p and limbs < 2^31, the product is < 1.03p; with the conditional 2p of
        // f9_mul_addred the bound values stay < 3.04p (< 2^256: they are stored as 8 x 32)
        F9 l0 = f9_mul_addred<P9, 4>(f9_sub<P9>(a2, a0), r32, a0);
        F9 l1 = f9_mul_addred<P9, 4>(f9_sub<P9>(a3, a1), r32, a1);
        F9 r0 = f9_mul_addred<P9, 4>(f9_sub<P9>(b2, b0), r32, b0);
        F9 r1 = f9_mul_addred<P9, 4>(f9_sub<P9>(b3, b1), r32, b1);
        if constexpr (CANON_OUT) {
            fe_store(L + i, f9_canon<P9>(l0)); fe_store(L + i + q, f9_canon<P9>(l1));
            fe_store(R + i, f9_canon<P9>(r0)); fe_store(R + i + q, f9_canon<P9>(r1));
        } else {
            fe_store(L + i, f9_to_fe(l0)); fe_store(L + i + q, f9_to_fe(l1));
            fe_store(R + i, f9_to_fe(r0)); fe_store(R + i + q, f9_to_fe(r1));
        }
        const F9 l2 = f9_norm(f9_add(l1, f9_sub<P9>(l1, l0))), r2 = f9_add(r1, f9_sub<P9>(r1, r0));      // r2: lazy limbs < 2^31.4
        acc0 = f9_mul_addred<P9>(l0, r0, acc0);
        acc2 = f9_mul_addred<P9>(l2, r2, acc2);
    }
    io.stamp(CH_EV_WORK_DONE);
    io.emit2(f9_norm_red<P9, 1>(acc0), f9_norm_red<P9, 1>(acc2));
    io.stamp(CH_EV_MAILED);
}

// The same pass for AT MOST 2^15 quads (the rounds of 2^17 .. 2^13 coefficients, where a launch is one quad per thread and the round is the
// latency of that thread's chain: 4 sparse + 2 full multiplications, ~3.5 us): TWO lanes per quad, as the tail kernel below does in LDS — lane
// 2i binds the two L pairs of quad i, lane 2i+1 the two R pairs (two sparse multiplications each), they swap the bound values by DPP, lane 2i
// multiplies l0 * r0 and lane 2i+1 l2 * r2: a chain of 2 sparse + 1 full multiplication.  Same sums (exact arithmetic in another order), same
// mail format: value 0 = the l0 r0 sum, value 1 = the l2 r2 sum of the workgroup, lazy limbs.
__device__ __forceinline__ F9 f9_dpp_swap1(const F9& a);
template <class IO>
__global__ __launch_bounds__(SC_THREADS) void k_dot_bind_eval2_f9_pair(Fr* L, Fr* R, size_t q, IO io) {
    using P9 = Fr9Params;
    const size_t i = ((size_t)blockIdx.x * SC_THREADS + threadIdx.x) >> 1;
    const uint32_t h = threadIdx.x & 1u;
    Fr* const S = h ? R : L;
    const bool live = i < q;
    Fe s0 = fe_zero(), s1 = fe_zero(), s2 = fe_zero(), s3 = fe_zero();
    if (live) { s0 = fe_load(S + i); s1 = fe_load(S + i + q); s2 = fe_load(S + i + 2 * q); s3 = fe_load(S + i + 3 * q); }     // in flight while the challenge is waited for
    Fr r_fe;
    if (!io.challenge(r_fe)) return;
    const F9 r32 = f9_shl5(f9_from_fe(r_fe));
    // every lane runs the arithmetic (the DPP moves need the whole wavefront; a lane beyond the end holds zeros and contributes multiples of p)
    const F9 a0 = f9_from_fe(s0), a1 = f9_from_fe(s1), a2 = f9_from_fe(s2), a3 = f9_from_fe(s3);
    const F9 v0 = f9_mul_addred<P9, 4>(f9_sub<P9>(a2, a0), r32, a0);
    const F9 v1 = f9_mul_addred<P9, 4>(f9_sub<P9>(a3, a1), r32, a1);
    if (live) { fe_store(S + i, f9_to_fe(v0)); fe_store(S + i + q, f9_to_fe(v1)); }
    const F9 w0 = f9_dpp_swap1(v0), w1 = f9_dpp_swap1(v1);        // the partner's side of the quad
    const F9 l2 = f9_norm(f9_add(w1, f9_sub<P9>(w1, w0))), r2 = f9_add(v1, f9_sub<P9>(v1, v0));      // (used by the odd lane: w = L side, v = R side)
    F9 x, y;
#pragma unroll
    for (int k = 0; k < 9; k++) { x.l[k] = h ? l2.l[k] : v0.l[k]; y.l[k] = h ? r2.l[k] : w0.l[k]; }
    F9 acc = f9_mul<P9>(x, y);
    // lanes of equal parity hold the same kind of product: plain integer sums (f9.hip.h), mailed as lazy limbs
    acc = f9_add_dpp<0x4e>(acc);
    acc = f9_add_dpp<0x124>(acc);
    acc = f9_add_dpp<0x128>(acc);
    acc = f9_norm(acc);
    acc = f9_add(acc, f9_shfl_xor(acc, 16));
    acc = f9_norm(f9_add(acc, f9_shfl_xor(acc, 32)));
    __shared__ F9 red9p[SC_THREADS / 64][2];
    __shared__ uint32_t stage9p[18];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 2) red9p[wave][lane] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        F9 t = f9_zero();
        if (threadIdx.x < 2) {
            t = red9p[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < SC_THREADS / 64; w++) t = f9_add(t, red9p[w][threadIdx.x]);
            t = f9_norm(t);
        }
        ch_mail_wave_f9(io.io, blockIdx.x * ch_stride(2), 2, t, stage9p);
    }
}

// ---- every bind pass of an instance in ONE resident launch ------------------------------------------------------------------------
// k_dot_bind_eval2_f9 / _pair are a launch per round: behind every round's arithmetic stood a kernel boundary (the grid drains, the next one
// is dispatched and its first workgroup starts polling: ~5 us), a launch call on the host thread that runs the transcript, and — under a
// tracer — the host's lateness.  Here the W workgroups of one launch stay resident over the rounds round0 .. round0 + n_mid - 1:
// per round every workgroup waits for the challenge (workgroup 0 polls the host slot and fans it out to the HBM replicas, ch_wait_r),
// binds its share IN PLACE and mails its two sums, exactly as the per-round kernels do — same products, same sums, same records:
//   * q = len / 4 >= W * 256 quads ("owned" rounds): thread g of the grid takes the quads i = g, g + G, ... (G = W * 256), all four
//     coefficients of a quad and both halves of what it leaves belong to the same thread in every such round (G divides q), so no round
//     needs anything another thread wrote: plain loads and stores, the next tile requested before the current one is consumed;
//   * fewer quads ("pair" rounds): two lanes per quad (lane 2i binds the L pairs, 2i + 1 the R pairs, one product each after a DPP swap),
//     quad i on workgroup 2 i / 256 — ownership moves between workgroups from round to round, so what a round stores is stored WRITE-THROUGH
//     (sc1) and drained before the workgroup mails, and read with sc1 loads (past the CU's L1): the host publishes the next challenge only
//     when it holds every workgroup's record, i.e. after every store of the round has left its CU (MI355X_MICROARCH: sc1 payload + drain).
//     The last owned round stores write-through as well (the first pair round reads it).  A workgroup without quads in a pair round
//     has none in any later round and leaves.
// The rounds below 2^tail_log coefficients stay with k_dot_tail2_f9 (one workgroup, rows in LDS).
struct DotResidentArgs {
    Fr* L; Fr* R;
    unsigned long long len;           // coefficients per operand when the launch starts (the challenge of round round0 - 1 is still to be bound)
    uint32_t round0, n_mid;           // message rounds served: round0 .. round0 + n_mid - 1 (round k binds challenge k - 1 first)
    Chunk* mail; uint32_t region_chunks;     // round round0 + t mails into mail + t * region_chunks, a record per workgroup (ch_stride(2) apart)
    const Chunk* r_host0; Chunk* r_dev0;     // slot / replicas of challenge round0 - 1; challenge j later: + j * CH_SLOT_CHUNKS / + j * dev_slot_chunks
    uint32_t dev_slot_chunks, r_replicas;
    uint32_t* abort_flag;
    uint32_t tag_mail0, tag_r0;       // tag of round round0's records / of challenge round0 - 1; later ones count up
};
typedef uint32_t rs_u32x4 __attribute__((ext_vector_type(4)));
// four coefficients past the L1 (sc1), all eight loads in flight before the one wait
__device__ __forceinline__ void rs_load4_sc1(const Fr* p0, const Fr* p1, const Fr* p2, const Fr* p3, Fe& a, Fe& b, Fe& c, Fe& d) {
    rs_u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %9, off sc1\n\tglobal_load_dwordx4 %3, %9, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %4, %10, off sc1\n\tglobal_load_dwordx4 %5, %10, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %6, %11, off sc1\n\tglobal_load_dwordx4 %7, %11, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
    a.v[0] = v0.x; a.v[1] = v0.y; a.v[2] = v0.z; a.v[3] = v0.w; a.v[4] = v1.x; a.v[5] = v1.y; a.v[6] = v1.z; a.v[7] = v1.w;
    b.v[0] = v2.x; b.v[1] = v2.y; b.v[2] = v2.z; b.v[3] = v2.w; b.v[4] = v3.x; b.v[5] = v3.y; b.v[6] = v3.z; b.v[7] = v3.w;
    c.v[0] = v4.x; c.v[1] = v4.y; c.v[2] = v4.z; c.v[3] = v4.w; c.v[4] = v5.x; c.v[5] = v5.y; c.v[6] = v5.z; c.v[7] = v5.w;
    d.v[0] = v6.x; d.v[1] = v6.y; d.v[2] = v6.z; d.v[3] = v6.w; d.v[4] = v7.x; d.v[5] = v7.y; d.v[6] = v7.z; d.v[7] = v7.w;
}
__device__ __forceinline__ void rs_store_sc1(Fr* p, const Fe& a) {
    const rs_u32x4 lo = {a.v[0], a.v[1], a.v[2], a.v[3]}, hi = {a.v[4], a.v[5], a.v[6], a.v[7]};
    // (s_nop 1 closes the string: a store of more than 8 bytes reads its data registers a cycle after it issues, and hipcc pads nothing
    // behind inline asm — without it the next instruction overwrote v[hi] and every bound value of an owned write-through round went out wrong)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\ts_nop 1" ::"v"(p), "v"(lo), "v"(hi) : "memory");
}
__device__ __forceinline__ F9 f9_dpp_swap1(const F9& a);
static __global__ __launch_bounds__(SC_THREADS) void k_dot_resident2_f9(DotResidentArgs A) {
    using P9 = Fr9Params;
    __shared__ F9 red9r[SC_THREADS / 64][2];
    __shared__ uint32_t stage9r[18];
    const size_t G = (size_t)gridDim.x * SC_THREADS, g = (size_t)blockIdx.x * SC_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t len = (size_t)A.len;
    for (uint32_t t = 0; t < A.n_mid; t++, len >>= 1) {
        const size_t q = len / 4;
        const bool owned = q >= G;
        if (!owned && (size_t)blockIdx.x * SC_THREADS >= 2 * q) return;        // no quad for this workgroup in this round, nor in any later one
        RoundIo io;
        io.mail = A.mail + (size_t)t * A.region_chunks;
        io.r_host = A.r_host0 + (size_t)t * CH_SLOT_CHUNKS; io.r_dev = A.r_dev0 + (size_t)t * A.dev_slot_chunks; io.r_replicas = A.r_replicas;
        io.abort_flag = A.abort_flag; io.tag_mail = A.tag_mail0 + t; io.tag_r = A.tag_r0 + t; io.tag_step = 1;
        F9 acc0 = f9_zero(), acc2 = f9_zero();
        if (owned) {
            // the LAST owned round stores write-through: the pair round that follows reads it from other CUs.  (Only that one: a write-through
            // store does not refresh the line this CU's L1 holds from its own earlier plain load, so an owned round that read sc1-stored data
            // with plain loads saw the values of two rounds before — the n = 20 instance of tests/test_gpu_full_size.py.)
            const bool wt = q < 2 * G;
            size_t i = g;
            Fe x0, x1, x2, x3, y0, y1, y2, y3;
            x0 = fe_load(A.L + i); x1 = fe_load(A.L + i + q); x2 = fe_load(A.L + i + 2 * q); x3 = fe_load(A.L + i + 3 * q);       // in flight under the wait
            y0 = fe_load(A.R + i); y1 = fe_load(A.R + i + q); y2 = fe_load(A.R + i + 2 * q); y3 = fe_load(A.R + i + 3 * q);
            uint64_t lo, hi;
            if (!ch_wait_r(io, lo, hi)) return;
            const F9 r32 = f9_shl5(f9_from_fe(challenge_to_mont(lo, hi, 0)));
            for (; i < q; i += G) {
                const F9 a0 = f9_from_fe(x0), a1 = f9_from_fe(x1), a2 = f9_from_fe(x2), a3 = f9_from_fe(x3);
                const F9 b0 = f9_from_fe(y0), b1 = f9_from_fe(y1), b2 = f9_from_fe(y2), b3 = f9_from_fe(y3);
                const size_t nx = i + G;
                if (nx < q) {
                    x0 = fe_load(A.L + nx); x1 = fe_load(A.L + nx + q); x2 = fe_load(A.L + nx + 2 * q); x3 = fe_load(A.L + nx + 3 * q);
                    y0 = fe_load(A.R + nx); y1 = fe_load(A.R + nx + q); y2 = fe_load(A.R + nx + 2 * q); y3 = fe_load(A.R + nx + 3 * q);
                }
                const F9 l0 = f9_mul_addred<P9, 4>(f9_sub<P9>(a2, a0), r32, a0), l1 = f9_mul_addred<P9, 4>(f9_sub<P9>(a3, a1), r32, a1);
                const F9 r0 = f9_mul_addred<P9, 4>(f9_sub<P9>(b2, b0), r32, b0), r1 = f9_mul_addred<P9, 4>(f9_sub<P9>(b3, b1), r32, b1);
                if (wt) {
                    rs_store_sc1(A.L + i, f9_to_fe(l0)); rs_store_sc1(A.L + i + q, f9_to_fe(l1));
                    rs_store_sc1(A.R + i, f9_to_fe(r0)); rs_store_sc1(A.R + i + q, f9_to_fe(r1));
                } else {
                    fe_store(A.L + i, f9_to_fe(l0)); fe_store(A.L + i + q, f9_to_fe(l1));
                    fe_store(A.R + i, f9_to_fe(r0)); fe_store(A.R + i + q, f9_to_fe(r1));
                }
                const F9 l2 = f9_norm(f9_add(l1, f9_sub<P9>(l1, l0))), r2 = f9_add(r1, f9_sub<P9>(r1, r0));
                acc0 = f9_mul_addred<P9>(l0, r0, acc0);
                acc2 = f9_mul_addred<P9>(l2, r2, acc2);
            }
            if (wt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wavefront's stores have left the CU before its workgroup mails
            acc0 = f9_wave_sum_lazy(f9_norm_red<P9, 1>(acc0));
            acc2 = f9_wave_sum_lazy(f9_norm_red<P9, 1>(acc2));
            if (lane == 0) { red9r[wave][0] = acc0; red9r[wave][1] = acc2; }
        } else {
            const size_t i = g >> 1;
            const uint32_t h = threadIdx.x & 1u;
            Fr* const S = h ? A.R : A.L;
            const bool live = i < q;
            uint64_t lo, hi;
            if (!ch_wait_r(io, lo, hi)) return;                                   // (the loads wait for the challenge: what they read was stored in the round before)
            const F9 r32 = f9_shl5(f9_from_fe(challenge_to_mont(lo, hi, 0)));
            Fe s0 = fe_zero(), s1 = fe_zero(), s2 = fe_zero(), s3 = fe_zero();
            const size_t ii = live ? i : 0;
            rs_load4_sc1(S + ii, S + ii + q, S + ii + 2 * q, S + ii + 3 * q, s0, s1, s2, s3);
            if (!live) { s0 = fe_zero(); s1 = fe_zero(); s2 = fe_zero(); s3 = fe_zero(); }
            const F9 a0 = f9_from_fe(s0), a1 = f9_from_fe(s1), a2 = f9_from_fe(s2), a3 = f9_from_fe(s3);
            const F9 v0 = f9_mul_addred<P9, 4>(f9_sub<P9>(a2, a0), r32, a0);
            const F9 v1 = f9_mul_addred<P9, 4>(f9_sub<P9>(a3, a1), r32, a1);
            if (live) { rs_store_sc1(S + i, f9_to_fe(v0)); rs_store_sc1(S + i + q, f9_to_fe(v1)); }
            const F9 w0 = f9_dpp_swap1(v0), w1 = f9_dpp_swap1(v1);
            const F9 l2 = f9_norm(f9_add(w1, f9_sub<P9>(w1, w0))), r2 = f9_add(v1, f9_sub<P9>(v1, v0));
            F9 x, y;
#pragma unroll
            for (int k = 0; k < 9; k++) { x.l[k] = h ? l2.l[k] : v0.l[k]; y.l[k] = h ? r2.l[k] : w0.l[k]; }
            F9 acc = f9_mul<P9>(x, y);
            acc = f9_add_dpp<0x4e>(acc);
            acc = f9_add_dpp<0x124>(acc);
            acc = f9_add_dpp<0x128>(acc);
            acc = f9_norm(acc);
            acc = f9_add(acc, f9_shfl_xor(acc, 16));
            acc = f9_norm(f9_add(acc, f9_shfl_xor(acc, 32)));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the stores above have left the CU
            if (lane < 2) red9r[wave][lane] = acc;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            F9 sres = f9_zero();
            if (threadIdx.x < 2) {
                sres = red9r[0][threadIdx.x];
#pragma unroll
                for (int w = 1; w < SC_THREADS / 64; w++) sres = f9_add(sres, red9r[w][threadIdx.x]);
                sres = f9_norm(sres);
            }
            ch_mail_wave_f9(io, blockIdx.x * ch_stride(2), 2, sres, stage9r);
        }
        __syncthreads();                                                          // red9r / ch_wait_r's slot are reused by the next round
    }
}

// ---- degree-2 tail over the round channel on the lazy limbs ---------------------------------------
// Same contract as k_dot_tail_ch<2> (EqSchedule::None, challenge mode 0): every remaining round of an
// instance of <= 2^11 coefficients in one resident launch, transcript on the host.  Laid out for the
// latency of ONE round, which is all that is left at this size: the work of coefficient index i is
// split over a lane pair — lane 2i binds the two L pairs, lane 2i+1 the two R pairs (two sparse
// multiplications each), they swap results through DPP, then lane 2i multiplies l0*r0 and lane 2i+1
// l2*r2 — so a round is 2 sparse + 1 full multiplication deep instead of 4 + 2.  Sums stay lazy
// (9 x 29 bits, value < 2.1p, scaled by 2^-5) all the way to the host, which reduces them once.
__device__ __forceinline__ F9 f9_dpp_swap1(const F9& a) {       // value of lane ^ 1
    F9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[i], 0xB1, 0xf, 0xf, true);
    return o;
}

static __global__ __launch_bounds__(SC_TAIL_THREADS) void k_dot_tail2_f9(TailChArgs A, ScConsts K) {
    using P9 = Fr9Params;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Fr* sL = reinterpret_cast<Fr*>(smem_raw);
    Fr* sR = sL + (1u << A.cap_log);
    __shared__ F9 red9t[SC_TAIL_THREADS / 64][2];
    __shared__ uint64_t s_ch[3];
    __shared__ uint32_t stage9t[27];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = tid & 1;

    uint32_t len = A.len;
    for (uint32_t i = tid; i < len; i += SC_TAIL_THREADS) {
        if (A.src_i32) {
            sL[i] = fr_from_i32(reinterpret_cast<const int32_t*>(A.L)[i], K.k32);
            sR[i] = fr_from_i32(reinterpret_cast<const int32_t*>(A.R)[i], K.k32);
        } else {
            sL[i] = fe_load(reinterpret_cast<const Fr*>(A.L) + i);
            sR[i] = fe_load(reinterpret_cast<const Fr*>(A.R) + i);
        }
    }
    __syncthreads();

    Fr* const S = h ? sR : sL;
    int pending = A.pending_bind;
    uint32_t round = A.round0;
    for (;;) {
        F9 acc = f9_zero();
        uint32_t n_lanes = 0;                  // lanes that hold a product this round
        if (pending) {
            const uint32_t prev = round - 1;
            if (tid == 0) {
                uint64_t l = 0, hh = 0;
                const bool ok = ch_poll_slot(A.r_host + (size_t)prev * A.r_slot_chunks, A.tag_r0 + prev, A.abort_flag, l, hh);
                s_ch[0] = l; s_ch[1] = hh; s_ch[2] = ok ? 1 : 0;
            }
            __syncthreads();
            if (!s_ch[2]) return;
            const F9 r32 = f9_shl5(f9_from_fe(challenge_to_mont(s_ch[0], s_ch[1], 0)));
            if (len == 2) {                    // last bind: no message follows
                if (tid < 2) {
                    const F9 a = f9_from_fe(S[0]), b = f9_from_fe(S[1]);
                    S[0] = f9_to_fe(f9_norm_red<P9>(f9_add(a, f9_mul<P9, 4>(f9_sub<P9>(b, a), r32))));
                }
                len = 1;
                __syncthreads();
            } else {
                // ingest_challenge(r_{round-1}) fused with compute_message(round) (dot.rs:290-375)
                const uint32_t q = len / 4;
                n_lanes = 2 * q < SC_TAIL_THREADS ? 2 * q : SC_TAIL_THREADS;
                for (uint32_t idx = tid; idx < 2 * q; idx += SC_TAIL_THREADS) {
                    const uint32_t i = idx >> 1;
                    const F9 s0 = f9_from_fe(S[i]), s1 = f9_from_fe(S[i + q]), s2 = f9_from_fe(S[i + 2 * q]), s3 = f9_from_fe(S[i + 3 * q]);
                    const F9 v0 = f9_norm_red<P9>(f9_add(s0, f9_mul<P9, 4>(f9_sub<P9>(s2, s0), r32)));
                    const F9 v1 = f9_norm_red<P9>(f9_add(s1, f9_mul<P9, 4>(f9_sub<P9>(s3, s1), r32)));
                    S[i] = f9_to_fe(v0); S[i + q] = f9_to_fe(v1);
                    const F9 w0 = f9_dpp_swap1(v0), w1 = f9_dpp_swap1(v1);       // the partner's side
                    F9 x, y;
                    if (h == 0) { x = v0; y = w0; }                                  // l0 * r0
                    else { x = f9_norm(f9_add(w1, f9_sub<P9>(w1, w0))); y = f9_norm(f9_add(v1, f9_sub<P9>(v1, v0))); }   // l2 * r2
                    acc = f9_norm_red<P9>(f9_add(acc, f9_mul<P9>(x, y)));
                }
                len /= 2;
            }
        } else if (round < A.n_rounds) {
            // compute_message(round) over untouched operands (a fresh instance that fits the tail)
            const uint32_t half = len / 2;
            n_lanes = 2 * half < SC_TAIL_THREADS ? 2 * half : SC_TAIL_THREADS;
            for (uint32_t idx = tid; idx < 2 * half; idx += SC_TAIL_THREADS) {
                const uint32_t i = idx >> 1;
                const F9 l0 = f9_from_fe(sL[i]), r0 = f9_from_fe(sR[i]);
                F9 x = l0, y = r0;
                if (h) {
                    const F9 l1 = f9_from_fe(sL[i + half]), r1 = f9_from_fe(sR[i + half]);
                    x = f9_norm(f9_add(l1, f9_sub<P9>(l1, l0))); y = f9_norm(f9_add(r1, f9_sub<P9>(r1, r0)));
                }
                acc = f9_norm_red<P9>(f9_add(acc, f9_mul<P9>(x, y)));
            }
        }
        if (round == A.n_rounds) break;

        // lanes of equal parity hold the same kind of product: their sum over the wavefront, as a plain integer sum with carry passes (the host
        // reduces the mailed limbs once: no reduction here; DPP inside the 16-lane row).  Whole wavefronts take the branch: DPP needs every lane.
        const uint32_t n_waves = (n_lanes + 63) / 64;
        if (wave < n_waves) {
            acc = f9_add_dpp<0x4e>(acc);                      // + lane ^ 2
            acc = f9_add_dpp<0x124>(acc);                     // + 4 lanes round the row
            acc = f9_add_dpp<0x128>(acc);                     // + 8 lanes round the row: the 8 lanes of this parity in the row
            acc = f9_norm(acc);
            acc = f9_add(acc, f9_shfl_xor(acc, 16));
            acc = f9_norm(f9_add(acc, f9_shfl_xor(acc, 32)));
        }
        const RoundIo io{A.mail, nullptr, nullptr, 1u, A.abort_flag, A.tag_mail0 + round, 0u};
        if (n_waves == 1) {
            if (wave == 0) ch_mail_wave_f9(io, (round - A.round0) * ch_stride(2), 2, acc, stage9t);
        } else {
            if (wave < n_waves && lane < 2) red9t[wave][h] = acc;
            __syncthreads();
            if (wave == 0) {                                  // lane 2 w + h holds wavefront w's sum (< 64 p, normalized): at most 16 of them per parity
                F9 t = lane < 2 * n_waves ? red9t[lane >> 1][h] : f9_zero();
                t = f9_add_dpp<0x4e>(t);
                t = f9_add_dpp<0x124>(t);
                t = f9_add_dpp<0x128>(t);
                t = f9_norm(t);
                t = f9_norm(f9_add(t, f9_shfl_xor(t, 16)));     // < 1100 p: the top limb fits its word
                ch_mail_wave_f9(io, (round - A.round0) * ch_stride(2), 2, t, stage9t);
            }
        }
        pending = 1;
        round += 1;
    }
    // final_claim()s cached by cache_openings (dot.rs:377-400): canonical residues
    if (tid < 64) {
        const RoundIo io{A.mail, nullptr, nullptr, 1u, A.abort_flag, A.tag_mail0 + A.n_rounds, 0u};
        const Fr f = tid < 2 ? f9_canon<P9>(f9_from_fe(S[0])) : fr_one();
        ch_mail_wave_fe(io, (A.n_rounds - A.round0) * ch_stride(2), 3, f, stage9t);
    }
}

}  // namespace atlas
