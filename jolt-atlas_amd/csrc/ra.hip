// The one-hot "ra" sumcheck instances of the lookup arguments, as
// host-stepped SumcheckInstanceProver objects (instance.hpp) over device-resident polynomials.
// Device counterparts of (paths under the jolt-atlas tree, joltworks/src/):
//   RaPolynomial                 poly/ra_poly.rs:21-110      ra(j) = F[idx_j] (0 for None)
//   RaSumcheckProver             subprotocols/ra_virtual.rs:97-145
//   compute_mles_product_sum     subprotocols/mles_product_sum.rs:15-131
//   BooleanitySumcheckProver     subprotocols/booleanity.rs:169-348
//   HammingWeightSumcheckProver  subprotocols/hamming_weight.rs:106-148
//
// Layout: the d polynomials of one instance live in one allocation, row i at i * stride, and are
// bound LowToHigh out of place between two buffers (T and T/2 coefficients per row).  The
// reference keeps RaPolynomial as (index, 16-entry table) for the first three binds to save host
// memory; the MLE it represents is the gathered vector, which is what is materialised here.
//
// Work split: everything that is O(T) runs in kernels (gather, product grid, booleanity fold,
// binds); everything that is O(d * 2^log_k_chunk) or O(degree^3) (phase-1 booleanity, hamming
// weight, Toom interpolation, Gruen's cubic) is host arithmetic between launches — those
// tables have 16 (at most 256) entries.
#include "ra_common.hip.h"

namespace {

template <int D, int K0, int KN>
__global__ __launch_bounds__(RA_THREADS) void k_ra_prod_f9(const Fr* __restrict__ ra, size_t stride, SplitEqView E,
                                                           size_t n_groups, Fr* __restrict__ partials, MailTail tail) {
    using P9 = Fr9Params;
    const size_t gidx = (size_t)blockIdx.x * RA_THREADS + threadIdx.x;
    F9 prod[KN];
#pragma unroll
    for (int k = 0; k < KN; k++) prod[k] = f9_zero();
    if (gidx < n_groups) {
        const size_t mask = ((size_t)1 << E.in_bits) - 1;
        const F9 wgt = f9_mul<P9>(f9_load(E.e_out + (gidx >> E.in_bits)), f9_load(E.e_in + (gidx & mask)));
        // iteration D multiplies the weight in (kept inside the one loop: the running products stay in VGPRs)
#pragma unroll 1
        for (int i = 0; i <= D; i++) {
            const Fr* row = ra + (size_t)(i < D ? i : 0) * stride + 2 * gidx;
            const F9 a0 = f9_load(row), a1 = f9_load(row + 1);
            const F9 dl = f9_norm_red<P9, 2>(f9_sub<P9>(a1, a0));          // a1 - a0 (+4p), < 2.1p
            // p_i(x) = a0 + x dl is the LAZY operand of f9_mul (normalized limbs, a value of several p): a carry pass per
            // column, no reduction.  (Skipping the carry pass on some columns made the compiler spill the products.)
            F9 cur = f9_axpy_small(a0, dl, K0 + 1);                             // p_i(K0 + 1)
#pragma unroll
            for (int k = 0; k < KN; k++) {
                const bool inf = (K0 + k == D - 1);                             // column D-1 = X -> inf
                const F9 val = i == D ? wgt : (inf ? dl : cur);
                if (!inf) cur = f9_norm(f9_add(cur, dl));
                prod[k] = i == 0 ? val : f9_mul<P9>(prod[k], val);
            }
        }
    }
    // sums over the workgroup as plain integer sums (f9_wave_sum_lazy: DPP in the row, carry passes), ONE reduction per column at the end
    __shared__ F9 red9[RA_THREADS / 64][KN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KN; k++) {
        const F9 sres = f9_wave_sum_lazy(prod[k]);
        if (lane == 0) red9[wave][k] = sres;
    }
    __syncthreads();
    if (threadIdx.x < KN) {
        F9 sres = red9[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < RA_THREADS / 64; w++) sres = f9_add(sres, red9[w][threadIdx.x]);        // four normalized values: limbs < 2^31
        fe_store(partials + (size_t)blockIdx.x * D + K0 + threadIdx.x, f9_canon<P9>(f9_reduce_lazy<P9>(f9_norm(sres))));
    }
    mail_tail(partials, tail);
}

// the same sums with one grid column per thread (blockIdx.y = column): for short instances the chain of D
// multiplications per column is the latency of the round, so the columns go to different threads
template <int D>
__global__ __launch_bounds__(RA_THREADS) void k_ra_prod_f9_col(const Fr* __restrict__ ra, size_t stride, SplitEqView E,
                                                               size_t n_groups, Fr* __restrict__ partials /* [gridDim.x][D] */, MailTail tail) {
    using P9 = Fr9Params;
    const size_t gidx = (size_t)blockIdx.x * RA_THREADS + threadIdx.x;
    const int k = blockIdx.y;
    F9 prod = f9_zero();
    if (gidx < n_groups) {
        const size_t mask = ((size_t)1 << E.in_bits) - 1;
        prod = f9_mul<P9>(f9_load(E.e_out + (gidx >> E.in_bits)), f9_load(E.e_in + (gidx & mask)));
        // rows in batches of RB: the 2 RB loads of a batch are in flight together, so a thread waits for memory
        // ceil(D / RB) times instead of D times (24 us -> 9 us per round at d = 16 below 2^13 pairs)
        constexpr int RB = D < 4 ? D : 4;
#pragma unroll 1
        for (int i0 = 0; i0 < D; i0 += RB) {
            Fr a0[RB], a1[RB];
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const int i = i0 + u < D ? i0 + u : D - 1;
                const Fr* row = ra + (size_t)i * stride + 2 * gidx;
                a0[u] = fe_load(row); a1[u] = fe_load(row + 1);
            }
#pragma unroll
            for (int u = 0; u < RB; u++) {
                if (i0 + u >= D) break;
                const F9 x0 = f9_from_fe(a0[u]), x1 = f9_from_fe(a1[u]);
                const F9 dl = f9_norm_red<P9, 2>(f9_sub<P9>(x1, x0));
                const F9 val = k == D - 1 ? dl : f9_axpy_small(x0, dl, (uint32_t)k + 1);   // column D-1: X -> inf; else p_i(k + 1), lazy
                prod = f9_mul<P9>(prod, val);
            }
        }
    }
    __shared__ F9 red9[RA_THREADS / 64];
    const F9 sres = f9_wave_sum_lazy(prod);
    if ((threadIdx.x & 63) == 0) red9[threadIdx.x >> 6] = sres;
    __syncthreads();
    if (threadIdx.x == 0) {
        F9 t = red9[0];
#pragma unroll
        for (int w = 1; w < RA_THREADS / 64; w++) t = f9_add(t, red9[w]);
        const Fr v = f9_canon<P9>(f9_reduce_lazy<P9>(f9_norm(t)));
        if (tail.tagged) tail_put(tail, blockIdx.x, (uint32_t)k, v);
        else fe_store(partials + (size_t)blockIdx.x * D + k, v);
    }
    mail_tail(partials, tail);
}

// d = 16 as a tree of half products (the count of mles_product_sum.rs:871-949 — eval_prod_16_assign over eval_linear_prod_8_internal / _4_internal —
// restated for lazy limbs): the product of two lines on {1, 2, inf} (3 multiplications), of two such quadratics on {1..4, inf} (5), of two
// quartics on {1..8, inf} (9) and of the two halves on {1..15, inf} (16): 2 (4 * 3 + 2 * 5 + 9) + 16 = 78 multiplications per pair instead
// of 16 * 17, with the values a level needs beyond the ones it multiplied obtained from the constant top difference n! * c of a degree-n
// polynomial with leading coefficient c (small signed combinations: f9_lincomb, no field multiplication).
//   * TWO lanes per pair: lane h = 0 / 1 owns rows 0-7 / 8-15 (so a launch has twice the lanes of one-pair-per-thread; a round of 2^15
//     pairs fills the chip once).  Lane 0 folds E_out into its first row and lane 1 folds E_in into its (two multiplications each, not one
//     per column).  The last level is split by columns: in step i both lanes extend their half by one point (9 + i), lane 0 multiplies the
//     halves at point 1 + i (the oldest value of its window and of its neighbour's), lane 1 at point 9 + i (step 7: at infinity).
//   * every column of the result has the same 17 f9_mul behind it as the chain kernels (rows 0 and 8 carry one more, every product level
//     one): a stored sum is 32^-17 times the true one, as before.
//   * the body is three rolled loops (row pairs, last level, final sum): ~45 KB of code instead of ~100 KB unrolled.
//   * sums over pairs: limb-wise adds over the 8 lanes of a 16-lane row that own the same columns (DPP, no carries: 8 * 2^29 fits), then 16
//     row slots per column in LDS, added in 64-bit columns and reduced once (f9_reduce_i64).
// Measured (profiles/r04g_split_trace.txt; MI355X): 2^15 pairs 65 us against 190 us (k_ra_prod_f9_col<16>), 2^14 50 / 104, 2^13 44 / 62, 2^19 1.07 ms;
// a whole RaVirtual proof of d = 16: T = 2^16 0.97 -> 0.76 ms, 2^18 1.85 -> 1.18, 2^20 5.89 -> 3.01 (the rounds under 2^12 pairs are unchanged:
// they are the latency of a round trip, not multiplications).  Per wavefront the kernel takes ~65 us at any occupancy: 41 f9_mul are ~20 us of
// that (tools/exp_mad.hip: 0.47 us per multiplication per wavefront), the rest is the 23 small combinations, carries and moves.
constexpr int RA_SPLIT_PAIRS = RA_THREADS / 2;
// MODE 0: the rows are in memory.  MODE 1 / 2 (lazy rounds 0 / 1, ra_common.hip.h): `lookups` = the packed chunk indices of the cycles, `ra` =
// the table the row values come from — F (stride 16) in round 0, T1 (stride 256) in round 1; the loads change, the arithmetic does not.
template <bool TIMING, int MODE = 0>
__global__ __launch_bounds__(RA_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ra_prod16_split(const Fr* __restrict__ ra, size_t stride, SplitEqView E, size_t n_groups,
                                                                Fr* __restrict__ partials /* [gridDim.x][16] */, MailTail tail, const uint64_t* __restrict__ lookups = nullptr) {
    using P9 = Fr9Params;
    __shared__ F9 red[RA_THREADS / 16][16];
    // TIMING (ATLAS_RA_SPLIT_TIMING=1, diagnosis): the first wavefront of three workgroups prints its shader-clock and 100 MHz wall-clock stamps per phase
    uint64_t ck[6], wk[6];
#define RA_STAMP(i) if constexpr (TIMING) { ck[i] = clock64(); wk[i] = wall_clock64(); }
    RA_STAMP(0)
    const uint32_t h = threadIdx.x & 1u;
    size_t gidx = (size_t)blockIdx.x * RA_SPLIT_PAIRS + (threadIdx.x >> 1);
    const bool live = gidx < n_groups;
    if (!live) gidx = 0;
    const size_t mask = ((size_t)1 << E.in_bits) - 1;
    F9 wgt = f9_load(h ? E.e_in + (gidx & mask) : E.e_out + (gidx >> E.in_bits));
    if (!live) wgt = f9_zero();                                   // a pair beyond the end contributes multiples of p
    // Live state: the rolled loop carries what it declares outside across every iteration, so the half itself (9 values) is formed after
    // it, and the first quartic stays on five points until then: 256 registers, no scratch.  (Formed inside the loop: 381-409 registers with
    // the overflow in AGPRs, one wavefront per SIMD — the same times: a wavefront of f9_mul chains keeps its SIMD's multiplier busy alone,
    // tools/exp_mad.hip.)
    F9 P[5], Pp[5], Q0[5];                                        // this row pair / quartic, the previous row pair and the first quartic, on {1..4, inf}
    uint64_t lw[MODE == 2 ? 4 : MODE == 1 ? 2 : 1] = {};
    if constexpr (MODE == 1) { lw[0] = lookups[2 * gidx]; lw[1] = lookups[2 * gidx + 1]; }
    if constexpr (MODE == 2) { lw[0] = lookups[4 * gidx]; lw[1] = lookups[4 * gidx + 1]; lw[2] = lookups[4 * gidx + 2]; lw[3] = lookups[4 * gidx + 3]; }
#pragma unroll 1
    for (int t = 0; t < 4; t++) {
        // (loading rows 2 t + 2, 2 t + 3 ahead of this iteration's products changed nothing: the kernel is bound by its own instructions)
        F9 a0, a1, b0, b1;
        if constexpr (MODE == 0) {
            const Fr* row = ra + (size_t)(8 * h + 2 * t) * stride + 2 * gidx;
            a0 = f9_load(row); a1 = f9_load(row + 1); b0 = f9_load(row + stride); b1 = f9_load(row + stride + 1);
        } else {
            const uint32_t i = 8 * h + 2 * t, sa = 4 * (15 - i), sb = sa - 4;      // chunk i = nibble 15 - i of a cycle's word (d = 16)
            const Fr* ta = ra + (size_t)i * stride; const Fr* tb = ta + stride;
            if constexpr (MODE == 1) {
                a0 = f9_load(ta + ((lw[0] >> sa) & 15u)); a1 = f9_load(ta + ((lw[1] >> sa) & 15u));
                b0 = f9_load(tb + ((lw[0] >> sb) & 15u)); b1 = f9_load(tb + ((lw[1] >> sb) & 15u));
            } else {
                a0 = f9_load(ta + (((lw[0] >> sa) & 15u) * 16u + ((lw[1] >> sa) & 15u))); a1 = f9_load(ta + (((lw[2] >> sa) & 15u) * 16u + ((lw[3] >> sa) & 15u)));
                b0 = f9_load(tb + (((lw[0] >> sb) & 15u) * 16u + ((lw[1] >> sb) & 15u))); b1 = f9_load(tb + (((lw[2] >> sb) & 15u) * 16u + ((lw[3] >> sb) & 15u)));
            }
        }
        F9 da = f9_norm_red<P9, 2>(f9_sub<P9>(a1, a0)), db = f9_norm_red<P9, 2>(f9_sub<P9>(b1, b0));
        if (t == 0) { a0 = f9_mul<P9>(a0, wgt); da = f9_mul<P9>(da, wgt); }
        const F9 A1 = f9_norm(f9_add(a0, da)), A2 = f9_norm(f9_add(A1, da));            // lazy operands: < 6.5 p
        const F9 B1 = f9_norm(f9_add(b0, db)), B2 = f9_norm(f9_add(B1, db));
        P[0] = f9_mul<P9>(A1, B1); P[1] = f9_mul<P9>(A2, B2); P[4] = f9_mul<P9>(da, db);
        {   // f(x + 2) = 2 c - f(x) + 2 f(x + 1), from the base values
            const F9* const v[3] = {&P[0], &P[1], &P[4]};
            P[2] = f9_lincomb<P9, 3>(v, {-1, 2, 2});
            P[3] = f9_lincomb<P9, 3>(v, {-2, 3, 6});
        }
        if (t & 1) {
#pragma unroll
            for (int k = 0; k < 5; k++) P[k] = f9_mul<P9>(P[k], Pp[k]);               // the quartic of rows 4 (t - 1) / 2 .. + 3 on {1..4, inf}
            if (t == 1) {
#pragma unroll
                for (int k = 0; k < 5; k++) Q0[k] = P[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) Pp[k] = P[k];
        }
    }
    RA_STAMP(1)
    F9 Qp[9];                                                     // the half = the two quartics multiplied on {1..8, inf}
    {
        // points 5..8 of either quartic: the cubic through f(1..4) carried on, plus c (x - 1)(x - 2)(x - 3)(x - 4)
        const F9* const v[5] = {&P[0], &P[1], &P[2], &P[3], &P[4]};
        const F9* const v0[5] = {&Q0[0], &Q0[1], &Q0[2], &Q0[3], &Q0[4]};
#define RA_OCT(k, ...) { const F9 e1 = f9_lincomb<P9, 5>(v, __VA_ARGS__), e0 = f9_lincomb<P9, 5>(v0, __VA_ARGS__); Qp[k] = f9_mul<P9>(e1, e0); }
        RA_OCT(4, {-1, 4, -6, 4, 24}) RA_OCT(5, {-4, 15, -20, 10, 120}) RA_OCT(6, {-10, 36, -45, 20, 360}) RA_OCT(7, {-20, 70, -84, 35, 840})
#undef RA_OCT
#define RA_OCT(k, j) Qp[k] = f9_mul<P9>(P[j], Q0[j]);
        RA_OCT(0, 0) RA_OCT(1, 1) RA_OCT(2, 2) RA_OCT(3, 3) RA_OCT(8, 4)
#undef RA_OCT
    }
    RA_STAMP(2)
    // Qp = this lane's half on {1..8, inf}.  Window w = the half at 8 consecutive points; f(x + 8) = 8! c - f(x) + 8 f(x+1) - 28 f(x+2) + ...
    const int lane = threadIdx.x & 63, slot = (threadIdx.x >> 6) * 4 + (lane >> 4);
#pragma unroll 1
    for (int i = 0; i < 8; i++) {
        F9 nw = Qp[8];
        if (i < 7) {
            const F9* const v[9] = {&Qp[0], &Qp[1], &Qp[2], &Qp[3], &Qp[4], &Qp[5], &Qp[6], &Qp[7], &Qp[8]};
            nw = f9_lincomb<P9, 9>(v, {-1, 8, -28, 56, -70, 56, -28, 8, 40320});
        }
        F9 keep, send;
#pragma unroll
        for (int l = 0; l < 9; l++) { keep.l[l] = h ? nw.l[l] : Qp[0].l[l]; send.l[l] = h ? Qp[0].l[l] : nw.l[l]; }
        F9 out = f9_mul<P9>(keep, f9_dpp<0xb1>(send));            // the neighbour lane (xor 1)
        out = f9_add(out, f9_dpp<0x4e>(out));                     // + lane xor 2, + the lanes 4, 8, 12 further round the row: the 8 lanes of this half
        out = f9_add(out, f9_dpp<0x124>(out));
        out = f9_add(out, f9_dpp<0x128>(out));
        if ((lane & 15) < 2) red[slot][8 * h + i] = out;
#pragma unroll
        for (int k = 0; k < 7; k++) Qp[k] = Qp[k + 1];
        Qp[7] = nw;
    }
    RA_STAMP(3)
    __syncthreads();
    // 16 row slots per (column, limb): thread 9 c + l adds the slots of limb l of column c (16 LDS reads each, not 144 by one thread per
    // column), thread c then reduces its nine 64-bit columns
    __shared__ uint64_t cols[16][9];
    if (threadIdx.x < 144) {
        const uint32_t c = threadIdx.x / 9, l = threadIdx.x % 9;
        uint64_t acc = 0;
#pragma unroll
        for (int r = 0; r < RA_THREADS / 16; r++) acc += red[r][c].l[l];
        cols[c][l] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        int64_t t[9];
#pragma unroll
        for (int l = 0; l < 9; l++) t[l] = (int64_t)cols[threadIdx.x][l];
        const Fr v = f9_canon<P9>(f9_reduce_i64<P9>(t));
        if (tail.tagged) tail_put(tail, blockIdx.x, threadIdx.x, v);
        else fe_store(partials + (size_t)blockIdx.x * 16 + threadIdx.x, v);
    }
    RA_STAMP(4)
    mail_tail(partials, tail);
    RA_STAMP(5)
#undef RA_STAMP
    if constexpr (TIMING) {
        if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1))
            printf("[split timing] wg %u of %u: shader clocks rows %llu half %llu last level %llu sums %llu tail %llu | wall 10 ns ticks %llu %llu %llu %llu %llu\n", blockIdx.x, gridDim.x,
                   (unsigned long long)(ck[1] - ck[0]), (unsigned long long)(ck[2] - ck[1]), (unsigned long long)(ck[3] - ck[2]), (unsigned long long)(ck[4] - ck[3]), (unsigned long long)(ck[5] - ck[4]),
                   (unsigned long long)(wk[1] - wk[0]), (unsigned long long)(wk[2] - wk[1]), (unsigned long long)(wk[3] - wk[2]), (unsigned long long)(wk[4] - wk[3]), (unsigned long long)(wk[5] - wk[4]));
    }
}

// Bind and product of a round in ONE launch, for rounds of at most RA_FUSE_MAX pairs (the cycle rounds of every lookup of T <= 2^13,
// and the later rounds of the larger ones): a workgroup takes 8 pairs, the two lanes of (pair, k) bind row k of their pair — four coefficients
// of the previous round's rows -> the two of this round's, stored for the next round — and leave (x0, x1 - x0) as lazy limbs in LDS; after
// the barrier they are column k of the pair and run the chain of k_ra_prod_f9_col over the rows in LDS, half of the rows each (see below).  Against k_ra_bind_ch +
// k_ra_prod_f9_col the round loses a kernel boundary, the trip of the bound rows through HBM and the four batches of row loads: RaVirtual is
// the long lane of a one-hot batch, so this is the batch's round time.  Same sums: the chain is weight x the d values, d + 1
// multiplications of 2^-5 each as before; the partial rows are canonical.
constexpr size_t RA_FUSE_MAX = 4096;
constexpr int RA_FUSE_PAIRS = 8;
// Layout: thread (pair p, column k, half h) = 32 lanes per pair, 8 pairs per workgroup.  The round is the latency of one thread's chain, so the
// chain is cut in two: in the bind, half 0 binds the first and half 1 the second coefficient pair of row k (one multiplication each instead of
// two; the difference x1 - x0 is formed after a DPP swap); in the product, half 1 multiplies the weight and rows D/2 .. D-1 of column k, half 0
// rows 0 .. D/2-1, and half 0 multiplies the two (after a DPP swap): 1 + D/2 + 1 multiplications deep instead of 1 + D.  The number of f9_mul
// behind a column is unchanged (half 0 starts from its first value: D/2 - 1; half 1: 1 + (D - D/2); the join: 1), so is the host's 32^(D+1).
__global__ __launch_bounds__(RA_THREADS) void k_ra_bind_prod_f9(const Fr* __restrict__ src, size_t src_stride, Fr* __restrict__ dst, size_t dst_stride,
                                                                uint32_t D, SplitEqView E, size_t n_groups, Fr* __restrict__ partials /* [gridDim.x][D] */,
                                                                ChanIo io, int r_hi_only, MailTail tail) {
    using P9 = Fr9Params;
    __shared__ F9 sh_x0[RA_FUSE_PAIRS][16], sh_dl[RA_FUSE_PAIRS][16], sh_red[RA_FUSE_PAIRS][16];
    const uint32_t p = threadIdx.x >> 5, k = (threadIdx.x >> 1) & 15u, h = threadIdx.x & 1u;
    const size_t gidx = (size_t)blockIdx.x * RA_FUSE_PAIRS + p;
    const bool live = gidx < n_groups && k < D;
    const size_t emask = ((size_t)1 << E.in_bits) - 1;
    // requested before the challenge is waited for: the two coefficients this lane binds, and (half 1) the weight's factors
    Fr a0 = fe_zero(), a1 = fe_zero(), w_out = fe_zero(), w_in = fe_zero();
    if (live) {
        const Fr* s2 = src + (size_t)k * src_stride + 4 * gidx + 2 * h;
        a0 = fe_load(s2); a1 = fe_load(s2 + 1);
        if (h) { w_out = fe_load(E.e_out + (gidx >> E.in_bits)); w_in = fe_load(E.e_in + (gidx & emask)); }
    }
    Fr r;
    if (!io.challenge(r)) return;
    // every lane runs the arithmetic (the DPP swaps need whole wavefronts); a lane beyond the end holds zeros and stores nothing
    const Fr b = bind_pair(a0, a1, r, r_hi_only != 0);
    if (live) fe_store(dst + (size_t)k * dst_stride + 2 * gidx + h, b);
    const F9 x = f9_from_fe(b), xo = f9_dpp<0xb1>(x);             // this lane's bound coefficient and its neighbour's
    if (live) {
        if (h == 0) sh_x0[p][k] = x;
        else sh_dl[p][k] = f9_norm_red<P9, 2>(f9_sub<P9>(x, xo)); // x1 - x0
    }
    __syncthreads();
    const uint32_t hd = D / 2, i0 = h ? hd : 0, i1 = h ? D : hd;  // D = 1: half 0 has no row, its product is the neutral start below
    F9 prod = f9_zero();
    if (live) {
        if (h) prod = f9_mul<P9>(f9_from_fe(w_out), f9_from_fe(w_in));
#pragma unroll 1
        for (uint32_t i = i0; i < i1; i++) {
            const F9 x0 = sh_x0[p][i], dl = sh_dl[p][i];
            const F9 val = k == D - 1 ? dl : f9_axpy_small(x0, dl, k + 1);          // column D-1: X -> inf; else p_i(k + 1), lazy
            prod = (h == 0 && i == 0) ? f9_norm(val) : f9_mul<P9>(prod, val);      // half 0 starts from its first value
        }
    }
    const F9 other = f9_dpp<0xb1>(prod);
    if (live && h == 0) prod = hd ? f9_mul<P9>(prod, other) : other;               // the join (D = 1: half 1 holds the whole product)
    if (h == 0) sh_red[p][k] = live ? prod : f9_zero();
    __syncthreads();
    if (threadIdx.x < D) {                                    // the pairs' products of column k: 64-bit limb sums, one reduction
        int64_t t[9];
#pragma unroll
        for (int l = 0; l < 9; l++) t[l] = 0;
        for (int q = 0; q < RA_FUSE_PAIRS; q++) {
#pragma unroll
            for (int l = 0; l < 9; l++) t[l] += (int64_t)sh_red[q][threadIdx.x].l[l];
        }
        const Fr v = f9_canon<P9>(f9_reduce_i64<P9>(t));
        if (tail.tagged) tail_put(tail, blockIdx.x, threadIdx.x, v);
        else fe_store(partials + (size_t)blockIdx.x * D + threadIdx.x, v);
    }
    mail_tail(partials, tail);
}

// (Measured and removed: the last rounds — rows of at most 64 coefficients — as ONE workgroup per round that binds, multiplies
// with 2 or 4 lanes sharing a pair, and mails, for RaVirtual and for the booleanity fold.  No gain for RaVirtual (0.267 against
// 0.272 ms at d = 16, T = 2^6; the Einsum node unchanged) and 20-30 % slower for booleanity (0.32 against 0.25 ms): sixteen
// wavefronts in one workgroup leave 128 registers per lane for the f9_mul chains, and the round is bounded by the two host
// link crossings, not by the second launch.)
// booleanity phase 2 (booleanity.rs:254-276): per pair index j
//   c = sum_i gamma_i h0 (h0 - 1),  e = sum_i gamma_i (h1 - h0)^2, folded with E_out * E_in
// on the 29-bit lazy limbs: h0 - 1 and h1 - h0 enter their products unreduced (the lazy operand of f9_mul), the row
// sums are limb-wise adds with a carry + reduce pass every fourth row.  Four f9_mul stand behind every term (gamma h,
// the second factor, E_out E_in, the weighting): a stored sum is 32^-4 times the true one, undone on the host.
// LAZY (the FIRST cycle round of an instance over packed 4-bit chunk indices, Booleanity::lazy): every H_i reads the same 16-entry table F, so
// the two terms of a (row, pair) take 16 / 256 values — gamma_i F[a] (F[a] - 1) and gamma_i (F[b] - F[a])^2, built by k_bool_lazy_tables with
// the same two f9_mul each — and the round is two table reads and two additions per (row, pair) instead of four multiplications.
// Hp = Ctab [d][16], gammas = Etab [d][256], lookups = the cycles' words (chunk i = nibble d - 1 - i).
template <bool LAZY = false>
__global__ __launch_bounds__(RA_THREADS) void k_bool_fold(const Fr* __restrict__ Hp, size_t stride, uint32_t d,
                                                          const Fr* __restrict__ gammas, SplitEqView E, size_t n_groups,
                                                          Fr* __restrict__ partials, MailTail tail, const uint64_t* __restrict__ lookups = nullptr) {
    using P9 = Fr9Params;
    F9 acc0 = f9_zero(), acc1 = f9_zero();
    const F9 one = f9_from_fe(fr_one());
    for (size_t j = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; j < n_groups; j += (size_t)gridDim.x * RA_THREADS) {
        F9 c = f9_zero(), e = f9_zero();
        // blockIdx.y splits the d rows (the sums are additive): short instances put one row per thread
        const uint32_t i0 = (uint32_t)(((uint64_t)d * blockIdx.y) / gridDim.y), i1 = (uint32_t)(((uint64_t)d * (blockIdx.y + 1)) / gridDim.y);
        uint64_t w0 = 0, w1 = 0;
        if constexpr (LAZY) { w0 = lookups[2 * j]; w1 = lookups[2 * j + 1]; }
#pragma unroll 1
        for (uint32_t i = i0; i < i1; i++) {
            if constexpr (LAZY) {
                const uint32_t sh = 4 * (d - 1 - i), a = (uint32_t)((w0 >> sh) & 15u), b = (uint32_t)((w1 >> sh) & 15u);
                c = f9_add(c, f9_load(Hp + i * 16 + a));
                e = f9_add(e, f9_load(gammas + i * 256 + a * 16 + b));
                if ((i - i0) % 4 == 3) { c = f9_norm_red<P9, 4>(c); e = f9_norm_red<P9, 4>(e); }
                continue;
            }
            const Fr* row = Hp + (size_t)i * stride;
            const F9 h0 = f9_load(row + 2 * j), h1 = f9_load(row + 2 * j + 1);
            const F9 gm = f9_load(gammas + i);
            const F9 b = f9_sub<P9>(h1, h0);                       // lazy: + 4p, limbs < 2^31
            const F9 m1 = f9_sub<P9>(h0, one);
            c = f9_add(c, f9_mul<P9>(f9_mul<P9>(gm, h0), m1));
            e = f9_add(e, f9_mul<P9>(f9_mul<P9>(gm, b), b));
            if ((i - i0) % 4 == 3) { c = f9_norm_red<P9, 4>(c); e = f9_norm_red<P9, 4>(e); }
        }
        c = f9_norm_red<P9, 4>(c); e = f9_norm_red<P9, 4>(e);
        const F9 wgt = f9_mul<P9>(f9_load(E.e_out + (j >> E.in_bits)), f9_load(E.e_in + (j & (((size_t)1 << E.in_bits) - 1))));
        acc0 = f9_norm_red<P9>(f9_add(acc0, f9_mul<P9>(wgt, c)));
        acc1 = f9_norm_red<P9>(f9_add(acc1, f9_mul<P9>(wgt, e)));
    }
    __shared__ F9 red9[RA_THREADS / 64][2];
    acc0 = f9_wave_sum_lazy(acc0); acc1 = f9_wave_sum_lazy(acc1);      // plain integer sums; one reduction per term below
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red9[wave][0] = acc0; red9[wave][1] = acc1; }
    __syncthreads();
    if (threadIdx.x < 2) {
        F9 s = red9[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < RA_THREADS / 64; w++) s = f9_add(s, red9[w][threadIdx.x]);
        const Fr v = f9_canon<P9>(f9_reduce_lazy<P9>(f9_norm(s)));
        if (tail.tagged) tail_put(tail, blockIdx.y * gridDim.x + blockIdx.x, threadIdx.x, v);
        else fe_store(partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + threadIdx.x, v);
    }
    mail_tail(partials, tail);
}

// the two tables of k_bool_fold<true>: the SAME products the rows path forms per (row, pair), once per (row, value) / (row, value pair)
__global__ __launch_bounds__(RA_THREADS) void k_bool_lazy_tables(const Fr* __restrict__ F /* 16 */, const Fr* __restrict__ gammas, uint32_t d,
                                                                 Fr* __restrict__ Ctab /* [d][16] */, Fr* __restrict__ Etab /* [d][256] */) {
    using P9 = Fr9Params;
    const F9 one = f9_from_fe(fr_one());
    for (uint32_t t = blockIdx.x * RA_THREADS + threadIdx.x; t < d * 256u; t += gridDim.x * RA_THREADS) {
        const uint32_t i = t >> 8, a = (t >> 4) & 15u, b = t & 15u;
        const F9 h0 = f9_load(F + a), h1 = f9_load(F + b), gm = f9_load(gammas + i);
        const F9 df = f9_sub<P9>(h1, h0);
        fe_store(Etab + t, f9_canon<P9>(f9_reduce_lazy<P9>(f9_norm(f9_mul<P9>(f9_mul<P9>(gm, df), df)))));
        if (b == 0) fe_store(Ctab + i * 16 + a, f9_canon<P9>(f9_reduce_lazy<P9>(f9_norm(f9_mul<P9>(f9_mul<P9>(gm, h0), f9_sub<P9>(h0, one))))));
    }
}
// the rows of cycle round 1 (T / 2 long): out[i][j] = bind(F[chunk i of cycle 2j], F[chunk i of cycle 2j + 1]; r); grid (x, d), waits for the challenge
__global__ __launch_bounds__(RA_THREADS) void k_bool_lazy_rows(const uint64_t* __restrict__ lookups, const Fr* __restrict__ F, uint32_t d, size_t half,
                                                               Fr* __restrict__ out, size_t out_stride, ChanIo io, int r_hi_only) {
    Fr r;
    if (!io.challenge(r)) return;
    const uint32_t i = blockIdx.y, sh = 4 * (d - 1 - i);
    for (size_t j = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; j < half; j += (size_t)gridDim.x * RA_THREADS)
        fe_store(out + (size_t)i * out_stride + j, bind_pair(fe_load(F + ((lookups[2 * j] >> sh) & 15u)), fe_load(F + ((lookups[2 * j + 1] >> sh) & 15u)), r, r_hi_only != 0));
}

// The bind of a booleanity cycle round and its fold in ONE launch, one (row, pair) per thread (the latency regime of k_bool_fold, at most
// RA_FUSE_MAX pairs): four coefficients of the previous round's row -> the two of this round's, stored for the next round, and the same
// two terms — the same four f9_mul behind each.  With RaVirtual's bind fused into its product (k_ra_bind_prod_f9) this lane would
// otherwise be the long one of a one-hot batch.
__global__ __launch_bounds__(RA_THREADS) void k_bool_bind_fold(const Fr* __restrict__ src, size_t src_stride, Fr* __restrict__ dst, size_t dst_stride,
                                                               const Fr* __restrict__ gammas, SplitEqView E, size_t n_groups, Fr* __restrict__ partials,
                                                               ChanIo io, int r_hi_only, MailTail tail) {
    // TWO lanes per (row, pair) — the round is the latency of one thread's chain: lane h binds coefficient pair h of the four (one multiplication
    // instead of two), the two swap their bound values by DPP, lane 0 carries the term gamma h0 (h0 - 1) and lane 1 the term gamma (h1 - h0)^2,
    // each with the weight E_out E_in: bind + 4 multiplications deep instead of 2 + 7.  The same four f9_mul stand behind each term.
    using P9 = Fr9Params;
    const size_t j = ((size_t)blockIdx.x * RA_THREADS + threadIdx.x) >> 1;
    const uint32_t i = blockIdx.y, h = threadIdx.x & 1u;
    const bool live = j < n_groups;
    Fr a0 = fe_zero(), a1 = fe_zero(), w_out = fe_zero(), w_in = fe_zero(), gmf = fe_zero();
    if (live) {                                                    // requested before the challenge is waited for
        const Fr* s2 = src + (size_t)i * src_stride + 4 * j + 2 * h;
        a0 = fe_load(s2); a1 = fe_load(s2 + 1);
        w_out = fe_load(E.e_out + (j >> E.in_bits)); w_in = fe_load(E.e_in + (j & (((size_t)1 << E.in_bits) - 1)));
        gmf = fe_load(gammas + i);
    }
    Fr r;
    if (!io.challenge(r)) return;
    const Fr bnd = bind_pair(a0, a1, r, r_hi_only != 0);
    if (live) fe_store(dst + (size_t)i * dst_stride + 2 * j + h, bnd);
    const F9 x = f9_from_fe(bnd), xo = f9_dpp<0xb1>(x);            // every lane: the swap needs whole wavefronts (a lane beyond the end holds zeros)
    const F9 h0 = h ? xo : x, h1 = h ? x : xo;
    const F9 one = f9_from_fe(fr_one()), gm = f9_from_fe(gmf);
    const F9 bd = f9_sub<P9>(h1, h0), m1 = f9_sub<P9>(h0, one);    // lazy: + 4p, limbs < 2^31
    F9 u, v;
#pragma unroll
    for (int l = 0; l < 9; l++) { u.l[l] = h ? bd.l[l] : h0.l[l]; v.l[l] = h ? bd.l[l] : m1.l[l]; }
    const F9 term = f9_norm_red<P9, 4>(f9_mul<P9>(f9_mul<P9>(gm, u), v));          // lane 0: gamma h0 (h0 - 1); lane 1: gamma (h1 - h0)^2
    const F9 wgt = f9_mul<P9>(f9_from_fe(w_out), f9_from_fe(w_in));
    F9 acc = f9_mul<P9>(wgt, term);                               // normalized, < 1.1 p; zero weight beyond the end
    // sums by parity over the workgroup as plain integer sums (f9.hip.h), one reduction per term
    acc = f9_add_dpp<0x4e>(acc);
    acc = f9_add_dpp<0x124>(acc);
    acc = f9_add_dpp<0x128>(acc);
    acc = f9_norm(acc);
    acc = f9_add(acc, f9_shfl_xor(acc, 16));
    acc = f9_norm(f9_add(acc, f9_shfl_xor(acc, 32)));
    __shared__ F9 red9[RA_THREADS / 64][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 2) red9[wave][lane] = acc;
    __syncthreads();
    if (threadIdx.x < 2) {
        F9 s = red9[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < RA_THREADS / 64; w++) s = f9_add(s, red9[w][threadIdx.x]);
        const Fr v = f9_canon<P9>(f9_reduce_lazy<P9>(f9_norm(s)));
        if (tail.tagged) tail_put(tail, blockIdx.y * gridDim.x + blockIdx.x, threadIdx.x, v);
        else fe_store(partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + threadIdx.x, v);
    }
    mail_tail(partials, tail);
}

// out[k] = sum_p partials[p * K + k]; one workgroup per column
// `blocks` = ceil(n_groups / RA_THREADS) on entry; on return the number of partial rows the launch wrote
template <int D>
void launch_prod(const Fr* buf, size_t stride, Fr* partials, const SplitEqView& E, size_t n_groups, unsigned& blocks, MailTail tail) {
    if constexpr (D == 16) {
        // the split product from RA_SPLIT_MIN pairs up (below that the round is the latency of one thread's chain, and sixteen short chains
        // per pair beat two long ones).  ATLAS_RA_SPLIT_MIN=<log2> moves the switch (experiments; 31 = never)
        static const size_t split_min = [] { const char* e = getenv("ATLAS_RA_SPLIT_MIN"); int v = e ? atoi(e) : -1; return (size_t)1 << (v >= 0 && v <= 31 ? v : 12); }();
        if (n_groups >= split_min) {
            blocks = (unsigned)((n_groups + RA_SPLIT_PAIRS - 1) / RA_SPLIT_PAIRS);
            tail.n_rows = blocks; tail.K = 16;
            static const bool no_tail16 = getenv("ATLAS_NO_MAIL_TAIL") != nullptr;
            static const bool timing = getenv("ATLAS_RA_SPLIT_TIMING") != nullptr;
            if (no_tail16 && tail.counter) {
                k_ra_prod16_split<false><<<blocks, RA_THREADS, 0, rt().stream>>>(buf, stride, E, n_groups, partials, MailTail{tail.io, nullptr, 0, 0});
                k_col_reduce_mail<<<1, RA_THREADS, 0, rt().stream>>>(partials, blocks, 16u, tail.io);
            } else if (timing)
                k_ra_prod16_split<true><<<blocks, RA_THREADS, 0, rt().stream>>>(buf, stride, E, n_groups, partials, tail);
            else
                k_ra_prod16_split<false><<<blocks, RA_THREADS, 0, rt().stream>>>(buf, stride, E, n_groups, partials, tail);
            return;
        }
    }
    tail.n_rows = blocks; tail.K = D;
    const MailTail none{tail.io, nullptr, 0, 0};
    static const bool no_tail = getenv("ATLAS_NO_MAIL_TAIL") != nullptr;      // diagnosis (tools/stress_lanes.py): the column sums in a launch of their own
    if (no_tail && tail.counter) {
        if (n_groups <= ((size_t)1 << 15)) k_ra_prod_f9_col<D><<<dim3(blocks, D), RA_THREADS, 0, rt().stream>>>(buf, stride, E, n_groups, partials, none);
        else {
            constexpr int KA0 = D < 8 ? D : 8;
            k_ra_prod_f9<D, 0, KA0><<<blocks, RA_THREADS, 0, rt().stream>>>(buf, stride, E, n_groups, partials, none);
            if constexpr (D > 8) k_ra_prod_f9<D, 8, D - 8><<<blocks, RA_THREADS, 0, rt().stream>>>(buf, stride, E, n_groups, partials, none);
        }
        k_col_reduce_mail<<<1, RA_THREADS, 0, rt().stream>>>(partials, blocks, (uint32_t)D, tail.io);
        return;
    }
    // one column per thread up to 2^15 pairs: 16 x more workgroups than the register-tiled kernel, which runs at one wavefront
    // per SIMD there (measured on the Einsum node: 2^15 is 2 % ahead of 2^13, 2^16 no better)
    static const size_t col_log = [] { const char* e = getenv("ATLAS_RA_COL_LOG"); int v = e ? atoi(e) : 0; return (size_t)(v >= 8 && v <= 24 ? v : 15); }();   // experiments
    if (n_groups <= ((size_t)1 << col_log)) {
        k_ra_prod_f9_col<D><<<dim3(blocks, D), RA_THREADS, 0, rt().stream>>>(buf, stride, E, n_groups, partials, tail);
        return;
    }
    constexpr int KA = D < 8 ? D : 8;
    tail.tagged = nullptr;                     // (its rows are assembled by two launches: plain rows behind the arrival counter)
    k_ra_prod_f9<D, 0, KA><<<blocks, RA_THREADS, 0, rt().stream>>>(buf, stride, E, n_groups, partials, D > 8 ? none : tail);
    if constexpr (D > 8)
        k_ra_prod_f9<D, 8, D - 8><<<blocks, RA_THREADS, 0, rt().stream>>>(buf, stride, E, n_groups, partials, tail);   // mails all D columns
}
int launch_prod_d(size_t d, const Fr* buf, size_t stride, Fr* partials, const SplitEqView& E, size_t n_groups, unsigned& blocks, const MailTail& tail) {
    switch (d) {
#define RA_CASE(D) case D: launch_prod<D>(buf, stride, partials, E, n_groups, blocks, tail); break;
        RA_CASE(1) RA_CASE(2) RA_CASE(3) RA_CASE(4) RA_CASE(5) RA_CASE(6) RA_CASE(7) RA_CASE(8)
        RA_CASE(9) RA_CASE(10) RA_CASE(11) RA_CASE(12) RA_CASE(13) RA_CASE(14) RA_CASE(15) RA_CASE(16)
#undef RA_CASE
        default: return fail(ATLAS_EINVAL, "ra_virtual: d > 16");
    }
    return ATLAS_OK;
}

// ---------------------------------------------------------------- RaSumcheckProver
struct RaVirtual : atlas_instance {
    RaRows rows;
    GseDev eq;
    size_t log_T = 0, round_next = 0;
    // lazy rounds 0 and 1 (ra_common.hip.h; poly/ra_poly.rs:21-110): d = 16, log_k = 4, T >= 2^RA_LAZY_LOG, stepped through the round channel.  The rows
    // are allocated from T / 4 on (buf[0]: round 2, buf[1]: round 3); `lazy_lk` = the packed chunk indices (borrowed device lookups or `lazy_own`).
    bool lazy = false;
    const uint64_t* lazy_lk = nullptr;
    uint64_t* lazy_own = nullptr;
    Fr* d_F = nullptr;            // [16][16]
    Fr* d_T1 = nullptr;           // [16][256], built by round 1's launch
    ~RaVirtual() override { rows.release(); eq.release(); if (lazy_own) hipFree(lazy_own); if (d_F) hipFree(d_F); if (d_T1) hipFree(d_T1); }
    // a host-stepped caller of a lazy instance: the gathered rows after all (callers hold rt().mu)
    int unlazy() {
        if (!lazy) return ATLAS_OK;
        if (round_next != 0) return fail(ATLAS_ESTATE, "ra_virtual: a lazy instance cannot change to host stepping after its first round");
        const size_t T = (size_t)1 << log_T;
        for (auto& b : rows.buf) { if (b) hipFree(b); b = nullptr; }
        HIP_TRY(hipMalloc(&rows.buf[0], rows.d * T * sizeof(Fr)));
        HIP_TRY(hipMalloc(&rows.buf[1], rows.d * (T / 2) * sizeof(Fr)));
        rows.stride[0] = T; rows.stride[1] = T / 2; rows.len = T; rows.cur = 0;
        size_t gb = (T + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
        k_ra_gather_lk<<<dim3((unsigned)gb, (unsigned)rows.d), RA_THREADS, 0, rt().stream>>>(lazy_lk, d_F, 16u, T, (uint32_t)rows.d, 4u, rows.buf[0]);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "ra_virtual: gather", e);
        lazy = false;
        return ATLAS_OK;
    }
    size_t rounds() const override { return log_T; }
    size_t degree() const override { return rows.d + 1; }
    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= log_T) return fail(ATLAS_ESTATE, "ra_virtual: round out of order");
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        int rc = unlazy();
        if (rc) return rc;
        const size_t n_groups = rows.len / 2;
        unsigned blocks = (unsigned)((n_groups + RA_THREADS - 1) / RA_THREADS);
        rc = launch_prod_d(rows.d, rows.buf[rows.cur], rows.stride[rows.cur], rows.partials, eq.view(), n_groups, blocks, MailTail{{}, nullptr, 0, 0});
        if (rc) return rc;
        std::vector<H::Fr> sums(rows.d);
        rc = rows.reduce_to_host(blocks, (uint32_t)rows.d, sums.data());
        if (rc) return rc;
        return finish_sums(sums, claim, coeffs);
    }
    int finish_sums(std::vector<H::Fr>& sums, const H::Fr& claim, std::vector<H::Fr>& coeffs) {
        H::Fr fix = H::one();                                        // 32^(d+1): the 2^-5 per f9_mul
        for (size_t k = 0; k < rows.d + 1; k++) fix = H::mul(fix, H::from_u64(32));
        fix = H::mul(fix, eq.st.scalar);                             // mles_product_sum.rs:131
        for (auto& s : sums) s = H::mul(s, fix);
        coeffs = H::finish_product_sum(sums, claim, eq.st);
        return ATLAS_OK;
    }
    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= log_T) return fail(ATLAS_ESTATE, "ra_virtual: round out of order");
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        int rc = rows.bind(r);
        if (rc) return rc;
        eq.st.bind(H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode));
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override {
        if (have_finals) { out = mailed_finals; return ATLAS_OK; }
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        return rows.finals(out);
    }

    // ---- round-channel stepping (instance.hpp).  Rows of round k live in buf[k & 1] with stride T >> k.
    bool have_finals = false;
    std::vector<H::Fr> mailed_finals;
    bool pipelined() const override { return log_T >= 1; }
    static bool fuse_off() { static const bool v = getenv("ATLAS_RA_NO_FUSE") != nullptr; return v; }   // diagnosis / A-B: bind and product as two launches
    bool fused(size_t round) const { return round >= 1 && round < log_T && (((size_t)1 << log_T) >> round) / 2 <= RA_FUSE_MAX && !fuse_off(); }
    bool wide_wait(size_t round) const override {                 // the bind of `round`: ceil(len / RA_THREADS) x d workgroups (fused: a workgroup per 8 pairs); the finals: one
        if (round >= log_T) return false;
        const size_t len = ((size_t)1 << log_T) >> round;
        if (fused(round)) return (len / 2 + RA_FUSE_PAIRS - 1) / RA_FUSE_PAIRS > WIDE_WAIT_WGS;
        return ((len + RA_THREADS - 1) / RA_THREADS) * rows.d > WIDE_WAIT_WGS;
    }
    int enqueue(size_t round, const atlas::RoundIo& io, bool bind_prev, atlas_mail_ref& mail) override {
        if (round >= log_T || (round > 0) != bind_prev) return fail(ATLAS_ESTATE, "ra_virtual: enqueue out of order");
        const size_t T = (size_t)1 << log_T, len = T >> round, n_groups = len / 2;
        const ChanIo cio{io, rt().challenge_mode};
        size_t ot, it;
        eq.st.tops_after(round, ot, it);
        if (lazy && round <= 2) {
            const int hi = rt().challenge_mode == 0 ? 1 : 0;
            if (round < 2) {                                      // the split product over the packed indices and a table
                if (round == 1) k_ra_lazy_t1<<<1, RA_THREADS, 0, rt().stream>>>(d_F, 16u, 16u, d_T1, cio, hi);
                const unsigned blocks = (unsigned)((n_groups + RA_SPLIT_PAIRS - 1) / RA_SPLIT_PAIRS);
                const MailTail tail{io, rows.d_counter, blocks, 16u, rows.tg()};
                if (round == 0) k_ra_prod16_split<false, 1><<<blocks, RA_THREADS, 0, rt().stream>>>(d_F, 16, eq.view_at(ot, it), n_groups, rows.partials, tail, lazy_lk);
                else k_ra_prod16_split<false, 2><<<blocks, RA_THREADS, 0, rt().stream>>>(d_T1, 256, eq.view_at(ot, it), n_groups, rows.partials, tail, lazy_lk);
            } else {                                              // the rows at last, bound through r1 (T / 4 long), and the product over them
                size_t gb = (len + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
                k_ra_lazy_rows<<<dim3((unsigned)gb, 16u), RA_THREADS, 0, rt().stream>>>(lazy_lk, d_T1, 16u, len, rows.buf[0], len, cio, hi);
                unsigned blocks = (unsigned)((n_groups + RA_THREADS - 1) / RA_THREADS);
                int rc = launch_prod_d(rows.d, rows.buf[0], len, rows.partials, eq.view_at(ot, it), n_groups, blocks, MailTail{io, rows.d_counter, 0, 0, rows.tg()});
                if (rc) return rc;
            }
        } else if (fused(round)) {                                // bind + product in one launch (k_ra_bind_prod_f9)
            const unsigned fb = (unsigned)((n_groups + RA_FUSE_PAIRS - 1) / RA_FUSE_PAIRS);
            k_ra_bind_prod_f9<<<fb, RA_THREADS, 0, rt().stream>>>(rows.buf[(round - 1) & 1], T >> (round - 1), rows.buf[round & 1], len, (uint32_t)rows.d, eq.view_at(ot, it),
                                                               n_groups, rows.partials, cio, rt().challenge_mode == 0 ? 1 : 0, MailTail{io, rows.d_counter, fb, (uint32_t)rows.d, rows.tg()});
        } else {
            if (bind_prev) {
                size_t gb = (len + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
                k_ra_bind_ch<<<dim3((unsigned)gb, (unsigned)rows.d), RA_THREADS, 0, rt().stream>>>(rows.buf[(round - 1) & 1], T >> (round - 1), rows.buf[round & 1], len, len,
                                                                                              cio, rt().challenge_mode == 0 ? 1 : 0);
            }
            unsigned blocks = (unsigned)((n_groups + RA_THREADS - 1) / RA_THREADS);
            int rc = launch_prod_d(rows.d, rows.buf[round & 1], len, rows.partials, eq.view_at(ot, it), n_groups, blocks, MailTail{io, rows.d_counter, 0, 0, rows.tg()});
            if (rc) return rc;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "ra_virtual: launch", e);
        mail.base = io.mail; mail.blocks = 1; mail.n_vals = (int)rows.d; mail.radix = 32; mail.shl = 0;
        return ATLAS_OK;
    }
    void prepare(size_t round) override { if (round == round_next) eq.st.prepare_inverses(true); }
    int finish(size_t round, const H::Fr& claim, const H::Fr* s, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= log_T) return fail(ATLAS_ESTATE, "ra_virtual: round out of order");
        std::vector<H::Fr> sums(s, s + rows.d);
        return finish_sums(sums, claim, coeffs);
    }
    int host_ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= log_T) return fail(ATLAS_ESTATE, "ra_virtual: round out of order");
        eq.st.bind(H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode));
        rows.cur = (int)((round + 1) & 1); rows.len = ((size_t)1 << log_T) >> (round + 1); rows.stride[rows.cur] = rows.len;
        round_next++;
        return ATLAS_OK;
    }
    int enqueue_finals(const atlas::RoundIo& io, atlas_mail_ref& mail) override {
        const size_t T = (size_t)1 << log_T;
        k_rows_final_ch<<<1, 64, 0, rt().stream>>>(rows.buf[(log_T - 1) & 1], T >> (log_T - 1), (uint32_t)rows.d, ChanIo{io, rt().challenge_mode}, rt().challenge_mode == 0 ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "ra_virtual: launch", e);
        mail.base = io.mail; mail.blocks = 1; mail.n_vals = (int)rows.d; mail.radix = 32; mail.shl = 0;
        return ATLAS_OK;
    }
    int set_finals(const H::Fr* vals, size_t n) override {
        if (n != rows.d) return fail(ATLAS_EINVAL, "ra_virtual: final claims");
        mailed_finals.assign(vals, vals + n); have_finals = true;
        return ATLAS_OK;
    }
};

// F_i = EqPolynomial::evals(r_address chunk i) for d chunks of log_k <= 8 challenges, built on the device from the d log_k
// challenges (carried as a kernel argument): thread (i, k) multiplies the log_k factors of entry k, variable 0 = most
// significant bit of k (eq_poly.rs:77-101: big-endian).  Same residues as the host table (field arithmetic is exact).
__global__ __launch_bounds__(RA_THREADS) void k_ra_eq_tables(FrArgs r, uint32_t d, uint32_t log_k, Fr* __restrict__ out /* [d][2^log_k] */) {
    const uint32_t K = 1u << log_k;
    for (uint32_t t = blockIdx.x * RA_THREADS + threadIdx.x; t < d * K; t += gridDim.x * RA_THREADS) {
        const uint32_t i = t / K, k = t % K;
        Fr acc = fr_one();
        for (uint32_t b = 0; b < log_k; b++) {
            const Fr x = r.v[i * log_k + b];
            acc = fr_mul(acc, ((k >> (log_k - 1 - b)) & 1u) ? x : fr_sub(fr_one(), x));
        }
        fe_store(out + t, acc);
    }
}

// upload d host tables of K Fr each
int upload_tables(const std::vector<std::vector<H::Fr>>& t, size_t K, Fr** out) {
    Fr* d = nullptr;
    HIP_TRY(hipMalloc(&d, t.size() * K * sizeof(Fr)));
    for (size_t i = 0; i < t.size(); i++)
        HIP_TRY(hipMemcpyAsync(d + i * K, t[i].data(), K * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    *out = d;
    return ATLAS_OK;
}

// ExpandingTable::update (LowToHigh) on the device for the pipelined Booleanity: F has n entries, the challenge of the
// round just closed doubles it.  One workgroup; n <= 2^15.
__global__ __launch_bounds__(RA_THREADS) void k_bool_expand_ch(Fr* F, uint32_t n, ChanIo io) {
    Fr r;
    if (!io.challenge(r)) return;
    for (uint32_t x = threadIdx.x; x < n; x += RA_THREADS) {
        const Fr f = fe_load(F + x), hi = fr_mul(f, r);
        fe_store(F + n + x, hi);
        fe_store(F + x, fr_sub(f, hi));
    }
}
__global__ void k_bool_expand_init(Fr* F) { if (threadIdx.x == 0) fe_store(F, fr_one()); }
// The same table for ALL n <= 8 address challenges in ONE launch (every lookup of the graph prover: chunks of 4 bits): lane k of the first
// wavefront polls slot k — one crossing of the host link, not n in turn — and the table doubles in LDS.  A launch per address round was five
// launches (~5 us of the host thread each) in rounds that are host arithmetic otherwise, i.e. on the batch's critical path.
struct BoolSlots { const Chunk* host[8]; uint32_t tag[8]; uint32_t n; uint32_t* abort_flag; int challenge_mode; };
__global__ __launch_bounds__(RA_THREADS) void k_bool_expand_all_ch(Fr* F, BoolSlots S) {
    __shared__ uint64_t s_r[8][2];
    __shared__ uint32_t s_bad;
    __shared__ Fr tab[RA_THREADS];
    if (threadIdx.x == 0) { s_bad = 0; tab[0] = fr_one(); }
    __syncthreads();
    if (threadIdx.x < S.n) {
        uint64_t lo = 0, hi = 0;
        if (!ch_poll_slot<true>(S.host[threadIdx.x], S.tag[threadIdx.x], S.abort_flag, lo, hi)) atomicOr(&s_bad, 1u);
        s_r[threadIdx.x][0] = lo; s_r[threadIdx.x][1] = hi;
    }
    __syncthreads();
    if (s_bad) return;
    for (uint32_t k = 0; k < S.n; k++) {                 // ExpandingTable::update, LowToHigh: F[n + x] = r F[x], F[x] -= F[n + x]
        const Fr r = challenge_to_mont(s_r[k][0], s_r[k][1], S.challenge_mode);
        const uint32_t n = 1u << k;
        if (threadIdx.x < n) { const Fr f = tab[threadIdx.x], hi = fr_mul(f, r); tab[n + threadIdx.x] = hi; tab[threadIdx.x] = fr_sub(f, hi); }
        __syncthreads();
    }
    if (threadIdx.x < (1u << S.n)) fe_store(F + threadIdx.x, tab[threadIdx.x]);
}

// ---------------------------------------------------------------- BooleanitySumcheckProver
struct Booleanity : atlas_instance {
    size_t d = 0, log_k = 0, log_T = 0, round_next = 0;
    std::vector<std::vector<H::Fr>> G;                 // d x 2^log_k (host: 16 entries each)
    std::vector<H::Fr> Ggam;                           // sum_i gamma_i G_i: what phase 1 reads
    std::vector<H::Fr> gammas, F;                      // F = ExpandingTable values
    H::GseState B;
    std::vector<std::vector<H::Fr>> B_out, B_in;       // host prefix tables of B
    GseDev D;
    RaRows rows;
    Fr *d_gammas = nullptr, *d_F = nullptr;          // d_F: 2^log_k entries, the device ExpandingTable of the pipelined path
    H::Fr eq_r_r = H::zero(), eq_r_r_inv = H::zero();
    bool have_eq_r_r_inv = false;
    // lazy first cycle round (k_bool_fold<true>): log_k = 4, device-resident lookup words, T >= 2^18 (ATLAS_BOOL_LAZY_LOG), the round channel.  No gather: cycle round 0
    // reads the words and two tables, cycle round 1's rows (T / 2) come from the words and F bound with its challenge; buf[1] holds T / 2, buf[0] T / 4.
    bool lazy = false;
    Fr* d_tabs = nullptr;                              // Ctab [d][16] then Etab [d][256]
    ~Booleanity() override { rows.release(); D.release(); if (d_gammas) hipFree(d_gammas); if (d_F) hipFree(d_F); if (d_tabs) hipFree(d_tabs); }
    int unlazy() {                                     // a host-stepped caller: the full-length buffers after all (callers hold rt().mu)
        if (!lazy) return ATLAS_OK;
        const size_t T = (size_t)1 << log_T;
        for (auto& b : rows.buf) { if (b) hipFree(b); b = nullptr; }
        HIP_TRY(hipMalloc(&rows.buf[0], d * T * sizeof(Fr)));
        HIP_TRY(hipMalloc(&rows.buf[1], d * (T / 2) * sizeof(Fr)));
        rows.stride[0] = T; rows.stride[1] = T / 2; rows.len = T; rows.cur = 0;
        lazy = false;
        return ATLAS_OK;
    }
    size_t rounds() const override { return log_k + log_T; }
    size_t degree() const override { return 3; }

    int phase1_message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) {      // compute_phase1_message
        const size_t m = round + 1;
        const auto& e_out = B_out[B.out_top]; const auto& e_in = B_in[B.in_top];
        H::Fr q0 = H::zero(), qinf = H::zero();
        for (size_t xo = 0; xo < e_out.size(); xo++) {
            H::Fr i0 = H::zero(), i1 = H::zero();
            for (size_t xi = 0; xi < e_in.size(); xi++) {
                const size_t kp = (xo << B.in_top) | xi;
                // sum_i gamma_i sum_k G_i[k] (F^2 - F)[k]: the expanding table F is the same for every chunk i, so the chunks meet once, in
                // Ggam[k] = sum_i gamma_i G_i[k] (built with the instance), and a round is 2 multiplications per address instead of 2 d + 2
                H::Fr c0 = H::zero(), c1 = H::zero();
                for (size_t k = 0; k < ((size_t)1 << m); k++) {
                    const H::Fr& Gk = Ggam[(kp << m) + k];
                    if (H::detail::is_zero4(Gk.l)) continue;              // (an address no lookup reads adds nothing: GatherSmall's table has 2^16 addresses and 16 lookups)
                    const H::Fr& Fk = F[k % ((size_t)1 << (m - 1))];
                    const H::Fr gf = H::mul(Gk, Fk), ei = H::mul(gf, Fk);
                    if ((k >> (m - 1)) == 0) c0 = H::add(c0, H::sub(ei, gf));
                    c1 = H::add(c1, ei);
                }
                i0 = H::add(i0, H::mul(e_in[xi], c0)); i1 = H::add(i1, H::mul(e_in[xi], c1));
            }
            q0 = H::add(q0, H::mul(e_out[xo], i0)); qinf = H::add(qinf, H::mul(e_out[xo], i1));
        }
        H::gruen_deg3(B, q0, qinf, claim, coeffs.data());
        return ATLAS_OK;
    }
    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "booleanity: round out of order");
        coeffs.assign(4, H::zero());
        if (round < log_k) return phase1_message(round, claim, coeffs);
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);                        // compute_phase2_message
        const size_t n_groups = rows.len / 2;
        const uint32_t n_part = launch_fold(rows.buf[rows.cur], rows.stride[rows.cur], D.view(), n_groups);
        H::Fr s[2];
        int rc = rows.reduce_to_host(n_part, 2, s);
        if (rc) return rc;
        return finish_phase2(s, claim, coeffs);
    }
    uint32_t launch_fold(const Fr* buf, size_t stride, const SplitEqView& E, size_t n_groups, const atlas::RoundIo* io = nullptr) {
        size_t blocks = (n_groups + RA_THREADS - 1) / RA_THREADS; if (blocks > 2048) blocks = 2048;
        static const size_t split_log = [] { const char* e = getenv("ATLAS_BOOL_SPLIT_LOG"); int v = e ? atoi(e) : 0; return (size_t)(v >= 8 && v <= 24 ? v : 13); }();   // experiments
        const unsigned ysplit = n_groups <= ((size_t)1 << split_log) ? (unsigned)d : 1u;   // latency regime: one row per thread
        const MailTail tail = io ? MailTail{*io, rows.d_counter, (uint32_t)(blocks * ysplit), 2u, rows.tg()} : MailTail{{}, nullptr, 0, 0};
        static const bool no_tail = getenv("ATLAS_NO_MAIL_TAIL") != nullptr;  // diagnosis (tools/stress_lanes.py)
        if (no_tail && io) {
            k_bool_fold<false><<<dim3((unsigned)blocks, ysplit), RA_THREADS, 0, rt().stream>>>(buf, stride, (uint32_t)d, d_gammas, E, n_groups, rows.partials, MailTail{{}, nullptr, 0, 0});
            k_col_reduce_mail<<<1, RA_THREADS, 0, rt().stream>>>(rows.partials, (uint32_t)(blocks * ysplit), 2u, *io);
            return (uint32_t)(blocks * ysplit);
        }
        k_bool_fold<false><<<dim3((unsigned)blocks, ysplit), RA_THREADS, 0, rt().stream>>>(buf, stride, (uint32_t)d, d_gammas, E, n_groups, rows.partials, tail);
        return (uint32_t)(blocks * ysplit);
    }
    int finish_phase2(const H::Fr* sums, const H::Fr& claim, std::vector<H::Fr>& coeffs) {
        static const H::Fr fix = H::from_u64(1048576);               // 32^4: the 2^-5 of each of the four f9_mul behind a term
        const H::Fr s0 = H::mul(sums[0], fix), s1 = H::mul(sums[1], fix);
        if (!have_eq_r_r_inv) { eq_r_r_inv = H::inv(eq_r_r); have_eq_r_r_inv = true; }   // constant over phase 2
        const H::Fr adj = H::mul(claim, eq_r_r_inv);
        H::gruen_deg3(D.st, s0, s1, adj, coeffs.data());
        for (auto& c : coeffs) c = H::mul(c, eq_r_r);                // gruen_poly * eq_r_r (from_coeff)
        H::trim(coeffs);
        return ATLAS_OK;
    }

    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "booleanity: round out of order");
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode);
        if (round < log_k) {
            B.bind(rf);
            const size_t n = F.size();                               // ExpandingTable::update, LowToHigh
            F.resize(2 * n);
            for (size_t x = 0; x < n; x++) { F[n + x] = H::mul(F[x], rf); F[x] = H::sub(F[x], F[n + x]); }
            if (round == log_k - 1) {
                std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
                eq_r_r = B.scalar;
                Fr* d_Fh = nullptr;
                HIP_TRY(hipMalloc(&d_Fh, F.size() * sizeof(Fr)));
                HIP_TRY(hipMemcpyAsync(d_Fh, F.data(), F.size() * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
                int rc = unlazy();
                if (!rc) rc = rows.gather(d_Fh, 0);                   // every H_i reads the same table F
                hipFree(d_Fh);
                if (rc) return rc;
                G.clear();
            }
        } else {
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            D.st.bind(rf);
            int rc = rows.bind(r);
            if (rc) return rc;
        }
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override {
        if (have_finals) { out = mailed_finals; return ATLAS_OK; }
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        return rows.finals(out);
    }

    // ---- round-channel stepping (instance.hpp).  Phase 1 (the log_k address rounds) is host arithmetic: its enqueues
    // only keep the device copy of the expanding table F in step (k_bool_expand_ch applies the challenge of the round
    // just closed), so that the enqueue of the first cycle round can gather H_i = F[idx_i] without the host.  Phase-2
    // round p has its rows in buf[p & 1] with stride T >> p.
    bool have_finals = false;
    std::vector<H::Fr> mailed_finals;
    bool pipelined() const override { return log_k <= 15 && log_T >= 1; }
    BoolSlots slots{};
    bool one_expand() const { static const bool off = getenv("ATLAS_BOOL_EXPAND_PER_ROUND") != nullptr; return log_k >= 1 && log_k <= 8 && !off; }     // A-B
    bool silent_round(size_t round) const override { return round < log_k && one_expand(); }     // the address rounds launch nothing
    size_t host_prefix() const override { return one_expand() ? log_k : 0; }
    // cycle round p >= 1 in one launch (k_bool_bind_fold); ATLAS_RA_NO_FUSE / ATLAS_NO_MAIL_TAIL: the separate launches (diagnosis, A-B)
    bool fused(size_t p) const {
        static const bool off = getenv("ATLAS_RA_NO_FUSE") != nullptr || getenv("ATLAS_BOOL_NO_FUSE") != nullptr || getenv("ATLAS_NO_MAIL_TAIL") != nullptr;
        return p >= 1 && p < log_T && (((size_t)1 << log_T) >> p) / 2 <= RA_FUSE_MAX && !off;
    }
    bool wide_wait(size_t round) const override {                 // address rounds: one workgroup steps F; cycle round p >= 1: the bind
        if (round <= log_k || round >= rounds()) return false;
        const size_t len = ((size_t)1 << log_T) >> (round - log_k);
        return ((len + RA_THREADS - 1) / RA_THREADS) * d > WIDE_WAIT_WGS;          // (the fused launch has two lanes per pair: the same count)
    }
    int enqueue(size_t round, const atlas::RoundIo& io, bool bind_prev, atlas_mail_ref& mail) override {
        if (round >= rounds() || (round > 0) != bind_prev) return fail(ATLAS_ESTATE, "booleanity: enqueue out of order");
        const ChanIo cio{io, rt().challenge_mode};
        mail.base = io.mail; mail.blocks = 0; mail.n_vals = 0; mail.radix = 32; mail.shl = 0;
        if (round == 0 && !d_F) HIP_TRY(hipMalloc(&d_F, ((size_t)1 << log_k) * sizeof(Fr)));
        if (one_expand()) {                                           // the expanding table in one launch, with the first cycle round (k_bool_expand_all_ch)
            if (round >= 1 && round <= log_k) { slots.host[round - 1] = io.r_host; slots.tag[round - 1] = io.tag_r; }
            if (round == log_k) {
                slots.n = (uint32_t)log_k; slots.abort_flag = io.abort_flag; slots.challenge_mode = rt().challenge_mode;
                k_bool_expand_all_ch<<<1, RA_THREADS, 0, rt().stream>>>(d_F, slots);
            }
        } else {
            if (round == 0) k_bool_expand_init<<<1, 64, 0, rt().stream>>>(d_F);
            if (round >= 1 && round <= log_k) k_bool_expand_ch<<<1, RA_THREADS, 0, rt().stream>>>(d_F, 1u << (round - 1), cio);
        }
        if (round < log_k) return ATLAS_OK;
        const size_t T = (size_t)1 << log_T, p = round - log_k, len = T >> p, n_groups = len / 2;
        if (lazy && p <= 1) {
            size_t ot0, it0;
            D.st.tops_after(p, ot0, it0);
            if (p == 0) {                                             // two tables from F and the gammas, then the fold over the words
                k_bool_lazy_tables<<<(unsigned)((d * 256 + RA_THREADS - 1) / RA_THREADS), RA_THREADS, 0, rt().stream>>>(d_F, d_gammas, (uint32_t)d, d_tabs, d_tabs + d * 16);
                size_t blocks = (n_groups + RA_THREADS - 1) / RA_THREADS; if (blocks > 2048) blocks = 2048;
                k_bool_fold<true><<<dim3((unsigned)blocks, 1u), RA_THREADS, 0, rt().stream>>>(d_tabs, 0, (uint32_t)d, d_tabs + d * 16, D.view_at(ot0, it0), n_groups, rows.partials,
                                                                                      MailTail{io, rows.d_counter, (uint32_t)blocks, 2u, rows.tg()}, rows.lk);
            } else {                                                  // the rows of cycle round 1 straight from the words, bound with round 0's challenge
                size_t gb = (len + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
                k_bool_lazy_rows<<<dim3((unsigned)gb, (unsigned)d), RA_THREADS, 0, rt().stream>>>(rows.lk, d_F, (uint32_t)d, len, rows.buf[1], len, cio, rt().challenge_mode == 0 ? 1 : 0);
                launch_fold(rows.buf[1], len, D.view_at(ot0, it0), n_groups, &io);
            }
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(ATLAS_ENODEV, "booleanity: launch", e);
            mail.blocks = 1; mail.n_vals = 2;
            return ATLAS_OK;
        }
        if (p == 0) {
            if (!rows.d_idx && !rows.lk) return fail(ATLAS_ESTATE, "booleanity: indices not uploaded");
            size_t gb = (T + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
            if (rows.lk) k_ra_gather_lk<<<dim3((unsigned)gb, (unsigned)d), RA_THREADS, 0, rt().stream>>>(rows.lk, d_F, 0u, T, (uint32_t)d, rows.lk_log, rows.buf[0]);
            else k_ra_gather<<<dim3((unsigned)gb, (unsigned)d), RA_THREADS, 0, rt().stream>>>(rows.d_idx, d_F, 0u, T, rows.buf[0]);   // every H_i reads the same table F
        } else if (!fused(p)) {
            size_t gb = (len + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
            k_ra_bind_ch<<<dim3((unsigned)gb, (unsigned)d), RA_THREADS, 0, rt().stream>>>(rows.buf[(p - 1) & 1], T >> (p - 1), rows.buf[p & 1], len, len, cio,
                                                                                     rt().challenge_mode == 0 ? 1 : 0);
        }
        size_t ot, it;
        D.st.tops_after(p, ot, it);
        if (fused(p)) {
            const unsigned fb = (unsigned)((2 * n_groups + RA_THREADS - 1) / RA_THREADS);       // two lanes per pair
            k_bool_bind_fold<<<dim3(fb, (unsigned)d), RA_THREADS, 0, rt().stream>>>(rows.buf[(p - 1) & 1], T >> (p - 1), rows.buf[p & 1], len, d_gammas, D.view_at(ot, it), n_groups,
                                                                                rows.partials, cio, rt().challenge_mode == 0 ? 1 : 0,
                                                                                MailTail{io, rows.d_counter, (uint32_t)(fb * d), 2u, rows.tg()});
        } else
            launch_fold(rows.buf[p & 1], len, D.view_at(ot, it), n_groups, &io);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "booleanity: launch", e);
        mail.blocks = 1; mail.n_vals = 2;
        return ATLAS_OK;
    }
    void prepare(size_t round) override { if (round == round_next && round > log_k) D.st.prepare_inverses(false); }
    int finish(size_t round, const H::Fr& claim, const H::Fr* s, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "booleanity: round out of order");
        coeffs.assign(4, H::zero());
        return round < log_k ? phase1_message(round, claim, coeffs) : finish_phase2(s, claim, coeffs);
    }
    int host_ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "booleanity: round out of order");
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode);
        if (round < log_k) {
            B.bind(rf);
            const size_t n = F.size();                               // ExpandingTable::update, LowToHigh (the host copy feeds phase1_message)
            F.resize(2 * n);
            for (size_t x = 0; x < n; x++) { F[n + x] = H::mul(F[x], rf); F[x] = H::sub(F[x], F[n + x]); }
            if (round == log_k - 1) { eq_r_r = B.scalar; G.clear(); rows.cur = 0; rows.len = (size_t)1 << log_T; rows.stride[0] = rows.len; }
        } else {
            D.st.bind(rf);
            const size_t p = round - log_k;
            rows.cur = (int)((p + 1) & 1); rows.len = ((size_t)1 << log_T) >> (p + 1); rows.stride[rows.cur] = rows.len;
        }
        round_next++;
        return ATLAS_OK;
    }
    int enqueue_finals(const atlas::RoundIo& io, atlas_mail_ref& mail) override {
        const size_t T = (size_t)1 << log_T;
        k_rows_final_ch<<<1, 64, 0, rt().stream>>>(rows.buf[(log_T - 1) & 1], T >> (log_T - 1), (uint32_t)d, ChanIo{io, rt().challenge_mode}, rt().challenge_mode == 0 ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "booleanity: launch", e);
        mail.base = io.mail; mail.blocks = 1; mail.n_vals = (int)d; mail.radix = 32; mail.shl = 0;
        return ATLAS_OK;
    }
    int set_finals(const H::Fr* vals, size_t n) override {
        if (n != d) return fail(ATLAS_EINVAL, "booleanity: final claims");
        mailed_finals.assign(vals, vals + n); have_finals = true;
        return ATLAS_OK;
    }
};

// ---------------------------------------------------------------- HammingWeightSumcheckProver (host: d x 2^log_k)
struct HammingWeight : atlas_instance {
    size_t log_k = 0, round_next = 0;
    std::vector<std::vector<H::Fr>> ra;
    std::vector<H::Fr> gamma_powers;
    size_t rounds() const override { return log_k; }
    size_t degree() const override { return 1; }
    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= log_k) return fail(ATLAS_ESTATE, "hamming_weight: round out of order");
        H::Fr e0 = H::zero();
        for (size_t i = 0; i < ra.size(); i++) {
            H::Fr s = H::zero();
            for (size_t k = 0; k < ra[i].size() / 2; k++) s = H::add(s, ra[i][2 * k]);
            e0 = H::add(e0, H::mul(s, gamma_powers[i]));
        }
        const H::Fr e1 = H::sub(claim, e0);
        coeffs = {e0, H::sub(e1, e0)};                                // from_evals (2 points) -> from_coeff
        H::trim(coeffs);
        return ATLAS_OK;
    }
    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= log_k) return fail(ATLAS_ESTATE, "hamming_weight: round out of order");
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode);
        for (auto& p : ra) {
            const size_t half = p.size() / 2;
            for (size_t k = 0; k < half; k++) {
                if (H::detail::is_zero4(p[2 * k].l) && H::detail::is_zero4(p[2 * k + 1].l)) { p[k] = H::zero(); continue; }     // (GatherSmall: 2^16 addresses, a handful read)
                p[k] = H::add(p[2 * k], H::mul(rf, H::sub(p[2 * k + 1], p[2 * k])));
            }
            p.resize(half);
        }
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override {
        out.clear();
        for (auto& p : ra) { if (p.size() != 1) return fail(ATLAS_ESTATE, "final_claims: rounds remaining"); out.push_back(p[0]); }
        return ATLAS_OK;
    }
    // round-channel stepping: nothing to launch and nothing to collect, the rounds are host arithmetic
    bool pipelined() const override { return true; }
    bool wide_wait(size_t) const override { return false; }
    bool silent_lane() const override { return true; }
    size_t host_prefix() const override { return log_k; }
    bool silent_round(size_t) const override { return true; }
    int enqueue(size_t, const atlas::RoundIo& io, bool, atlas_mail_ref& mail) override { mail.base = io.mail; mail.blocks = 0; mail.n_vals = 0; return ATLAS_OK; }
    int finish(size_t round, const H::Fr& claim, const H::Fr*, std::vector<H::Fr>& coeffs) override { return message(round, claim, coeffs); }
    int host_ingest(const atlas_u128_t& r, size_t round) override { return ingest(r, round); }
    int enqueue_finals(const atlas::RoundIo& io, atlas_mail_ref& mail) override { mail.base = io.mail; mail.blocks = 0; mail.n_vals = 0; return ATLAS_OK; }
    int set_finals(const H::Fr*, size_t) override { return ATLAS_OK; }
};

}  // namespace

extern "C" {

static int ra_virtual_build(const int32_t* const* H_indices, const uint64_t* lookups, size_t d, size_t log_k_chunk, size_t log_T,
                            const atlas_fr_t* r_address_chunks, const atlas_fr_t* r_cycle, atlas_instance_t* out);

int atlas_ra_virtual_new(const int32_t* const* H_indices, size_t d, size_t log_k_chunk, size_t log_T,
                         const atlas_fr_t* r_address_chunks, const atlas_fr_t* r_cycle, atlas_instance_t* out) {
    NEED_INIT();
    if (!H_indices) return fail(ATLAS_EINVAL, "ra_virtual_new: null argument");
    return ra_virtual_build(H_indices, nullptr, d, log_k_chunk, log_T, r_address_chunks, r_cycle, out);
}

// RaSumcheckProver::gen from the lookup indices and r_address: OneHotParams::{lookup_index_chunk, compute_r_address_chunks}
// (config.rs:73-100) done here — d = ceil(log_K / log_k_chunk), r_address left-padded with zeros to a multiple of the chunk
int atlas_ra_virtual_from_lookups_new(const uint64_t* lookup_indices, size_t log_T, size_t log_K, size_t log_k_chunk,
                                      const atlas_fr_t* r_address, const atlas_fr_t* r_cycle, atlas_instance_t* out) {
    PROF("atlas_ra_virtual_from_lookups_new");
    NEED_INIT();
    if (!lookup_indices || !r_address || log_k_chunk == 0 || log_K == 0) return fail(ATLAS_EINVAL, "ra_virtual_from_lookups_new: null argument");
    const size_t d = (log_K + log_k_chunk - 1) / log_k_chunk, pad = d * log_k_chunk - log_K;
    std::vector<atlas_fr_t> chunks(d * log_k_chunk);
    std::memset(chunks.data(), 0, pad * sizeof(atlas_fr_t));
    std::memcpy(chunks.data() + pad, r_address, log_K * sizeof(atlas_fr_t));
    return ra_virtual_build(nullptr, lookup_indices, d, log_k_chunk, log_T, chunks.data(), r_cycle, out);
}

static int ra_virtual_build(const int32_t* const* H_indices, const uint64_t* lookups, size_t d, size_t log_k_chunk, size_t log_T,
                            const atlas_fr_t* r_address_chunks, const atlas_fr_t* r_cycle, atlas_instance_t* out) {
    if (!r_address_chunks || (!r_cycle && log_T) || !out) return fail(ATLAS_EINVAL, "ra_virtual_new: null argument");
    if (d == 0 || d > RA_MAX_D) return fail(ATLAS_EINVAL, "ra_virtual_new: d must be in 1..16");
    // log_T == 0: ONE cycle — a member without rounds (its claim is the product of the d gathered values); stepped by the host, never started
    if (log_k_chunk > 16 || log_T > 25) return fail(ATLAS_EINVAL, "ra_virtual_new: log_k_chunk <= 16, log_T <= 25");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    RaVirtual* P = new RaVirtual();
    P->log_T = log_T;
    const size_t K = (size_t)1 << log_k_chunk, T = (size_t)1 << log_T;
    const H::Fr* ch = reinterpret_cast<const H::Fr*>(r_address_chunks);
    Fr* d_tabs = nullptr;
    int rc = ATLAS_OK;
    if (log_k_chunk <= 8 && d * log_k_chunk <= 64) {                 // ra_virtual.rs:113-116, on the device
        HIP_TRY(hipMalloc(&d_tabs, d * K * sizeof(Fr)));
        FrArgs a;
        std::memcpy(a.v, ch, d * log_k_chunk * sizeof(Fr));
        k_ra_eq_tables<<<(unsigned)((d * K + RA_THREADS - 1) / RA_THREADS), RA_THREADS, 0, rt().stream>>>(a, (uint32_t)d, (uint32_t)log_k_chunk, d_tabs);
    } else {
        std::vector<std::vector<H::Fr>> tabs(d);
        for (size_t i = 0; i < d; i++) tabs[i] = H::eq_evals(ch + i * log_k_chunk, log_k_chunk);
        rc = upload_tables(tabs, K, &d_tabs);
    }
    // lazy rounds 0 and 1 (RaVirtual::lazy): 16 chunks of 4 bits, T >= 2^18 (ATLAS_RA_LAZY_LOG moves it, 31 = never), the round channel
    static const size_t lazy_log = [] { const char* e = getenv("ATLAS_RA_LAZY_LOG"); const int v = e ? atoi(e) : 18; return (size_t)(v >= 14 && v <= 31 ? v : 18); }();
    static const bool no_pipe = getenv("ATLAS_NO_PIPELINE") != nullptr;
    bool lazy = !rc && d == 16 && log_k_chunk == 4 && log_T >= lazy_log && d_tabs && rt().fs_mode == ATLAS_FS_HOST && !no_pipe;
    if (lazy) {
        P->rows.d = d; P->rows.len = T;
        if (H_indices) {                                          // host chunk rows: packed on the device; an index outside 0..15 (None) -> gathered rows
            rc = P->rows.upload_indices(H_indices);
            uint32_t* d_bad = nullptr;
            if (!rc) { HIP_TRY(hipMalloc(&P->lazy_own, T * sizeof(uint64_t))); HIP_TRY(hipMalloc(&d_bad, 4)); HIP_TRY(hipMemsetAsync(d_bad, 0, 4, rt().stream)); }
            uint32_t bad = 1;
            if (!rc) {
                size_t gb = (T + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
                k_ra_pack_nibbles<<<(unsigned)gb, RA_THREADS, 0, rt().stream>>>(P->rows.d_idx, T, (uint32_t)d, P->lazy_own, d_bad);
                HIP_TRY(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, rt().stream));
                HIP_TRY(hipStreamSynchronize(rt().stream));
            }
            if (d_bad) hipFree(d_bad);
            if (bad) { lazy = false; hipFree(P->lazy_own); P->lazy_own = nullptr; }
            else { hipFree(P->rows.d_idx); P->rows.d_idx = nullptr; P->lazy_lk = P->lazy_own; }
        } else {
            rc = P->rows.upload_lookups(lookups, (uint32_t)log_k_chunk);
            if (!rc && P->rows.lk) P->lazy_lk = P->rows.lk; else lazy = false;
        }
    }
    if (!rc && lazy) {
        rc = P->rows.alloc(d, T / 4, 2, T);
        P->rows.len = T;
        if (!rc) { HIP_TRY(hipMalloc(&P->d_T1, 16 * 256 * sizeof(Fr))); }
        P->d_F = d_tabs; d_tabs = nullptr;
        P->lazy = true;
        static const bool trace = getenv("ATLAS_TRACE") != nullptr;
        if (trace) fprintf(stderr, "[atlas trace] ra_virtual d=16 T=2^%zu: lazy rounds 0-1 (packed indices + tables), rows from round 2\n", log_T);
    } else if (!rc) {
        const bool have_idx = P->rows.d_idx != nullptr || P->rows.lk != nullptr;      // (the lazy attempt above may have uploaded them already)
        int32_t* keep_idx = P->rows.d_idx; const uint64_t* keep_lk = P->rows.lk; const uint32_t keep_log = P->rows.lk_log;
        rc = P->rows.alloc(d, T);
        P->rows.d_idx = keep_idx; P->rows.lk = keep_lk; P->rows.lk_log = keep_log;
        if (!rc && !have_idx) rc = H_indices ? P->rows.upload_indices(H_indices) : P->rows.upload_lookups(lookups, (uint32_t)log_k_chunk);
        if (!rc) rc = P->rows.gather(d_tabs, (uint32_t)K);
    }
    if (d_tabs) hipFree(d_tabs);
    if (!rc && log_T) rc = P->eq.init(reinterpret_cast<const H::Fr*>(r_cycle), log_T);
    if (rc) { delete P; return rc; }
    *out = P;
    return ATLAS_OK;
}

static int booleanity_build(const atlas_fr_t* G, const int32_t* const* H_indices, const uint64_t* lookups, size_t d, size_t log_k_chunk,
                            size_t log_T, const atlas_fr_t* gammas, const atlas_fr_t* r_address, const atlas_fr_t* r_cycle,
                            atlas_instance_t* out);

int atlas_booleanity_new(const atlas_fr_t* G, const int32_t* const* H_indices, size_t d, size_t log_k_chunk, size_t log_T,
                         const atlas_fr_t* gammas, const atlas_fr_t* r_address, const atlas_fr_t* r_cycle, atlas_instance_t* out) {
    NEED_INIT();
    if (!H_indices) return fail(ATLAS_EINVAL, "booleanity_new: null argument");
    return booleanity_build(G, H_indices, nullptr, d, log_k_chunk, log_T, gammas, r_address, r_cycle, out);
}

int atlas_booleanity_from_lookups_new(const atlas_fr_t* G, const uint64_t* lookup_indices, size_t log_T, size_t log_K, size_t log_k_chunk,
                                      const atlas_fr_t* gammas, const atlas_fr_t* r_address, const atlas_fr_t* r_cycle,
                                      atlas_instance_t* out) {
    PROF("atlas_booleanity_from_lookups_new");
    NEED_INIT();
    if (!lookup_indices || log_k_chunk == 0 || log_K == 0) return fail(ATLAS_EINVAL, "booleanity_from_lookups_new: null argument");
    return booleanity_build(G, nullptr, lookup_indices, (log_K + log_k_chunk - 1) / log_k_chunk, log_k_chunk, log_T, gammas, r_address, r_cycle, out);
}

static int booleanity_build(const atlas_fr_t* G, const int32_t* const* H_indices, const uint64_t* lookups, size_t d, size_t log_k_chunk,
                            size_t log_T, const atlas_fr_t* gammas, const atlas_fr_t* r_address, const atlas_fr_t* r_cycle,
                            atlas_instance_t* out) {
    if (!G || !gammas || !r_address || (!r_cycle && log_T) || !out) return fail(ATLAS_EINVAL, "booleanity_new: null argument");
    // log_T == 0: ONE cycle — the log_k address rounds only (host arithmetic), then H_i = F[idx_i] gathered as the final claims
    if (d == 0 || log_k_chunk == 0 || log_k_chunk > 16 || log_T > 25)
        return fail(ATLAS_EINVAL, "booleanity_new: 1 <= log_k_chunk <= 16, log_T <= 25");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    Booleanity* P = new Booleanity();
    P->d = d; P->log_k = log_k_chunk; P->log_T = log_T;
    const size_t K = (size_t)1 << log_k_chunk, T = (size_t)1 << log_T;
    const H::Fr* Gh = reinterpret_cast<const H::Fr*>(G);
    P->G.resize(d);
    for (size_t i = 0; i < d; i++) P->G[i].assign(Gh + i * K, Gh + (i + 1) * K);
    P->gammas.assign(reinterpret_cast<const H::Fr*>(gammas), reinterpret_cast<const H::Fr*>(gammas) + d);
    P->Ggam.assign(K, H::zero());
    for (size_t i = 0; i < d; i++)
        for (size_t k = 0; k < K; k++)
            if (!H::detail::is_zero4(P->G[i][k].l)) P->Ggam[k] = H::add(P->Ggam[k], H::mul(P->gammas[i], P->G[i][k]));
    P->F = {H::one()};
    const H::Fr* ra = reinterpret_cast<const H::Fr*>(r_address);
    P->B.init(ra, log_k_chunk);
    P->B_out = H::eq_cached(ra, P->B.k_out);
    P->B_in = H::eq_cached(ra + P->B.m, P->B.k_in);
    int rc = log_T ? P->D.init(reinterpret_cast<const H::Fr*>(r_cycle), log_T) : ATLAS_OK;
    static const size_t lazy_log = [] { const char* e = getenv("ATLAS_BOOL_LAZY_LOG"); const int v = e ? atoi(e) : 18; return (size_t)(v >= 15 && v <= 31 ? v : 18); }();   // 31 = never; from 2^15 on: no faster on the T = 2^16 nodes (the RaVirtual lane is the batch's long one), so it is used where the rows' memory counts
    static const bool no_pipe = getenv("ATLAS_NO_PIPELINE") != nullptr;
    bool lazy = !rc && !H_indices && log_k_chunk == 4 && d * 4 <= 64 && log_T >= lazy_log && rt().fs_mode == ATLAS_FS_HOST && !no_pipe;
    if (lazy) {
        P->rows.d = d; P->rows.len = T;
        rc = P->rows.upload_lookups(lookups, (uint32_t)log_k_chunk);
        if (rc || !P->rows.lk) lazy = false;              // host lookups are cut into index rows: the gathered path
    }
    if (!rc && lazy) {
        int32_t* keep_idx = P->rows.d_idx; const uint64_t* keep_lk = P->rows.lk; const uint32_t keep_log = P->rows.lk_log;
        rc = P->rows.alloc(d, T / 2, 2, T);              // T / 2 and T / 4 coefficients per row ...
        std::swap(P->rows.buf[0], P->rows.buf[1]);       // ... the longer in buf[1] (cycle round 1), the shorter in buf[0] (round 2)
        P->rows.d_idx = keep_idx; P->rows.lk = keep_lk; P->rows.lk_log = keep_log; P->rows.len = T;
        if (!rc) { hipError_t e = hipMalloc(&P->d_tabs, d * (16 + 256) * sizeof(Fr)); if (e != hipSuccess) rc = fail(ATLAS_ENODEV, "booleanity_new: tables", e); }
        P->lazy = !rc;
    } else if (!rc) {
        const bool have = P->rows.d_idx != nullptr || P->rows.lk != nullptr;
        int32_t* keep_idx = P->rows.d_idx; const uint64_t* keep_lk = P->rows.lk; const uint32_t keep_log = P->rows.lk_log;
        rc = P->rows.alloc(d, T);
        P->rows.d_idx = keep_idx; P->rows.lk = keep_lk; P->rows.lk_log = keep_log;
        if (!rc && !have) rc = H_indices ? P->rows.upload_indices(H_indices) : P->rows.upload_lookups(lookups, (uint32_t)log_k_chunk);   // resident until the phase-2 gather
    }
    if (!rc) {
        hipError_t e = hipMalloc(&P->d_gammas, d * sizeof(Fr));
        if (e != hipSuccess) rc = fail(ATLAS_ENODEV, "booleanity_new: gammas", e);
        else rc = store_small(reinterpret_cast<const H::Fr*>(gammas), d, P->d_gammas);
    }
    if (!rc && P->pipelined()) {          // the device copy of the expanding table of the round-channel path
        hipError_t e = hipMalloc(&P->d_F, K * sizeof(Fr));
        if (e != hipSuccess) rc = fail(ATLAS_ENODEV, "booleanity_new: expanding table", e);
    }
    if (rc) { delete P; return rc; }
    *out = P;
    return ATLAS_OK;
}

int atlas_hamming_weight_new(const atlas_fr_t* G, size_t d, size_t log_k_chunk, const atlas_fr_t* gamma_powers, atlas_instance_t* out) {
    PROF("atlas_hamming_weight_new");
    if (!G || !gamma_powers || !out || d == 0 || log_k_chunk > 20) return fail(ATLAS_EINVAL, "hamming_weight_new");
    HammingWeight* P = new HammingWeight();
    P->log_k = log_k_chunk;
    const size_t K = (size_t)1 << log_k_chunk;
    const H::Fr* Gh = reinterpret_cast<const H::Fr*>(G);
    P->ra.resize(d);
    for (size_t i = 0; i < d; i++) P->ra[i].assign(Gh + i * K, Gh + (i + 1) * K);
    P->gamma_powers.assign(reinterpret_cast<const H::Fr*>(gamma_powers), reinterpret_cast<const H::Fr*>(gamma_powers) + d);
    *out = P;
    return ATLAS_OK;
}

}  // extern "C"
