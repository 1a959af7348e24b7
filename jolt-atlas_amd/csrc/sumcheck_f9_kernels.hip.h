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
    acc0 = f9_wave_sum<P9>(acc0);
    acc2 = f9_wave_sum<P9>(acc2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red9[wave][0] = acc0; red9[wave][1] = acc2; }
    __syncthreads();
    if (threadIdx.x < 2) {
        F9 s = red9[0][threadIdx.x];
        for (int w = 1; w < SC_THREADS / 64; w++) s = f9_norm_red<P9>(f9_add(s, red9[w][threadIdx.x]));
        fe_store(partials + (size_t)blockIdx.x * 2 + threadIdx.x, f9_canon_x32<P9>(s));
    }
}

// round-0 message over untouched (canonical) operands
__global__ __launch_bounds__(SC_THREADS) void k_dot_eval2_f9(const Fr* __restrict__ L, const Fr* __restrict__ R,
                                                             size_t half, Fr* partials) {
    using P9 = Fr9Params;
    F9 acc0 = f9_zero(), acc2 = f9_zero();
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < half; i += (size_t)gridDim.x * SC_THREADS) {
        const F9 l0 = f9_load(L + i), l1 = f9_load(L + i + half);
        const F9 r0 = f9_load(R + i), r1 = f9_load(R + i + half);
        const F9 l2 = f9_norm(f9_add(l1, f9_sub<P9>(l1, l0))), r2 = f9_norm(f9_add(r1, f9_sub<P9>(r1, r0)));
        acc0 = f9_norm_red<P9>(f9_add(acc0, f9_mul<P9>(l0, r0)));
        acc2 = f9_norm_red<P9>(f9_add(acc2, f9_mul<P9>(l2, r2)));
    }
    f9_block_reduce_store2(acc0, acc2, partials);
}

// fused ingest_challenge(r_j) + compute_message(j+1); operands bound in place (thread i owns
// i, i+q, i+2q, i+3q).  The next iteration's eight coefficients are requested before the
// current ones are consumed, so HBM latency sits under ~700 multiply-adds.
template <bool CANON_OUT>
__global__ __launch_bounds__(SC_THREADS) void k_dot_bind_eval2_f9(Fr* L, Fr* R, size_t q, const ScCtx* cx,
                                                                  Fr* partials) {
    using P9 = Fr9Params;
    const F9 r32 = f9_shl5(f9_from_fe(fe_load(&cx->r)));     // 32 * r: limbs 0..3 stay zero
    F9 acc0 = f9_zero(), acc2 = f9_zero();
    size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * SC_THREADS;
    Fe x0, x1, x2, x3, y0, y1, y2, y3;
    if (i < q) {
        x0 = fe_load(L + i); x1 = fe_load(L + i + q); x2 = fe_load(L + i + 2 * q); x3 = fe_load(L + i + 3 * q);
        y0 = fe_load(R + i); y1 = fe_load(R + i + q); y2 = fe_load(R + i + 2 * q); y3 = fe_load(R + i + 3 * q);
    }
    for (; i < q; i += stride) {
        const F9 a0 = f9_from_fe(x0), a1 = f9_from_fe(x1), a2 = f9_from_fe(x2), a3 = f9_from_fe(x3);
        const F9 b0 = f9_from_fe(y0), b1 = f9_from_fe(y1), b2 = f9_from_fe(y2), b3 = f9_from_fe(y3);
        const size_t nx = i + stride;
        if (nx < q) {
            x0 = fe_load(L + nx); x1 = fe_load(L + nx + q); x2 = fe_load(L + nx + 2 * q); x3 = fe_load(L + nx + 3 * q);
            y0 = fe_load(R + nx); y1 = fe_load(R + nx + q); y2 = fe_load(R + nx + 2 * q); y3 = fe_load(R + nx + 3 * q);
        }
        // a + r*(b - a): the difference carries +4p and limbs < 2^31, the product is < 1.03p
        F9 l0 = f9_norm_red<P9>(f9_add(a0, f9_mul<P9, 4>(f9_sub<P9>(a2, a0), r32)));
        F9 l1 = f9_norm_red<P9>(f9_add(a1, f9_mul<P9, 4>(f9_sub<P9>(a3, a1), r32)));
        F9 r0 = f9_norm_red<P9>(f9_add(b0, f9_mul<P9, 4>(f9_sub<P9>(b2, b0), r32)));
        F9 r1 = f9_norm_red<P9>(f9_add(b1, f9_mul<P9, 4>(f9_sub<P9>(b3, b1), r32)));
        if constexpr (CANON_OUT) {
            fe_store(L + i, f9_canon<P9>(l0)); fe_store(L + i + q, f9_canon<P9>(l1));
            fe_store(R + i, f9_canon<P9>(r0)); fe_store(R + i + q, f9_canon<P9>(r1));
        } else {
            fe_store(L + i, f9_to_fe(l0)); fe_store(L + i + q, f9_to_fe(l1));
            fe_store(R + i, f9_to_fe(r0)); fe_store(R + i + q, f9_to_fe(r1));
        }
        const F9 l2 = f9_norm(f9_add(l1, f9_sub<P9>(l1, l0))), r2 = f9_norm(f9_add(r1, f9_sub<P9>(r1, r0)));
        acc0 = f9_norm_red<P9>(f9_add(acc0, f9_mul<P9>(l0, r0)));
        acc2 = f9_norm_red<P9>(f9_add(acc2, f9_mul<P9>(l2, r2)));
    }
    f9_block_reduce_store2(acc0, acc2, partials);
}

}  // namespace atlas
