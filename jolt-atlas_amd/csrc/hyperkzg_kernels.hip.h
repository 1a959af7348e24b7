// Kernels for HyperKZG::open (joltworks/src/poly/commitment/hyperkzg/mod.rs:400-447).
//
// The reference's open has three serial scans per polynomial — the Horner evaluations
// `UniPoly::eval_as_univariate` (poly/unipoly.rs:247-259) and the quotient recurrence
// `compute_witness_polynomial` (hyperkzg/mod.rs:213-229, h[i-1] = f[i] + u*h[i]).  Both are
// the same suffix recurrence  g(i) = f[i] + u*g(i+1);  exact field arithmetic makes any
// re-association value-identical, so they run as a three-level blocked suffix scan:
// thread (HK_CH coefficients, Horner) -> workgroup (weighted Hillis-Steele in LDS) -> grid
// (one small scan over the workgroup totals), for the three points u = (r, -r, r^2) at once
// so each coefficient is read from HBM once per pass.
#pragma once
#include "field.hip.h"

namespace atlas {

constexpr int HK_THREADS = 256;
constexpr int HK_CH = 16;                          // coefficients per thread
constexpr int HK_BLOCK = HK_THREADS * HK_CH;       // 4096 coefficients per workgroup
constexpr int HK_NP = 3;                           // evaluation points

struct HkPowers {          // per point: u^(2^k), k < 32 ; all Montgomery Fr
    Fr p2[HK_NP][32];
};

// Phase-1 fold of HyperKZG::open: Pi[j] = x*(p[2j+1]-p[2j]) + p[2j]  (hyperkzg/mod.rs:413-428)
__global__ __launch_bounds__(HK_THREADS) void k_hk_fold(const Fr* __restrict__ p, Fr* __restrict__ out, size_t half,
                                                        Fr x, int x_hi_only) {
    for (size_t j = (size_t)blockIdx.x * HK_THREADS + threadIdx.x; j < half; j += (size_t)gridDim.x * HK_THREADS) {
        Fr a = fe_load(p + 2 * j), b = fe_load(p + 2 * j + 1);
        Fr d = fr_sub(b, a);
        Fr m = x_hi_only ? fr_mul_hi(d, x) : fr_mul(d, x);
        fe_store(out + j, fr_add(m, a));
    }
}

// B[i] = sum_j q[j] * polys_j[i] over the polys that are long enough
// (DensePolynomial::linear_combination, dense_mlpoly.rs:444-499).  polys_j starts at
// offset off(j) = 2n - 2n/2^j of one contiguous buffer, length n >> j.
__global__ __launch_bounds__(HK_THREADS) void k_hk_lincomb(const Fr* __restrict__ polys, size_t n, uint32_t ell,
                                                           const Fr* __restrict__ q, Fr* __restrict__ B) {
    for (size_t i = (size_t)blockIdx.x * HK_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * HK_THREADS) {
        Fr acc = fe_zero();
        size_t off = 0, len = n;
        for (uint32_t j = 0; j < ell && i < len; j++) {
            acc = fr_add(acc, fr_mul(fe_load(polys + off + i), fe_load(q + j)));
            off += len;
            len >>= 1;
        }
        fe_store(B + i, acc);
    }
}

// u^e by the binary table
__device__ __forceinline__ Fr hk_pow(const HkPowers& P, int k, uint32_t e) {
    Fr acc = fr_one();
    for (int b = 0; e; b++, e >>= 1)
        if (e & 1u) acc = fr_mul(acc, P.p2[k][b]);
    return acc;
}

// Level 1+2: per-thread Horner totals and the weighted in-workgroup suffix scan.
//   xincl[k][blk*256 + t] = sum_{i >= s_t, i in workgroup} f[i] u_k^(i - s_t),  s_t = blk*4096 + 16 t
//   blocktot[k][blk]      = xincl[k][blk*256 + 0]
// xincl may be null (evaluation only).
__global__ __launch_bounds__(HK_THREADS) void k_hk_scan_blocks(const Fr* __restrict__ f, size_t len, HkPowers P,
                                                               Fr* __restrict__ xincl, size_t xincl_stride,
                                                               Fr* __restrict__ blocktot, size_t n_blocks) {
    __shared__ Fr sm[HK_NP][HK_THREADS];
    const uint32_t t = threadIdx.x;
    const size_t s = (size_t)blockIdx.x * HK_BLOCK + (size_t)t * HK_CH;
    Fr tot[HK_NP];
#pragma unroll
    for (int k = 0; k < HK_NP; k++) tot[k] = fe_zero();
    for (int i = HK_CH - 1; i >= 0; i--) {
        const size_t idx = s + i;
        const Fr c = idx < len ? fe_load(f + idx) : fe_zero();
#pragma unroll
        for (int k = 0; k < HK_NP; k++) tot[k] = fr_add(fr_mul(tot[k], P.p2[k][0]), c);
    }
#pragma unroll
    for (int k = 0; k < HK_NP; k++) sm[k][t] = tot[k];
    __syncthreads();
    // x[t] += u^(16 d) * x[t + d],  d = 1, 2, 4, ..., 128   (u^16 = p2[4], u^(16 d) = p2[4 + log2 d])
    for (int lg = 0; lg < 8; lg++) {
        const uint32_t d = 1u << lg;
        Fr add[HK_NP];
        const bool on = t + d < HK_THREADS;
#pragma unroll
        for (int k = 0; k < HK_NP; k++) add[k] = on ? fr_mul(sm[k][t + d], P.p2[k][4 + lg]) : fe_zero();
        __syncthreads();
#pragma unroll
        for (int k = 0; k < HK_NP; k++) sm[k][t] = fr_add(sm[k][t], add[k]);
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < HK_NP; k++) {
        if (xincl) fe_store(xincl + (size_t)k * xincl_stride + (size_t)blockIdx.x * HK_THREADS + t, sm[k][t]);
        if (t == 0) fe_store(blocktot + (size_t)k * n_blocks + blockIdx.x, sm[k][0]);
    }
}

// Phase 3a of open: v[k][j] = Pi_j(u_k) for ALL the folded polynomials in two launches.  Pi_j starts at
// 2n - 2n/2^j of `polys` and has n >> j coefficients; workgroup b of the grid owns 4096 coefficients of
// one Pi_j (short polynomials take one workgroup each).  part[k][b] = u_k^(4096 b_local) * (the
// workgroup's Horner total), so that Pi_j(u_k) is the plain sum of its workgroups' entries.
__device__ __forceinline__ void hk_locate(size_t n, size_t& b, size_t& off, size_t& len, uint32_t& j) {
    off = 0; len = n; j = 0;
    for (;; j++) {
        const size_t nb = (len + HK_BLOCK - 1) / HK_BLOCK;
        if (b < nb) return;
        b -= nb; off += len; len >>= 1;
    }
}

__global__ __launch_bounds__(HK_THREADS) void k_hk_eval_blocks(const Fr* __restrict__ polys, size_t n, HkPowers P,
                                                               Fr* __restrict__ part, size_t total_blocks) {
    __shared__ Fr sm[HK_NP][HK_THREADS];
    const uint32_t t = threadIdx.x;
    size_t b = blockIdx.x, off, len; uint32_t j;
    hk_locate(n, b, off, len, j);
    const Fr* f = polys + off;
    const size_t s = b * HK_BLOCK + (size_t)t * HK_CH;
    Fr tot[HK_NP];
#pragma unroll
    for (int k = 0; k < HK_NP; k++) tot[k] = fe_zero();
    for (int i = HK_CH - 1; i >= 0; i--) {
        const size_t idx = s + i;
        const Fr c = idx < len ? fe_load(f + idx) : fe_zero();
#pragma unroll
        for (int k = 0; k < HK_NP; k++) tot[k] = fr_add(fr_mul(tot[k], P.p2[k][0]), c);
    }
#pragma unroll
    for (int k = 0; k < HK_NP; k++) sm[k][t] = tot[k];
    __syncthreads();
    // weighted tree: x[t] += u^(16 d) x[t + d] for t a multiple of 2d  (only x[0] is needed)
    for (int lg = 0; lg < 8; lg++) {
        const uint32_t d = 1u << lg;
        if ((t & (2 * d - 1)) == 0) {
#pragma unroll
            for (int k = 0; k < HK_NP; k++) sm[k][t] = fr_add(sm[k][t], fr_mul(sm[k][t + d], P.p2[k][4 + lg]));
        }
        __syncthreads();
    }
    if (t < HK_NP) fe_store(part + (size_t)t * total_blocks + blockIdx.x, fr_mul(sm[t][0], hk_pow(P, (int)t, (uint32_t)(b * HK_BLOCK))));
}

// one workgroup per polynomial: out[k * ell + j] = sum of part[k][.] over Pi_j's workgroups
__global__ __launch_bounds__(HK_THREADS) void k_hk_eval_sum(const Fr* __restrict__ part, size_t total_blocks, size_t n,
                                                            uint32_t ell, Fr* __restrict__ out) {
    __shared__ Fr sm[HK_NP][HK_THREADS];
    const uint32_t t = threadIdx.x, j = blockIdx.x;
    size_t start = 0, len = n;
    for (uint32_t q = 0; q < j; q++) { start += (len + HK_BLOCK - 1) / HK_BLOCK; len >>= 1; }
    const size_t nb = (len + HK_BLOCK - 1) / HK_BLOCK;
#pragma unroll
    for (int k = 0; k < HK_NP; k++) {
        Fr acc = fe_zero();
        for (size_t b = t; b < nb; b += HK_THREADS) acc = fr_add(acc, fe_load(part + (size_t)k * total_blocks + start + b));
        sm[k][t] = acc;
    }
    __syncthreads();
    for (uint32_t d = HK_THREADS / 2; d >= 1; d >>= 1) {
        if (t < d) {
#pragma unroll
            for (int k = 0; k < HK_NP; k++) sm[k][t] = fr_add(sm[k][t], sm[k][t + d]);
        }
        __syncthreads();
    }
    if (t < HK_NP) fe_store(out + (size_t)t * ell + j, sm[t][0]);
}

// Level 3: suffix scan over workgroup totals (one workgroup, serial segments per thread).
//   G[k][b] = g_k(end of workgroup b) = sum_{b' > b} blocktot[k][b'] u_k^(4096 (b' - b - 1))
//   total[k] = g_k(0) = blocktot[k][0] + u_k^4096 * G[k][0]            (the evaluation f(u_k))
__global__ __launch_bounds__(HK_THREADS) void k_hk_scan_grid(const Fr* __restrict__ blocktot, size_t n_blocks,
                                                             HkPowers P, Fr* __restrict__ G, Fr* __restrict__ total) {
    __shared__ Fr sm[HK_NP][HK_THREADS];
    const uint32_t t = threadIdx.x;
    const size_t per = (n_blocks + HK_THREADS - 1) / HK_THREADS;
    const size_t lo = (size_t)t * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    // segment total with weights relative to the segment start: sum_{b in seg} tot[b] w^(b - lo), w = u^4096
    Fr seg[HK_NP];
#pragma unroll
    for (int k = 0; k < HK_NP; k++) seg[k] = fe_zero();
    for (size_t b = hi; b-- > lo;)
#pragma unroll
        for (int k = 0; k < HK_NP; k++)
            seg[k] = fr_add(fr_mul(seg[k], P.p2[k][12]), fe_load(blocktot + (size_t)k * n_blocks + b));
#pragma unroll
    for (int k = 0; k < HK_NP; k++) sm[k][t] = (lo < n_blocks) ? seg[k] : fe_zero();
    __syncthreads();
    // inclusive weighted suffix scan over segments: weight between neighbours = w^per
    Fr wper[HK_NP];
    // w^per = u^(4096*per): compute by repeated squaring of the table entry
#pragma unroll
    for (int k = 0; k < HK_NP; k++) {
        Fr acc = fr_one();
        uint64_t e = (uint64_t)per;          // exponent of w = u^(2^12)
        for (int b = 0; e; b++, e >>= 1)
            if (e & 1u) acc = fr_mul(acc, P.p2[k][12 + b]);
        wper[k] = acc;
    }
    Fr wd[HK_NP];
#pragma unroll
    for (int k = 0; k < HK_NP; k++) wd[k] = wper[k];
    for (uint32_t d = 1; d < HK_THREADS; d <<= 1) {
        Fr add[HK_NP];
        const bool on = t + d < HK_THREADS;
#pragma unroll
        for (int k = 0; k < HK_NP; k++) add[k] = on ? fr_mul(sm[k][t + d], wd[k]) : fe_zero();
        __syncthreads();
#pragma unroll
        for (int k = 0; k < HK_NP; k++) { sm[k][t] = fr_add(sm[k][t], add[k]); wd[k] = fr_mul(wd[k], wd[k]); }
        __syncthreads();
    }
    // sm[k][t] = sum_{b >= lo_t} tot[b] w^(b - lo_t).  Walk the segment from its end:
    // carry = g at the end of the segment's last workgroup = sm[k][t+1] (next segment's inclusive sum)
    Fr carry[HK_NP];
#pragma unroll
    for (int k = 0; k < HK_NP; k++) carry[k] = (t + 1 < HK_THREADS) ? sm[k][t + 1] : fe_zero();
    for (size_t b = hi; b-- > lo;) {
#pragma unroll
        for (int k = 0; k < HK_NP; k++) {
            fe_store(G + (size_t)k * n_blocks + b, carry[k]);
            carry[k] = fr_add(fr_mul(carry[k], P.p2[k][12]), fe_load(blocktot + (size_t)k * n_blocks + b));
        }
    }
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < HK_NP; k++) fe_store(total + k, carry[k]);
    }
}

// compute_witness_polynomial for the three points: h_k[i-1] = g_k(i), h_k[n-1] = 0.
// pw16[k][j] = u_k^(16 j), j <= 256.
__global__ __launch_bounds__(HK_THREADS) void k_hk_witness(const Fr* __restrict__ f, size_t len, HkPowers P,
                                                           const Fr* __restrict__ xincl, size_t xincl_stride,
                                                           const Fr* __restrict__ G, size_t n_blocks,
                                                           const Fr* __restrict__ pw16, Fr* __restrict__ h,
                                                           size_t h_stride) {
    const uint32_t t = threadIdx.x;
    const size_t blk = blockIdx.x;
    const size_t s = blk * HK_BLOCK + (size_t)t * HK_CH;
    if (s >= len) return;
    Fr g[HK_NP];     // g_k(s + 16)
#pragma unroll
    for (int k = 0; k < HK_NP; k++) {
        Fr inblk = (t + 1 < HK_THREADS) ? fe_load(xincl + (size_t)k * xincl_stride + blk * HK_THREADS + t + 1) : fe_zero();
        Fr far = fr_mul(fe_load(G + (size_t)k * n_blocks + blk), fe_load(pw16 + (size_t)k * 257 + (HK_THREADS - 1 - t)));
        g[k] = fr_add(inblk, far);
    }
    for (int i = HK_CH - 1; i >= 0; i--) {
        const size_t idx = s + i;
        if (idx >= len) continue;                 // beyond the polynomial: g stays 0 there
        const Fr c = fe_load(f + idx);
#pragma unroll
        for (int k = 0; k < HK_NP; k++) {
            if (idx == len - 1) fe_store(h + (size_t)k * h_stride + idx, fe_zero());
            g[k] = fr_add(fr_mul(g[k], P.p2[k][0]), c);
            if (idx >= 1) fe_store(h + (size_t)k * h_stride + idx - 1, g[k]);
        }
    }
}

}  // namespace atlas
