// Fixed-base Pippenger: the SRS is fixed for the life of a prover key (KZGProverKey::g1_powers,
// joltworks/src/poly/commitment/hyperkzg/kzg.rs:107-143), so its multiples 2^(c j) G_i are computed once and kept in HBM
// next to it (n * ceil(255 / c) * 64 B: 3.5 GB for 2^22 points at c = 20, out of 288 GB).  With the multiples at hand
// digit j of scalar i is just another point, T_j[i] = 2^(c j) G_i, and ALL digits of a scalar vector fall into ONE set of
// 2^(c-1) buckets: the per-window bucket folds disappear, so the digit width can grow from 13 bits (20 digits per scalar)
// to 20 (13 digits) — 35 % fewer point additions, which is where an MSM spends its time (the bucket accumulation runs
// at 0.8 of the chip's integer-multiplier rate; bench.py `msm.roofline`).
//
// A narrower digit c' = c / q still reads the same table: digit m sits at bit c' m = c (m / q) + c' (m % q), so it
// takes level m / q of the table and bucket set m % q; the q sets are combined by Horner on the host as before.
//
// The sort of (scalar, digit) pairs by bucket has up to 2^21 keys per vector and is done in two passes:
//   1. partition by (vector, set, high bucket bits) — "bins" of 2^lo_bits buckets; the digits are recomputed from
//      the scalars in each kernel (32 B read per scalar beats 8 B per digit written and read back);
//   2. counting sort of a bin's entries by the low bucket bits in LDS, exactly like the per-window sort of the
//      variable-base path (msm_kernels.hip.h), over fixed chunks of the partitioned array.
// From the sorted entries on (segment counts, accumulation, bucket reduction, fold) the kernels are the variable-base ones.
#pragma once
#include "msm_kernels.hip.h"

namespace atlas {

struct TabShape {
    uint32_t c;             // digit width c' (<= 24)
    uint32_t n_digits;      // ceil(255 / c')
    uint32_t q;             // bucket sets per vector
    uint32_t lo_bits;       // bucket bits sorted in LDS (pass 2)
    uint32_t hi_bits;       // c' - 1 - lo_bits, resolved by the partition (pass 1)
    uint32_t level_stride;  // points per table level
};

constexpr uint32_t TAB_LO_BITS = 12;
constexpr uint32_t TAB_MAX_BINS = 4096;      // bins of one vector (q << hi_bits), counted in LDS
constexpr uint32_t TAB_TILE = 4096;          // scalars per workgroup in pass 1
constexpr uint32_t TAB_CHUNK = 8192;         // entries per workgroup in pass 2 (measured: 32768 is slower, 1.5 ms against 0.93 ms at 2^22)
constexpr int TAB_UNROLL = 8;                // entries a thread loads before it touches the LDS counters
constexpr uint32_t TAB_TINY = 256;           // a bin segment this short skips the LDS counters

struct TabTile {
    uint32_t begin, end;    // scalars [begin, end) of the concatenated scalar array: one vector
    uint32_t bin_base;      // first bin of that vector
    uint32_t pt_base;       // table index of scalar i at level 0 = i + pt_base (mod 2^32)
};

// F(digit index m, magnitude - 1, negative)
template <class F>
__device__ __forceinline__ void tab_for_each_digit(const Fr& k, const TabShape S, F&& f) {
    uint32_t carry = 0;
    const uint32_t half = 1u << (S.c - 1);
    for (uint32_t m = 0; m < S.n_digits; m++) {
        const uint32_t lo = m * S.c;
        uint32_t d = (lo < 256 ? bits_at(k.v, lo, S.c) : 0u) + carry;
        carry = 0;
        if (d >= half) { d = (1u << S.c) - d; carry = 1; if (d) f(m, d - 1, true); }
        else if (d) f(m, d - 1, false);
    }
}

// table level l+1 from level l: c doublings, back to affine with one inversion per TAB_INV_BATCH points
constexpr int TAB_INV_BATCH = 4;
__global__ __launch_bounds__(MSM_THREADS) void k_tab_next_level(const G1Affine* __restrict__ prev, G1Affine* __restrict__ next,
                                                                size_t n, uint32_t c) {
    const size_t stride = (size_t)gridDim.x * MSM_THREADS;
    for (size_t i0 = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; i0 < n; i0 += stride * TAB_INV_BATCH) {
        G1Xyzz p[TAB_INV_BATCH];
        Fq pre[TAB_INV_BATCH];              // prefix products of the zzz of the finite points
        Fq run = fq_one();
        for (int b = 0; b < TAB_INV_BATCH; b++) {
            const size_t i = i0 + (size_t)b * stride;
            p[b] = g1_inf();
            if (i < n) {
                const G1Affine a = g1_aff_load(prev + i);
                if (!g1_aff_is_inf(a)) {
                    p[b] = g1_dbl_aff(a);
                    for (uint32_t d = 1; d < c; d++) p[b] = g1_dbl(p[b]);
                }
            }
            pre[b] = run;
            if (!g1_is_inf(p[b])) run = fq_mul(run, p[b].zzz);
        }
        Fq inv = fq_inv(run);
#pragma unroll
        for (int b = TAB_INV_BATCH - 1; b >= 0; b--) {
            const size_t i = i0 + (size_t)b * stride;
            if (i >= n) continue;
            G1Affine o;
            if (g1_is_inf(p[b])) { o.x = fe_zero(); o.y = fe_zero(); }
            else {
                const Fq izzz = fq_mul(inv, pre[b]);
                inv = fq_mul(inv, p[b].zzz);
                const Fq izz = fq_sqr(fq_mul(izzz, p[b].zz));     // zz / zzz = 1 / z
                o.x = fq_mul(p[b].x, izz);
                o.y = fq_mul(p[b].y, izzz);
            }
            fe_store(&next[i].x, o.x);
            fe_store(&next[i].y, o.y);
        }
    }
}

// ---- pass 1: partition by bin ------------------------------------------------------------------------------------
__global__ __launch_bounds__(MSM_THREADS) void k_tab_part_hist(const Fr* __restrict__ s, const TabTile* __restrict__ tiles, TabShape S,
                                                               uint32_t* bin_counts) {
    __shared__ uint32_t h[TAB_MAX_BINS];
    const TabTile tl = tiles[blockIdx.x];
    const uint32_t nb = S.q << S.hi_bits;
    for (uint32_t b = threadIdx.x; b < nb; b += MSM_THREADS) h[b] = 0;
    __syncthreads();
    for (uint32_t i = tl.begin + threadIdx.x; i < tl.end; i += MSM_THREADS) {
        const Fr k = fe_from_mont<FrParams>(fe_load(s + i));
        uint32_t set = 0;                 // m % q, advanced with m (tab_for_each_digit visits m in order but skips zeros)
        uint32_t m_seen = 0;
        tab_for_each_digit(k, S, [&](uint32_t m, uint32_t mag1, bool) {
            set += m - m_seen; m_seen = m;
            while (set >= S.q) set -= S.q;
            atomicAdd(&h[(set << S.hi_bits) | (mag1 >> S.lo_bits)], 1u);
        });
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += MSM_THREADS)
        if (h[b]) atomicAdd(&bin_counts[tl.bin_base + b], h[b]);
}

// entry = {table index | sign << 31, low bucket bits}
__global__ __launch_bounds__(MSM_THREADS) void k_tab_part_scatter(const Fr* __restrict__ s, const TabTile* __restrict__ tiles, TabShape S,
                                                                  uint32_t* bin_cursor, uint2* __restrict__ ent) {
    __shared__ uint32_t cnt[TAB_MAX_BINS];
    __shared__ uint32_t base[TAB_MAX_BINS];
    const TabTile tl = tiles[blockIdx.x];
    const uint32_t nb = S.q << S.hi_bits;
    const uint32_t lo_mask = (1u << S.lo_bits) - 1u;
    for (uint32_t b = threadIdx.x; b < nb; b += MSM_THREADS) cnt[b] = 0;
    __syncthreads();
    for (uint32_t i = tl.begin + threadIdx.x; i < tl.end; i += MSM_THREADS) {
        const Fr k = fe_from_mont<FrParams>(fe_load(s + i));
        uint32_t set = 0, m_seen = 0;
        tab_for_each_digit(k, S, [&](uint32_t m, uint32_t mag1, bool) {
            set += m - m_seen; m_seen = m;
            while (set >= S.q) set -= S.q;
            atomicAdd(&cnt[(set << S.hi_bits) | (mag1 >> S.lo_bits)], 1u);
        });
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += MSM_THREADS) {
        const uint32_t c = cnt[b];
        base[b] = c ? atomicAdd(&bin_cursor[tl.bin_base + b], c) : 0u;
        cnt[b] = 0;
    }
    __syncthreads();
    for (uint32_t i = tl.begin + threadIdx.x; i < tl.end; i += MSM_THREADS) {
        const Fr k = fe_from_mont<FrParams>(fe_load(s + i));
        uint32_t set = 0, level = 0, m_seen = 0;
        tab_for_each_digit(k, S, [&](uint32_t m, uint32_t mag1, bool neg) {
            set += m - m_seen; m_seen = m;
            while (set >= S.q) { set -= S.q; level++; }
            const uint32_t b = (set << S.hi_bits) | (mag1 >> S.lo_bits);
            const uint32_t r = atomicAdd(&cnt[b], 1u);
            ent[base[b] + r] = make_uint2((level * S.level_stride + (i + tl.pt_base)) | (neg ? 0x80000000u : 0u), mag1 & lo_mask);
        });
    }
}

// ---- pass 2: counting sort inside the bins, over fixed chunks of the partitioned array -----------------------------
// largest b with bin_off[b] <= pos (bin_off ascending, empty bins repeat a value: the last one is the bin holding pos)
__device__ __forceinline__ uint32_t tab_find_bin(const uint32_t* __restrict__ bin_off, uint32_t n_bins, uint32_t pos) {
    uint32_t lo = 0, hi = n_bins;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (bin_off[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

// Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8); each XCD has its own 4 MB L2.  A chunk's scatter writes
// 4-byte entries in ~8-byte runs all over its bin's output region (1.7 MB at 2^22), so the L2 must collect the runs of
// many chunks of the SAME bin before a line leaves for HBM.  Chunk index = xcd * (chunks / 8) + k keeps every XCD on its
// own contiguous eighth of the partitioned array (~3 bins in flight per XCD instead of ~25 spread over all of them).
__device__ __forceinline__ uint32_t tab_chunk_of_block() {
    const uint32_t per = (gridDim.x + 7) / 8;
    return (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
}

__global__ __launch_bounds__(MSM_THREADS) void k_tab_bin_hist(const uint2* __restrict__ ent, const uint32_t* __restrict__ bin_off,
                                                              uint32_t n_bins, uint32_t lo_bits, uint32_t* counts) {
    __shared__ uint32_t h[1u << TAB_LO_BITS];
    const uint32_t total = bin_off[n_bins];
    const uint64_t start64 = (uint64_t)tab_chunk_of_block() * TAB_CHUNK;
    if (start64 >= total) return;
    const uint32_t start = (uint32_t)start64;
    const uint32_t end = start + TAB_CHUNK < total ? start + TAB_CHUNK : total;
    const uint32_t nlo = 1u << lo_bits;
    uint32_t b = tab_find_bin(bin_off, n_bins, start);
    for (uint32_t pos = start; pos < end;) {
        while (bin_off[b + 1] <= pos) b++;
        const uint32_t seg_end = end < bin_off[b + 1] ? end : bin_off[b + 1];
        uint32_t* cb = counts + (size_t)b * nlo;
        if (seg_end - pos <= TAB_TINY) {
            for (uint32_t i = pos + threadIdx.x; i < seg_end; i += MSM_THREADS) atomicAdd(&cb[ent[i].y], 1u);
        } else {
            for (uint32_t l = threadIdx.x; l < nlo; l += MSM_THREADS) h[l] = 0;
            __syncthreads();
            for (uint32_t i0 = pos + threadIdx.x; i0 < seg_end; i0 += MSM_THREADS * TAB_UNROLL) {
                uint32_t y[TAB_UNROLL];
#pragma unroll
                for (int u = 0; u < TAB_UNROLL; u++) { const uint32_t i = i0 + u * MSM_THREADS; y[u] = i < seg_end ? ent[i].y : 0xffffffffu; }
#pragma unroll
                for (int u = 0; u < TAB_UNROLL; u++) if (y[u] != 0xffffffffu) atomicAdd(&h[y[u]], 1u);
            }
            __syncthreads();
            for (uint32_t l = threadIdx.x; l < nlo; l += MSM_THREADS)
                if (h[l]) atomicAdd(&cb[l], h[l]);
            __syncthreads();
        }
        pos = seg_end;
    }
}

__global__ __launch_bounds__(MSM_THREADS) void k_tab_bin_scatter(const uint2* __restrict__ ent, const uint32_t* __restrict__ bin_off,
                                                                 uint32_t n_bins, uint32_t lo_bits, uint32_t* cursor,
                                                                 uint32_t* __restrict__ sorted) {
    __shared__ uint32_t cnt[1u << TAB_LO_BITS];
    __shared__ uint32_t base[1u << TAB_LO_BITS];
    const uint32_t total = bin_off[n_bins];
    const uint64_t start64 = (uint64_t)tab_chunk_of_block() * TAB_CHUNK;
    if (start64 >= total) return;
    const uint32_t start = (uint32_t)start64;
    const uint32_t end = start + TAB_CHUNK < total ? start + TAB_CHUNK : total;
    const uint32_t nlo = 1u << lo_bits;
    uint32_t b = tab_find_bin(bin_off, n_bins, start);
    for (uint32_t pos = start; pos < end;) {
        while (bin_off[b + 1] <= pos) b++;
        const uint32_t seg_end = end < bin_off[b + 1] ? end : bin_off[b + 1];
        uint32_t* cb = cursor + (size_t)b * nlo;
        if (seg_end - pos <= TAB_TINY) {
            for (uint32_t i = pos + threadIdx.x; i < seg_end; i += MSM_THREADS) {
                const uint2 e = ent[i];
                sorted[atomicAdd(&cb[e.y], 1u)] = e.x;
            }
        } else {
            for (uint32_t l = threadIdx.x; l < nlo; l += MSM_THREADS) cnt[l] = 0;
            __syncthreads();
            for (uint32_t i0 = pos + threadIdx.x; i0 < seg_end; i0 += MSM_THREADS * TAB_UNROLL) {
                uint32_t y[TAB_UNROLL];
#pragma unroll
                for (int u = 0; u < TAB_UNROLL; u++) { const uint32_t i = i0 + u * MSM_THREADS; y[u] = i < seg_end ? ent[i].y : 0xffffffffu; }
#pragma unroll
                for (int u = 0; u < TAB_UNROLL; u++) if (y[u] != 0xffffffffu) atomicAdd(&cnt[y[u]], 1u);
            }
            __syncthreads();
            for (uint32_t l = threadIdx.x; l < nlo; l += MSM_THREADS) {
                const uint32_t c = cnt[l];
                base[l] = c ? atomicAdd(&cb[l], c) : 0u;
                cnt[l] = 0;
            }
            __syncthreads();
            for (uint32_t i0 = pos + threadIdx.x; i0 < seg_end; i0 += MSM_THREADS * TAB_UNROLL) {
                uint2 e[TAB_UNROLL];
#pragma unroll
                for (int u = 0; u < TAB_UNROLL; u++) { const uint32_t i = i0 + u * MSM_THREADS; e[u] = i < seg_end ? ent[i] : make_uint2(0u, 0xffffffffu); }
#pragma unroll
                for (int u = 0; u < TAB_UNROLL; u++)
                    if (e[u].y != 0xffffffffu) sorted[base[e[u].y] + atomicAdd(&cnt[e[u].y], 1u)] = e[u].x;
            }
            __syncthreads();
        }
        pos = seg_end;
    }
}

}  // namespace atlas
