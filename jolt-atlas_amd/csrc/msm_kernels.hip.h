// Pippenger (bucket method) multi-scalar multiplication on gfx950.
//
// Device replacement for the arkworks call behind VariableBaseMSM::msm
// (joltworks/src/msm/mod.rs:27-181, called from UnivariateKZG::commit_* kzg.rs:195-298 and
// HyperKZG::open hyperkzg/mod.rs:400-447), and for jolt_optimizations::batch_g1_additions_multi
// behind HyperKZG::commit_one_hot (hyperkzg/mod.rs:520-554).
//
// Layout: bases are 64-byte affine points (x, y Montgomery Fq) resident in HBM (the SRS is
// uploaded once and reused across proofs); scalars are the 32-byte Montgomery Fr image.
// Signed c-bit windows; every (scalar, window) digit becomes one 4-byte entry
// (index | sign<<31) that a counting sort (histogram -> exclusive scan -> scatter) groups
// by bucket; one thread per bucket then streams its entries with mixed XYZZ additions;
// buckets are folded by the running-sum method in chunks; the last O(windows) doublings and
// the one inversion to affine run on the host.
#pragma once
#include <type_traits>
#include "curve.hip.h"
#include "curve_f9.hip.h"
#include "scan.hip.h"

namespace atlas {

constexpr int MSM_THREADS = 256;
constexpr int MSM_CHUNK = 4;           // buckets folded by one thread in the reduce step (same-box A/B: 4 is 1-3 % ahead of 8, 2 is worse)

struct MsmShape {
    uint32_t c;          // window bits (<= 16)
    uint32_t n_windows;  // ceil(255 / c)
    uint32_t bpw;        // buckets per window = 2^(c-1)
};

// Montgomery Fr -> canonical integer limbs
__global__ __launch_bounds__(MSM_THREADS) void k_msm_canon(const Fr* __restrict__ s, Fr* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * MSM_THREADS)
        fe_store(out + i, fe_from_mont<FrParams>(fe_load(s + i)));
}

// c-bit field of a 256-bit little-endian integer starting at bit `lo`
__device__ __forceinline__ uint32_t bits_at(const uint32_t k[8], uint32_t lo, uint32_t c) {
    const uint32_t w = lo >> 5, sh = lo & 31;
    uint64_t v = k[w];
    if (w + 1 < 8) v |= (uint64_t)k[w + 1] << 32;
    return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}

// Walk the signed digits of one scalar; F(window, bucket_in_window, negative)
template <class F>
__device__ __forceinline__ void for_each_digit(const Fr& k, const MsmShape S, F&& f) {
    uint32_t carry = 0;
    const uint32_t half = 1u << (S.c - 1);
    for (uint32_t w = 0; w < S.n_windows; w++) {
        const uint32_t lo = w * S.c;
        uint32_t d = (lo < 256 ? bits_at(k.v, lo, S.c) : 0u) + carry;
        carry = 0;
        // d >= half goes negative: positive digits <= half-1, negative magnitudes <= half (fits int16 at c = 16)
        if (d >= half) { d = (1u << S.c) - d; carry = 1; if (d) f(w, d - 1, true); }
        else if (d) f(w, d - 1, false);
    }
}

__global__ __launch_bounds__(MSM_THREADS) void k_msm_hist(const Fr* __restrict__ canon, size_t n, MsmShape S,
                                                          uint32_t* counts) {
    for (size_t i = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * MSM_THREADS) {
        const Fr k = fe_load(canon + i);
        for_each_digit(k, S, [&](uint32_t w, uint32_t b, bool) { atomicAdd(&counts[w * S.bpw + b], 1u); });
    }
}

// (k_exclusive_scan lives in scan.hip.h)

// ---- window-major digit pipeline -------------------------------------------------------
// Signed digits are computed once (Montgomery -> canonical -> 16 x int16 per scalar) and stored
// window-major (digits[w*n + i]), so the histogram / scatter of ONE window touches n*2 bytes
// of input and a bucket range of 2^(c-1) counters + ~n entries of output: the random part of
// the counting sort stays inside the 32 MiB of L2 instead of spraying the whole N*W*4 B array.
__global__ __launch_bounds__(MSM_THREADS) void k_msm_digits(const Fr* __restrict__ s, size_t n, MsmShape S,
                                                            int16_t* __restrict__ digits) {
    for (size_t i = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * MSM_THREADS) {
        const Fr k = fe_from_mont<FrParams>(fe_load(s + i));
        for (uint32_t w = 0; w < S.n_windows; w++) digits[(size_t)w * n + i] = 0;
        for_each_digit(k, S, [&](uint32_t w, uint32_t b, bool neg) {
            digits[(size_t)w * n + i] = neg ? (int16_t)(-(int32_t)(b + 1)) : (int16_t)(b + 1);
        });
    }
}

// blockIdx.y = window
__global__ __launch_bounds__(MSM_THREADS) void k_msm_hist_w(const int16_t* __restrict__ digits, size_t n, MsmShape S,
                                                            uint32_t* counts) {
    const uint32_t w = blockIdx.y;
    const int16_t* dw = digits + (size_t)w * n;
    uint32_t* cw = counts + (size_t)w * S.bpw;
    for (size_t i = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * MSM_THREADS) {
        const int32_t d = dw[i];
        if (d) atomicAdd(&cw[(d < 0 ? -d : d) - 1], 1u);
    }
}

// ---- few-bucket plans (narrow scalars, bpw <= 4096): the global counters of a window would be hit
// by every element, so counts and ranks are taken in LDS per workgroup tile and only one global
// atomic per (workgroup, non-empty bucket) remains.
constexpr uint32_t MSM_LDS_BPW = 8192;
constexpr uint32_t MSM_TILE = 8192;       // elements per workgroup tile (measured 4096 .. 262144: 8192 is 3-4 % ahead of 16384, larger is worse)

// a tile = at most MSM_TILE consecutive scalars of ONE scalar vector; several vectors over the same bases
// (the shrinking polynomials of HyperKZG::open) run through one pipeline as extra "virtual windows":
// vector k, window w owns buckets [bucket_base + w * bpw, +bpw)
struct MsmTile {
    uint32_t begin, end;        // range in the concatenated scalar index space
    uint32_t bucket_base;       // k * n_windows * bpw
    uint32_t local_off;         // offset of vector k: sorted entries carry (i - local_off), the base index
};

__global__ __launch_bounds__(MSM_THREADS) void k_msm_hist_lds(const int16_t* __restrict__ digits, size_t n_tot,
                                                              const MsmTile* __restrict__ tiles, MsmShape S, uint32_t* counts) {
    __shared__ uint32_t h[MSM_LDS_BPW];
    const uint32_t w = blockIdx.y;
    const MsmTile tl = tiles[blockIdx.x];
    for (uint32_t b = threadIdx.x; b < S.bpw; b += MSM_THREADS) h[b] = 0;
    __syncthreads();
    const int16_t* dw = digits + (size_t)w * n_tot;
    for (uint32_t i = tl.begin + threadIdx.x; i < tl.end; i += MSM_THREADS) {
        const int32_t d = dw[i];
        if (d) atomicAdd(&h[(d < 0 ? -d : d) - 1], 1u);
    }
    __syncthreads();
    uint32_t* cw = counts + tl.bucket_base + (size_t)w * S.bpw;
    for (uint32_t b = threadIdx.x; b < S.bpw; b += MSM_THREADS)
        if (h[b]) atomicAdd(&cw[b], h[b]);
}

__global__ __launch_bounds__(MSM_THREADS) void k_msm_scatter_lds(const int16_t* __restrict__ digits, size_t n_tot,
                                                                 const MsmTile* __restrict__ tiles, uint32_t bpw,
                                                                 uint32_t* cursor, uint32_t* __restrict__ sorted) {
    __shared__ uint32_t cnt[MSM_LDS_BPW];
    __shared__ uint32_t base[MSM_LDS_BPW];
    const MsmTile tl = tiles[blockIdx.x];
    const int16_t* dw = digits + (size_t)blockIdx.y * n_tot;          // blockIdx.y = window
    uint32_t* cursor_w = cursor + tl.bucket_base + (size_t)blockIdx.y * bpw;
    for (uint32_t b = threadIdx.x; b < bpw; b += MSM_THREADS) cnt[b] = 0;
    __syncthreads();
    for (uint32_t i = tl.begin + threadIdx.x; i < tl.end; i += MSM_THREADS) {
        const int32_t d = dw[i];
        if (d) atomicAdd(&cnt[(d < 0 ? -d : d) - 1], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < bpw; b += MSM_THREADS) {
        const uint32_t c = cnt[b];
        base[b] = c ? atomicAdd(&cursor_w[b], c) : 0u;
        cnt[b] = 0;
    }
    __syncthreads();
    for (uint32_t i = tl.begin + threadIdx.x; i < tl.end; i += MSM_THREADS) {
        const int32_t d = dw[i];
        if (d) {
            const uint32_t b = (uint32_t)((d < 0 ? -d : d) - 1);
            const uint32_t r = atomicAdd(&cnt[b], 1u);
            sorted[base[b] + r] = (i - tl.local_off) | (d < 0 ? 0x80000000u : 0u);
        }
    }
}

// one window per launch
__global__ __launch_bounds__(MSM_THREADS) void k_msm_scatter_w(const int16_t* __restrict__ dw, size_t n,
                                                               uint32_t* cursor_w, uint32_t* __restrict__ sorted) {
    for (size_t i = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * MSM_THREADS) {
        const int32_t d = dw[i];
        if (d) {
            const uint32_t pos = atomicAdd(&cursor_w[(d < 0 ? -d : d) - 1], 1u);
            sorted[pos] = (uint32_t)i | (d < 0 ? 0x80000000u : 0u);
        }
    }
}

__global__ __launch_bounds__(MSM_THREADS) void k_msm_scatter(const Fr* __restrict__ canon, size_t n, MsmShape S,
                                                             uint32_t* cursor, uint32_t* __restrict__ sorted) {
    for (size_t i = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * MSM_THREADS) {
        const Fr k = fe_load(canon + i);
        for_each_digit(k, S, [&](uint32_t w, uint32_t b, bool neg) {
            const uint32_t pos = atomicAdd(&cursor[w * S.bpw + b], 1u);
            sorted[pos] = (uint32_t)i | (neg ? 0x80000000u : 0u);
        });
    }
}

// one thread per bucket: stream its sorted entries with mixed additions
__global__ __launch_bounds__(MSM_THREADS) void k_msm_accumulate(const G1Affine* __restrict__ bases,
                                                                const uint32_t* __restrict__ sorted,
                                                                const uint32_t* __restrict__ offsets,
                                                                uint32_t n_buckets, uint32_t seg,
                                                                G1Xyzz* __restrict__ buckets) {
    const uint32_t t = blockIdx.x * MSM_THREADS + threadIdx.x;
    const uint32_t b = t / seg;
    if (b >= n_buckets) return;
    uint32_t lo = offsets[b], hi = offsets[b + 1];
    if (seg > 1) {   // this thread's slice of the bucket
        const uint32_t len = hi - lo, s = t % seg;
        const uint32_t a = (uint32_t)(((uint64_t)len * s) / seg), e = (uint32_t)(((uint64_t)len * (s + 1)) / seg);
        hi = lo + e; lo = lo + a;
    }
    // 29-bit lazy-limb accumulator (curve_f9.hip.h); canonical XYZZ written once per bucket
    G1Xyzz9 acc;
    acc.inf = true;
    acc.x = f9_zero(); acc.y = f9_zero(); acc.zz = f9_zero(); acc.zzz = f9_zero();
    for (uint32_t j = lo; j < hi; j++) {
        const uint32_t v = sorted[j];
        const G1Affine p = g1_aff_load(bases + (v & 0x7fffffffu));
        if (g1_aff_is_inf(p)) continue;
        g1_madd_f9(acc, p, (v >> 31) != 0);
    }
    g1_store(buckets + t, g1_from_f9(acc));
}

// ---- load-balanced accumulation: a bucket is cut into segments of at most MSM_SEG_LEN sorted
// entries; thread t owns segment t (bucket found by binary search in the segment offsets), so a
// skewed window (the short top window of a 254-bit scalar, or narrow activations) cannot leave one
// thread with a million additions.
constexpr uint32_t MSM_SEG_LEN = 128;     // upper bound; small problems use shorter segments (more threads, shorter chains)

__global__ __launch_bounds__(MSM_THREADS) void k_msm_seg_counts(const uint32_t* __restrict__ offsets, uint32_t n_buckets,
                                                                uint32_t seg_len, uint32_t* __restrict__ segc) {
    const uint32_t b = blockIdx.x * MSM_THREADS + threadIdx.x;
    if (b < n_buckets) segc[b] = (offsets[b + 1] - offsets[b] + seg_len - 1) / seg_len;
}

__global__ __launch_bounds__(MSM_THREADS) void k_msm_accumulate_seg(const G1Affine* __restrict__ bases,
                                                                    const uint32_t* __restrict__ sorted,
                                                                    const uint32_t* __restrict__ offsets,
                                                                    const uint32_t* __restrict__ seg_off, uint32_t n_buckets,
                                                                    uint32_t seg_len, G1Xyzz* __restrict__ partial) {
    const uint32_t t = blockIdx.x * MSM_THREADS + threadIdx.x;
    if (t >= seg_off[n_buckets]) return;
    uint32_t lo_b = 0, hi_b = n_buckets;              // largest b with seg_off[b] <= t
    while (hi_b - lo_b > 1) {
        const uint32_t mid = (lo_b + hi_b) >> 1;
        if (seg_off[mid] <= t) lo_b = mid; else hi_b = mid;
    }
    const uint32_t b = lo_b;
    const uint32_t lo = offsets[b] + (t - seg_off[b]) * seg_len;
    const uint32_t end = offsets[b + 1];
    const uint32_t hi = lo + seg_len < end ? lo + seg_len : end;
    G1Xyzz9 acc;
    acc.inf = true;
    acc.x = f9_zero(); acc.y = f9_zero(); acc.zz = f9_zero(); acc.zzz = f9_zero();
    for (uint32_t j = lo; j < hi; j++) {
        const uint32_t v = sorted[j];
        const G1Affine p = g1_aff_load(bases + (v & 0x7fffffffu));
        if (g1_aff_is_inf(p)) continue;
        g1_madd_f9(acc, p, (v >> 31) != 0);
    }
    g1_store(partial + t, g1_from_f9(acc));
}

// buckets with at most MSM_SMALL_SEGS segments: one thread sums them; the others are put on a list for the
// workgroup-per-bucket kernel (n_big must be zero on entry)
constexpr uint32_t MSM_SMALL_SEGS = 12;
__global__ __launch_bounds__(MSM_THREADS) void k_msm_bucket_reduce_small(const G1Xyzz* __restrict__ partial,
                                                                         const uint32_t* __restrict__ seg_off, uint32_t n_buckets,
                                                                         G1Xyzz* __restrict__ buckets, uint32_t* __restrict__ big_list,
                                                                         uint32_t* n_big) {
    const uint32_t b = blockIdx.x * MSM_THREADS + threadIdx.x;
    if (b >= n_buckets) return;
    const uint32_t s0 = seg_off[b], cnt = seg_off[b + 1] - s0;
    if (cnt > MSM_SMALL_SEGS) { big_list[atomicAdd(n_big, 1u)] = b; return; }
    G1Xyzz acc = g1_inf();
    if (cnt) acc = g1_load(partial + s0);
    for (uint32_t s = 1; s < cnt; s++) acc = g1_add(acc, g1_load(partial + s0 + s));
    g1_store(buckets + b, acc);
}

// the long ones: a workgroup per listed bucket, the grid strides over the list.  (One workgroup per bucket of the
// whole table spent 0.27 ms at 2^22 launching 82 k workgroups that left at once; a grid striding over all buckets
// spent 1.4 ms in the batched HyperKZG pipeline on 200 dependent loads per workgroup just to skip them.)
__global__ __launch_bounds__(MSM_THREADS) void k_msm_bucket_reduce_big(const G1Xyzz* __restrict__ partial,
                                                                       const uint32_t* __restrict__ seg_off,
                                                                       const uint32_t* __restrict__ big_list,
                                                                       const uint32_t* __restrict__ n_big,
                                                                       G1Xyzz* __restrict__ buckets) {
    __shared__ G1Xyzz sm[MSM_THREADS];
    const uint32_t n_list = *n_big;
    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        const uint32_t b = big_list[li];
        const uint32_t s0 = seg_off[b], cnt = seg_off[b + 1] - s0;
        G1Xyzz acc = g1_inf();
        for (uint32_t i = threadIdx.x; i < cnt; i += MSM_THREADS) acc = g1_add(acc, g1_load(partial + s0 + i));
        sm[threadIdx.x] = acc;
        __syncthreads();
        uint32_t top = MSM_THREADS / 2;                          // tree only as deep as the occupied slots
        while (top >= cnt && top > 1) top >>= 1;
        for (uint32_t d = top; d >= 1; d >>= 1) {
            if (threadIdx.x < d) sm[threadIdx.x] = g1_add(sm[threadIdx.x], sm[threadIdx.x + d]);
            __syncthreads();
        }
        if (threadIdx.x == 0) g1_store(buckets + b, sm[0]);
        __syncthreads();
    }
}

// buckets[b] = sum of its `seg` partials
__global__ __launch_bounds__(MSM_THREADS) void k_g1_seg_sum(const G1Xyzz* __restrict__ partial, uint32_t seg,
                                                            uint32_t n_buckets, G1Xyzz* __restrict__ buckets) {
    const uint32_t b = blockIdx.x * MSM_THREADS + threadIdx.x;
    if (b >= n_buckets) return;
    G1Xyzz acc = g1_load(partial + (size_t)b * seg);
    for (uint32_t s = 1; s < seg; s++) acc = g1_add(acc, g1_load(partial + (size_t)b * seg + s));
    g1_store(buckets + b, acc);
}

// narrow scalars: |s| as a 64-bit magnitude
template <typename T>
__device__ __forceinline__ uint64_t small_abs(T v, bool& neg) {
    if constexpr (std::is_signed<T>::value) {
        neg = v < 0;
        const uint64_t u = (uint64_t)(int64_t)v;
        return neg ? (uint64_t)0 - u : u;
    } else {
        neg = false;
        return (uint64_t)v;
    }
}

template <typename T>
__global__ __launch_bounds__(MSM_THREADS) void k_abs_max(const T* __restrict__ s, size_t n, unsigned long long* out) {
    unsigned long long m = 0;
    for (size_t i = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * MSM_THREADS) {
        bool neg;
        const unsigned long long a = small_abs<T>(s[i], neg);
        m = a > m ? a : m;
    }
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long x = __shfl_xor(m, o);
        m = x > m ? x : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

template <typename T>
__global__ __launch_bounds__(MSM_THREADS) void k_msm_digits_small(const T* __restrict__ s, size_t n, MsmShape S,
                                                                  int16_t* __restrict__ digits) {
    for (size_t i = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * MSM_THREADS) {
        bool sneg;
        const uint64_t a = small_abs<T>(s[i], sneg);
        Fr k;
        k.v[0] = (uint32_t)a; k.v[1] = (uint32_t)(a >> 32);
        for (int j = 2; j < 8; j++) k.v[j] = 0;
        for (uint32_t w = 0; w < S.n_windows; w++) digits[(size_t)w * n + i] = 0;
        for_each_digit(k, S, [&](uint32_t w, uint32_t b, bool neg) {
            digits[(size_t)w * n + i] = (neg != sneg) ? (int16_t)(-(int32_t)(b + 1)) : (int16_t)(b + 1);
        });
    }
}

// small * P by double-and-add
__device__ inline G1Xyzz g1_mul_small(const G1Xyzz& p, uint32_t s) {
    G1Xyzz acc = g1_inf();
    for (int i = 31 - __clz((int)(s | 1u)); i >= 0; i--) {
        acc = g1_dbl(acc);
        if ((s >> i) & 1u) acc = g1_add(acc, p);
    }
    return s ? acc : g1_inf();
}

// fold MSM_CHUNK buckets: sum_{k in chunk} (k+1) * B_k, k = index within the window
// (bucket k holds digit magnitude k+1).  One thread per chunk.
__global__ __launch_bounds__(MSM_THREADS) void k_msm_fold_chunks(const G1Xyzz* __restrict__ buckets, MsmShape S,
                                                                 uint32_t chunk, uint32_t n_chunks_total,
                                                                 G1Xyzz* __restrict__ out) {
    const uint32_t t = blockIdx.x * MSM_THREADS + threadIdx.x;
    if (t >= n_chunks_total) return;
    const uint32_t chunks_per_window = S.bpw / chunk;
    const uint32_t w = t / chunks_per_window, j = t % chunks_per_window;
    const G1Xyzz* base = buckets + (size_t)w * S.bpw + (size_t)j * chunk;
    G1Xyzz run = g1_inf(), acc = g1_inf();
    for (int k = (int)chunk - 1; k >= 0; k--) {
        run = g1_add(run, g1_load(base + k));
        acc = g1_add(acc, run);
    }
    // acc = sum (k_local + 1) B ; add (j*chunk) * run for the chunk's offset
    acc = g1_add(acc, g1_mul_small(run, j * chunk));
    g1_store(out + t, acc);
}

// sum groups of `per_group` points: one workgroup per group, tree in LDS
__global__ __launch_bounds__(MSM_THREADS) void k_g1_group_sum(const G1Xyzz* __restrict__ pts, uint32_t per_group,
                                                              G1Xyzz* __restrict__ out) {
    __shared__ G1Xyzz sm[MSM_THREADS];
    const G1Xyzz* base = pts + (size_t)blockIdx.x * per_group;
    G1Xyzz acc = g1_inf();
    for (uint32_t i = threadIdx.x; i < per_group; i += MSM_THREADS) acc = g1_add(acc, g1_load(base + i));
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = MSM_THREADS / 2; d >= 1; d >>= 1) {
        if (threadIdx.x < d) sm[threadIdx.x] = g1_add(sm[threadIdx.x], sm[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) g1_store(out + blockIdx.x, sm[0]);
}

// one-hot commit: sum of bases[idx[i]] — each thread folds a strided slice, then the
// workgroup tree; one partial per workgroup (hyperkzg/mod.rs:520-554: T additions)
__global__ __launch_bounds__(MSM_THREADS) void k_g1_sum_indexed(const G1Affine* __restrict__ bases,
                                                                const uint32_t* __restrict__ idx, size_t n,
                                                                G1Xyzz* __restrict__ out) {
    __shared__ G1Xyzz sm[MSM_THREADS];
    G1Xyzz acc = g1_inf();
    for (size_t i = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * MSM_THREADS) {
        const G1Affine p = g1_aff_load(bases + idx[i]);
        if (!g1_aff_is_inf(p)) acc = g1_madd(acc, p, false);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = MSM_THREADS / 2; d >= 1; d >>= 1) {
        if (threadIdx.x < d) sm[threadIdx.x] = g1_add(sm[threadIdx.x], sm[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) g1_store(out + blockIdx.x, sm[0]);
}

// HyperKZG::batch_commit_one_hot (hyperkzg/mod.rs:558-596): R one-hot polynomials in one launch.  blockIdx.y = polynomial;
// polynomial r has T[r] cycles whose addresses sit at idx[off[r] .. off[r] + T[r]) (negative = None); its non-zero
// coefficient of cycle t is the SRS point k * T + t.  One partial per workgroup: out[r * gridDim.x + blockIdx.x].
struct OneHotRowDesc { uint32_t off, T; };
__global__ __launch_bounds__(MSM_THREADS) void k_g1_sum_onehot_rows(const G1Affine* __restrict__ bases, const int32_t* __restrict__ idx,
                                                                    const OneHotRowDesc* __restrict__ rows, G1Xyzz* __restrict__ out) {
    __shared__ G1Xyzz sm[MSM_THREADS];
    const OneHotRowDesc rd = rows[blockIdx.y];
    G1Xyzz acc = g1_inf();
    for (size_t t = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; t < rd.T; t += (size_t)gridDim.x * MSM_THREADS) {
        const int32_t k = idx[rd.off + t];
        if (k < 0) continue;
        const G1Affine p = g1_aff_load(bases + (size_t)k * rd.T + t);
        if (!g1_aff_is_inf(p)) acc = g1_madd(acc, p, false);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = MSM_THREADS / 2; d >= 1; d >>= 1) {
        if (threadIdx.x < d) sm[threadIdx.x] = g1_add(sm[threadIdx.x], sm[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) g1_store(out + (size_t)blockIdx.y * gridDim.x + blockIdx.x, sm[0]);
}

// SRS generation (SRS::setup, hyperkzg/kzg.rs:26-93): out[i] = tau^(i+1) * G.
// tau_pow2[j] = tau^(2^j) (Montgomery Fr), dbl_table[j] = 2^j * G (affine).
__global__ __launch_bounds__(MSM_THREADS) void k_srs_generate(const Fr* __restrict__ tau_pow2,
                                                              const G1Affine* __restrict__ dbl_table, size_t n,
                                                              G1Affine* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * MSM_THREADS) {
        uint64_t e = (uint64_t)i + 1;
        Fr s = fr_one();
        for (int j = 0; e; j++, e >>= 1)
            if (e & 1) s = fr_mul(s, fe_load(tau_pow2 + j));
        const Fr k = fe_from_mont<FrParams>(s);
        G1Xyzz acc = g1_inf();
        for (int b = 0; b < 254; b++)
            if ((k.v[b >> 5] >> (b & 31)) & 1u) acc = g1_madd(acc, g1_aff_load(dbl_table + b), false);
        const G1Affine a = g1_to_aff(acc);
        fe_store(&out[i].x, a.x);
        fe_store(&out[i].y, a.y);
    }
}

}  // namespace atlas
