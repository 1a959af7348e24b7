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

// ---- even segments: thread t owns sorted entries [t * S, (t + 1) * S) whatever buckets they belong to and writes one
// partial per bucket it touches, at slot t + b.  Slots are unique and ascending (thread t leaving bucket b for b + 1
// writes t + b and t + b + 1; thread t + 1 continues bucket b + 1 at t + b + 2), so bucket b's partials are the
// contiguous slots [offsets[b] / S + b, (offsets[b+1] - 1) / S + b].  Every thread does exactly S additions: no lane
// waits for a longer segment of its wavefront, and no segment-count scan is needed.
__global__ __launch_bounds__(MSM_THREADS) void k_msm_accumulate_even(const G1Affine* __restrict__ bases,
                                                                     const uint32_t* __restrict__ sorted,
                                                                     const uint32_t* __restrict__ offsets, uint32_t n_buckets,
                                                                     uint32_t seg_log, G1Xyzz* __restrict__ partial) {
    const uint32_t t = blockIdx.x * MSM_THREADS + threadIdx.x;
    const uint32_t total = offsets[n_buckets];
    const uint64_t lo64 = (uint64_t)t << seg_log;
    if (lo64 >= total) return;
    const uint32_t lo = (uint32_t)lo64;
    const uint32_t hi = lo64 + (1u << seg_log) < total ? lo + (1u << seg_log) : total;
    uint32_t lo_b = 0, hi_b = n_buckets;              // largest b with offsets[b] <= lo: the bucket holding entry lo
    while (hi_b - lo_b > 1) {
        const uint32_t mid = (lo_b + hi_b) >> 1;
        if (offsets[mid] <= lo) lo_b = mid; else hi_b = mid;
    }
    uint32_t b = lo_b, next = offsets[b + 1];
    G1Xyzz9 acc;
    acc.inf = true;
    acc.x = f9_zero(); acc.y = f9_zero(); acc.zz = f9_zero(); acc.zzz = f9_zero();
    for (uint32_t j = lo; j < hi; j++) {
        if (j == next) {
            g1_store(partial + (size_t)t + b, g1_from_f9(acc));
            acc.inf = true;
            acc.x = f9_zero(); acc.y = f9_zero(); acc.zz = f9_zero(); acc.zzz = f9_zero();
            do { b++; next = offsets[b + 1]; } while (next <= j);
        }
        const uint32_t v = sorted[j];
        const G1Affine p = g1_aff_load(bases + (v & 0x7fffffffu));
        if (g1_aff_is_inf(p)) continue;
        g1_madd_f9(acc, p, (v >> 31) != 0);
    }
    g1_store(partial + (size_t)t + b, g1_from_f9(acc));
}

// where bucket b's partials are: the scanned segment counts of k_msm_accumulate_seg, or the slots of k_msm_accumulate_even
struct SegMap {
    const uint32_t* seg_off;      // non-null: partials [seg_off[b], seg_off[b+1])
    const uint32_t* offsets;      // else: from the bucket offsets and the segment length 2^seg_log
    uint32_t seg_log;
    __device__ __forceinline__ void get(uint32_t b, uint32_t& s0, uint32_t& cnt) const {
        if (seg_off) { s0 = seg_off[b]; cnt = seg_off[b + 1] - s0; return; }
        const uint32_t a = offsets[b], e = offsets[b + 1];
        if (e == a) { s0 = 0; cnt = 0; return; }
        s0 = (a >> seg_log) + b;
        cnt = ((e - 1) >> seg_log) + b - s0 + 1;
    }
};

// ---- bucket reduction: buckets[b] = sum of bucket b's partials.  Three regimes so that no distribution of scalars falls
// off a cliff (uniform 254-bit scalars, a short top digit that fills a handful of buckets with a quarter of all entries,
// field elements that are really small integers and all land in one bucket):
//   <= MSM_SMALL_SEGS partials: one thread sums them (k_msm_bucket_reduce_small, which also sorts the rest into lists);
//   <= MSM_MEDIUM_SEGS: 16 lanes per bucket, <= 16 partials per lane then a 4-level tree through shuffles;
//   more: the bucket's partials are cut into runs of MSM_BIG_RUN, one workgroup per run wherever it lies (stage 1), then
//         one workgroup per bucket sums the run results (stage 2) — a bucket holding every entry of a 2^24 MSM costs
//         two ~0.2 ms kernels instead of a 10 ms serial chain in one workgroup.
constexpr uint32_t MSM_SMALL_SEGS = 16;
constexpr uint32_t MSM_MEDIUM_SEGS = 256;
constexpr uint32_t MSM_BIG_RUN = 1024;

__global__ __launch_bounds__(MSM_THREADS) void k_msm_bucket_reduce_small(const G1Xyzz* __restrict__ partial,
                                                                         const SegMap sm, uint32_t n_buckets,
                                                                         G1Xyzz* __restrict__ buckets, uint32_t* __restrict__ medium_list,
                                                                         uint32_t* __restrict__ big_list, uint32_t* n_lists) {
    const uint32_t b = blockIdx.x * MSM_THREADS + threadIdx.x;
    if (b >= n_buckets) return;
    uint32_t s0, cnt;
    sm.get(b, s0, cnt);
    if (cnt > MSM_MEDIUM_SEGS) { big_list[atomicAdd(n_lists + 1, 1u)] = b; return; }
    if (cnt > MSM_SMALL_SEGS) { medium_list[atomicAdd(n_lists, 1u)] = b; return; }
    G1Xyzz acc = g1_inf();
    if (cnt) acc = g1_load(partial + s0);
    for (uint32_t s = 1; s < cnt; s++) acc = g1_add(acc, g1_load(partial + s0 + s));
    g1_store(buckets + b, acc);
}

__device__ __forceinline__ G1Xyzz g1_shfl_down16(const G1Xyzz& p, uint32_t d) {
    G1Xyzz o;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        o.x.v[i] = __shfl_down(p.x.v[i], d, 16);
        o.y.v[i] = __shfl_down(p.y.v[i], d, 16);
        o.zz.v[i] = __shfl_down(p.zz.v[i], d, 16);
        o.zzz.v[i] = __shfl_down(p.zzz.v[i], d, 16);
    }
    return o;
}

__global__ __launch_bounds__(MSM_THREADS) void k_msm_bucket_reduce_medium(const G1Xyzz* __restrict__ partial, const SegMap map,
                                                                          const uint32_t* __restrict__ medium_list,
                                                                          const uint32_t* __restrict__ n_lists,
                                                                          G1Xyzz* __restrict__ buckets) {
    const uint32_t n_list = n_lists[0];
    const uint32_t lane = threadIdx.x & 15u;
    const uint32_t groups = gridDim.x * (MSM_THREADS / 16);
    // every 16-lane group of a wavefront runs the same number of rounds (shuffles need the whole wavefront)
    const uint32_t g0 = (blockIdx.x * MSM_THREADS + threadIdx.x) / 16;
    const uint32_t wave_first = g0 & ~3u;
    for (uint32_t base = wave_first; base < n_list; base += groups) {
        const uint32_t li = base + (g0 & 3u);
        uint32_t b = 0, s0 = 0, cnt = 0;
        if (li < n_list) { b = medium_list[li]; map.get(b, s0, cnt); }
        G1Xyzz acc = g1_inf();
        for (uint32_t i = lane; i < cnt; i += 16) acc = g1_add(acc, g1_load(partial + s0 + i));
        for (uint32_t d = 8; d >= 1; d >>= 1) {
            const G1Xyzz o = g1_shfl_down16(acc, d);
            if (lane < d) acc = g1_add(acc, o);
        }
        if (lane == 0 && li < n_list) g1_store(buckets + b, acc);
    }
}

// big_run_off[li] = number of runs of the big buckets before list entry li (exclusive), [n_big] = total.  One workgroup.
__global__ __launch_bounds__(1024) void k_msm_big_prefix(const SegMap map, const uint32_t* __restrict__ big_list,
                                                         const uint32_t* __restrict__ n_lists, uint32_t* __restrict__ big_run_off) {
    __shared__ uint32_t part[1024];
    const uint32_t n = n_lists[1], t = threadIdx.x;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) { uint32_t s0, cnt; map.get(big_list[i], s0, cnt); s += (cnt + MSM_BIG_RUN - 1) / MSM_BIG_RUN; }
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = t ? part[t - 1] : 0;
    for (uint32_t i = lo; i < hi; i++) {
        big_run_off[i] = run;
        uint32_t s0, cnt; map.get(big_list[i], s0, cnt);
        run += (cnt + MSM_BIG_RUN - 1) / MSM_BIG_RUN;
    }
    if (t == 1023) big_run_off[n] = part[1023];
}

__device__ __forceinline__ G1Xyzz g1_block_sum(G1Xyzz acc, G1Xyzz* sm) {      // result valid in thread 0
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = MSM_THREADS / 2; d >= 1; d >>= 1) {
        if (threadIdx.x < d) sm[threadIdx.x] = g1_add(sm[threadIdx.x], sm[threadIdx.x + d]);
        __syncthreads();
    }
    const G1Xyzz r = sm[0];
    __syncthreads();
    return r;
}

// stage 1: run w of the big buckets (found by binary search in big_run_off) -> run_sums[w]
__global__ __launch_bounds__(MSM_THREADS) void k_msm_bucket_reduce_big1(const G1Xyzz* __restrict__ partial, const SegMap map,
                                                                        const uint32_t* __restrict__ big_list,
                                                                        const uint32_t* __restrict__ n_lists,
                                                                        const uint32_t* __restrict__ big_run_off,
                                                                        G1Xyzz* __restrict__ run_sums) {
    __shared__ G1Xyzz sm[MSM_THREADS];
    const uint32_t n_big = n_lists[1];
    const uint32_t n_runs = big_run_off[n_big];
    for (uint32_t w = blockIdx.x; w < n_runs; w += gridDim.x) {
        uint32_t lo = 0, hi = n_big;                  // largest li with big_run_off[li] <= w
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (big_run_off[mid] <= w) lo = mid; else hi = mid;
        }
        uint32_t s0, cnt;
        map.get(big_list[lo], s0, cnt);
        const uint32_t r0 = (w - big_run_off[lo]) * MSM_BIG_RUN;
        const uint32_t r1 = r0 + MSM_BIG_RUN < cnt ? r0 + MSM_BIG_RUN : cnt;
        G1Xyzz acc = g1_inf();
        for (uint32_t i = r0 + threadIdx.x; i < r1; i += MSM_THREADS) acc = g1_add(acc, g1_load(partial + s0 + i));
        const G1Xyzz r = g1_block_sum(acc, sm);
        if (threadIdx.x == 0) g1_store(run_sums + w, r);
    }
}

// stage 2: a workgroup per big bucket sums its run results
__global__ __launch_bounds__(MSM_THREADS) void k_msm_bucket_reduce_big2(const G1Xyzz* __restrict__ run_sums,
                                                                        const uint32_t* __restrict__ big_list,
                                                                        const uint32_t* __restrict__ n_lists,
                                                                        const uint32_t* __restrict__ big_run_off,
                                                                        G1Xyzz* __restrict__ buckets) {
    __shared__ G1Xyzz sm[MSM_THREADS];
    const uint32_t n_big = n_lists[1];
    for (uint32_t li = blockIdx.x; li < n_big; li += gridDim.x) {
        const uint32_t w0 = big_run_off[li], w1 = big_run_off[li + 1];
        G1Xyzz acc = g1_inf();
        for (uint32_t w = w0 + threadIdx.x; w < w1; w += MSM_THREADS) acc = g1_add(acc, g1_load(run_sums + w));
        const G1Xyzz r = g1_block_sum(acc, sm);
        if (threadIdx.x == 0) g1_store(buckets + big_list[li], r);
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

// ---- the fold of a wide bucket set (fixed-base path: 2^19 buckets in ONE set), without scalar multiples ----------------
// sum_k (k+1) B_k over bpw buckets, chunk = C consecutive buckets per thread, j = chunk index:
//     = sum_j acc_j + C * sum_j j * S_j,   acc_j = sum_i (i+1) B_{jC+i},  S_j = sum_i B_{jC+i}           (level 1, 2C additions deep)
//     sum_j j * S_j = sum_b 2^b * M_b,     M_b = sum of the S_j whose index has bit b set                (plain masked sums)
// k_msm_fold_chunks pays ~19 doublings + as many (divergent) additions per thread for j*C * S_j — 3/4 of its time — and the
// additions of a point chain cost ~12 us each whatever the occupancy (14 dependent field products), so what counts here is the
// DEPTH: level 1 (2C), two tree levels of 8 (k_msm_fold_groups over 256 chunks, k_msm_fold_sets over the groups), and the
// Horner over the ~17 masked sums on the host (0.4 us per operation there) or, for many sets, one thread per set.
__global__ __launch_bounds__(MSM_THREADS) void k_msm_fold_pairs(const G1Xyzz* __restrict__ buckets, uint32_t chunk, uint32_t n_chunks_total,
                                                                G1Xyzz* __restrict__ acc_out, G1Xyzz* __restrict__ run_out) {
    const uint32_t t = blockIdx.x * MSM_THREADS + threadIdx.x;
    if (t >= n_chunks_total) return;
    const G1Xyzz* base = buckets + (size_t)t * chunk;
    G1Xyzz run = g1_load(base + chunk - 1), acc = run;
    for (int k = (int)chunk - 2; k >= 0; k--) {
        run = g1_add(run, g1_load(base + k));
        acc = g1_add(acc, run);
    }
    g1_store(acc_out + t, acc);
    g1_store(run_out + t, run);
}

// One tree over 256 points p_t that yields their sum AND the eight masked sums M_b = sum of the p_t with bit b of t set: a node of
// level L (2^L consecutive points) carries its sum and M_0 .. M_{L-1}; merging two level-b nodes costs b + 1 additions (M_i = M_i' + M_i''
// for i < b, M_b = the right node's sum, S = S' + S''), one per lane, so a level is ONE addition deep and at most two wavefronts wide
// (128, 128, 96, 64, 40, 24, 14, 8 lanes) — ten separate masked trees were 2.3 x the wavefront-additions of this one and took 0.54 ms.
// sS: 256 points (in: the points; out: [0] = sum), sM: 2 x 128 points (out: sM[0 .. 8) = M_0 .. M_7).  Every thread calls.
__device__ __forceinline__ void g1_block_bit_sums(G1Xyzz* sS, G1Xyzz* sM) {
#pragma unroll 1
    for (uint32_t b = 0; b < 8; b++) {
        const uint32_t items = (128u >> b) * (b + 1);
        G1Xyzz* oldM = sM + (b & 1u) * 128u;
        G1Xyzz* newM = sM + ((b + 1) & 1u) * 128u;
        if (threadIdx.x < items) {
            const uint32_t m = threadIdx.x / (b + 1), i = threadIdx.x % (b + 1);
            const bool mm = i < b;                  // ONE addition per lane and level, whichever kind (two call sites would run one after the other)
            const G1Xyzz* pa = mm ? oldM + (2 * m) * b + i : sS + ((2 * m) << b);
            const G1Xyzz* pb = mm ? oldM + (2 * m + 1) * b + i : sS + ((2 * m + 1) << b);
            const G1Xyzz A = *pa, B = *pb;
            const G1Xyzz r = g1_add(A, B);
            if (mm) newM[m * (b + 1) + i] = r;
            else { newM[m * (b + 1) + b] = B; sS[(2 * m) << b] = r; }
        }
        __syncthreads();
    }
}

// grid (groups of 256 chunks, 2, sets): y = 0: the sum of the group's acc -> row 0; y = 1: the sum of its S -> row 1 and the masked
// sums by bit b of the chunk index -> row 2 + b.  out[(set * 10 + row) * groups + group]
__global__ __launch_bounds__(MSM_THREADS) void k_msm_fold_groups(const G1Xyzz* __restrict__ acc_in, const G1Xyzz* __restrict__ run_in,
                                                                 uint32_t chunks_per_set, G1Xyzz* __restrict__ out) {
    extern __shared__ G1Xyzz sm[];          // 256 points (y = 0) or 512 (y = 1)
    const uint32_t grp = blockIdx.x, set = blockIdx.z, groups = gridDim.x;
    const size_t t = (size_t)set * chunks_per_set + (size_t)grp * MSM_THREADS + threadIdx.x;
    G1Xyzz* o = out + (size_t)set * 10 * groups + grp;
    if (blockIdx.y == 0) {
        const G1Xyzz r = g1_block_sum(g1_load(acc_in + t), sm);
        if (threadIdx.x == 0) g1_store(o, r);
        return;
    }
    sm[threadIdx.x] = g1_load(run_in + t);
    __syncthreads();
    g1_block_bit_sums(sm, sm + MSM_THREADS);
    if (threadIdx.x == 0) g1_store(o + groups, sm[0]);
    else if (threadIdx.x <= 8) g1_store(o + (size_t)(1 + threadIdx.x) * groups, sm[MSM_THREADS + threadIdx.x - 1]);
}

// grid (10 + max(hi_bits - 8, 0), sets): x < 9: the plain sums over the groups of rows 0 (-> fin 0) and 2 + b (-> fin 1 + b); x = 9: the masked
// sums of row 1 by bits 0..7 of the group index -> fin 9 + h; x = 10 + e: by bit 8 + e (more than 256 groups) -> fin 17 + e.
__global__ __launch_bounds__(MSM_THREADS) void k_msm_fold_sets(const G1Xyzz* __restrict__ part, uint32_t groups, uint32_t hi_bits,
                                                               G1Xyzz* __restrict__ fin) {
    extern __shared__ G1Xyzz sm[];
    const uint32_t x = blockIdx.x, set = blockIdx.y;
    const uint32_t row = x == 0 ? 0u : x < 9 ? x + 1u : 1u;
    const uint32_t mask = x > 9 ? 1u << (x - 2) : 0u;
    const G1Xyzz* src = part + ((size_t)set * 10 + row) * groups;
    G1Xyzz* f = fin + (size_t)set * (9 + hi_bits);
    G1Xyzz p = g1_inf();
    for (uint32_t gi = threadIdx.x; gi < groups; gi += MSM_THREADS)
        if (!mask || (gi & mask)) p = gi < MSM_THREADS ? g1_load(src + gi) : g1_add(p, g1_load(src + gi));
    if (x != 9) {
        const G1Xyzz r = g1_block_sum(p, sm);
        if (threadIdx.x == 0) g1_store(f + (x < 9 ? x : x + 7), r);
        return;
    }
    sm[threadIdx.x] = p;
    __syncthreads();
    g1_block_bit_sums(sm, sm + MSM_THREADS);
    if (threadIdx.x < (hi_bits < 8 ? hi_bits : 8u)) g1_store(f + 9 + threadIdx.x, sm[MSM_THREADS + threadIdx.x]);
}

// many sets: the Horner over a set's masked sums, one thread per set (a few sets: msm_finish does it on the host)
__global__ __launch_bounds__(64) void k_msm_fold_horner(const G1Xyzz* __restrict__ fin, uint32_t n_sets, uint32_t n_bits, uint32_t chunk_log,
                                                        G1Xyzz* __restrict__ wsum) {
    const uint32_t set = blockIdx.x * 64 + threadIdx.x;
    if (set >= n_sets) return;
    const G1Xyzz* f = fin + (size_t)set * (1 + n_bits);
    G1Xyzz acc = g1_inf();
    for (int b = (int)n_bits - 1; b >= 0; b--) acc = g1_add(g1_dbl(acc), g1_load(f + 1 + b));
    for (uint32_t d = 0; d < chunk_log; d++) acc = g1_dbl(acc);
    g1_store(wsum + set, g1_add(acc, g1_load(f)));
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
    G1Xyzz9 acc = g1_inf_f9();          // the lazy-limb mixed addition of the bucket accumulation (curve_f9.hip.h: 1.6 x the multiplication rate of the 8 x 32 one)
    for (size_t i = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * MSM_THREADS) {
        const G1Affine p = g1_aff_load(bases + idx[i]);
        if (!g1_aff_is_inf(p)) g1_madd_f9(acc, p, false);
    }
    sm[threadIdx.x] = g1_from_f9(acc);
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
    G1Xyzz9 acc = g1_inf_f9();
    for (size_t t = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; t < rd.T; t += (size_t)gridDim.x * MSM_THREADS) {
        const int32_t k = idx[rd.off + t];
        if (k < 0) continue;
        const G1Affine p = g1_aff_load(bases + (size_t)k * rd.T + t);
        if (!g1_aff_is_inf(p)) g1_madd_f9(acc, p, false);
    }
    sm[threadIdx.x] = g1_from_f9(acc);
    __syncthreads();
    for (uint32_t d = MSM_THREADS / 2; d >= 1; d >>= 1) {
        if (threadIdx.x < d) sm[threadIdx.x] = g1_add(sm[threadIdx.x], sm[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) g1_store(out + (size_t)blockIdx.y * gridDim.x + blockIdx.x, sm[0]);
}

// The RaD witness commitments of a lookup (witness.rs:136-200 builds one OneHotPolynomial per 4-bit chunk of the lookup
// index and commits it with commit_one_hot): chunk i of cycle t selects SRS point chunk_i(idx_t) * T + t.  The chunks are cut
// in the kernel, so the d commitments need only the T lookup indices already on the device.  blockIdx.y = chunk.
__global__ __launch_bounds__(MSM_THREADS) void k_g1_sum_lookup_chunks(const G1Affine* __restrict__ bases, const uint64_t* __restrict__ lookups,
                                                                      uint32_t T, uint32_t d, uint32_t log_k_chunk, G1Xyzz* __restrict__ out) {
    __shared__ G1Xyzz sm[MSM_THREADS];
    const uint32_t i = blockIdx.y, shift = log_k_chunk * (d - 1 - i);
    const uint64_t mask = ((uint64_t)1 << log_k_chunk) - 1;
    G1Xyzz9 acc = g1_inf_f9();
    for (size_t t = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; t < T; t += (size_t)gridDim.x * MSM_THREADS) {
        const uint64_t k = shift >= 64 ? 0 : ((lookups[t] >> shift) & mask);
        const G1Affine p = g1_aff_load(bases + (size_t)k * T + t);
        if (!g1_aff_is_inf(p)) g1_madd_f9(acc, p, false);
    }
    sm[threadIdx.x] = g1_from_f9(acc);
    __syncthreads();
    for (uint32_t s2 = MSM_THREADS / 2; s2 >= 1; s2 >>= 1) {
        if (threadIdx.x < s2) sm[threadIdx.x] = g1_add(sm[threadIdx.x], sm[threadIdx.x + s2]);
        __syncthreads();
    }
    if (threadIdx.x == 0) g1_store(out + (size_t)blockIdx.y * gridDim.x + blockIdx.x, sm[0]);
}

// the same for the chunk polynomials of MANY lookups in one launch (a model graph commits thousands): blockIdx.y = chunk polynomial
struct LookupChunkRow { const uint64_t* lookups; uint32_t T, shift; };
__global__ __launch_bounds__(MSM_THREADS) void k_g1_sum_lookup_rows(const G1Affine* __restrict__ bases, const LookupChunkRow* __restrict__ rows, uint32_t mask,
                                                                    G1Xyzz* __restrict__ out) {
    __shared__ G1Xyzz sm[MSM_THREADS];
    const LookupChunkRow R = rows[blockIdx.y];
    G1Xyzz9 acc = g1_inf_f9();
    for (size_t t = (size_t)blockIdx.x * MSM_THREADS + threadIdx.x; t < R.T; t += (size_t)gridDim.x * MSM_THREADS) {
        const uint64_t k = R.shift >= 64 ? 0 : ((R.lookups[t] >> R.shift) & mask);
        const G1Affine p = g1_aff_load(bases + (size_t)k * R.T + t);
        if (!g1_aff_is_inf(p)) g1_madd_f9(acc, p, false);
    }
    sm[threadIdx.x] = g1_from_f9(acc);
    __syncthreads();
    for (uint32_t s2 = MSM_THREADS / 2; s2 >= 1; s2 >>= 1) {
        if (threadIdx.x < s2) sm[threadIdx.x] = g1_add(sm[threadIdx.x], sm[threadIdx.x + s2]);
        __syncthreads();
    }
    if (threadIdx.x == 0) g1_store(out + (size_t)blockIdx.y * gridDim.x + blockIdx.x, sm[0]);
}

// Which consecutive chunk polynomials of a lookup family are THE SAME polynomial: bit c of diff[f] is set iff chunk c and chunk c + 1 (chunk c is
// cut at shift = log_k_chunk (d - 1 - c)) differ at some cycle.  A 32-bit value sign-extended into a 64-bit lookup index has eight identical top
// nibbles, and the nibbles above the operands' magnitude repeat the sign as well: 11 of the 16 chunk polynomials of a typical clamp / ReLU family
// are one polynomial — one commitment, computed once (atlas_commit_lookup_chunks_multi).  One workgroup per family.
struct LookupFamilyDev { const uint64_t* lookups; uint32_t T, d; };
__global__ __launch_bounds__(MSM_THREADS) void k_lookup_chunk_diff(const LookupFamilyDev* __restrict__ fams, uint32_t log_k_chunk, uint32_t* __restrict__ diff) {
    __shared__ uint32_t sm;
    const LookupFamilyDev F = fams[blockIdx.x];
    if (threadIdx.x == 0) sm = 0;
    __syncthreads();
    const uint64_t mask = ((uint64_t)1 << log_k_chunk) - 1;
    uint32_t acc = 0;
    for (uint32_t t = threadIdx.x; t < F.T; t += MSM_THREADS) {
        const uint64_t v = F.lookups[t];
        for (uint32_t c = 0; c + 1 < F.d; c++) {
            const uint32_t s0 = log_k_chunk * (F.d - 1 - c), s1 = s0 - log_k_chunk;
            const uint64_t a = s0 >= 64 ? 0 : ((v >> s0) & mask), b = (v >> s1) & mask;
            acc |= (a != b ? 1u : 0u) << c;
        }
    }
    if (acc) atomicOr(&sm, acc);
    __syncthreads();
    if (threadIdx.x == 0) diff[blockIdx.x] = sm;
}

// The same with the slices cut PER ROW (a graph's rows run from 2^6 to 2^20 cycles): a workgroup takes slice `slice` of `n_slices` of its row, a
// thread ~32 points of it.  With one slice count for the whole launch (sized for the longest row) a row of 2^14 cycles was spread over 64
// workgroups of ONE point per thread, each followed by the 8-level tree of full additions: the trees were the launch — 8823 rows x 64
// workgroups, 117 ms for the witness commitments of the GPT-2-shaped graph.
struct LookupSlice { uint32_t row, slice, n_slices; };
__global__ __launch_bounds__(MSM_THREADS) void k_g1_sum_lookup_slices(const G1Affine* __restrict__ bases, const LookupChunkRow* __restrict__ rows,
                                                                      const LookupSlice* __restrict__ slices, uint32_t mask, G1Xyzz* __restrict__ out) {
    __shared__ G1Xyzz sm[MSM_THREADS];
    const LookupSlice S = slices[blockIdx.x];
    const LookupChunkRow R = rows[S.row];
    G1Xyzz9 acc = g1_inf_f9();
    for (size_t t = (size_t)S.slice * MSM_THREADS + threadIdx.x; t < R.T; t += (size_t)S.n_slices * MSM_THREADS) {
        const uint64_t k = R.shift >= 64 ? 0 : ((R.lookups[t] >> R.shift) & mask);
        const G1Affine p = g1_aff_load(bases + (size_t)k * R.T + t);
        if (!g1_aff_is_inf(p)) g1_madd_f9(acc, p, false);
    }
    sm[threadIdx.x] = g1_from_f9(acc);
    __syncthreads();
    for (uint32_t s2 = MSM_THREADS / 2; s2 >= 1; s2 >>= 1) {
        if (threadIdx.x < s2) sm[threadIdx.x] = g1_add(sm[threadIdx.x], sm[threadIdx.x + s2]);
        __syncthreads();
    }
    if (threadIdx.x == 0) g1_store(out + blockIdx.x, sm[0]);
}
// out[r] = sum of pts[off[r] .. off[r + 1]) (at most 64 partials of a row): one wavefront per row
__global__ __launch_bounds__(64) void k_g1_group_sum_var(const G1Xyzz* __restrict__ pts, const uint32_t* __restrict__ off, G1Xyzz* __restrict__ out) {
    __shared__ G1Xyzz sm[64];
    const uint32_t lo = off[blockIdx.x], hi = off[blockIdx.x + 1];
    G1Xyzz acc = g1_inf();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 64) acc = g1_add(acc, g1_load(pts + i));
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = 32; d >= 1; d >>= 1) {
        if (threadIdx.x < d && lo + threadIdx.x + d < hi) sm[threadIdx.x] = g1_add(sm[threadIdx.x], sm[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) g1_store(out + blockIdx.x, sm[0]);
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
