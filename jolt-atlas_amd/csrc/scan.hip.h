// Exclusive scan of u32 counts (bucket offsets of the counting sorts in msm.hip / shout.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace atlas {

// offsets[0..n] (offsets[n] = total) and a copy in cursor.  One workgroup of 1024 threads,
// each scanning a contiguous run of ceil(n/1024) counts.
static __global__ __launch_bounds__(1024) void k_exclusive_scan(const uint32_t* __restrict__ counts, uint32_t n,
                                                                uint32_t* offsets, uint32_t* cursor) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += counts[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {      // Hillis-Steele inclusive scan
        uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = t ? part[t - 1] : 0;
    for (uint32_t i = lo; i < hi; i++) {
        offsets[i] = run; cursor[i] = run;
        run += counts[i];
    }
    if (t == 1023) offsets[n] = part[1023];
}

}  // namespace atlas

namespace atlas {

// ---- multi-workgroup exclusive scan (3 launches) for large bucket arrays -----------------
constexpr int SCAN_BLOCK = 2048;      // counts per workgroup (256 threads x 8)

// block sums
static __global__ __launch_bounds__(256) void k_scan_block_sums(const uint32_t* __restrict__ counts, uint32_t n,
                                                                uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t sm[256];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * 8;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += base + k < n ? counts[base + k] : 0u;
    sm[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t d = 128; d >= 1; d >>= 1) {
        if (threadIdx.x < d) sm[threadIdx.x] += sm[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sm[0];
}

// per-block exclusive scan with the block's base offset (block_offsets = exclusive scan of block_sums)
static __global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* __restrict__ counts, uint32_t n,
                                                           const uint32_t* __restrict__ block_offsets,
                                                           uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor,
                                                           uint32_t n_blocks) {
    __shared__ uint32_t sm[256];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * 8;
    uint32_t v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { v[k] = base + k < n ? counts[base + k] : 0u; s += v[k]; }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t t = threadIdx.x >= d ? sm[threadIdx.x - d] : 0;
        __syncthreads();
        sm[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = block_offsets[blockIdx.x] + (threadIdx.x ? sm[threadIdx.x - 1] : 0);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (base + k < n) { offsets[base + k] = run; cursor[base + k] = run; }
        run += v[k];
    }
    if (blockIdx.x == n_blocks - 1 && threadIdx.x == 255) offsets[n] = run;
}

}  // namespace atlas
