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
