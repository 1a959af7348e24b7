// Experiment: latency of the pieces of one transcript round on a single wavefront.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "sc_consts.hpp"
using namespace atlas;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_absorbs(DevTranscript* d, int n, ScConsts K) {
    const uint32_t lane = threadIdx.x & 63;
    if (threadIdx.x >= 64) return;
    const uint32_t vz = vgpr_zero();
    WaveTranscript T = wt_load(d, lane, vz);
    WaveBlakeSched W = wave_blake_sched(lane & 3);
    for (int i = 0; i < n; i++) wt_append_label(T, nullptr, W, lane, K.lbl_begin);
    wt_store(d, T, lane);
}
__global__ void k_field(Fr* io, int n, ScConsts K) {
    if (threadIdx.x >= 64) return;
    const uint32_t vz = vgpr_zero();
    Fr c0 = fe_load(io + vz), c1 = fe_load(io + 1 + vz), c2 = fe_load(io + 2 + vz), r = fe_load(io + 3 + vz);
    r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0;
    Fr acc = c0;
    for (int i = 0; i < n; i++) {
        Fr a = fs_from_mont(acc), b = fs_from_mont(c2);
        using P9 = Fr9Params;
        const F9 r32 = f9_shl5(f9_from_fe(r));
        F9 x = f9_from_fe(c2);
        x = f9_add(f9_mul<P9, 4>(x, r32), f9_from_fe(c1));
        x = f9_add(f9_mul<P9, 4>(x, r32), f9_from_fe(fr_add(a, b)));
        Fr ev = f9_to_fe(f9_norm(x));
        fe_cond_sub_p<FrParams>(ev.v); fe_cond_sub_p<FrParams>(ev.v); fe_cond_sub_p<FrParams>(ev.v);
        acc = fr_half(fr_add(fr_sub(fr_sub(ev, c1), c1), c2));
        c1 = fr_sub(fr_sub(c1, ev), acc);
    }
    if (threadIdx.x == 0) fe_store(io, acc);
}
__global__ void k_empty() {}

int main() {
    const ScConsts K = make_consts();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
    DevTranscript* d; Fr* io; CK(hipMalloc(&d, sizeof(DevTranscript))); CK(hipMalloc(&io, 4 * sizeof(Fr)));
    CK(hipMemset(d, 1, sizeof(DevTranscript))); CK(hipMemset(io, 3, 4 * sizeof(Fr)));
    for (int n : {1, 5, 50, 500}) {
        k_absorbs<<<1, 256>>>(d, n, K); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k_absorbs<<<1, 256>>>(d, n, K); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("absorbs n=%3d: %8.2f us total, %6.3f us each\n", n, ms * 1e3, ms * 1e3 / n);
    }
    for (int n : {1, 10, 100}) {
        k_field<<<1, 256>>>(io, n, K); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k_field<<<1, 256>>>(io, n, K); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("field-round n=%3d: %8.2f us total, %6.3f us each\n", n, ms * 1e3, ms * 1e3 / n);
    }
    CK(hipEventRecord(e0)); for (int i = 0; i < 100; i++) k_empty<<<1, 64>>>(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty launch chain: %.2f us each\n", ms * 10);
    return 0;
}
