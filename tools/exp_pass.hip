// Experiment harness: time the sumcheck data-pass kernels in isolation.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I jolt-atlas_amd/csrc tools/exp_pass.hip -o tools/bin/exp_pass
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "sc_consts.hpp"
using namespace atlas;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    const ScConsts K = make_consts();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    Fr *L, *R, *part; ScCtx* cx;
    const size_t N = (size_t)1 << 22;
    CK(hipMalloc(&L, N * sizeof(Fr))); CK(hipMalloc(&R, N * sizeof(Fr))); CK(hipMalloc(&part, 4096 * 3 * sizeof(Fr))); CK(hipMalloc(&cx, sizeof(ScCtx)));
    std::vector<uint32_t> h(N * 8); for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345u) & ((i & 7) == 7 ? 0x0fffffffu : 0xffffffffu);
    CK(hipMemcpy(L, h.data(), N * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(R, h.data(), N * 32, hipMemcpyHostToDevice));
    ScCtx hc{}; for (int i = 4; i < 8; i++) hc.r.v[i] = 0x01234567u * (i + 1); hc.r.v[7] &= 0x0fffffffu;
    CK(hipMemcpy(cx, &hc, sizeof hc, hipMemcpyHostToDevice));
    EqView eq{}; eq.mode = EQ_NONE;
    for (int lg : {22, 20, 18, 16}) {
        const size_t len = (size_t)1 << lg, q = len / 4;
        for (int blocks : {512, 1024, 2048, 4096}) {
            int grid = (int)((q + 255) / 256); if (grid > blocks) grid = blocks;
            k_dot_bind_eval<2, Fr, false><<<grid, 256>>>(L, R, L, R, nullptr, eq, q, cx, part, K, 1); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) k_dot_bind_eval<2, Fr, false><<<grid, 256>>>(L, R, L, R, nullptr, eq, q, cx, part, K, 1);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("bind_eval<2> len 2^%d grid %5d: %8.2f us  %7.0f GB/s\n", lg, grid, ms / 5 * 1e3, 96.0 * len / (ms / 5) * 1e-6);
        }
    }
    {
        const size_t len = N, half = len / 2;
        for (int blocks : {1024, 2048, 4096}) {
            k_dot_eval<2, Fr><<<blocks, 256>>>(L, R, eq, half, part, K); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) k_dot_eval<2, Fr><<<blocks, 256>>>(L, R, eq, half, part, K);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("eval<2> len 2^22 grid %5d: %8.2f us  %7.0f GB/s\n", blocks, ms / 5 * 1e3, 64.0 * len / (ms / 5) * 1e-6);
        }
    }
    return 0;
}
