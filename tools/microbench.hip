// Micro-benchmarks that size the two ceilings of the sumcheck data pass on gfx950:
// 32-bit integer multiply-add throughput (v_mad_u64_u32), Montgomery multiplication
// throughput of field.hip.h, and the streaming bandwidth of the pass's access pattern.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I jolt-atlas_amd/csrc tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "field.hip.h"
using namespace atlas;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_mad(uint64_t* out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 1, b = blockIdx.x * 40503u + 7;
    uint64_t x0 = a, x1 = b, x2 = a ^ b, x3 = a + b, x4 = 5, x5 = 6, x6 = 7, x7 = 8;
    for (int i = 0; i < iters; i++) {
        x0 = (uint64_t)(uint32_t)x0 * a + x0; x1 = (uint64_t)(uint32_t)x1 * b + x1;
        x2 = (uint64_t)(uint32_t)x2 * a + x2; x3 = (uint64_t)(uint32_t)x3 * b + x3;
        x4 = (uint64_t)(uint32_t)x4 * a + x4; x5 = (uint64_t)(uint32_t)x5 * b + x5;
        x6 = (uint64_t)(uint32_t)x6 * a + x6; x7 = (uint64_t)(uint32_t)x7 * b + x7;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

template <int MODE>
__global__ void k_fr(Fr* out, const Fr* in, int iters) {
    const size_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fr x = fe_load(in + (t & 1023)), y = fe_load(in + ((t + 1) & 1023)), z = fe_load(in + ((t + 2) & 1023)), w = fe_load(in + ((t + 3) & 1023));
    Fr c = fe_load(in + 5); c.v[0] = c.v[1] = c.v[2] = c.v[3] = 0;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { x = fr_mul(x, y); y = fr_mul(y, z); z = fr_mul(z, w); w = fr_mul(w, x); }
        if (MODE == 1) { x = fr_mul_hi(x, c); y = fr_mul_hi(y, c); z = fr_mul_hi(z, c); w = fr_mul_hi(w, c); }
        if (MODE == 2) { x = fr_add(x, y); y = fr_sub(y, z); z = fr_add(z, w); w = fr_sub(w, x); }
    }
    fe_store(out + t, fr_add(fr_add(x, y), fr_add(z, w)));
}

// the fused pass's traffic shape without arithmetic: read i, i+q, i+2q, i+3q of two arrays, write i, i+q
__global__ void k_stream(Fr* L, Fr* R, size_t q) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
        Fr a0 = fe_load(L + i), a1 = fe_load(L + i + q), a2 = fe_load(L + i + 2 * q), a3 = fe_load(L + i + 3 * q);
        Fr b0 = fe_load(R + i), b1 = fe_load(R + i + q), b2 = fe_load(R + i + 2 * q), b3 = fe_load(R + i + 3 * q);
        for (int k = 0; k < 8; k++) { a0.v[k] ^= a2.v[k]; a1.v[k] ^= a3.v[k]; b0.v[k] ^= b2.v[k]; b1.v[k] ^= b3.v[k]; }
        fe_store(L + i, a0); fe_store(L + i + q, a1); fe_store(R + i, b0); fe_store(R + i + q, b1);
    }
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    const int blocks = 256 * 8, threads = 256;
    uint64_t* d64; CK(hipMalloc(&d64, (size_t)blocks * threads * 8));
    Fr *din, *dout; CK(hipMalloc(&din, 1024 * sizeof(Fr))); CK(hipMalloc(&dout, (size_t)blocks * threads * sizeof(Fr)));
    std::vector<uint32_t> h(1024 * 8); for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) & 0x0fffffffu;
    CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    {   // raw mad
        const int iters = 4096;
        k_mad<<<blocks, threads>>>(d64, 16); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k_mad<<<blocks, threads>>>(d64, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        double n = (double)blocks * threads * iters * 8;
        printf("v_mad_u64_u32: %.2f Tmad/s (%.3f ms)\n", n / ms * 1e-9, ms);
    }
    const char* names[3] = {"fr_mul (full CIOS)", "fr_mul_hi (challenge, LO=4)", "fr_add/sub"};
    for (int mode = 0; mode < 3; mode++) {
        const int iters = mode == 2 ? 2048 : 512;
        if (mode == 0) k_fr<0><<<blocks, threads>>>(dout, din, 4); if (mode == 1) k_fr<1><<<blocks, threads>>>(dout, din, 4); if (mode == 2) k_fr<2><<<blocks, threads>>>(dout, din, 4);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (mode == 0) k_fr<0><<<blocks, threads>>>(dout, din, iters); if (mode == 1) k_fr<1><<<blocks, threads>>>(dout, din, iters); if (mode == 2) k_fr<2><<<blocks, threads>>>(dout, din, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        double n = (double)blocks * threads * iters * 4;
        printf("%s: %.2f Gop/s (%.3f ms)\n", names[mode], n / ms * 1e-6, ms);
    }
    {   // stream
        const size_t len = (size_t)1 << 22, q = len / 4;
        Fr *L, *R; CK(hipMalloc(&L, len * sizeof(Fr))); CK(hipMalloc(&R, len * sizeof(Fr)));
        CK(hipMemset(L, 1, len * sizeof(Fr))); CK(hipMemset(R, 2, len * sizeof(Fr)));
        for (int g : {1024, 2048, 4096, 8192}) {
            k_stream<<<g, 256>>>(L, R, q); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); for (int r = 0; r < 10; r++) k_stream<<<g, 256>>>(L, R, q); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            double bytes = 10.0 * (2.0 * len * 32 + 2.0 * (len / 2) * 32);
            printf("stream pass shape 2^22 grid %d: %.0f GB/s (%.3f ms/pass)\n", g, bytes / ms * 1e-6, ms / 10);
        }
    }
    return 0;
}
