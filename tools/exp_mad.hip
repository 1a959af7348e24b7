// Issue rate of v_mad_u64_u32 (the multiplier of f9.hip.h), of a 32-bit v_mad_u32_u24 / v_add_u32 for comparison, and of one f9_mul,
// per wavefront, at 1 / 2 / 4 / 8 wavefronts per SIMD:   hipcc --offload-arch=gfx950 -O3 -I jolt-atlas_amd/csrc -I include tools/exp_mad.hip -o /tmp/exp_mad
#include <hip/hip_runtime.h>
#include <cstdio>
#include "f9.hip.h"
using namespace atlas;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(uint64_t* out, uint32_t a0, uint32_t b0, int iters) {
    uint64_t t[9];
    uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
    for (int j = 0; j < 9; j++) t[j] = j;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int j = 0; j < 9; j++) {
                if (MODE == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(t[j]) : "v"(a), "v"(b) : "vcc");
                else if (MODE == 1) { uint32_t lo = (uint32_t)t[j]; asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b)); t[j] = lo; }
                else if (MODE == 2) { uint32_t lo = (uint32_t)t[j]; asm volatile("v_add_u32 %0, %1, %0" : "+v"(lo) : "v"(a)); t[j] = lo; }
                else if (MODE == 4) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(t[j]) : "v"(t[(j + 1) % 9]));
                else if (MODE == 5) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(t[j]));
                else if (MODE == 6) asm volatile("v_ashrrev_i64 %0, 29, %0" : "+v"(t[j]));
                else if (MODE == 7) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(t[j]) : "v"(a), "v"(b) : "vcc");
                else if (MODE == 8) { uint32_t lo = (uint32_t)t[j]; asm volatile("v_mov_b32 %0, %1" : "=v"(lo) : "v"(a + (uint32_t)j)); t[j] = lo; }
                else if (MODE == 9) { uint32_t lo = (uint32_t)t[j]; asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(lo) : "v"(a)); t[j] = lo; }
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) s += t[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_mul(F9* out, const F9* in, int iters) {
    F9 a = in[threadIdx.x & 63], b = in[(threadIdx.x + 1) & 63];
    for (int i = 0; i < iters; i++) a = f9_mul<Fr9Params>(a, b);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int main() {
    uint64_t* out; F9* fo; F9* fi;
    CK(hipMalloc(&out, 8 * 256 * 4096)); CK(hipMalloc(&fo, sizeof(F9) * 256 * 4096)); CK(hipMalloc(&fi, sizeof(F9) * 64)); CK(hipMemset(fi, 1, sizeof(F9) * 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {                       // wavefronts per SIMD: 256 CUs * 4 SIMDs; a 256-thread workgroup is one wavefront per SIMD of its CU
        const unsigned grid = 256 * wps;
        static const char* names[] = {"v_mad_u64_u32", "v_mad_u32_u24", "v_add_u32", "f9_mul", "v_lshl_add_u64", "v_lshrrev_b64", "v_ashrrev_i64", "v_mad_i64_i32", "v_mov_b32", "v_mov_b32_dpp"};
        for (int mode = 0; mode < 10; mode++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                switch (mode) {
                    case 0: k_rate<0><<<grid, 256>>>(out, 3, 5, iters); break;
                    case 1: k_rate<1><<<grid, 256>>>(out, 3, 5, iters); break;
                    case 2: k_rate<2><<<grid, 256>>>(out, 3, 5, iters); break;
                    case 3: k_mul<<<grid, 256>>>(fo, fi, iters / 4); break;
                    case 4: k_rate<4><<<grid, 256>>>(out, 3, 5, iters); break;
                    case 5: k_rate<5><<<grid, 256>>>(out, 3, 5, iters); break;
                    case 6: k_rate<6><<<grid, 256>>>(out, 3, 5, iters); break;
                    case 7: k_rate<7><<<grid, 256>>>(out, 3, 5, iters); break;
                    case 8: k_rate<8><<<grid, 256>>>(out, 3, 5, iters); break;
                    default: k_rate<9><<<grid, 256>>>(out, 3, 5, iters); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double per_wave = mode != 3 ? (double)iters * 72 : iters / 4;       // instructions (or multiplications) per wavefront
            // cycles per instruction per SIMD at 2.4 GHz = time * clock / (instructions per wavefront * wavefronts per SIMD)
            printf("waves/SIMD %d  %-14s %8.3f ms  -> %7.2f cycles per wave-%s at 2.4 GHz\n", wps, names[mode], ms, ms * 1e-3 * 2.4e9 / (per_wave * wps),
                   mode != 3 ? "instruction" : "multiplication");
        }
    }
    return 0;
}
