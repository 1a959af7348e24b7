// Self-test + throughput of f9.hip.h, and the F9 variant of the fused sumcheck pass.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "sc_consts.hpp"
#include "f9.hip.h"
using namespace atlas;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class P9>
__global__ void k_check(const Fe* in, int* bad) {
    using B = typename P9::Base;
    const size_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fe a = fe_load(in + (t & 1023)), b = fe_load(in + ((t * 7 + 3) & 1023)), c = fe_load(in + ((t * 13 + 5) & 1023));
    fe_cond_sub_p<B>(a.v); fe_cond_sub_p<B>(b.v); fe_cond_sub_p<B>(c.v);
    F9 A = f9_from_fe(a), Bb = f9_from_fe(b), C = f9_from_fe(c);
    // (1) product: canon_x32(mont9(a,b)) == fe_mul(a,b)
    if (!fe_eq(f9_canon_x32<P9>(f9_mul<P9>(A, Bb)), fe_mul<B>(a, b))) atomicAdd(bad, 1);
    // (2) lazy chain: ((a - b) + c) * (b + c)  with norm_red in between
    F9 d = f9_add(f9_sub<P9>(A, Bb), C);               // limbs < 2^31.2, value < 6p
    F9 e = f9_add(Bb, C);                              // limbs < 2^30
    F9 pr = f9_mul<P9>(f9_norm(d), e);
    Fe want = fe_mul<B>(fe_add<B>(fe_sub<B>(a, b), c), fe_add<B>(b, c));
    if (!fe_eq(f9_canon_x32<P9>(pr), want)) atomicAdd(bad + 1, 1);
    // (3) norm_red keeps the residue and bounds the value
    F9 s = f9_add(f9_add(A, Bb), C);                   // < 3p
    F9 n = f9_norm_red<P9>(s);
    if (!fe_eq(f9_canon<P9>(n), fe_add<B>(fe_add<B>(a, b), c))) atomicAdd(bad + 2, 1);
    if (n.l[8] > P9::TOP2P + 3) atomicAdd(bad + 3, 1);
    // (4) sparse multiplier pre-scaled by 32: f9_mul<4>(x, shl5(r)) == fe_mul_hi(x, r) (mod p), no x32 needed
    Fe rr = c; rr.v[0] = rr.v[1] = rr.v[2] = rr.v[3] = 0; rr.v[7] &= 0x1fffffffu;
    F9 r32 = f9_shl5(f9_from_fe(rr));
    if (!fe_eq(f9_canon<P9>(f9_mul<P9, 4>(d, r32)), fe_mul<B, 4>(fe_add<B>(fe_sub<B>(a, b), c), rr))) atomicAdd(bad + 4, 1);
    // (5) wave sum
    F9 ws = f9_wave_sum<P9>(A);
    Fe wf = a;
    for (int m = 32; m >= 1; m >>= 1) wf = fe_add<B>(wf, fe_shfl_xor(wf, m));
    if (!fe_eq(f9_canon<P9>(ws), wf)) atomicAdd(bad + 5, 1);
}

// F9 variant of the fused pass (degree 2, no eq): bind with r (pre-scaled), accumulate next message
__global__ __launch_bounds__(256) void k_pass_f9(Fe* L, Fe* R, size_t q, const ScCtx* cx, Fe* partials) {
    using P9 = Fr9Params;
    const F9 r32 = f9_shl5(f9_from_fe(fe_load(&cx->r)));
    F9 acc0 = f9_zero(), acc2 = f9_zero();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < q; i += (size_t)gridDim.x * 256) {
        F9 a0 = f9_load(L + i), a1 = f9_load(L + i + q), a2 = f9_load(L + i + 2 * q), a3 = f9_load(L + i + 3 * q);
        F9 l0 = f9_add(a0, f9_mul<P9, 4>(f9_sub<P9>(a2, a0), r32));
        F9 l1 = f9_add(a1, f9_mul<P9, 4>(f9_sub<P9>(a3, a1), r32));
        F9 b0 = f9_load(R + i), b1 = f9_load(R + i + q), b2 = f9_load(R + i + 2 * q), b3 = f9_load(R + i + 3 * q);
        F9 r0 = f9_add(b0, f9_mul<P9, 4>(f9_sub<P9>(b2, b0), r32));
        F9 r1 = f9_add(b1, f9_mul<P9, 4>(f9_sub<P9>(b3, b1), r32));
        l0 = f9_norm_red<P9>(l0); l1 = f9_norm_red<P9>(l1); r0 = f9_norm_red<P9>(r0); r1 = f9_norm_red<P9>(r1);
        fe_store(L + i, f9_to_fe(l0)); fe_store(L + i + q, f9_to_fe(l1));
        fe_store(R + i, f9_to_fe(r0)); fe_store(R + i + q, f9_to_fe(r1));
        F9 l2 = f9_norm(f9_add(l1, f9_sub<P9>(l1, l0))), r2 = f9_norm(f9_add(r1, f9_sub<P9>(r1, r0)));
        acc0 = f9_norm_red<P9>(f9_add(acc0, f9_mul<P9>(l0, r0)));
        acc2 = f9_norm_red<P9>(f9_add(acc2, f9_mul<P9>(l2, r2)));
    }
    __shared__ Fe red[4][2];
    acc0 = f9_wave_sum<P9>(acc0); acc2 = f9_wave_sum<P9>(acc2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = f9_canon_x32<P9>(acc0); red[wave][1] = f9_canon_x32<P9>(acc2); }
    __syncthreads();
    if (threadIdx.x < 2) {
        Fe s = red[0][threadIdx.x];
        for (int w = 1; w < 4; w++) s = fr_add(s, red[w][threadIdx.x]);
        fe_store(partials + (size_t)blockIdx.x * 2 + threadIdx.x, s);
    }
}

template <int VAR>
__global__ __launch_bounds__(256) void k_pass_f9v(Fe* L, Fe* R, size_t q, const ScCtx* cx, Fe* partials) {
    using P9 = Fr9Params;
    const F9 r32 = f9_shl5(f9_from_fe(fe_load(&cx->r)));
    F9 acc0 = f9_zero(), acc2 = f9_zero();
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    Fe x0, x1, x2, x3, y0, y1, y2, y3;
    if (i < q) { x0 = fe_load(L + i); x1 = fe_load(L + i + q); x2 = fe_load(L + i + 2 * q); x3 = fe_load(L + i + 3 * q);
                 y0 = fe_load(R + i); y1 = fe_load(R + i + q); y2 = fe_load(R + i + 2 * q); y3 = fe_load(R + i + 3 * q); }
    for (; i < q; i += stride) {
        F9 a0 = f9_from_fe(x0), a1 = f9_from_fe(x1), a2 = f9_from_fe(x2), a3 = f9_from_fe(x3);
        F9 b0 = f9_from_fe(y0), b1 = f9_from_fe(y1), b2 = f9_from_fe(y2), b3 = f9_from_fe(y3);
        const size_t nx = i + stride;
        if (VAR == 1 && nx < q) {      // prefetch the next iteration's operands before computing
            x0 = fe_load(L + nx); x1 = fe_load(L + nx + q); x2 = fe_load(L + nx + 2 * q); x3 = fe_load(L + nx + 3 * q);
            y0 = fe_load(R + nx); y1 = fe_load(R + nx + q); y2 = fe_load(R + nx + 2 * q); y3 = fe_load(R + nx + 3 * q);
        }
        F9 l0 = f9_add(a0, f9_mul<P9, 4>(f9_sub<P9>(a2, a0), r32));
        F9 l1 = f9_add(a1, f9_mul<P9, 4>(f9_sub<P9>(a3, a1), r32));
        F9 r0 = f9_add(b0, f9_mul<P9, 4>(f9_sub<P9>(b2, b0), r32));
        F9 r1 = f9_add(b1, f9_mul<P9, 4>(f9_sub<P9>(b3, b1), r32));
        l0 = f9_norm_red<P9>(l0); l1 = f9_norm_red<P9>(l1); r0 = f9_norm_red<P9>(r0); r1 = f9_norm_red<P9>(r1);
        if (VAR != 2) {
            fe_store(L + i, f9_to_fe(l0)); fe_store(L + i + q, f9_to_fe(l1));
            fe_store(R + i, f9_to_fe(r0)); fe_store(R + i + q, f9_to_fe(r1));
        } else {                        // compute only: feed results back instead of touching memory
            x0 = f9_to_fe(l0); x1 = f9_to_fe(l1); y0 = f9_to_fe(r0); y1 = f9_to_fe(r1);
        }
        F9 l2 = f9_norm(f9_add(l1, f9_sub<P9>(l1, l0))), r2 = f9_norm(f9_add(r1, f9_sub<P9>(r1, r0)));
        acc0 = f9_norm_red<P9>(f9_add(acc0, f9_mul<P9>(l0, r0)));
        acc2 = f9_norm_red<P9>(f9_add(acc2, f9_mul<P9>(l2, r2)));
    }
    __shared__ Fe red[4][2];
    acc0 = f9_wave_sum<P9>(acc0); acc2 = f9_wave_sum<P9>(acc2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = f9_canon_x32<P9>(acc0); red[wave][1] = f9_canon_x32<P9>(acc2); }
    __syncthreads();
    if (threadIdx.x < 2) {
        Fe s = red[0][threadIdx.x];
        for (int w = 1; w < 4; w++) s = fr_add(s, red[w][threadIdx.x]);
        fe_store(partials + (size_t)blockIdx.x * 2 + threadIdx.x, s);
    }
}

__global__ void k_sum_partials(const Fe* p, int n, Fe* out) {   // serial, exact
    if (threadIdx.x || blockIdx.x) return;
    Fe a = fe_zero(), b = fe_zero();
    for (int i = 0; i < n; i++) { a = fr_add(a, fe_load(p + 2 * i)); b = fr_add(b, fe_load(p + 2 * i + 1)); }
    fe_store(out, a); fe_store(out + 1, b);
}
__global__ void k_canon_cmp(const Fe* x, const Fe* y, size_t n, int* bad) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe a = fe_load(x + i), b = fe_load(y + i);
    Fe one = fr_one();
    a = fr_mul(a, one); b = fr_mul(b, one);     // canonical residues of both (x R R^-1)
    if (!fe_eq(a, b)) atomicAdd(bad, 1);
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    Fe* din; int* dbad; int hbad[8];
    CK(hipMalloc(&din, 1024 * sizeof(Fe))); CK(hipMalloc(&dbad, 32));
    std::vector<uint32_t> h(1024 * 8); for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 99) & ((i & 7) == 7 ? 0x1fffffffu : 0xffffffffu);
    CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dbad, 0, 32)); k_check<Fr9Params><<<64, 256>>>(din, dbad); CK(hipMemcpy(hbad, dbad, 32, hipMemcpyDeviceToHost));
    printf("Fr9 selftest mismatches: %d %d %d %d %d %d\n", hbad[0], hbad[1], hbad[2], hbad[3], hbad[4], hbad[5]);
    CK(hipMemset(dbad, 0, 32)); k_check<Fq9Params><<<64, 256>>>(din, dbad); CK(hipMemcpy(hbad, dbad, 32, hipMemcpyDeviceToHost));
    printf("Fq9 selftest mismatches: %d %d %d %d %d %d\n", hbad[0], hbad[1], hbad[2], hbad[3], hbad[4], hbad[5]);

    // fused pass: F9 variant vs the exact kernel (same inputs), compare sums + bound arrays
    const ScConsts K = make_consts();
    const size_t N = (size_t)1 << 22;
    Fe *L, *R, *L2, *R2, *part, *part2, *sums; ScCtx* cx;
    CK(hipMalloc(&L, N * 32)); CK(hipMalloc(&R, N * 32)); CK(hipMalloc(&L2, N * 32)); CK(hipMalloc(&R2, N * 32));
    CK(hipMalloc(&part, 4096 * 3 * 32)); CK(hipMalloc(&part2, 4096 * 3 * 32)); CK(hipMalloc(&sums, 4 * 32)); CK(hipMalloc(&cx, sizeof(ScCtx)));
    std::vector<uint32_t> hh(N * 8); for (size_t i = 0; i < hh.size(); i++) hh[i] = (uint32_t)(i * 2654435761u + 12345u) & ((i & 7) == 7 ? 0x0fffffffu : 0xffffffffu);
    ScCtx hc{}; for (int i = 4; i < 8; i++) hc.r.v[i] = 0x01234567u * (i + 1); hc.r.v[7] &= 0x0fffffffu;
    CK(hipMemcpy(cx, &hc, sizeof hc, hipMemcpyHostToDevice));
    EqView eq{}; eq.mode = EQ_NONE;
    for (int lg : {22, 20, 16}) {
        const size_t len = (size_t)1 << lg, q = len / 4;
        CK(hipMemcpy(L, hh.data(), len * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(R, hh.data() + 8 * 1000, len * 32, hipMemcpyHostToDevice));
        CK(hipMemcpy(L2, L, len * 32, hipMemcpyDeviceToDevice)); CK(hipMemcpy(R2, R, len * 32, hipMemcpyDeviceToDevice));
        int grid = (int)((q + 255) / 256); if (grid > 1024) grid = 1024;
        k_dot_bind_eval<2, Fr, false><<<grid, 256>>>(L, R, L, R, nullptr, eq, q, cx, part, K, 1);
        k_pass_f9<<<grid, 256>>>(L2, R2, q, cx, part2);
        k_sum_partials<<<1, 1>>>(part, grid, sums); k_sum_partials<<<1, 1>>>(part2, grid, sums + 2);
        Fe hs[4]; CK(hipMemcpy(hs, sums, 4 * 32, hipMemcpyDeviceToHost));
        CK(hipMemset(dbad, 0, 32));
        k_canon_cmp<<<(unsigned)((len / 2 + 255) / 256), 256>>>(L, L2, len / 2, dbad); k_canon_cmp<<<(unsigned)((len / 2 + 255) / 256), 256>>>(R, R2, len / 2, dbad + 1);
        CK(hipMemcpy(hbad, dbad, 32, hipMemcpyDeviceToHost));
        printf("2^%d: sums equal %d %d, bound arrays mismatches %d %d\n", lg, !memcmp(&hs[0], &hs[2], 32), !memcmp(&hs[1], &hs[3], 32), hbad[0], hbad[1]);
        for (int var = 1; var <= 2; var++) for (int blocks : {256, 512, 1024}) {
            int g2 = (int)((q + 255) / 256); if (g2 > blocks) g2 = blocks;
            if (var == 1) k_pass_f9v<1><<<g2, 256>>>(L2, R2, q, cx, part2); else k_pass_f9v<2><<<g2, 256>>>(L2, R2, q, cx, part2);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) { if (var == 1) k_pass_f9v<1><<<g2, 256>>>(L2, R2, q, cx, part2); else k_pass_f9v<2><<<g2, 256>>>(L2, R2, q, cx, part2); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  %s len 2^%d grid %5d: %8.2f us  %7.0f GB/s\n", var == 1 ? "prefetch    " : "compute-only", lg, g2, ms / 5 * 1e3, 96.0 * len / (ms / 5) * 1e-6);
        }
        for (int blocks : {512, 1024, 2048}) {
            int g2 = (int)((q + 255) / 256); if (g2 > blocks) g2 = blocks;
            k_pass_f9<<<g2, 256>>>(L2, R2, q, cx, part2); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) k_pass_f9<<<g2, 256>>>(L2, R2, q, cx, part2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  pass_f9 len 2^%d grid %5d: %8.2f us  %7.0f GB/s\n", lg, g2, ms / 5 * 1e3, 96.0 * len / (ms / 5) * 1e-6);
        }
    }
    return 0;
}
