// Einsum operand folds: i32 matrix x Fr vector.
//
// Device counterpart of `EinsumLayout::fold` for the standard matmul layout
//   jolt-atlas-core/src/onnx_proof/ops/einsum/mk_kn_mn.rs:47-79
//     left[j]  = sum_i from_i32(A[i*k + j]) * eq_m[i]      (column fold of the m x k operand)
//     right[j] = sum_h from_i32(B[j*n + h]) * eq_n[h]      (row fold of the k x n operand)
// which for GPT-2's logits (k = 768, n = 65536) reads 192 MB of i32 and is the largest single
// data pass of the IOP.  Bound: HBM (4 B per term); per term the work is nine signed
// multiply-adds: the Montgomery residue of eq is split into 29-bit limbs and a_ij * limb is
// accumulated in 64-bit columns (v_mad_i64_i32) with NO modular reduction in the loop — the
// sum of small-integer multiples of residues is reduced once per output (one REDC, one
// multiplication by 2^522 mod p, one canonicalisation).  Results equal the reference's
// field-element sums exactly (linear map, exact arithmetic).
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/atlas_hip.h"
#include "f9.hip.h"
#include "runtime.hpp"

using namespace atlas;
using atlas_rt::fail;
using atlas_rt::rt;

namespace {

constexpr int FOLD_THREADS = 256;
constexpr int FOLD_NORM_EVERY = 4;       // |a| < 2^31, limb < 2^29: 4 terms stay below 2^63

struct Cols {            // signed 64-bit columns of base 2^29
    int64_t c[9];
    int64_t top;         // overflow beyond limb 8 collected during normalisation (units of 2^261)
};

__device__ __forceinline__ void cols_zero(Cols& s) {
#pragma unroll
    for (int i = 0; i < 9; i++) s.c[i] = 0;
    s.top = 0;
}
__device__ __forceinline__ void cols_mad(Cols& s, int32_t a, const F9& e) {
#pragma unroll
    for (int i = 0; i < 9; i++) s.c[i] += (int64_t)a * (int64_t)(int32_t)e.l[i];
}
// signed carry propagation: columns -> [0, 2^29), carries into `top`
__device__ __forceinline__ void cols_norm(Cols& s) {
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c += s.c[i];
        s.c[i] = c & (int64_t)F9_MASK;
        c >>= 29;
    }
    s.top += c;
}

// S = sum c[i] 2^(29 i) + top 2^261  (|S| < p 2^55)  ->  F9 value congruent to S * 2^-261, < 2p
__device__ __forceinline__ F9 cols_redc(Cols s) {
    using P9 = Fr9Params;
    cols_norm(s);
    // make it non-negative: add p << 56 (12 normalized limbs), then split `top` into limbs 9..11
    constexpr uint32_t OFFS[12] = {0x00000000u, 0x08000000u, 0x1c000000u, 0x07c3eb27u, 0x1b970914u, 0x19f4243cu,
                                   0x056174a0u, 0x08b6d030u, 0x0029b850u, 0x1397098du, 0x000c1913u, 0x00000000u};
    uint64_t t[13];
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c += s.c[i] + (int64_t)OFFS[i];
        t[i] = (uint64_t)(c & (int64_t)F9_MASK);
        c >>= 29;
    }
    c += s.top;
#pragma unroll
    for (int i = 9; i < 12; i++) {
        c += (int64_t)OFFS[i];
        t[i] = (uint64_t)(c & (int64_t)F9_MASK);
        c >>= 29;
    }
    t[12] = (uint64_t)c;        // >= 0 by construction
    // nine radix-2^29 Montgomery steps over the low limbs
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t m = ((uint32_t)t[0] * P9::INV29) & F9_MASK;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)m * P9::p(j);
        const uint64_t carry = t[0] >> 29;
#pragma unroll
        for (int j = 0; j < 12; j++) t[j] = t[j + 1];
        t[0] += carry;
        t[12] = 0;
    }
    F9 o;
    uint64_t cc = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { cc += t[j]; o.l[j] = (uint32_t)cc & F9_MASK; cc >>= 29; }
    o.l[8] = (uint32_t)(cc + t[8]);
    return o;
}

// undo the 2^-261 of cols_redc and canonicalise: x * 2^522 * 2^-261
__device__ __forceinline__ Fe fold_finish(const F9& x) {
    using P9 = Fr9Params;
    constexpr uint32_t C522[9] = {0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu,
                                  0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au};
    F9 k;
#pragma unroll
    for (int i = 0; i < 9; i++) k.l[i] = C522[i];
    return f9_canon<P9>(f9_mul<P9>(x, k));
}

// out[row] = sum_h M[row*cols + h] * e[h] : one workgroup per row, lanes stride the row
__global__ __launch_bounds__(FOLD_THREADS) void k_fold_rows(const int32_t* __restrict__ M, const Fe* __restrict__ e,
                                                            size_t cols, Fe* __restrict__ out) {
    using P9 = Fr9Params;
    const int32_t* row = M + (size_t)blockIdx.x * cols;
    Cols s; cols_zero(s);
    int since = 0;
    for (size_t h = threadIdx.x; h < cols; h += FOLD_THREADS) {
        cols_mad(s, row[h], f9_load(e + h));
        if (++since == FOLD_NORM_EVERY) { cols_norm(s); since = 0; }
    }
    F9 v = f9_wave_sum<P9>(cols_redc(s));
    __shared__ F9 red[FOLD_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        F9 t = red[0];
        for (int w = 1; w < FOLD_THREADS / 64; w++) t = f9_norm_red<P9>(f9_add(t, red[w]));
        fe_store(out + blockIdx.x, fold_finish(t));
    }
}

// batched row fold (bmk/mbk/rbmk layouts): out[o0*t0 + o1*t1] = sum_{h < R} M[o0*s0 + o1*s1 + h] * e[h];
// one workgroup per (o0, o1), lanes stride the contiguous reduce axis
__global__ __launch_bounds__(FOLD_THREADS) void k_fold_rows_batched(const int32_t* __restrict__ M, const Fe* __restrict__ e,
                                                                    size_t n1, size_t s0, size_t s1, size_t R, size_t t0, size_t t1,
                                                                    Fe* __restrict__ out) {
    using P9 = Fr9Params;
    const size_t o0 = blockIdx.x / n1, o1 = blockIdx.x % n1;
    const int32_t* row = M + o0 * s0 + o1 * s1;
    Cols s; cols_zero(s);
    int since = 0;
    for (size_t h = threadIdx.x; h < R; h += FOLD_THREADS) {
        cols_mad(s, row[h], f9_load(e + h));
        if (++since == FOLD_NORM_EVERY) { cols_norm(s); since = 0; }
    }
    F9 v = f9_wave_sum<P9>(cols_redc(s));
    __shared__ F9 red[FOLD_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        F9 t = red[0];
        for (int w = 1; w < FOLD_THREADS / 64; w++) t = f9_norm_red<P9>(f9_add(t, red[w]));
        fe_store(out + o0 * t0 + o1 * t1, fold_finish(t));
    }
}

// batched column fold: out[z*tB + j*tC] = sum_{i < R} M[z*sB + i*sR + j] * e[i]; one thread per (z, j),
// j contiguous in memory so a wavefront reads 256 B per reduce step
__global__ __launch_bounds__(FOLD_THREADS) void k_fold_cols_batched(const int32_t* __restrict__ M, const Fe* __restrict__ e,
                                                                    size_t sB, size_t R, size_t sR, size_t C, size_t tB, size_t tC,
                                                                    size_t B1, size_t sB1, size_t tB1, Fe* __restrict__ out) {
    using P9 = Fr9Params;
    // blockIdx.y = z0 * B1 + z1: two batch axes with independent input and output strides (B1 = 1: one axis)
    const size_t j = (size_t)blockIdx.x * FOLD_THREADS + threadIdx.x, z0 = blockIdx.y / B1, z1 = blockIdx.y % B1;
    if (j >= C) return;
    const int32_t* base = M + z0 * sB + z1 * sB1 + j;
    Cols s; cols_zero(s);
    int since = 0;
    for (size_t i = 0; i < R; i++) {
        cols_mad(s, base[i * sR], f9_load(e + i));
        if (++since == FOLD_NORM_EVERY) { cols_norm(s); since = 0; }
    }
    fe_store(out + z0 * tB + z1 * tB1 + j * tC, fold_finish(f9_norm_red<P9>(cols_redc(s))));
}

// out[j] = sum_i M[i*cols + j] * e[i] : one thread per column j, rows split over blockIdx.y
// (partial[blockIdx.y][j] as F9-in-Fe; combined by k_fold_cols_finish)
__global__ __launch_bounds__(FOLD_THREADS) void k_fold_cols(const int32_t* __restrict__ M, const Fe* __restrict__ e,
                                                            size_t rows, size_t cols, size_t rows_per_slab,
                                                            Fe* __restrict__ partial) {
    const size_t j = (size_t)blockIdx.x * FOLD_THREADS + threadIdx.x;
    if (j >= cols) return;
    const size_t r0 = (size_t)blockIdx.y * rows_per_slab;
    const size_t r1 = r0 + rows_per_slab < rows ? r0 + rows_per_slab : rows;
    Cols s; cols_zero(s);
    int since = 0;
    for (size_t i = r0; i < r1; i++) {
        cols_mad(s, M[i * cols + j], f9_load(e + i));      // e[i] is wave-uniform: one broadcast load
        if (++since == FOLD_NORM_EVERY) { cols_norm(s); since = 0; }
    }
    fe_store(partial + (size_t)blockIdx.y * cols + j, f9_to_fe(f9_norm_red<Fr9Params>(cols_redc(s))));
}

__global__ __launch_bounds__(FOLD_THREADS) void k_fold_cols_finish(const Fe* __restrict__ partial, size_t cols,
                                                                   size_t n_slabs, Fe* __restrict__ out) {
    using P9 = Fr9Params;
    const size_t j = (size_t)blockIdx.x * FOLD_THREADS + threadIdx.x;
    if (j >= cols) return;
    F9 t = f9_load(partial + j);
    for (size_t sl = 1; sl < n_slabs; sl++) t = f9_norm_red<P9>(f9_add(t, f9_load(partial + sl * cols + j)));
    fe_store(out + j, fold_finish(t));
}

int make_poly(size_t len, atlas_poly_t* out) {
    atlas_poly* p = new atlas_poly();
    hipError_t e = hipMalloc(&p->d, len * sizeof(Fe));
    if (e != hipSuccess) { delete p; return fail(ATLAS_ENOMEM, "hipMalloc(fold)", e); }
    p->len = len; p->cap_bytes = len * sizeof(Fe); p->is_i32 = false; p->owned = true;
    *out = p;
    return ATLAS_OK;
}

inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }

}  // namespace

extern "C" {

int atlas_fold_i32_rows(const int32_t* d_matrix, size_t rows, size_t cols, atlas_poly_t eq, atlas_poly_t* out) {
    NEED_INIT();
    if (!d_matrix || !eq || !out || !is_pow2(rows) || cols == 0) return fail(ATLAS_EINVAL, "fold_i32_rows: rows must be a power of two");
    if (eq->is_i32 || eq->len != cols) return fail(ATLAS_EINVAL, "fold_i32_rows: eq table length != cols");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    int rc = make_poly(rows, out);
    if (rc) return rc;
    k_fold_rows<<<(unsigned)rows, FOLD_THREADS, 0, rt().stream>>>(d_matrix, (const Fe*)eq->d, cols, (Fe*)(*out)->d);
    HIP_TRY(hipStreamSynchronize(rt().stream));
    return ATLAS_OK;
}

int atlas_fold_i32_cols(const int32_t* d_matrix, size_t rows, size_t cols, atlas_poly_t eq, atlas_poly_t* out) {
    NEED_INIT();
    if (!d_matrix || !eq || !out || !is_pow2(cols) || rows == 0) return fail(ATLAS_EINVAL, "fold_i32_cols: cols must be a power of two");
    if (eq->is_i32 || eq->len != rows) return fail(ATLAS_EINVAL, "fold_i32_cols: eq table length != rows");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    int rc = make_poly(cols, out);
    if (rc) return rc;
    // split the rows into slabs so that ~2048 workgroups are in flight
    const size_t col_blocks = (cols + FOLD_THREADS - 1) / FOLD_THREADS;
    size_t n_slabs = 2048 / col_blocks;
    if (n_slabs < 1) n_slabs = 1;
    if (n_slabs > rows) n_slabs = rows;
    const size_t rows_per_slab = (rows + n_slabs - 1) / n_slabs;
    n_slabs = (rows + rows_per_slab - 1) / rows_per_slab;
    Fe* partial = nullptr;
    HIP_TRY(hipMalloc(&partial, n_slabs * cols * sizeof(Fe)));
    k_fold_cols<<<dim3((unsigned)col_blocks, (unsigned)n_slabs), FOLD_THREADS, 0, rt().stream>>>(
        d_matrix, (const Fe*)eq->d, rows, cols, rows_per_slab, partial);
    k_fold_cols_finish<<<(unsigned)col_blocks, FOLD_THREADS, 0, rt().stream>>>(partial, cols, n_slabs, (Fe*)(*out)->d);
    hipError_t e = hipStreamSynchronize(rt().stream);
    hipFree(partial);
    if (e != hipSuccess) return fail(ATLAS_ENODEV, "fold_i32_cols", e);
    return ATLAS_OK;
}

int atlas_fold_i32_rows_batched(const int32_t* d_matrix, size_t n0, size_t n1, size_t s0, size_t s1, size_t R, size_t t0,
                                size_t t1, atlas_poly_t eq, atlas_poly_t* out) {
    NEED_INIT();
    if (!d_matrix || !eq || !out || n0 == 0 || n1 == 0 || R == 0 || !is_pow2(n0 * n1)) return fail(ATLAS_EINVAL, "fold_i32_rows_batched: n0*n1 must be a power of two");
    if (eq->is_i32 || eq->len != R) return fail(ATLAS_EINVAL, "fold_i32_rows_batched: eq table length != R");
    if ((n0 - 1) * t0 + (n1 - 1) * t1 >= n0 * n1) return fail(ATLAS_EINVAL, "fold_i32_rows_batched: output strides leave the n0*n1 range");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    int rc = make_poly(n0 * n1, out);
    if (rc) return rc;
    k_fold_rows_batched<<<(unsigned)(n0 * n1), FOLD_THREADS, 0, rt().stream>>>(d_matrix, (const Fe*)eq->d, n1, s0, s1, R, t0, t1, (Fe*)(*out)->d);
    HIP_TRY(hipStreamSynchronize(rt().stream));
    return ATLAS_OK;
}

int atlas_fold_i32_cols_batched(const int32_t* d_matrix, size_t B, size_t sB, size_t R, size_t sR, size_t C, size_t tB, size_t tC,
                                atlas_poly_t eq, atlas_poly_t* out) {
    NEED_INIT();
    if (!d_matrix || !eq || !out || B == 0 || C == 0 || R == 0 || !is_pow2(B * C)) return fail(ATLAS_EINVAL, "fold_i32_cols_batched: B*C must be a power of two");
    if (eq->is_i32 || eq->len != R) return fail(ATLAS_EINVAL, "fold_i32_cols_batched: eq table length != R");
    if ((B - 1) * tB + (C - 1) * tC >= B * C) return fail(ATLAS_EINVAL, "fold_i32_cols_batched: output strides leave the B*C range");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    int rc = make_poly(B * C, out);
    if (rc) return rc;
    k_fold_cols_batched<<<dim3((unsigned)((C + FOLD_THREADS - 1) / FOLD_THREADS), (unsigned)B), FOLD_THREADS, 0, rt().stream>>>(
        d_matrix, (const Fe*)eq->d, sB, R, sR, C, tB, tC, 1, 0, 0, (Fe*)(*out)->d);
    HIP_TRY(hipStreamSynchronize(rt().stream));
    return ATLAS_OK;
}

int atlas_fold_i32_cols_batched2(const int32_t* d_matrix, size_t B0, size_t sB0, size_t tB0, size_t B1, size_t sB1, size_t tB1, size_t R,
                                 size_t sR, size_t C, size_t tC, atlas_poly_t eq, atlas_poly_t* out) {
    NEED_INIT();
    if (!d_matrix || !eq || !out || B0 == 0 || B1 == 0 || C == 0 || R == 0 || !is_pow2(B0 * B1 * C)) return fail(ATLAS_EINVAL, "fold_i32_cols_batched2: B0*B1*C must be a power of two");
    if (eq->is_i32 || eq->len != R) return fail(ATLAS_EINVAL, "fold_i32_cols_batched2: eq table length != R");
    if ((B0 - 1) * tB0 + (B1 - 1) * tB1 + (C - 1) * tC >= B0 * B1 * C) return fail(ATLAS_EINVAL, "fold_i32_cols_batched2: output strides leave the output range");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    int rc = make_poly(B0 * B1 * C, out);
    if (rc) return rc;
    k_fold_cols_batched<<<dim3((unsigned)((C + FOLD_THREADS - 1) / FOLD_THREADS), (unsigned)(B0 * B1)), FOLD_THREADS, 0, rt().stream>>>(
        d_matrix, (const Fe*)eq->d, sB0, R, sR, C, tB0, tC, B1, sB1, tB1, (Fe*)(*out)->d);
    HIP_TRY(hipStreamSynchronize(rt().stream));
    return ATLAS_OK;
}

// out[(r * repeat + q) * row_len + j] = base[r * row_len + j]: the broadcast of a folded operand over the batch axes it
// does not depend on (build_acbmk_kcn_cbmn's `right`, ops/einsum/rbmk_rbnk_bmn.rs:270-289)
__global__ __launch_bounds__(FOLD_THREADS) void k_repeat_rows(const Fe* __restrict__ base, size_t rows, size_t row_len, size_t repeat,
                                                              Fe* __restrict__ out) {
    const size_t total = rows * repeat * row_len;
    for (size_t i = (size_t)blockIdx.x * FOLD_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * FOLD_THREADS) {
        const size_t j = i % row_len, r = i / (row_len * repeat);
        fe_store(out + i, fe_load(base + r * row_len + j));
    }
}

int atlas_poly_repeat_rows(atlas_poly_t base, size_t rows, size_t row_len, size_t repeat, atlas_poly_t* out) {
    NEED_INIT();
    if (!base || !out || base->is_i32 || base->len != rows * row_len || repeat == 0 || !is_pow2(rows * row_len * repeat))
        return fail(ATLAS_EINVAL, "poly_repeat_rows: base must hold rows*row_len Fr and the output length be a power of two");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    int rc = make_poly(rows * row_len * repeat, out);
    if (rc) return rc;
    const size_t total = rows * row_len * repeat;
    size_t gb = (total + FOLD_THREADS - 1) / FOLD_THREADS; if (gb > 4096) gb = 4096;
    k_repeat_rows<<<(unsigned)gb, FOLD_THREADS, 0, rt().stream>>>((const Fe*)base->d, rows, row_len, repeat, (Fe*)(*out)->d);
    HIP_TRY(hipStreamSynchronize(rt().stream));
    return ATLAS_OK;
}


// EinsumLayout::fold for every layout the reference registers (ops/einsum/mod.rs:71-115): which strided fold each
// operand takes, with the transposes of utils/dims.rs:658-690 folded into the output strides.  dims per layout:
//   MK_KN_MN (m,k,n)  BMK_BKN_MBN / BMK_KBN_MBN / MBK_BNK_BMN / MBK_NBK_BMN (b,m,k,n)  K_NK_N (k,n)
//   ABMK_ABNK_ABMN (a,b,m,n,k)  ACBMK_KCN_CBMN (a,c,b,m,n,k)  CBMK_CBKN_AMN (cb,m,n,k)
int atlas_einsum_fold(int layout, const size_t* dims, size_t n_dims, const int32_t* d_left, const int32_t* d_right,
                      atlas_poly_t eq_r_m, atlas_poly_t eq_r_n, atlas_poly_t* left_out, atlas_poly_t* right_out) {
    PROF("atlas_einsum_fold");
    NEED_INIT();
    static const size_t want[] = {3, 4, 4, 4, 4, 2, 5, 6, 4};
    if (layout < 0 || layout > ATLAS_EINSUM_CBMK_CBKN_AMN || !dims || n_dims != want[layout]) return fail(ATLAS_EINVAL, "einsum_fold: layout / dims");
    if (!d_right || !eq_r_n || !right_out || (layout != ATLAS_EINSUM_K_NK_N && (!d_left || !eq_r_m || !left_out)))
        return fail(ATLAS_EINVAL, "einsum_fold: null argument");
    int rc = ATLAS_OK;
    atlas_poly_t lo = nullptr, ro = nullptr;
    switch (layout) {
        case ATLAS_EINSUM_MK_KN_MN: {                       // mk_kn_mn.rs:47-79
            const size_t m = dims[0], k = dims[1], n = dims[2];
            rc = atlas_fold_i32_cols(d_left, m, k, eq_r_m, &lo);
            if (!rc) rc = atlas_fold_i32_rows(d_right, k, n, eq_r_n, &ro);
            break;
        }
        case ATLAS_EINSUM_BMK_BKN_MBN: case ATLAS_EINSUM_BMK_KBN_MBN: {      // bmk_rhs_mbn.rs:78-110
            const size_t b = dims[0], m = dims[1], k = dims[2], n = dims[3];
            rc = atlas_fold_i32_cols_batched(d_left, b, k * m, m, k, k, 1, b, eq_r_m, &lo);                   // lo[h*k + j], transposed (b, k)
            if (!rc) rc = layout == ATLAS_EINSUM_BMK_KBN_MBN ? atlas_fold_i32_rows_batched(d_right, k, b, b * n, n, n, b, 1, eq_r_n, &ro)
                                                              : atlas_fold_i32_rows_batched(d_right, b, k, k * n, n, n, 1, b, eq_r_n, &ro);
            break;
        }
        case ATLAS_EINSUM_MBK_BNK_BMN: case ATLAS_EINSUM_MBK_NBK_BMN: {      // mbk_rhs_bmn.rs:78-119
            const size_t b = dims[0], m = dims[1], k = dims[2], n = dims[3];
            rc = atlas_fold_i32_cols_batched(d_left, b, k, m, k * b, k, k, 1, eq_r_m, &lo);                    // sum_i left[i*k*b + h*k + j]
            if (!rc) rc = layout == ATLAS_EINSUM_MBK_BNK_BMN ? atlas_fold_i32_cols_batched(d_right, b, n * k, n, k, k, k, 1, eq_r_n, &ro)
                                                              : atlas_fold_i32_cols_batched(d_right, b, k, n, k * b, k, k, 1, eq_r_n, &ro);
            break;
        }
        case ATLAS_EINSUM_K_NK_N: {                         // k_nk_n.rs:46-68: right[j] = sum_h B[h*k + j] eq[h]; the left operand is used as is
            const size_t k = dims[0], n = dims[1];
            rc = atlas_fold_i32_cols_batched(d_right, 1, 0, n, k, k, 0, 1, eq_r_n, &ro);
            break;
        }
        case ATLAS_EINSUM_ABMK_ABNK_ABMN: {                 // rbmk_rbnk_bmn.rs:163-217
            const size_t batch = dims[0] * dims[1], m = dims[2], n = dims[3], k = dims[4];
            rc = atlas_fold_i32_cols_batched(d_left, batch, m * k, m, k, k, k, 1, eq_r_m, &lo);
            if (!rc) rc = atlas_fold_i32_cols_batched(d_right, batch, n * k, n, k, k, k, 1, eq_r_n, &ro);
            break;
        }
        case ATLAS_EINSUM_ACBMK_KCN_CBMN: {                 // rbmk_rbnk_bmn.rs:219-290: left transposes the (a, cb) batch axes, right is broadcast over (b, a)
            const size_t a = dims[0], c = dims[1], b = dims[2], m = dims[3], n = dims[4], k = dims[5], cb = c * b;
            rc = atlas_fold_i32_cols_batched2(d_left, a, cb * m * k, k, cb, m * k, a * k, m, k, k, 1, eq_r_m, &lo);
            atlas_poly_t base = nullptr;
            if (!rc) rc = atlas_fold_i32_rows_batched(d_right, c, k, n, c * n, n, k, 1, eq_r_n, &base);
            if (!rc) rc = atlas_poly_repeat_rows(base, c, k, b * a, &ro);
            if (base) atlas_poly_free(base);
            break;
        }
        default: {                                          // cbmk,cbkn->amn (rbmk_rbnk_bmn.rs:292-338)
            const size_t cb = dims[0], m = dims[1], n = dims[2], k = dims[3];
            rc = atlas_fold_i32_cols_batched(d_left, cb, m * k, m, k, k, k, 1, eq_r_m, &lo);
            if (!rc) rc = atlas_fold_i32_rows_batched(d_right, cb * k, 1, n, 0, n, 1, 0, eq_r_n, &ro);
        }
    }
    if (rc) { if (lo) atlas_poly_free(lo); if (ro) atlas_poly_free(ro); return rc; }
    if (left_out) *left_out = lo;
    *right_out = ro;
    return ATLAS_OK;
}

// device int32 buffers for the operand tensors (Tensor<i32>): upload / free
int atlas_i32_upload(const int32_t* host, size_t n, int32_t** d_out) {
    NEED_INIT();
    if (!host || !d_out || n == 0) return fail(ATLAS_EINVAL, "i32_upload");
    int32_t* d = nullptr;
    hipError_t e = hipMalloc(&d, n * sizeof(int32_t));
    if (e != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(i32)", e);
    HIP_TRY(hipMemcpyAsync(d, host, n * sizeof(int32_t), hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    *d_out = d;
    return ATLAS_OK;
}

int atlas_i32_free(int32_t* d) {
    if (d) hipFree(d);
    return ATLAS_OK;
}

}  // extern "C"
