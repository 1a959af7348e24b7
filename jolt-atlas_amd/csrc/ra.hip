// The one-hot "ra" sumcheck instances of the lookup arguments, as
// host-stepped SumcheckInstanceProver objects (instance.hpp) over device-resident polynomials.
// Device counterparts of (paths under the jolt-atlas tree, joltworks/src/):
//   RaPolynomial                 poly/ra_poly.rs:21-110      ra(j) = F[idx_j] (0 for None)
//   RaSumcheckProver             subprotocols/ra_virtual.rs:97-145
//   compute_mles_product_sum     subprotocols/mles_product_sum.rs:15-131
//   BooleanitySumcheckProver     subprotocols/booleanity.rs:169-348
//   HammingWeightSumcheckProver  subprotocols/hamming_weight.rs:106-148
//
// Layout: the d polynomials of one instance live in one allocation, row i at i * stride, and are
// bound LowToHigh out of place between two buffers (T and T/2 coefficients per row).  The
// reference keeps RaPolynomial as (index, 16-entry table) for the first three binds to save host
// memory; the MLE it represents is the gathered vector, which is what is materialised here.
//
// Work split: everything that is O(T) runs in kernels (gather, product grid, booleanity fold,
// binds); everything that is O(d * 2^log_k_chunk) or O(degree^3) (phase-1 booleanity, hamming
// weight, Toom interpolation, Gruen's cubic) is host arithmetic between launches — those
// tables have 16 (at most 256) entries.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_poly.hpp"
#include "instance.hpp"
#include "runtime.hpp"
#include "sc_consts.hpp"
#include "spliteq_kernels.hip.h"
#include "f9.hip.h"

using namespace atlas;
namespace H = atlas_host;
using atlas_rt::fail;
using atlas_rt::g;

namespace {

constexpr int RA_THREADS = 256;
constexpr size_t RA_MAX_D = 16;

inline Fr to_dev(const H::Fr& a) { Fr o; std::memcpy(&o, &a, 32); return o; }

// ra_i[j] = idx_i[j] < 0 ? 0 : F_i[idx_i[j]]       (RaPolynomialRound1::get_bound_coeff)
__global__ __launch_bounds__(RA_THREADS) void k_ra_gather(const int32_t* __restrict__ idx, const Fr* __restrict__ F,
                                                          uint32_t f_stride, size_t T, Fr* __restrict__ out) {
    const uint32_t i = blockIdx.y;
    for (size_t j = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; j < T; j += (size_t)gridDim.x * RA_THREADS) {
        const int32_t k = idx[(size_t)i * T + j];
        fe_store(out + (size_t)i * T + j, k < 0 ? fe_zero() : fe_load(F + (size_t)i * f_stride + k));
    }
}

// compute_instruction_h_indices (shout.rs:532-547) / OneHotParams::lookup_index_chunk (config.rs:73-75):
// chunk i of a lookup index = (index >> (log_k_chunk * (d - 1 - i))) & (k_chunk - 1), i = 0 most significant
__global__ __launch_bounds__(RA_THREADS) void k_ra_chunk_indices(const uint64_t* __restrict__ lookups, size_t T, uint32_t d,
                                                                 uint32_t log_k_chunk, int32_t* __restrict__ out /* [d][T] */) {
    const uint64_t mask = ((uint64_t)1 << log_k_chunk) - 1;
    for (size_t j = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; j < T; j += (size_t)gridDim.x * RA_THREADS) {
        const uint64_t v = lookups[j];
        for (uint32_t i = 0; i < d; i++) {
            const uint32_t shift = log_k_chunk * (d - 1 - i);
            out[(size_t)i * T + j] = (int32_t)(shift >= 64 ? 0 : ((v >> shift) & mask));
        }
    }
}

// bind every row LowToHigh: dst[i][j] = src[i][2j] + r (src[i][2j+1] - src[i][2j])
__global__ __launch_bounds__(RA_THREADS) void k_ra_bind(const Fr* __restrict__ src, size_t src_stride, Fr* __restrict__ dst,
                                                        size_t dst_stride, size_t half, Fr r, int r_hi_only) {
    const Fr* s = src + (size_t)blockIdx.y * src_stride;
    Fr* d = dst + (size_t)blockIdx.y * dst_stride;
    for (size_t j = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; j < half; j += (size_t)gridDim.x * RA_THREADS)
        fe_store(d + j, bind_pair(fe_load(s + 2 * j), fe_load(s + 2 * j + 1), r, r_hi_only != 0));
}

__device__ __forceinline__ Fr gse_weight(const SplitEqView& E, size_t gidx) {
    return fr_mul(fe_load(E.e_out + (gidx >> E.in_bits)), fe_load(E.e_in + (gidx & (((size_t)1 << E.in_bits) - 1))));
}

// compute_mles_product_sum_evals_generic: per pair index g the product of the D lines
// p_i(X) = ra_i[2g] + X (ra_i[2g+1] - ra_i[2g]) on the grid [1, ..., D-1, inf], weighted by
// E_out * E_in.  One g per thread; the KN running products of grid columns [K0, K0 + KN) live in
// registers as 9 x 29-bit lazy limbs (f9.hip.h), so a launch covers at most 8 columns and D > 8
// takes two launches (the rows are re-read through L2).  Every f9_mul carries 2^-5 relative to
// the Montgomery radix: a stored sum is 32^-(D+1) times the true one, undone on the host.
template <int D, int K0, int KN>
__global__ __launch_bounds__(RA_THREADS) void k_ra_prod_f9(const Fr* __restrict__ ra, size_t stride, SplitEqView E,
                                                           size_t n_groups, Fr* __restrict__ partials) {
    using P9 = Fr9Params;
    const size_t gidx = (size_t)blockIdx.x * RA_THREADS + threadIdx.x;
    F9 prod[KN];
#pragma unroll
    for (int k = 0; k < KN; k++) prod[k] = f9_zero();
    if (gidx < n_groups) {
        const size_t mask = ((size_t)1 << E.in_bits) - 1;
        const F9 wgt = f9_mul<P9>(f9_load(E.e_out + (gidx >> E.in_bits)), f9_load(E.e_in + (gidx & mask)));
        // iteration D multiplies the weight in (kept inside the one loop: the running products stay in VGPRs)
#pragma unroll 1
        for (int i = 0; i <= D; i++) {
            const Fr* row = ra + (size_t)(i < D ? i : 0) * stride + 2 * gidx;
            const F9 a0 = f9_load(row), a1 = f9_load(row + 1);
            const F9 dl = f9_norm_red<P9, 2>(f9_sub<P9>(a1, a0));          // a1 - a0 (+4p), < 2.1p
            F9 cur = a0;
#pragma unroll
            for (int s = 0; s < K0 + 1; s++) cur = f9_norm_red<P9, 2>(f9_add(cur, dl));   // p_i(K0 + 1)
#pragma unroll
            for (int k = 0; k < KN; k++) {
                const bool inf = (K0 + k == D - 1);                             // column D-1 = X -> inf
                const F9 val = i == D ? wgt : (inf ? dl : cur);
                if (!inf) cur = f9_norm_red<P9, 2>(f9_add(cur, dl));
                prod[k] = i == 0 ? val : f9_mul<P9>(prod[k], val);
            }
        }
    }
    __shared__ F9 red9[RA_THREADS / 64][KN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KN; k++) {
        const F9 sres = f9_wave_sum<P9>(prod[k]);
        if (lane == 0) red9[wave][k] = sres;
    }
    __syncthreads();
    if (threadIdx.x < KN) {
        F9 sres = red9[0][threadIdx.x];
        for (int w = 1; w < RA_THREADS / 64; w++) sres = f9_norm_red<P9>(f9_add(sres, red9[w][threadIdx.x]));
        fe_store(partials + (size_t)blockIdx.x * D + K0 + threadIdx.x, f9_canon<P9>(sres));
    }
}

// the same sums with one grid column per thread (blockIdx.y = column): for short instances the chain of D
// multiplications per column is the latency of the round, so the columns go to different threads
template <int D>
__global__ __launch_bounds__(RA_THREADS) void k_ra_prod_f9_col(const Fr* __restrict__ ra, size_t stride, SplitEqView E,
                                                               size_t n_groups, Fr* __restrict__ partials /* [gridDim.x][D] */) {
    using P9 = Fr9Params;
    const size_t gidx = (size_t)blockIdx.x * RA_THREADS + threadIdx.x;
    const int k = blockIdx.y;
    F9 prod = f9_zero();
    if (gidx < n_groups) {
        const size_t mask = ((size_t)1 << E.in_bits) - 1;
        prod = f9_mul<P9>(f9_load(E.e_out + (gidx >> E.in_bits)), f9_load(E.e_in + (gidx & mask)));
#pragma unroll 1
        for (int i = 0; i < D; i++) {
            const Fr* row = ra + (size_t)i * stride + 2 * gidx;
            const F9 a0 = f9_load(row), a1 = f9_load(row + 1);
            const F9 dl = f9_norm_red<P9, 2>(f9_sub<P9>(a1, a0));
            F9 val = dl;                                                     // column D-1: X -> inf
            if (k != D - 1) {
                val = a0;
                for (int s2 = 0; s2 < k + 1; s2++) val = f9_norm_red<P9, 2>(f9_add(val, dl));   // p_i(k + 1)
            }
            prod = f9_mul<P9>(prod, val);
        }
    }
    __shared__ F9 red9[RA_THREADS / 64];
    const F9 sres = f9_wave_sum<P9>(prod);
    if ((threadIdx.x & 63) == 0) red9[threadIdx.x >> 6] = sres;
    __syncthreads();
    if (threadIdx.x == 0) {
        F9 t = red9[0];
        for (int w = 1; w < RA_THREADS / 64; w++) t = f9_norm_red<P9>(f9_add(t, red9[w]));
        fe_store(partials + (size_t)blockIdx.x * D + k, f9_canon<P9>(t));
    }
}

// booleanity phase 2 (booleanity.rs:254-276): per pair index j
//   c = sum_i gamma_i h0 (h0 - 1),  e = sum_i gamma_i (h1 - h0)^2, folded with E_out * E_in
__global__ __launch_bounds__(RA_THREADS) void k_bool_fold(const Fr* __restrict__ Hp, size_t stride, uint32_t d,
                                                          const Fr* __restrict__ gammas, SplitEqView E, size_t n_groups,
                                                          Fr* __restrict__ partials) {
    Fr acc[2];
    acc[0] = fe_zero(); acc[1] = fe_zero();
    const Fr one = fr_one();
    for (size_t j = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; j < n_groups; j += (size_t)gridDim.x * RA_THREADS) {
        Fr c = fe_zero(), e = fe_zero();
        // blockIdx.y splits the d rows (the sums are additive): short instances put one row per thread
        const uint32_t i0 = (uint32_t)(((uint64_t)d * blockIdx.y) / gridDim.y), i1 = (uint32_t)(((uint64_t)d * (blockIdx.y + 1)) / gridDim.y);
        for (uint32_t i = i0; i < i1; i++) {
            const Fr* row = Hp + (size_t)i * stride;
            const Fr h0 = fe_load(row + 2 * j), h1 = fe_load(row + 2 * j + 1);
            const Fr b = fr_sub(h1, h0);
            const Fr gm = fe_load(gammas + i);
            c = fr_add(c, fr_mul(fr_mul(gm, h0), fr_sub(h0, one)));
            e = fr_add(e, fr_mul(fr_mul(gm, b), b));
        }
        const Fr wgt = gse_weight(E, j);
        acc[0] = fr_add(acc[0], fr_mul(wgt, c));
        acc[1] = fr_add(acc[1], fr_mul(wgt, e));
    }
    block_reduce_store<2>(acc, partials + (size_t)blockIdx.y * gridDim.x * 2);
}

// out[k] = sum_p partials[p * K + k]; one workgroup per column
__global__ __launch_bounds__(RA_THREADS) void k_col_reduce(const Fr* __restrict__ partials, uint32_t n_partials, uint32_t K,
                                                           Fr* __restrict__ out) {
    __shared__ Fr red[RA_THREADS / 64];
    const uint32_t k = blockIdx.x;
    Fr acc = fe_zero();
    for (uint32_t p = threadIdx.x; p < n_partials; p += RA_THREADS) acc = fr_add(acc, fe_load(partials + (size_t)p * K + k));
    acc = fr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Fr s = red[0];
        for (int w = 1; w < RA_THREADS / 64; w++) s = fr_add(s, red[w]);
        fe_store(out + k, s);
    }
}

// device half of a LowToHigh GruenSplitEqPolynomial: the cached prefix tables
struct GseDev {
    H::GseState st;
    Fr *d_w = nullptr, *d_eout = nullptr, *d_ein = nullptr;
    int init(const H::Fr* w, size_t n) {
        st.init(w, n);
        if (st.k_out > 12 || st.k_in > 12) return fail(ATLAS_EINVAL, "split-eq: more than 25 variables not supported");
        HIP_TRY(hipMalloc(&d_w, (n ? n : 1) * sizeof(Fr)));
        HIP_TRY(hipMalloc(&d_eout, ((size_t)2 << st.k_out) * sizeof(Fr)));
        HIP_TRY(hipMalloc(&d_ein, ((size_t)2 << st.k_in) * sizeof(Fr)));
        if (n) HIP_TRY(hipMemcpyAsync(d_w, w, n * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
        k_eq_cached<<<1, 1024, 0, g.stream>>>(d_eout, d_w, (uint32_t)st.k_out);
        k_eq_cached<<<1, 1024, 0, g.stream>>>(d_ein, d_w + st.m, (uint32_t)st.k_in);
        HIP_TRY(hipStreamSynchronize(g.stream));
        return ATLAS_OK;
    }
    SplitEqView view() const {
        SplitEqView E;
        E.e_out = d_eout + (((size_t)1 << st.out_top) - 1);
        E.e_in = d_ein + (((size_t)1 << st.in_top) - 1);
        E.in_bits = (uint32_t)st.in_top;
        return E;
    }
    void release() { if (d_w) hipFree(d_w); if (d_eout) hipFree(d_eout); if (d_ein) hipFree(d_ein); d_w = d_eout = d_ein = nullptr; }
};

// d rows of one instance, ping-pong bound
struct RaRows {
    size_t d = 0, len = 0;
    Fr* buf[2] = {nullptr, nullptr};
    size_t stride[2] = {0, 0};
    int cur = 0;
    Fr* partials = nullptr;     // ceil(T/2 / RA_THREADS) * max(d, 2) Fr
    Fr* d_sums = nullptr;       // max(d, 2) Fr
    size_t K = 0;

    int alloc(size_t d_, size_t T) {
        d = d_; len = T; K = d > 2 ? d : 2;
        HIP_TRY(hipMalloc(&buf[0], d * T * sizeof(Fr)));
        HIP_TRY(hipMalloc(&buf[1], d * (T > 1 ? T / 2 : 1) * sizeof(Fr)));
        stride[0] = T; stride[1] = T > 1 ? T / 2 : 1;
        const size_t blocks = (T / 2 + RA_THREADS - 1) / RA_THREADS + 1;
        HIP_TRY(hipMalloc(&partials, (blocks * K > 4096 ? blocks * K : 4096) * sizeof(Fr)));   // room for the row-split launches of short instances
        HIP_TRY(hipMalloc(&d_sums, K * sizeof(Fr)));
        return ATLAS_OK;
    }
    // indices: d host rows of T int32 -> one device allocation (kept until the gather)
    int32_t* d_idx = nullptr;
    int upload_indices(const int32_t* const* H_indices) {
        HIP_TRY(hipMalloc(&d_idx, d * len * sizeof(int32_t)));
        for (size_t i = 0; i < d; i++)
            HIP_TRY(hipMemcpyAsync(d_idx + i * len, H_indices[i], len * sizeof(int32_t), hipMemcpyHostToDevice, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        return ATLAS_OK;
    }
    // the same from the T lookup indices themselves: the d chunk rows are cut on the device (8 T bytes over PCIe
    // instead of 4 d T)
    int upload_lookups(const uint64_t* lookups, uint32_t log_k_chunk) {
        uint64_t* d_l = nullptr;
        HIP_TRY(hipMalloc(&d_idx, d * len * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&d_l, len * sizeof(uint64_t)));
        hipError_t e = hipMemcpyAsync(d_l, lookups, len * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream);
        if (e == hipSuccess) {
            size_t gb = (len + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
            k_ra_chunk_indices<<<(unsigned)gb, RA_THREADS, 0, g.stream>>>(d_l, len, (uint32_t)d, log_k_chunk, d_idx);
            e = hipStreamSynchronize(g.stream);
        }
        hipFree(d_l);
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "ra lookups upload", e);
        return ATLAS_OK;
    }
    // ra_i[j] = table_i[idx_i[j]] from device tables (d rows of f_stride Fr; f_stride 0 = shared table)
    int gather(const Fr* d_tables, uint32_t f_stride) {
        if (!d_idx) return fail(ATLAS_ESTATE, "ra gather: indices not uploaded");
        size_t gb = (len + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
        k_ra_gather<<<dim3((unsigned)gb, (unsigned)d), RA_THREADS, 0, g.stream>>>(d_idx, d_tables, f_stride, len, buf[0]);
        hipError_t e = hipStreamSynchronize(g.stream);
        hipFree(d_idx); d_idx = nullptr;
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "ra gather", e);
        cur = 0; stride[0] = len;
        return ATLAS_OK;
    }
    int bind(const atlas_u128_t& r) {
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, g.challenge_mode);
        const size_t half = len / 2;
        const int nxt = cur ^ 1;
        stride[nxt] = half;
        size_t gb = (half + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096; if (gb < 1) gb = 1;
        k_ra_bind<<<dim3((unsigned)gb, (unsigned)d), RA_THREADS, 0, g.stream>>>(buf[cur], stride[cur], buf[nxt], stride[nxt], half,
                                                                               to_dev(rf), g.challenge_mode == 0 ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "ra bind", e);
        cur = nxt; len = half;
        return ATLAS_OK;
    }
    int reduce_to_host(uint32_t n_partials, uint32_t k, H::Fr* out) {
        if (n_partials > 1) k_col_reduce<<<k, RA_THREADS, 0, g.stream>>>(partials, n_partials, k, d_sums);
        HIP_TRY(hipMemcpyAsync(g.h_pinned, n_partials > 1 ? d_sums : partials, k * sizeof(Fr), hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        std::memcpy(out, g.h_pinned, k * sizeof(Fr));
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) {
        if (len != 1) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        out.resize(d);
        for (size_t i = 0; i < d; i++)
            HIP_TRY(hipMemcpyAsync((uint8_t*)g.h_pinned + 32 * i, buf[cur] + i * stride[cur], sizeof(Fr), hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        std::memcpy(out.data(), g.h_pinned, d * sizeof(Fr));
        return ATLAS_OK;
    }
    void release() { for (auto& b : buf) if (b) hipFree(b); if (partials) hipFree(partials); if (d_sums) hipFree(d_sums); if (d_idx) hipFree(d_idx); buf[0] = buf[1] = partials = d_sums = nullptr; d_idx = nullptr; }
};

template <int D>
void launch_prod(const RaRows& R, const SplitEqView& E, size_t n_groups, unsigned blocks) {
    if (n_groups <= ((size_t)1 << 13)) {      // latency regime: one column per thread
        k_ra_prod_f9_col<D><<<dim3(blocks, D), RA_THREADS, 0, g.stream>>>(R.buf[R.cur], R.stride[R.cur], E, n_groups, R.partials);
        return;
    }
    constexpr int KA = D < 8 ? D : 8;
    k_ra_prod_f9<D, 0, KA><<<blocks, RA_THREADS, 0, g.stream>>>(R.buf[R.cur], R.stride[R.cur], E, n_groups, R.partials);
    if constexpr (D > 8)
        k_ra_prod_f9<D, 8, D - 8><<<blocks, RA_THREADS, 0, g.stream>>>(R.buf[R.cur], R.stride[R.cur], E, n_groups, R.partials);
}

// ---------------------------------------------------------------- RaSumcheckProver
struct RaVirtual : atlas_instance {
    RaRows rows;
    GseDev eq;
    size_t log_T = 0, round_next = 0;
    ~RaVirtual() override { rows.release(); eq.release(); }
    size_t rounds() const override { return log_T; }
    size_t degree() const override { return rows.d + 1; }
    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= log_T) return fail(ATLAS_ESTATE, "ra_virtual: round out of order");
        std::lock_guard<std::mutex> lk(g.mu);
        const size_t n_groups = rows.len / 2;
        const unsigned blocks = (unsigned)((n_groups + RA_THREADS - 1) / RA_THREADS);
        const SplitEqView E = eq.view();
        switch (rows.d) {
#define RA_CASE(D) case D: launch_prod<D>(rows, E, n_groups, blocks); break;
            RA_CASE(1) RA_CASE(2) RA_CASE(3) RA_CASE(4) RA_CASE(5) RA_CASE(6) RA_CASE(7) RA_CASE(8)
            RA_CASE(9) RA_CASE(10) RA_CASE(11) RA_CASE(12) RA_CASE(13) RA_CASE(14) RA_CASE(15) RA_CASE(16)
#undef RA_CASE
            default: return fail(ATLAS_EINVAL, "ra_virtual: d > 16");
        }
        std::vector<H::Fr> sums(rows.d);
        int rc = rows.reduce_to_host(blocks, (uint32_t)rows.d, sums.data());
        if (rc) return rc;
        H::Fr fix = H::one();                                        // 32^(d+1): the 2^-5 per f9_mul
        for (size_t k = 0; k < rows.d + 1; k++) fix = H::mul(fix, H::from_u64(32));
        fix = H::mul(fix, eq.st.scalar);                             // mles_product_sum.rs:131
        for (auto& s : sums) s = H::mul(s, fix);
        coeffs = H::finish_product_sum(sums, claim, eq.st);
        return ATLAS_OK;
    }
    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= log_T) return fail(ATLAS_ESTATE, "ra_virtual: round out of order");
        std::lock_guard<std::mutex> lk(g.mu);
        int rc = rows.bind(r);
        if (rc) return rc;
        eq.st.bind(H::challenge_to_fr(r.lo, r.hi, g.challenge_mode));
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override { std::lock_guard<std::mutex> lk(g.mu); return rows.finals(out); }
};

// upload d host tables of K Fr each
int upload_tables(const std::vector<std::vector<H::Fr>>& t, size_t K, Fr** out) {
    Fr* d = nullptr;
    HIP_TRY(hipMalloc(&d, t.size() * K * sizeof(Fr)));
    for (size_t i = 0; i < t.size(); i++)
        HIP_TRY(hipMemcpyAsync(d + i * K, t[i].data(), K * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    *out = d;
    return ATLAS_OK;
}

// ---------------------------------------------------------------- BooleanitySumcheckProver
struct Booleanity : atlas_instance {
    size_t d = 0, log_k = 0, log_T = 0, round_next = 0;
    std::vector<std::vector<H::Fr>> G;                 // d x 2^log_k (host: 16 entries each)
    std::vector<H::Fr> gammas, F;                      // F = ExpandingTable values
    H::GseState B;
    std::vector<std::vector<H::Fr>> B_out, B_in;       // host prefix tables of B
    GseDev D;
    RaRows rows;
    Fr* d_gammas = nullptr;
    H::Fr eq_r_r = H::zero(), eq_r_r_inv = H::zero();
    bool have_eq_r_r_inv = false;
    ~Booleanity() override { rows.release(); D.release(); if (d_gammas) hipFree(d_gammas); }
    size_t rounds() const override { return log_k + log_T; }
    size_t degree() const override { return 3; }

    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "booleanity: round out of order");
        coeffs.assign(4, H::zero());
        if (round < log_k) {                                         // compute_phase1_message
            const size_t m = round + 1;
            const auto& e_out = B_out[B.out_top]; const auto& e_in = B_in[B.in_top];
            H::Fr q0 = H::zero(), qinf = H::zero();
            for (size_t xo = 0; xo < e_out.size(); xo++) {
                H::Fr i0 = H::zero(), i1 = H::zero();
                for (size_t xi = 0; xi < e_in.size(); xi++) {
                    const size_t kp = (xo << B.in_top) | xi;
                    H::Fr c0 = H::zero(), c1 = H::zero();
                    for (size_t i = 0; i < d; i++) {
                        H::Fr s0 = H::zero(), s1 = H::zero();
                        for (size_t k = 0; k < ((size_t)1 << m); k++) {
                            const H::Fr& Gk = G[i][(kp << m) + k];
                            const H::Fr& Fk = F[k % ((size_t)1 << (m - 1))];
                            const H::Fr gf = H::mul(Gk, Fk), ei = H::mul(gf, Fk);
                            if ((k >> (m - 1)) == 0) s0 = H::add(s0, H::sub(ei, gf));
                            s1 = H::add(s1, ei);
                        }
                        c0 = H::add(c0, H::mul(gammas[i], s0)); c1 = H::add(c1, H::mul(gammas[i], s1));
                    }
                    i0 = H::add(i0, H::mul(e_in[xi], c0)); i1 = H::add(i1, H::mul(e_in[xi], c1));
                }
                q0 = H::add(q0, H::mul(e_out[xo], i0)); qinf = H::add(qinf, H::mul(e_out[xo], i1));
            }
            H::gruen_deg3(B, q0, qinf, claim, coeffs.data());
            return ATLAS_OK;
        }
        std::lock_guard<std::mutex> lk(g.mu);                        // compute_phase2_message
        const size_t n_groups = rows.len / 2;
        size_t blocks = (n_groups + RA_THREADS - 1) / RA_THREADS; if (blocks > 2048) blocks = 2048;
        const unsigned ysplit = n_groups <= ((size_t)1 << 13) ? (unsigned)d : 1u;   // latency regime: one row per thread
        k_bool_fold<<<dim3((unsigned)blocks, ysplit), RA_THREADS, 0, g.stream>>>(rows.buf[rows.cur], rows.stride[rows.cur], (uint32_t)d, d_gammas,
                                                                             D.view(), n_groups, rows.partials);
        H::Fr s[2];
        int rc = rows.reduce_to_host((uint32_t)(blocks * ysplit), 2, s);
        if (rc) return rc;
        if (!have_eq_r_r_inv) { eq_r_r_inv = H::inv(eq_r_r); have_eq_r_r_inv = true; }   // constant over phase 2
        const H::Fr adj = H::mul(claim, eq_r_r_inv);
        H::gruen_deg3(D.st, s[0], s[1], adj, coeffs.data());
        for (auto& c : coeffs) c = H::mul(c, eq_r_r);                // gruen_poly * eq_r_r (from_coeff)
        H::trim(coeffs);
        return ATLAS_OK;
    }

    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "booleanity: round out of order");
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, g.challenge_mode);
        if (round < log_k) {
            B.bind(rf);
            const size_t n = F.size();                               // ExpandingTable::update, LowToHigh
            F.resize(2 * n);
            for (size_t x = 0; x < n; x++) { F[n + x] = H::mul(F[x], rf); F[x] = H::sub(F[x], F[n + x]); }
            if (round == log_k - 1) {
                std::lock_guard<std::mutex> lk(g.mu);
                eq_r_r = B.scalar;
                Fr* d_F = nullptr;
                HIP_TRY(hipMalloc(&d_F, F.size() * sizeof(Fr)));
                HIP_TRY(hipMemcpyAsync(d_F, F.data(), F.size() * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
                int rc = rows.gather(d_F, 0);                         // every H_i reads the same table F
                hipFree(d_F);
                if (rc) return rc;
                G.clear();
            }
        } else {
            std::lock_guard<std::mutex> lk(g.mu);
            D.st.bind(rf);
            int rc = rows.bind(r);
            if (rc) return rc;
        }
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override { std::lock_guard<std::mutex> lk(g.mu); return rows.finals(out); }
};

// ---------------------------------------------------------------- HammingWeightSumcheckProver (host: d x 2^log_k)
struct HammingWeight : atlas_instance {
    size_t log_k = 0, round_next = 0;
    std::vector<std::vector<H::Fr>> ra;
    std::vector<H::Fr> gamma_powers;
    size_t rounds() const override { return log_k; }
    size_t degree() const override { return 1; }
    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= log_k) return fail(ATLAS_ESTATE, "hamming_weight: round out of order");
        H::Fr e0 = H::zero();
        for (size_t i = 0; i < ra.size(); i++) {
            H::Fr s = H::zero();
            for (size_t k = 0; k < ra[i].size() / 2; k++) s = H::add(s, ra[i][2 * k]);
            e0 = H::add(e0, H::mul(s, gamma_powers[i]));
        }
        const H::Fr e1 = H::sub(claim, e0);
        coeffs = {e0, H::sub(e1, e0)};                                // from_evals (2 points) -> from_coeff
        H::trim(coeffs);
        return ATLAS_OK;
    }
    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= log_k) return fail(ATLAS_ESTATE, "hamming_weight: round out of order");
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, g.challenge_mode);
        for (auto& p : ra) {
            const size_t half = p.size() / 2;
            for (size_t k = 0; k < half; k++) p[k] = H::add(p[2 * k], H::mul(rf, H::sub(p[2 * k + 1], p[2 * k])));
            p.resize(half);
        }
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override {
        out.clear();
        for (auto& p : ra) { if (p.size() != 1) return fail(ATLAS_ESTATE, "final_claims: rounds remaining"); out.push_back(p[0]); }
        return ATLAS_OK;
    }
};

// ---------------------------------------------------------------- prefix-suffix Shout read-raf (unary, ReLU)
// ReadRafSumcheckProver over a unary lookup table (joltworks/src/subprotocols/ps_shout/mod.rs:227-560,
// unary.rs:45-148) for ReluTable<N> (lookup_tables/relu.rs) with the SignedIdentity RAF term
// (poly/signed_identity_poly.rs).  N address rounds in 8 phases of log_m = N/8 bits, then log_T cycle
// rounds (LowToHigh Gruen, degree 2).
//
// The summand is  eq(r_node, t) [k = idx_t] (Val(k) + gamma SId(k)).  With the index split as
// prefix | chunk_p | suffix in phase p, both Val and SId are  A(prefix, chunk) * 1 + B(prefix, chunk) *
// suffix  (relu.rs:55-59; signed_identity_poly.rs:136-160), so two suffix tables per phase carry the
// whole O(T) part:   Q1[y] = sum_{t : chunk_p(idx_t) = y} u_t,   Qs[y] = sum ... u_t * suffix_t,
// u_t = eq(r_node, t) * prod_{q < p} v_q[chunk_q(idx_t)]  (mod.rs:269-335).  They are built by
// k_ps_q (one workgroup per (bin, slice of T)); the per-round arithmetic over the 2^log_m entries —
// prefix evaluations, binding Q and the expanding table v_p — is host work.  The reference keeps the
// WordNoMSB suffix as u32, which is exact for N <= 32; N = 64 is refused here for that reason.
// spec of the suffix functions: mode 0/1 -> {1, suffix}; mode 2 (clamp, BOUND) additionally
// {HAZ_s, HAZ_s * lw_s, HAO_s, HAO_s * lw_s} with HAZ_s / HAO_s = "the suffix bits of significance >= BOUND are
// all zero / all one" and lw_s = the suffix bits below BOUND (suffixes/higher_all_zero.rs, hzero_mul_lword.rs,
// hone_mul_lword.rs)
template <int NQ>
__global__ __launch_bounds__(RA_THREADS) void k_ps_q(const uint64_t* __restrict__ idx, const Fr* __restrict__ u0,
                                                     const Fr* __restrict__ prod, size_t T, uint32_t suffix_len, uint32_t m_mask,
                                                     uint32_t bound, Fr* __restrict__ partials /* [slices][NQ m] */) {
    __shared__ Fr red[RA_THREADS / 64][NQ];
    const uint32_t y = blockIdx.x, slice = blockIdx.y, n_slices = gridDim.y;
    const size_t per = (T + n_slices - 1) / n_slices, t0 = (size_t)slice * per, t1 = t0 + per < T ? t0 + per : T;
    Fr acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = fe_zero();
    const uint64_t smask = suffix_len >= 64 ? ~0ull : (((uint64_t)1 << suffix_len) - 1);
    for (size_t t = t0 + threadIdx.x; t < t1; t += RA_THREADS) {
        const uint64_t k = idx[t];
        if (((uint32_t)(k >> suffix_len) & m_mask) != y) continue;
        const Fr u = fr_mul(fe_load(u0 + t), fe_load(prod + t));
        acc[0] = fr_add(acc[0], u);
        const uint64_t sb = k & smask;
        if constexpr (NQ == 4) {      // binary lookups: suffix = interleave(x, y) (utils/mod.rs:105-125 uninterleave_bits)
            uint64_t xb = (sb >> 1) & 0x5555555555555555ull, yb = sb & 0x5555555555555555ull;
            xb = (xb | (xb >> 1)) & 0x3333333333333333ull; xb = (xb | (xb >> 2)) & 0x0F0F0F0F0F0F0F0Full;
            xb = (xb | (xb >> 4)) & 0x00FF00FF00FF00FFull; xb = (xb | (xb >> 8)) & 0x0000FFFF0000FFFFull; xb = (xb | (xb >> 16)) & 0xFFFFFFFFull;
            yb = (yb | (yb >> 1)) & 0x3333333333333333ull; yb = (yb | (yb >> 2)) & 0x0F0F0F0F0F0F0F0Full;
            yb = (yb | (yb >> 4)) & 0x00FF00FF00FF00FFull; yb = (yb | (yb >> 8)) & 0x0000FFFF0000FFFFull; yb = (yb | (yb >> 16)) & 0xFFFFFFFFull;
            if (xb < yb) acc[1] = fr_add(acc[1], u);                              // LessThan suffix
            if (xb) acc[2] = fr_add(acc[2], fr_mul(u, fr_from_i64((int64_t)xb)));   // left operand of the suffix
            if (yb) acc[3] = fr_add(acc[3], fr_mul(u, fr_from_i64((int64_t)yb)));   // right operand
            continue;
        }
        if (sb) acc[1] = fr_add(acc[1], fr_mul(u, fr_from_i64((int64_t)sb)));
        if constexpr (NQ == 6) {
            bool haz = true, hao = true;
            uint64_t lw = sb;
            if (suffix_len > bound) {
                const uint64_t hi = sb >> bound, ones = (((uint64_t)1 << (suffix_len - bound)) - 1);
                haz = hi == 0; hao = hi == ones;
                lw = sb & (((uint64_t)1 << bound) - 1);
            }
            if (haz || hao) {
                const Fr ul = lw ? fr_mul(u, fr_from_i64((int64_t)lw)) : fe_zero();
                if (haz) { acc[2] = fr_add(acc[2], u); acc[3] = fr_add(acc[3], ul); }
                if (hao) { acc[4] = fr_add(acc[4], u); acc[5] = fr_add(acc[5], ul); }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const Fr sres = fr_wave_sum(acc[q]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = sres;
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        Fr sres = red[0][threadIdx.x];
        for (int w = 1; w < RA_THREADS / 64; w++) sres = fr_add(sres, red[w][threadIdx.x]);
        fe_store(partials + ((size_t)slice * NQ * (m_mask + 1)) + NQ * y + threadIdx.x, sres);
    }
}

// prod[t] *= v[chunk(idx_t)]   (u_evals rescale, mod.rs:275-284, and the ra product, :429-441)
__global__ __launch_bounds__(RA_THREADS) void k_ps_scale(const uint64_t* __restrict__ idx, const Fr* __restrict__ v, size_t T,
                                                         uint32_t shift, uint32_t m_mask, Fr* __restrict__ prod) {
    for (size_t t = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; t < T; t += (size_t)gridDim.x * RA_THREADS)
        fe_store(prod + t, fr_mul(fe_load(prod + t), fe_load(v + ((uint32_t)(idx[t] >> shift) & m_mask))));
}

__global__ __launch_bounds__(RA_THREADS) void k_ps_fill_one(Fr* p, size_t T) {
    const Fr one = fr_one();
    for (size_t t = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; t < T; t += (size_t)gridDim.x * RA_THREADS) fe_store(p + t, one);
}

// sum_j E_out E_in ra[2 j]   (mod.rs:463-484)
__global__ __launch_bounds__(RA_THREADS) void k_ps_fold(const Fr* __restrict__ ra, SplitEqView E, size_t n_groups, Fr* __restrict__ partials) {
    Fr acc[1];
    acc[0] = fe_zero();
    for (size_t j = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; j < n_groups; j += (size_t)gridDim.x * RA_THREADS)
        acc[0] = fr_add(acc[0], fr_mul(gse_weight(E, j), fe_load(ra + 2 * j)));
    block_reduce_store<1>(acc, partials);
}

struct PsRelu : atlas_instance {
    size_t N = 0, log_m = 0, m = 0, log_T = 0, T = 0, round_next = 0, phases = 8;   // N = LOG_K
    int mode = 0;                         // 0 = ReLU + gamma * SignedIdentity (unary read-raf), 1 = Identity (range check), 2 = clamp family
    size_t bound = 0; bool symmetric = true;   // ClampBoundedTable<N, BOUND, SYMMETRIC> (lookup_tables/clamp.rs)
    size_t nq() const { return mode == 2 ? 6 : mode == 3 ? 4 : 2; }
    H::Fr gamma = H::zero();
    uint64_t* d_idx = nullptr;
    Fr *d_u0 = nullptr, *d_v = nullptr, *d_qpart = nullptr;
    RaRows rows;                          // row 0 = running product of the v tables = ra at the end
    GseDev eq;
    std::vector<std::vector<H::Fr>> Q;    // current phase's suffix tables (bound HighToLow): 0 = One, 1 = suffix, 2..5 clamp
    std::vector<H::Fr> v;                 // expanding table of the phase
    H::Fr haz_acc = H::one(), hao_acc = H::one(), lw_acc = H::zero();
    H::Fr lt_acc = H::zero(), eq_acc = H::one(), lop_acc = H::zero(), rop_acc = H::zero();   // mode 3 (UnsignedLessThan, binary)
    std::vector<H::Fr> r_addr;
    H::Fr word_acc = H::zero(), sid_acc = H::zero(), wv = H::zero();
    static constexpr unsigned SLICES = 64;

    ~PsRelu() override { for (void* p : {(void*)d_idx, (void*)d_u0, (void*)d_v, (void*)d_qpart}) if (p) hipFree(p); rows.release(); eq.release(); }
    size_t rounds() const override { return N + log_T; }
    size_t degree() const override { return 2; }

    static H::Fr pow2(size_t k) { H::Fr o = H::one(); const H::Fr two = H::from_u64(2); for (size_t i = 0; i < k; i++) o = H::mul(o, two); return o; }
    H::Fr weight(size_t i) const { H::Fr w = pow2(N - 1 - i); return (i == 0 && mode != 1) ? H::sub(w, pow2(N)) : w; }   // (Signed)Identity coefficient of bit i

    int build_Q(size_t phase) {           // init_phase: Q tables of `phase` from the current products
        const uint32_t suffix_len = (uint32_t)((phases - 1 - phase) * log_m);
        const size_t NQ = nq();
        if (NQ == 4) k_ps_q<4><<<dim3((unsigned)m, SLICES), RA_THREADS, 0, g.stream>>>(d_idx, d_u0, rows.buf[0], T, suffix_len, (uint32_t)(m - 1), 0u, d_qpart);
        else if (NQ == 6) k_ps_q<6><<<dim3((unsigned)m, SLICES), RA_THREADS, 0, g.stream>>>(d_idx, d_u0, rows.buf[0], T, suffix_len, (uint32_t)(m - 1), (uint32_t)bound, d_qpart);
        else k_ps_q<2><<<dim3((unsigned)m, SLICES), RA_THREADS, 0, g.stream>>>(d_idx, d_u0, rows.buf[0], T, suffix_len, (uint32_t)(m - 1), 0u, d_qpart);
        k_col_reduce<<<(unsigned)(NQ * m), RA_THREADS, 0, g.stream>>>(d_qpart, SLICES, (uint32_t)(NQ * m), d_qpart + (size_t)SLICES * NQ * m);
        std::vector<H::Fr> q(NQ * m);
        HIP_TRY(hipMemcpyAsync(q.data(), d_qpart + (size_t)SLICES * NQ * m, NQ * m * sizeof(Fr), hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        Q.assign(NQ, std::vector<H::Fr>(m));
        for (size_t y = 0; y < m; y++) for (size_t k = 0; k < NQ; k++) Q[k][y] = q[NQ * y + k];
        v.assign(1, H::one());
        return ATLAS_OK;
    }

    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "ps_shout: round out of order");
        coeffs.assign(3, H::zero());
        if (round < N) {
            const size_t j = round, p = j / log_m, half = Q[0].size() / 2;
            const size_t suffix_len = N - (p + 1) * log_m;
            const size_t blen = log_m - (j % log_m) - 1;              // chunk bits still boolean after variable j
            const size_t hbits = mode == 2 ? N - bound : 0;           // variables 0..hbits-1 are the clamp's high bits
            const H::Fr sh = pow2(suffix_len), one = H::one();
            const H::Fr U = mode == 2 ? H::from_u64(((uint64_t)1 << bound) - 1) : H::zero();
            const H::Fr LC = symmetric ? H::add(H::add(U, U), one) : U;
            H::Fr ev[2];
            for (int ci = 0; ci < 2; ci++) {
                const H::Fr c = H::from_u64(ci ? 2 : 0);
                const H::Fr not_msb = j == 0 ? H::sub(one, c) : H::sub(one, r_addr[0]);
                const H::Fr word_c = j >= 1 ? H::add(word_acc, H::mul(c, pow2(N - 1 - j))) : H::zero();
                const H::Fr sid_c = H::add(sid_acc, H::mul(c, weight(j)));
                // clamp prefixes over the bound variables and c
                const H::Fr msb = j == 0 ? c : (r_addr.empty() ? c : r_addr[0]);
                const H::Fr haz_c = j < hbits ? H::mul(haz_acc, H::sub(one, c)) : haz_acc;
                const H::Fr hao_c = j < hbits ? H::mul(hao_acc, c) : hao_acc;
                const H::Fr lw_c = (mode == 2 && j >= hbits) ? H::add(lw_acc, H::mul(c, pow2(N - 1 - j))) : lw_acc;
                H::Fr acc = H::zero();
                for (size_t b = 0; b < half; b++) {
                    const H::Fr bs = H::mul(H::from_u64(b), sh);
                    auto qv = [&](size_t k) { return ci ? H::sub(H::add(Q[k][b + half], Q[k][b + half]), Q[k][b]) : Q[k][b]; };
                    if (mode == 3) {     // UnsignedLessThan over interleaved (x, y) pairs + gamma * Left + gamma^2 * Right
                        H::Fr lt = lt_acc, eq = eq_acc, lo = lop_acc, ro = rop_acc;
                        auto pair = [&](const H::Fr& x, const H::Fr& y) {              // unsigned_less_than.rs:33-41
                            lt = H::add(lt, H::mul(eq, H::mul(H::sub(one, x), y)));
                            eq = H::mul(eq, H::add(H::mul(x, y), H::mul(H::sub(one, x), H::sub(one, y))));
                        };
                        auto opw = [&](size_t var) { H::Fr w = pow2(31 - var / 2); return var < 2 ? H::sub(w, pow2(32)) : w; };
                        size_t q = 0;                                                   // next bit of b (MSB first)
                        auto bbit = [&](size_t qq) { return H::from_u64((b >> (blen - 1 - qq)) & 1); };
                        if (j % 2 == 0) {
                            const H::Fr y = bbit(0); q = 1;
                            pair(c, y);
                            lo = H::add(lo, H::mul(c, opw(j))); ro = H::add(ro, H::mul(y, opw(j + 1)));
                        } else {
                            pair(r_addr[j - 1], c);
                            ro = H::add(ro, H::mul(c, opw(j)));
                        }
                        for (; q + 1 < blen + 1 && q < blen; q += 2) {
                            const H::Fr x = bbit(q), y = bbit(q + 1);
                            pair(x, y);
                            lo = H::add(lo, H::mul(x, opw(j + 1 + q))); ro = H::add(ro, H::mul(y, opw(j + 2 + q)));
                        }
                        const H::Fr g2 = H::mul(gamma, gamma);
                        H::Fr val = H::add(H::mul(lt, qv(0)), H::mul(eq, qv(1)));
                        val = H::add(val, H::mul(gamma, H::add(H::mul(lo, qv(0)), qv(2))));
                        val = H::add(val, H::mul(g2, H::add(H::mul(ro, qv(0)), qv(3))));
                        acc = H::add(acc, val);
                        continue;
                    }
                    const H::Fr q1 = qv(0), qs = qv(1);
                    const H::Fr idt = H::add(H::mul(H::add(sid_c, bs), q1), qs);          // (Signed)Identity term
                    if (mode == 1) { acc = H::add(acc, idt); continue; }
                    if (mode == 0) {
                        const H::Fr val = H::mul(not_msb, H::add(H::mul(H::add(word_c, bs), q1), qs));
                        acc = H::add(acc, H::add(val, H::mul(gamma, idt)));
                        continue;
                    }
                    // clamp (clamp.rs:84-109): chunk bits of b at variable index j+1+q are high iff that index < hbits
                    bool z = true, o = true;
                    uint64_t lwb = 0;
                    for (size_t q = 0; q < blen; q++) {
                        const size_t var = j + 1 + q;
                        const uint64_t bit = (b >> (blen - 1 - q)) & 1;
                        if (var < hbits) { if (bit) z = false; else o = false; }
                        else lwb |= bit << (N - 1 - var);
                    }
                    H::Fr val = H::mul(H::sub(U, H::mul(msb, LC)), q1);
                    const H::Fr lw = H::add(lw_c, H::from_u64(lwb));
                    if (z) val = H::add(val, H::mul(haz_c, H::add(H::mul(H::sub(lw, U), qv(2)), qv(3))));
                    if (o && symmetric) val = H::add(val, H::mul(hao_c, H::add(H::mul(lw, qv(4)), qv(5))));
                    acc = H::add(acc, H::add(val, H::mul(gamma, idt)));
                }
                ev[ci] = acc;
            }
            H::unipoly_from_evals_and_hint(claim, ev, 2, coeffs.data());
            return ATLAS_OK;
        }
        std::lock_guard<std::mutex> lk(g.mu);
        const size_t n_groups = rows.len / 2;
        size_t blocks = (n_groups + RA_THREADS - 1) / RA_THREADS; if (blocks > 2048) blocks = 2048;
        k_ps_fold<<<(unsigned)blocks, RA_THREADS, 0, g.stream>>>(rows.buf[rows.cur], eq.view(), n_groups, rows.partials);
        H::Fr s;
        int rc = rows.reduce_to_host((uint32_t)blocks, 1, &s);
        if (rc) return rc;
        H::gruen_deg2(eq.st.scalar, eq.st.w_cur(), H::mul(s, wv), claim, coeffs.data());
        return ATLAS_OK;
    }

    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "ps_shout: round out of order");
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, g.challenge_mode);
        if (round < N) {
            const size_t j = round, p = j / log_m;
            const size_t half = Q[0].size() / 2;
            for (auto& q : Q) {                                       // suffix polys bind HighToLow
                for (size_t i = 0; i < half; i++) q[i] = H::add(q[i], H::mul(rf, H::sub(q[i + half], q[i])));
                q.resize(half);
            }
            std::vector<H::Fr> nv(2 * v.size());                      // ExpandingTable::update, HighToLow
            for (size_t i = 0; i < v.size(); i++) { nv[2 * i + 1] = H::mul(rf, v[i]); nv[2 * i] = H::sub(v[i], nv[2 * i + 1]); }
            v.swap(nv);
            if (j >= 1) word_acc = H::add(word_acc, H::mul(rf, pow2(N - 1 - j)));
            sid_acc = H::add(sid_acc, H::mul(rf, weight(j)));
            if (mode == 3) {
                const H::Fr w = j < 2 ? H::sub(pow2(31 - j / 2), pow2(32)) : pow2(31 - j / 2);
                if (j % 2 == 0) lop_acc = H::add(lop_acc, H::mul(rf, w));
                else {
                    rop_acc = H::add(rop_acc, H::mul(rf, w));
                    const H::Fr x = r_addr[j - 1], one = H::one();
                    lt_acc = H::add(lt_acc, H::mul(eq_acc, H::mul(H::sub(one, x), rf)));
                    eq_acc = H::mul(eq_acc, H::add(H::mul(x, rf), H::mul(H::sub(one, x), H::sub(one, rf))));
                }
            }
            if (mode == 2) {
                if (j < N - bound) { haz_acc = H::mul(haz_acc, H::sub(H::one(), rf)); hao_acc = H::mul(hao_acc, rf); }
                else lw_acc = H::add(lw_acc, H::mul(rf, pow2(N - 1 - j)));
            }
            r_addr.push_back(rf);
            if ((j + 1) % log_m == 0) {                               // phase boundary: fold v_p into the products
                std::lock_guard<std::mutex> lk(g.mu);
                HIP_TRY(hipMemcpyAsync(d_v, v.data(), m * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
                size_t gb = (T + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
                k_ps_scale<<<(unsigned)gb, RA_THREADS, 0, g.stream>>>(d_idx, d_v, T, (uint32_t)((phases - 1 - p) * log_m), (uint32_t)(m - 1), rows.buf[0]);
                HIP_TRY(hipStreamSynchronize(g.stream));
                if (p != phases - 1) { int rc = build_Q(p + 1); if (rc) return rc; }
            }
            if (j + 1 == N) {
                // val = Val~(r_address), raf_val = gamma * SId~(r_address)   (mod.rs:523-548)
                const H::Fr val = H::mul(H::sub(H::one(), r_addr[0]), word_acc);
                wv = mode == 1 ? sid_acc : H::add(val, H::mul(gamma, sid_acc));            // identity_range_check.rs:377-380
                if (mode == 3) wv = H::add(lt_acc, H::add(H::mul(gamma, lop_acc), H::mul(H::mul(gamma, gamma), rop_acc)));   // binary.rs:108-116
                if (mode == 2) {                                      // ClampBoundedTable::evaluate_mle at r_address
                    const H::Fr U = H::from_u64(((uint64_t)1 << bound) - 1);
                    const H::Fr LC = symmetric ? H::add(H::add(U, U), H::one()) : U;
                    H::Fr cv = H::add(H::sub(U, H::mul(r_addr[0], LC)), H::mul(haz_acc, H::sub(lw_acc, U)));
                    if (symmetric) cv = H::add(cv, H::mul(hao_acc, lw_acc));
                    wv = H::add(cv, H::mul(gamma, sid_acc));
                }
                rows.cur = 0; rows.stride[0] = T; rows.len = T;       // the products are ra (init_log_t_rounds)
            }
        } else {
            std::lock_guard<std::mutex> lk(g.mu);
            int rc = rows.bind(r);
            if (rc) return rc;
            eq.st.bind(rf);
        }
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override {
        if (round_next != rounds()) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        std::lock_guard<std::mutex> lk(g.mu);
        return rows.finals(out);
    }
};

}  // namespace

extern "C" {

static int ra_virtual_build(const int32_t* const* H_indices, const uint64_t* lookups, size_t d, size_t log_k_chunk, size_t log_T,
                            const atlas_fr_t* r_address_chunks, const atlas_fr_t* r_cycle, atlas_instance_t* out);

int atlas_ra_virtual_new(const int32_t* const* H_indices, size_t d, size_t log_k_chunk, size_t log_T,
                         const atlas_fr_t* r_address_chunks, const atlas_fr_t* r_cycle, atlas_instance_t* out) {
    NEED_INIT();
    if (!H_indices) return fail(ATLAS_EINVAL, "ra_virtual_new: null argument");
    return ra_virtual_build(H_indices, nullptr, d, log_k_chunk, log_T, r_address_chunks, r_cycle, out);
}

// RaSumcheckProver::gen from the lookup indices and r_address: OneHotParams::{lookup_index_chunk, compute_r_address_chunks}
// (config.rs:73-100) done here — d = ceil(log_K / log_k_chunk), r_address left-padded with zeros to a multiple of the chunk
int atlas_ra_virtual_from_lookups_new(const uint64_t* lookup_indices, size_t log_T, size_t log_K, size_t log_k_chunk,
                                      const atlas_fr_t* r_address, const atlas_fr_t* r_cycle, atlas_instance_t* out) {
    NEED_INIT();
    if (!lookup_indices || !r_address || log_k_chunk == 0 || log_K == 0) return fail(ATLAS_EINVAL, "ra_virtual_from_lookups_new: null argument");
    const size_t d = (log_K + log_k_chunk - 1) / log_k_chunk, pad = d * log_k_chunk - log_K;
    std::vector<atlas_fr_t> chunks(d * log_k_chunk);
    std::memset(chunks.data(), 0, pad * sizeof(atlas_fr_t));
    std::memcpy(chunks.data() + pad, r_address, log_K * sizeof(atlas_fr_t));
    return ra_virtual_build(nullptr, lookup_indices, d, log_k_chunk, log_T, chunks.data(), r_cycle, out);
}

static int ra_virtual_build(const int32_t* const* H_indices, const uint64_t* lookups, size_t d, size_t log_k_chunk, size_t log_T,
                            const atlas_fr_t* r_address_chunks, const atlas_fr_t* r_cycle, atlas_instance_t* out) {
    if (!r_address_chunks || (!r_cycle && log_T) || !out) return fail(ATLAS_EINVAL, "ra_virtual_new: null argument");
    if (d == 0 || d > RA_MAX_D) return fail(ATLAS_EINVAL, "ra_virtual_new: d must be in 1..16");
    if (log_k_chunk > 16 || log_T == 0 || log_T > 25) return fail(ATLAS_EINVAL, "ra_virtual_new: log_k_chunk <= 16, 1 <= log_T <= 25");
    std::lock_guard<std::mutex> lk(g.mu);
    RaVirtual* P = new RaVirtual();
    P->log_T = log_T;
    const size_t K = (size_t)1 << log_k_chunk, T = (size_t)1 << log_T;
    std::vector<std::vector<H::Fr>> tabs(d);
    const H::Fr* ch = reinterpret_cast<const H::Fr*>(r_address_chunks);
    for (size_t i = 0; i < d; i++) tabs[i] = H::eq_evals(ch + i * log_k_chunk, log_k_chunk);   // ra_virtual.rs:113-116
    Fr* d_tabs = nullptr;
    int rc = upload_tables(tabs, K, &d_tabs);
    if (!rc) rc = P->rows.alloc(d, T);
    if (!rc) rc = H_indices ? P->rows.upload_indices(H_indices) : P->rows.upload_lookups(lookups, (uint32_t)log_k_chunk);
    if (!rc) rc = P->rows.gather(d_tabs, (uint32_t)K);
    if (d_tabs) hipFree(d_tabs);
    if (!rc) rc = P->eq.init(reinterpret_cast<const H::Fr*>(r_cycle), log_T);
    if (rc) { delete P; return rc; }
    *out = P;
    return ATLAS_OK;
}

static int booleanity_build(const atlas_fr_t* G, const int32_t* const* H_indices, const uint64_t* lookups, size_t d, size_t log_k_chunk,
                            size_t log_T, const atlas_fr_t* gammas, const atlas_fr_t* r_address, const atlas_fr_t* r_cycle,
                            atlas_instance_t* out);

int atlas_booleanity_new(const atlas_fr_t* G, const int32_t* const* H_indices, size_t d, size_t log_k_chunk, size_t log_T,
                         const atlas_fr_t* gammas, const atlas_fr_t* r_address, const atlas_fr_t* r_cycle, atlas_instance_t* out) {
    NEED_INIT();
    if (!H_indices) return fail(ATLAS_EINVAL, "booleanity_new: null argument");
    return booleanity_build(G, H_indices, nullptr, d, log_k_chunk, log_T, gammas, r_address, r_cycle, out);
}

int atlas_booleanity_from_lookups_new(const atlas_fr_t* G, const uint64_t* lookup_indices, size_t log_T, size_t log_K, size_t log_k_chunk,
                                      const atlas_fr_t* gammas, const atlas_fr_t* r_address, const atlas_fr_t* r_cycle,
                                      atlas_instance_t* out) {
    NEED_INIT();
    if (!lookup_indices || log_k_chunk == 0 || log_K == 0) return fail(ATLAS_EINVAL, "booleanity_from_lookups_new: null argument");
    return booleanity_build(G, nullptr, lookup_indices, (log_K + log_k_chunk - 1) / log_k_chunk, log_k_chunk, log_T, gammas, r_address, r_cycle, out);
}

static int booleanity_build(const atlas_fr_t* G, const int32_t* const* H_indices, const uint64_t* lookups, size_t d, size_t log_k_chunk,
                            size_t log_T, const atlas_fr_t* gammas, const atlas_fr_t* r_address, const atlas_fr_t* r_cycle,
                            atlas_instance_t* out) {
    if (!G || !gammas || !r_address || (!r_cycle && log_T) || !out) return fail(ATLAS_EINVAL, "booleanity_new: null argument");
    if (d == 0 || log_k_chunk == 0 || log_k_chunk > 16 || log_T == 0 || log_T > 25)
        return fail(ATLAS_EINVAL, "booleanity_new: 1 <= log_k_chunk <= 16, 1 <= log_T <= 25");
    std::lock_guard<std::mutex> lk(g.mu);
    Booleanity* P = new Booleanity();
    P->d = d; P->log_k = log_k_chunk; P->log_T = log_T;
    const size_t K = (size_t)1 << log_k_chunk, T = (size_t)1 << log_T;
    const H::Fr* Gh = reinterpret_cast<const H::Fr*>(G);
    P->G.resize(d);
    for (size_t i = 0; i < d; i++) P->G[i].assign(Gh + i * K, Gh + (i + 1) * K);
    P->gammas.assign(reinterpret_cast<const H::Fr*>(gammas), reinterpret_cast<const H::Fr*>(gammas) + d);
    P->F = {H::one()};
    const H::Fr* ra = reinterpret_cast<const H::Fr*>(r_address);
    P->B.init(ra, log_k_chunk);
    P->B_out = H::eq_cached(ra, P->B.k_out);
    P->B_in = H::eq_cached(ra + P->B.m, P->B.k_in);
    int rc = P->D.init(reinterpret_cast<const H::Fr*>(r_cycle), log_T);
    if (!rc) rc = P->rows.alloc(d, T);
    if (!rc) rc = H_indices ? P->rows.upload_indices(H_indices) : P->rows.upload_lookups(lookups, (uint32_t)log_k_chunk);   // resident until the phase-2 gather
    if (!rc) {
        hipError_t e = hipMalloc(&P->d_gammas, d * sizeof(Fr));
        if (e == hipSuccess) e = hipMemcpyAsync(P->d_gammas, gammas, d * sizeof(Fr), hipMemcpyHostToDevice, g.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
        if (e != hipSuccess) rc = fail(ATLAS_ENODEV, "booleanity_new: gammas", e);
    }
    if (rc) { delete P; return rc; }
    *out = P;
    return ATLAS_OK;
}

int atlas_hamming_weight_new(const atlas_fr_t* G, size_t d, size_t log_k_chunk, const atlas_fr_t* gamma_powers, atlas_instance_t* out) {
    if (!G || !gamma_powers || !out || d == 0 || log_k_chunk > 20) return fail(ATLAS_EINVAL, "hamming_weight_new");
    HammingWeight* P = new HammingWeight();
    P->log_k = log_k_chunk;
    const size_t K = (size_t)1 << log_k_chunk;
    const H::Fr* Gh = reinterpret_cast<const H::Fr*>(G);
    P->ra.resize(d);
    for (size_t i = 0; i < d; i++) P->ra[i].assign(Gh + i * K, Gh + (i + 1) * K);
    P->gamma_powers.assign(reinterpret_cast<const H::Fr*>(gamma_powers), reinterpret_cast<const H::Fr*>(gamma_powers) + d);
    *out = P;
    return ATLAS_OK;
}

static int ps_new(const uint64_t* lookup_indices, size_t log_T, size_t log_K, size_t phases, int mode, const atlas_fr_t* r_node_output,
                  const atlas_fr_t* gamma, atlas_instance_t* out, size_t bound = 0, bool symmetric = true) {
    atlas_poly_t E = nullptr;
    int rc = atlas_eq_evals(r_node_output, log_T, nullptr, &E);      // u_evals = EqPolynomial::evals(r_node_output), mod.rs:234
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g.mu);
    PsRelu* P = new PsRelu();
    P->N = log_K; P->phases = phases; P->mode = mode; P->bound = bound; P->symmetric = symmetric;
    P->log_m = log_K / phases; P->m = (size_t)1 << P->log_m; P->log_T = log_T; P->T = (size_t)1 << log_T;
    if (gamma) std::memcpy(&P->gamma, gamma, 32);
    P->d_u0 = (Fr*)E->d; delete E;                                   // keep the table, drop the handle
    const size_t T = P->T, m = P->m;
    hipError_t e = hipMalloc(&P->d_idx, T * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(&P->d_v, m * sizeof(Fr));
    if (e == hipSuccess) e = hipMalloc(&P->d_qpart, ((size_t)PsRelu::SLICES + 1) * 6 * m * sizeof(Fr));
    if (e == hipSuccess) e = hipMemcpyAsync(P->d_idx, lookup_indices, T * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream);
    if (e != hipSuccess) { delete P; return fail(ATLAS_ENOMEM, "ps_shout_new", e); }
    rc = P->rows.alloc(1, T);
    if (!rc) {
        size_t gb = (T + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
        k_ps_fill_one<<<(unsigned)gb, RA_THREADS, 0, g.stream>>>(P->rows.buf[0], T);
        rc = P->eq.init(reinterpret_cast<const H::Fr*>(r_node_output), log_T);
    }
    if (!rc) rc = P->build_Q(0);
    if (rc) { delete P; return rc; }
    *out = P;
    return ATLAS_OK;
}

int atlas_ps_shout_relu_new(const uint64_t* lookup_indices, size_t log_T, size_t xlen, const atlas_fr_t* r_node_output,
                            const atlas_fr_t* gamma, atlas_instance_t* out) {
    NEED_INIT();
    if (!lookup_indices || !r_node_output || !gamma || !out) return fail(ATLAS_EINVAL, "ps_shout_relu_new: null argument");
    if (xlen != 16 && xlen != 32) return fail(ATLAS_EINVAL, "ps_shout_relu_new: X_LEN must be 16 or 32 (the reference's WordNoMSB suffix is a u32)");
    if (log_T == 0 || log_T > 25) return fail(ATLAS_EINVAL, "ps_shout_relu_new: 1 <= log_T <= 25");
    return ps_new(lookup_indices, log_T, xlen, 8, 0, r_node_output, gamma, out);
}

int atlas_ps_shout_clamp_new(const uint64_t* lookup_indices, size_t log_T, size_t xlen, size_t bound, int symmetric,
                             const atlas_fr_t* r_node_output, const atlas_fr_t* gamma, atlas_instance_t* out) {
    NEED_INIT();
    if (!lookup_indices || !r_node_output || !gamma || !out) return fail(ATLAS_EINVAL, "ps_shout_clamp_new: null argument");
    if (xlen != 16 && xlen != 32 && xlen != 64) return fail(ATLAS_EINVAL, "ps_shout_clamp_new: X_LEN must be 16, 32 or 64");
    if (bound == 0 || bound + 1 >= xlen || bound > 31) return fail(ATLAS_EINVAL, "ps_shout_clamp_new: 1 <= BOUND <= 31 and BOUND < X_LEN - 1");
    if (log_T == 0 || log_T > 25) return fail(ATLAS_EINVAL, "ps_shout_clamp_new: 1 <= log_T <= 25");
    return ps_new(lookup_indices, log_T, xlen, 8, 2, r_node_output, gamma, out, bound, symmetric != 0);
}

int atlas_ps_shout_ult_new(const uint64_t* lookup_indices, size_t log_T, const atlas_fr_t* r_node_output, const atlas_fr_t* gamma,
                           atlas_instance_t* out) {
    NEED_INIT();
    if (!lookup_indices || !r_node_output || !gamma || !out) return fail(ATLAS_EINVAL, "ps_shout_ult_new: null argument");
    if (log_T == 0 || log_T > 25) return fail(ATLAS_EINVAL, "ps_shout_ult_new: 1 <= log_T <= 25");
    return ps_new(lookup_indices, log_T, 64, 8, 3, r_node_output, gamma, out);
}

int atlas_identity_range_check_new(const uint64_t* lookup_indices, size_t log_T, size_t log_K, size_t phases,
                                   const atlas_fr_t* r_node_output, atlas_instance_t* out) {
    NEED_INIT();
    if (!lookup_indices || !r_node_output || !out) return fail(ATLAS_EINVAL, "identity_range_check_new: null argument");
    if (phases == 0 || log_K == 0 || log_K > 64 || log_K % phases || log_K / phases > 12)
        return fail(ATLAS_EINVAL, "identity_range_check_new: log_K must be a multiple of phases, chunks of at most 12 bits");
    if (log_T == 0 || log_T > 25) return fail(ATLAS_EINVAL, "identity_range_check_new: 1 <= log_T <= 25");
    return ps_new(lookup_indices, log_T, log_K, phases, 1, r_node_output, nullptr, out);
}

}  // extern "C"
