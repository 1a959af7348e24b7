// build_materialized_rlc (joltworks/src/poly/rlc_polynomial.rs:13-78): the joint
// polynomial the batched HyperKZG opening is run on,
//     joint[i] = sum_j coeff_j * dense_j[i]            (dense_j = LargeScalars or I32Scalars)
//     joint[k_j[t] * T_j + t] += coeff_j               (one-hot polynomials, k*T + t, no bit reversal)
// One streaming kernel for the dense part (HBM-bound: reads every dense coefficient once, writes
// joint once) and one scatter launch per group of one-hot polynomials that share T — within a
// group index k*T + t is only ever touched by thread t, so no atomics are needed (the reference
// runs this part as a serial loop, :62-75).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/atlas_hip.h"
#include "field.hip.h"
#include "host_field.hpp"
#include "sumcheck_kernels.hip.h"
#include "runtime.hpp"

using namespace atlas;
using atlas_rt::fail;
using atlas_rt::rt;

namespace {

constexpr int RLC_THREADS = 256;

struct RlcDense {
    const void* d;
    uint64_t len;
    uint32_t is_i32, pad;
    Fr coeff;       // Montgomery
    Fr coeff_s32;   // coeff * 2^32 (Montgomery): CIOS against a 1-limb integer gives coeff * v
};

__global__ __launch_bounds__(RLC_THREADS) void k_rlc_dense(const RlcDense* __restrict__ polys, uint32_t n_polys,
                                                           Fr* __restrict__ joint, size_t joint_len) {
    for (size_t i = (size_t)blockIdx.x * RLC_THREADS + threadIdx.x; i < joint_len; i += (size_t)gridDim.x * RLC_THREADS) {
        Fr acc = fe_zero();
        for (uint32_t j = 0; j < n_polys; j++) {
            const RlcDense P = polys[j];
            if (i >= P.len) continue;
            if (P.is_i32) {
                const int32_t v = ((const int32_t*)P.d)[i];
                if (v == 0) continue;
                const uint32_t mag = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
                const Fr t = fe_mul_small<FrParams, 1>(P.coeff_s32, &mag);   // Montgomery(coeff * |v|)
                acc = v < 0 ? fr_sub(acc, t) : fr_add(acc, t);
            } else {
                acc = fr_add(acc, fr_mul(P.coeff, fe_load((const Fr*)P.d + i)));
            }
        }
        fe_store(joint + i, acc);
    }
}

struct RlcOneHot {
    const int32_t* k;   // T entries on the device, negative = None
    Fr coeff;
    uint32_t K;         // address-space size: an index >= K is rejected (the reference panics on the slice bound)
    uint32_t pad;
};

__global__ __launch_bounds__(RLC_THREADS) void k_rlc_onehot(const RlcOneHot* __restrict__ polys, uint32_t n_polys, size_t T,
                                                            Fr* __restrict__ joint, uint32_t* __restrict__ bad) {
    for (size_t t = (size_t)blockIdx.x * RLC_THREADS + threadIdx.x; t < T; t += (size_t)gridDim.x * RLC_THREADS) {
        for (uint32_t j = 0; j < n_polys; j++) {
            const int32_t k = polys[j].k[t];
            if (k < 0) continue;
            if ((uint32_t)k >= polys[j].K) { *bad = 1u; continue; }
            Fr* dst = joint + (size_t)k * T + t;
            fe_store(dst, fr_add(fe_load(dst), polys[j].coeff));
        }
    }
}

// The rows with K <= 16 (the RaD chunk polynomials: all but a handful of a graph's thousands): thread (t, bin) adds up the coefficients of the rows
// whose index at t is its bin — in registers, ONE read-modify-write of the joint polynomial at the end.  k_rlc_onehot above walks the rows with a
// load-add-store of joint[k T + t] each, a chain of thousands of dependent memory round trips per thread: 9.9 ms for the 8292 rows of the
// GPT-2-shaped reduction (r06s).  The sums are exact field sums: their order changes no value.  Block = 64 cycles x 16 bins.
__global__ __launch_bounds__(1024) void k_rlc_onehot_bins(const RlcOneHot* __restrict__ polys, uint32_t n_polys, size_t T, size_t joint_len,
                                                          Fr* __restrict__ joint, uint32_t* __restrict__ bad) {
    const size_t t = (size_t)blockIdx.x * 64 + (threadIdx.x & 63u);
    const int32_t bin = (int32_t)(threadIdx.x >> 6);
    if (t >= T) return;
    Fr acc = fe_zero();
    bool touched = false;
    constexpr uint32_t RU = 8;                        // index loads in flight per thread (one per iteration: a memory round trip per row)
    for (uint32_t j0 = 0; j0 < n_polys; j0 += RU) {
        int32_t kx[RU];
#pragma unroll
        for (uint32_t u = 0; u < RU; u++) kx[u] = j0 + u < n_polys ? polys[j0 + u].k[t] : -1;
#pragma unroll
        for (uint32_t u = 0; u < RU; u++) {
            const int32_t k = kx[u];
            if (k < 0) continue;
            if ((uint32_t)k >= polys[j0 + u].K) { if (bin == 0) *bad = 1u; continue; }
            if (k != bin) continue;
            acc = fr_add(acc, polys[j0 + u].coeff);
            touched = true;
        }
    }
    if (!touched) return;
    const size_t at = (size_t)bin * T + t;
    if (at >= joint_len) return;                     // (unreachable: a matched row has K > bin and joint_len >= K T)
    Fr* dst = joint + at;
    fe_store(dst, fr_add(fe_load(dst), acc));
}

inline int grid_for(size_t work) {
    size_t b = (work + RLC_THREADS - 1) / RLC_THREADS;
    return (int)std::min<size_t>(std::max<size_t>(b, 1), 8192);
}

}  // namespace

extern "C" int atlas_rlc_build(const atlas_rlc_dense_t* dense, size_t n_dense, const atlas_rlc_onehot_t* onehot,
                               size_t n_onehot, atlas_poly_t* out) {
    PROF("atlas_rlc_build");
    NEED_INIT();
    if (!out || (n_dense && !dense) || (n_onehot && !onehot)) return fail(ATLAS_EINVAL, "rlc_build: null argument");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    size_t joint_len = 0;
    for (size_t j = 0; j < n_dense; j++) {
        if (!dense[j].poly) return fail(ATLAS_EINVAL, "rlc_build: null dense polynomial");
        joint_len = std::max(joint_len, dense[j].poly->len);
    }
    for (size_t j = 0; j < n_onehot; j++) {
        if (!onehot[j].k && onehot[j].T) return fail(ATLAS_EINVAL, "rlc_build: null one-hot indices");
        joint_len = std::max(joint_len, onehot[j].K * onehot[j].T);
    }
    // DensePolynomial::new asserts a power of two (dense_mlpoly.rs:33-45)
    if (joint_len == 0 || (joint_len & (joint_len - 1))) return fail(ATLAS_EINVAL, "rlc_build: joint length must be a power of two");

    atlas_poly* P = new atlas_poly();
    P->len = joint_len; P->cap_bytes = joint_len * sizeof(Fr);
    hipError_t e = hipMalloc(&P->d, P->cap_bytes);
    if (e != hipSuccess) { delete P; return fail(ATLAS_ENOMEM, "hipMalloc(rlc joint)", e); }
    Fr* joint = (Fr*)P->d;
    auto bail = [&](int rc) { hipFree(P->d); delete P; return rc; };

    // dense part (also zero-fills joint when there are no dense polynomials)
    std::vector<RlcDense> hd(n_dense);
    const atlas_host::Fr two32 = atlas_host::from_u64(1ull << 32);
    for (size_t j = 0; j < n_dense; j++) {
        hd[j].d = dense[j].poly->d; hd[j].len = dense[j].poly->len; hd[j].is_i32 = dense[j].poly->is_i32 ? 1 : 0; hd[j].pad = 0;
        std::memcpy(&hd[j].coeff, &dense[j].coeff, 32);
        atlas_host::Fr c; std::memcpy(&c, &dense[j].coeff, 32);
        const atlas_host::Fr cs = atlas_host::mul(c, two32);
        std::memcpy(&hd[j].coeff_s32, &cs, 32);
    }
    RlcDense* d_dense = nullptr;
    if (n_dense) {
        e = hipMalloc(&d_dense, n_dense * sizeof(RlcDense));
        if (e != hipSuccess) return bail(fail(ATLAS_ENOMEM, "hipMalloc(rlc table)", e));
        hipMemcpyAsync(d_dense, hd.data(), n_dense * sizeof(RlcDense), hipMemcpyHostToDevice, rt().stream);
    }
    k_rlc_dense<<<grid_for(joint_len), RLC_THREADS, 0, rt().stream>>>(d_dense, (uint32_t)n_dense, joint, joint_len);

    // one-hot part, grouped by T
    std::map<size_t, std::vector<size_t>> groups;
    for (size_t j = 0; j < n_onehot; j++)
        if (onehot[j].T) groups[onehot[j].T].push_back(j);
    std::vector<void*> to_free;
    int rc = ATLAS_OK;
    uint32_t* d_bad = nullptr;
    if (!groups.empty()) {
        e = hipMalloc(&d_bad, 4);
        if (e != hipSuccess) { if (d_dense) hipFree(d_dense); return bail(fail(ATLAS_ENOMEM, "hipMalloc(rlc flag)", e)); }
        to_free.push_back(d_bad);
        hipMemsetAsync(d_bad, 0, 4, rt().stream);
    }
    for (auto& kv : groups) {
        const size_t T = kv.first;
        std::vector<size_t>& members = kv.second;
        const size_t n_small = (size_t)(std::stable_partition(members.begin(), members.end(), [&](size_t j) { return onehot[j].K <= 16; }) - members.begin());   // the K <= 16 rows first
        std::vector<RlcOneHot> ho(members.size());
        for (size_t q = 0; q < members.size() && rc == ATLAS_OK; q++) {
            const atlas_rlc_onehot_t& O = onehot[members[q]];
            std::memcpy(&ho[q].coeff, &O.coeff, 32);
            ho[q].K = (uint32_t)O.K; ho[q].pad = 0;
            if (O.k_on_device) ho[q].k = O.k;
            else {
                int32_t* dk = nullptr;
                e = hipMalloc(&dk, T * 4);
                if (e != hipSuccess) { rc = fail(ATLAS_ENOMEM, "hipMalloc(rlc indices)", e); break; }
                to_free.push_back(dk);
                hipMemcpyAsync(dk, O.k, T * 4, hipMemcpyHostToDevice, rt().stream);
                ho[q].k = dk;
            }
        }
        if (rc) break;
        RlcOneHot* d_oh = nullptr;
        e = hipMalloc(&d_oh, ho.size() * sizeof(RlcOneHot));
        if (e != hipSuccess) { rc = fail(ATLAS_ENOMEM, "hipMalloc(rlc one-hot table)", e); break; }
        to_free.push_back(d_oh);
        hipMemcpyAsync(d_oh, ho.data(), ho.size() * sizeof(RlcOneHot), hipMemcpyHostToDevice, rt().stream);
        // the H2D copies above read pageable host memory that goes out of scope: drain before reuse
        hipStreamSynchronize(rt().stream);
        static const bool no_bins = getenv("ATLAS_RLC_NO_BINS") != nullptr;           // A/B: every row through the read-modify-write walk
        const size_t n_bins = no_bins ? 0 : n_small;
        if (n_bins) k_rlc_onehot_bins<<<(unsigned)((T + 63) / 64), 1024, 0, rt().stream>>>(d_oh, (uint32_t)n_bins, T, joint_len, joint, d_bad);
        if (n_bins < ho.size()) k_rlc_onehot<<<grid_for(T), RLC_THREADS, 0, rt().stream>>>(d_oh + n_bins, (uint32_t)(ho.size() - n_bins), T, joint, d_bad);
    }
    hipError_t le = hipGetLastError();
    uint32_t h_bad = 0;
    if (d_bad && rc == ATLAS_OK) hipMemcpyAsync(&h_bad, d_bad, 4, hipMemcpyDeviceToHost, rt().stream);
    hipStreamSynchronize(rt().stream);
    if (h_bad && rc == ATLAS_OK) rc = fail(ATLAS_EINVAL, "rlc_build: one-hot index >= K");
    for (void* p : to_free) hipFree(p);
    if (d_dense) hipFree(d_dense);
    if (rc) return bail(rc);
    if (le != hipSuccess) return bail(fail(ATLAS_ENODEV, "rlc launch", le));
    *out = P;
    return ATLAS_OK;
}
