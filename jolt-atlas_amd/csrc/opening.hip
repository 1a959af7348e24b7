// Opening-reduction sumcheck provers as host-stepped instances over device data.
// Device counterparts of (paths under the jolt-atlas tree, joltworks/src/):
//   DensePolynomialProverOpening::{compute_message,bind}   subprotocols/opening_reduction.rs:355-425
//   OneHotPolynomialProverOpening::{initialize,compute_message,bind}          :532-723
//   GruenSplitEqPolynomial (HighToLow), gruen_poly_deg_2   poly/split_eq_poly.rs:121-145,349-428
//   EqPolynomial::evals_cached_rev                         poly/eq_poly.rs:193-217
// The reference shares EqCycleState / EqAddressState / SharedDensePolynomial between openings at the
// same point so that a table is bound once; every opening here carries its own (identical values).
//
// O(T) work is in kernels: the weighted half-sum sum_{j < len/2} E_in[j_hi] E_out[j_lo] P[j], the
// HighToLow bind, the G histogram over the T indices and the gather H[j] = F[idx_j].  The K-sized
// address phase (K = 16 .. 256) is host arithmetic between launches.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_poly.hpp"
#include "instance.hpp"
#include "runtime.hpp"
#include "sc_consts.hpp"
#include "spliteq_kernels.hip.h"

using namespace atlas;
namespace H = atlas_host;
using atlas_rt::fail;
using atlas_rt::g;

namespace {

constexpr int OP_THREADS = 256;
inline Fr to_dev(const H::Fr& a) { Fr o; std::memcpy(&o, &a, 32); return o; }

// evals_cached_rev: table j = eq over the last j entries of r[0..k), stored at offset 2^j - 1;
// the variable added at step j becomes the top index bit (big-endian within the suffix).
__global__ __launch_bounds__(1024) void k_eq_cached_rev(Fr* tabs, const Fr* r, uint32_t k) {
    if (threadIdx.x == 0) fe_store(tabs, fr_one());
    __syncthreads();
    for (uint32_t j = 0; j < k; j++) {
        const Fr rv = fe_load(r + (k - 1 - j));
        const uint32_t size = 1u << j;
        const Fr* cur = tabs + (size - 1);
        Fr* nxt = tabs + (2 * size - 1);
        for (uint32_t i = threadIdx.x; i < size; i += 1024) {
            const Fr s = fe_load(cur + i);
            const Fr hi = fr_mul(s, rv);
            fe_store(nxt + i + size, hi);
            fe_store(nxt + i, fr_sub(s, hi));
        }
        __threadfence_block();
        __syncthreads();
    }
}

// sum_{j < half} E_hi[j >> lo_bits] * E_lo[j & mask] * P[j]
template <class T>
__global__ __launch_bounds__(OP_THREADS) void k_open_fold(const T* __restrict__ P, size_t half, SplitEqView E,
                                                          Fr* __restrict__ partials, ScConsts K) {
    Fr acc[1];
    acc[0] = fe_zero();
    const size_t mask = ((size_t)1 << E.in_bits) - 1;
    for (size_t j = (size_t)blockIdx.x * OP_THREADS + threadIdx.x; j < half; j += (size_t)gridDim.x * OP_THREADS) {
        const Fr w = fr_mul(fe_load(E.e_out + (j >> E.in_bits)), fe_load(E.e_in + (j & mask)));
        acc[0] = fr_add(acc[0], fr_mul(w, Src<T>::get(P, j, K)));
    }
    block_reduce_store<1>(acc, partials);
}

__global__ __launch_bounds__(OP_THREADS) void k_open_reduce(const Fr* __restrict__ partials, uint32_t n, Fr* out) {
    __shared__ Fr red[OP_THREADS / 64];
    Fr acc = fe_zero();
    for (uint32_t p = threadIdx.x; p < n; p += OP_THREADS) acc = fr_add(acc, fe_load(partials + p));
    acc = fr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Fr s = red[0];
        for (int w = 1; w < OP_THREADS / 64; w++) s = fr_add(s, red[w]);
        fe_store(out, s);
    }
}

// HighToLow bind in place: z[i] += r (z[i + half] - z[i])
__global__ __launch_bounds__(OP_THREADS) void k_open_bind_hi(Fr* z, size_t half, Fr r, int r_hi_only) {
    for (size_t i = (size_t)blockIdx.x * OP_THREADS + threadIdx.x; i < half; i += (size_t)gridDim.x * OP_THREADS)
        fe_store(z + i, bind_pair(fe_load(z + i), fe_load(z + i + half), r, r_hi_only != 0));
}

// G[k] = sum_{j : idx_j = k} E[j]: one workgroup per k (K <= 65536, T * K index reads)
__global__ __launch_bounds__(OP_THREADS) void k_onehot_G(const int32_t* __restrict__ idx, const Fr* __restrict__ E, size_t T,
                                                         Fr* __restrict__ G /* [gridDim.y][K] partial sums per slice of T */) {
    __shared__ Fr red[OP_THREADS / 64];
    const int32_t k = (int32_t)blockIdx.x;
    const size_t per = (T + gridDim.y - 1) / gridDim.y, j0 = (size_t)blockIdx.y * per, j1 = j0 + per < T ? j0 + per : T;
    G += (size_t)blockIdx.y * gridDim.x;
    Fr acc = fe_zero();
    for (size_t j = j0 + threadIdx.x; j < j1; j += OP_THREADS)
        if (idx[j] == k) acc = fr_add(acc, fe_load(E + j));
    acc = fr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Fr s = red[0];
        for (int w = 1; w < OP_THREADS / 64; w++) s = fr_add(s, red[w]);
        fe_store(G + k, s);
    }
}

__global__ __launch_bounds__(OP_THREADS) void k_onehot_gather(const int32_t* __restrict__ idx, const Fr* __restrict__ F, size_t T,
                                                              Fr* __restrict__ out) {
    for (size_t j = (size_t)blockIdx.x * OP_THREADS + threadIdx.x; j < T; j += (size_t)gridDim.x * OP_THREADS) {
        const int32_t k = idx[j];
        fe_store(out + j, k < 0 ? fe_zero() : fe_load(F + k));
    }
}

inline unsigned grid_for(size_t work, size_t cap = 2048) {
    size_t b = (work + OP_THREADS - 1) / OP_THREADS;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

// device half of the HighToLow split-eq: suffix tables of w_in and w_out
struct GseDevH {
    H::GseStateH st;
    Fr *d_w = nullptr, *d_ein = nullptr, *d_eout = nullptr, *d_part = nullptr, *d_sum = nullptr;
    int init(const H::Fr* w, size_t n) {
        st.init(w, n);
        if (st.k_in > 13 || st.k_out > 13) return fail(ATLAS_EINVAL, "split-eq (HighToLow): more than 26 variables not supported");
        HIP_TRY(hipMalloc(&d_w, (n ? n : 1) * sizeof(Fr)));
        HIP_TRY(hipMalloc(&d_ein, ((size_t)2 << st.k_in) * sizeof(Fr)));
        HIP_TRY(hipMalloc(&d_eout, ((size_t)2 << st.k_out) * sizeof(Fr)));
        HIP_TRY(hipMalloc(&d_part, 2048 * sizeof(Fr)));
        HIP_TRY(hipMalloc(&d_sum, sizeof(Fr)));
        if (n) HIP_TRY(hipMemcpyAsync(d_w, w, n * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
        k_eq_cached_rev<<<1, 1024, 0, g.stream>>>(d_ein, d_w + (n ? 1 : 0), (uint32_t)st.k_in);
        k_eq_cached_rev<<<1, 1024, 0, g.stream>>>(d_eout, d_w + (n ? 1 : 0) + st.k_in, (uint32_t)st.k_out);
        HIP_TRY(hipStreamSynchronize(g.stream));
        return ATLAS_OK;
    }
    SplitEqView view() const {     // high index bits <- w_in suffix, low bits <- w_out suffix
        SplitEqView E;
        E.e_out = d_ein + (((size_t)1 << st.in_top) - 1);
        E.e_in = d_eout + (((size_t)1 << st.out_top) - 1);
        E.in_bits = (uint32_t)st.out_top;
        return E;
    }
    // q(0) = sum over the lower half of P weighted by the current tables
    template <class T>
    int q0(const T* P, size_t half, H::Fr* out) {
        const unsigned grid = grid_for(half);
        k_open_fold<T><<<grid, OP_THREADS, 0, g.stream>>>(P, half, view(), d_part, make_consts());
        if (grid > 1) k_open_reduce<<<1, OP_THREADS, 0, g.stream>>>(d_part, grid, d_sum);
        HIP_TRY(hipMemcpyAsync(g.h_pinned, grid > 1 ? d_sum : d_part, sizeof(Fr), hipMemcpyDeviceToHost, g.stream));   // one workgroup: its partial is the sum
        HIP_TRY(hipStreamSynchronize(g.stream));
        std::memcpy(out, g.h_pinned, sizeof(Fr));
        return ATLAS_OK;
    }
    void release() { for (Fr* p : {d_w, d_ein, d_eout, d_part, d_sum}) if (p) hipFree(p); d_w = d_ein = d_eout = d_part = d_sum = nullptr; }
};

// ---------------------------------------------------------------- DensePolynomialProverOpening
struct DenseOpening : atlas_instance {
    atlas_poly_t P = nullptr;
    GseDevH D;
    size_t n = 0, round_next = 0;
    ~DenseOpening() override { if (P) atlas_poly_free(P); D.release(); }
    size_t rounds() const override { return n; }
    size_t degree() const override { return 2; }
    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= n) return fail(ATLAS_ESTATE, "dense_opening: round out of order");
        H::Fr q0;
        {
            std::lock_guard<atlas_rt::Mutex> lk(g.mu);
            const int rc = P->is_i32 ? D.q0<int32_t>((const int32_t*)P->d, P->len / 2, &q0) : D.q0<Fr>((const Fr*)P->d, P->len / 2, &q0);
            if (rc) return rc;
        }
        coeffs.resize(3);
        H::gruen_deg2(D.st.scalar, D.st.w_cur(), q0, claim, coeffs.data());
        return ATLAS_OK;
    }
    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= n) return fail(ATLAS_ESTATE, "dense_opening: round out of order");
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, g.challenge_mode);
        if (P->is_i32) {                                            // CompactPolynomial first bind: promotes to Fr (and syncs)
            int rc = atlas_poly_bind(P, &r, ATLAS_HIGH_TO_LOW);
            if (rc) return rc;
        } else {                                                    // in place, stream-ordered, no host wait
            std::lock_guard<atlas_rt::Mutex> lk(g.mu);
            const size_t half = P->len / 2;
            k_open_bind_hi<<<grid_for(half, 4096), OP_THREADS, 0, g.stream>>>((Fr*)P->d, half, to_dev(rf), g.challenge_mode == 0 ? 1 : 0);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(ATLAS_ENODEV, "dense_opening: bind", e);
            P->len = half;
        }
        D.st.bind(rf);
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override {
        out.resize(1);
        return atlas_poly_final_claim(P, (atlas_fr_t*)out.data());
    }
};

// ---------------------------------------------------------------- OneHotPolynomialProverOpening
struct OneHotOpening : atlas_instance {
    size_t log_K = 0, log_T = 0, round_next = 0;
    H::Fr eqa_inv_ = H::zero(); bool have_inv_ = false;
    const H::Fr& eqa_inv() { if (!have_inv_) { eqa_inv_ = H::inv(B[0]); have_inv_ = true; } return eqa_inv_; }   // B is fully bound
    std::vector<H::Fr> B, F, G;          // eq(r_address, .) bound HighToLow; expanding table; histogram
    int32_t* d_idx = nullptr;
    Fr* d_H = nullptr;
    size_t H_len = 0;
    GseDevH D;
    ~OneHotOpening() override { if (d_idx) hipFree(d_idx); if (d_H) hipFree(d_H); D.release(); }
    size_t rounds() const override { return log_K + log_T; }
    size_t degree() const override { return 2; }

    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "onehot_opening: round out of order");
        coeffs.assign(3, H::zero());
        if (round < log_K) {                                         // opening_reduction.rs:583-633
            const size_t unbound = log_K - round, K = (size_t)1 << log_K, half = B.size() / 2;
            H::Fr e0 = H::zero(), e2 = H::zero();
            for (size_t kp = 0; kp < half; kp++) {
                const H::Fr b0 = B[kp], b2 = H::add(B[kp + half], H::sub(B[kp + half], b0));
                H::Fr s0 = H::zero(), s2 = H::zero();
                for (size_t k = kp; k < K; k += half) {
                    const H::Fr gf = H::mul(G[k], F[k >> unbound]);
                    if (((k >> (unbound - 1)) & 1) == 0) { s0 = H::add(s0, gf); s2 = H::sub(s2, gf); }
                    else s2 = H::add(s2, H::add(gf, gf));
                }
                e0 = H::add(e0, H::mul(b0, s0)); e2 = H::add(e2, H::mul(b2, s2));
            }
            const H::Fr ev[2] = {e0, e2};
            H::unipoly_from_evals_and_hint(claim, ev, 2, coeffs.data());
            return ATLAS_OK;
        }
        H::Fr q0;                                                    // :634-676
        {
            std::lock_guard<atlas_rt::Mutex> lk(g.mu);
            const int rc = D.q0<Fr>(d_H, H_len / 2, &q0);
            if (rc) return rc;
        }
        const H::Fr eqa = B[0];
        H::gruen_deg2(D.st.scalar, D.st.w_cur(), q0, H::mul(claim, eqa_inv()), coeffs.data());
        for (auto& c : coeffs) c = H::mul(c, eqa);                   // UniPoly * F -> from_coeff
        H::trim(coeffs);
        return ATLAS_OK;
    }

    int ingest(const atlas_u128_t& r, size_t round) override {      // :679-718
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "onehot_opening: round out of order");
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, g.challenge_mode);
        if (round < log_K) {
            const size_t half = B.size() / 2;
            for (size_t i = 0; i < half; i++) B[i] = H::add(B[i], H::mul(rf, H::sub(B[i + half], B[i])));
            B.resize(half);
            std::vector<H::Fr> nf(2 * F.size());                     // ExpandingTable::update, HighToLow
            for (size_t i = 0; i < F.size(); i++) { nf[2 * i + 1] = H::mul(rf, F[i]); nf[2 * i] = H::sub(F[i], nf[2 * i + 1]); }
            F.swap(nf);
            if (round == log_K - 1) {
                std::lock_guard<atlas_rt::Mutex> lk(g.mu);
                const size_t T = (size_t)1 << log_T;
                DevBuf Fb;
                HIP_TRY(Fb.alloc(F.size() * sizeof(Fr)));
                HIP_TRY(hipMemcpyAsync(Fb.p, F.data(), F.size() * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
                k_onehot_gather<<<grid_for(T, 4096), OP_THREADS, 0, g.stream>>>(d_idx, Fb.as<Fr>(), T, d_H);
                hipError_t e = hipStreamSynchronize(g.stream);
                if (e != hipSuccess) return fail(ATLAS_ENODEV, "onehot_opening: gather", e);
                H_len = T;
                G.clear();
            }
        } else {
            std::lock_guard<atlas_rt::Mutex> lk(g.mu);
            const size_t half = H_len / 2;
            k_open_bind_hi<<<grid_for(half, 4096), OP_THREADS, 0, g.stream>>>(d_H, half, to_dev(rf), g.challenge_mode == 0 ? 1 : 0);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(ATLAS_ENODEV, "onehot_opening: bind", e);
            H_len = half;
            D.st.bind(rf);
        }
        round_next++;
        return ATLAS_OK;
    }

    int finals(std::vector<H::Fr>& out) override {
        if (round_next != rounds()) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        std::lock_guard<atlas_rt::Mutex> lk(g.mu);
        out.resize(1);
        HIP_TRY(hipMemcpyAsync(g.h_pinned, d_H, sizeof(Fr), hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        std::memcpy(out.data(), g.h_pinned, sizeof(Fr));
        return ATLAS_OK;
    }
};

// ---------------------------------------------------------------- one-hot openings that share r_cycle
// The reference keeps one EqCycleState per distinct r_cycle and binds it once per round for every
// opening that refers to it (opening_reduction.rs:686-698, opening_proof.rs:339-343).  Here the rows of
// such a group also share their launches: one fold, one bind and one reduction per round for all R
// polynomials (a lookup op registers d = 16 of them at a time), instead of R of each.
__global__ __launch_bounds__(OP_THREADS) void k_onehot_G_rows(const int32_t* __restrict__ idx, const Fr* __restrict__ E, size_t T,
                                                              uint32_t K, Fr* __restrict__ G /* [R][K] */) {
    __shared__ Fr red[OP_THREADS / 64];
    const int32_t k = (int32_t)blockIdx.x;
    const size_t row = blockIdx.y;
    const int32_t* ix = idx + row * T;
    const size_t per = (T + gridDim.z - 1) / gridDim.z, j0 = (size_t)blockIdx.z * per, j1 = j0 + per < T ? j0 + per : T;
    G += (size_t)blockIdx.z * gridDim.y * K;                  // partial tables per slice of T
    Fr acc = fe_zero();
    for (size_t j = j0 + threadIdx.x; j < j1; j += OP_THREADS)
        if (ix[j] == k) acc = fr_add(acc, fe_load(E + j));
    acc = fr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Fr s = red[0];
        for (int w = 1; w < OP_THREADS / 64; w++) s = fr_add(s, red[w]);
        fe_store(G + row * K + k, s);
    }
}

__global__ __launch_bounds__(OP_THREADS) void k_open_fold_rows(const Fr* __restrict__ H, size_t stride, size_t half, SplitEqView E,
                                                               Fr* __restrict__ partials /* [R][gridDim.x] */) {
    Fr acc[1];
    acc[0] = fe_zero();
    const Fr* row = H + (size_t)blockIdx.y * stride;
    const size_t mask = ((size_t)1 << E.in_bits) - 1;
    for (size_t j = (size_t)blockIdx.x * OP_THREADS + threadIdx.x; j < half; j += (size_t)gridDim.x * OP_THREADS) {
        const Fr w = fr_mul(fe_load(E.e_out + (j >> E.in_bits)), fe_load(E.e_in + (j & mask)));
        acc[0] = fr_add(acc[0], fr_mul(w, fe_load(row + j)));
    }
    block_reduce_store<1>(acc, partials + (size_t)blockIdx.y * gridDim.x);
}

__global__ __launch_bounds__(OP_THREADS) void k_open_reduce_rows(const Fr* __restrict__ partials, uint32_t n, Fr* out) {
    __shared__ Fr red[OP_THREADS / 64];
    const Fr* p = partials + (size_t)blockIdx.x * n;
    Fr acc = fe_zero();
    for (uint32_t i = threadIdx.x; i < n; i += OP_THREADS) acc = fr_add(acc, fe_load(p + i));
    acc = fr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Fr s = red[0];
        for (int w = 1; w < OP_THREADS / 64; w++) s = fr_add(s, red[w]);
        fe_store(out + blockIdx.x, s);
    }
}

__global__ __launch_bounds__(OP_THREADS) void k_open_bind_hi_rows(Fr* H, size_t stride, size_t half, Fr r, int r_hi_only) {
    Fr* z = H + (size_t)blockIdx.y * stride;
    for (size_t i = (size_t)blockIdx.x * OP_THREADS + threadIdx.x; i < half; i += (size_t)gridDim.x * OP_THREADS)
        fe_store(z + i, bind_pair(fe_load(z + i), fe_load(z + i + half), r, r_hi_only != 0));
}

struct OneHotGroup {
    size_t R = 0, log_K = 0, log_T = 0, T = 0, H_len = 0;
    int32_t* d_idx = nullptr;            // [R][T]
    Fr *d_H = nullptr, *d_part = nullptr, *d_q0 = nullptr;
    GseDevH D;
    std::vector<std::vector<H::Fr>> G;   // [R][K]
    std::vector<H::Fr> q0;               // cycle round cache: one fold per round for all rows
    H::Fr q0_scalar = H::one(), q0_w = H::zero();   // the split-eq's scalar and w of that round
    size_t q0_round = (size_t)-1, bound_rounds = 0;   // cycle rounds whose challenge has been applied
    size_t refs = 0;
    ~OneHotGroup() { for (void* p : {(void*)d_idx, (void*)d_H, (void*)d_part, (void*)d_q0}) if (p) hipFree(p); D.release(); }

    int fold_all(size_t cycle_round) {   // H rows at the current length, all rows at once
        if (q0_round == cycle_round) return ATLAS_OK;
        if (bound_rounds != cycle_round) return fail(ATLAS_ESTATE, "onehot group: rows of a group must advance round by round together");
        const size_t half = H_len / 2;
        const unsigned grid = grid_for(half, 512);
        k_open_fold_rows<<<dim3(grid, (unsigned)R), OP_THREADS, 0, g.stream>>>(d_H, T, half, D.view(), d_part);
        k_open_reduce_rows<<<(unsigned)R, OP_THREADS, 0, g.stream>>>(d_part, grid, d_q0);
        q0.resize(R);
        HIP_TRY(hipMemcpyAsync(q0.data(), d_q0, R * sizeof(Fr), hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        q0_round = cycle_round; q0_scalar = D.st.scalar; q0_w = D.st.w_cur();
        return ATLAS_OK;
    }
    int bind_all(size_t cycle_round, const H::Fr& rf) {   // "if num_variables_bound <= round" (opening_reduction.rs:686-698)
        if (bound_rounds > cycle_round) return ATLAS_OK;
        const size_t half = H_len / 2;
        k_open_bind_hi_rows<<<dim3(grid_for(half, 1024), (unsigned)R), OP_THREADS, 0, g.stream>>>(d_H, T, half, to_dev(rf), g.challenge_mode == 0 ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "onehot group bind", e);
        H_len = half;
        D.st.bind(rf);
        bound_rounds = cycle_round + 1;
        return ATLAS_OK;
    }
};

struct OneHotRow : atlas_instance {
    OneHotGroup* grp = nullptr;
    size_t row = 0, round_next = 0;
    H::Fr eqa_inv_ = H::zero(); bool have_inv_ = false;
    const H::Fr& eqa_inv() { if (!have_inv_) { eqa_inv_ = H::inv(B[0]); have_inv_ = true; } return eqa_inv_; }
    std::vector<H::Fr> B, F;
    ~OneHotRow() override { if (grp && --grp->refs == 0) delete grp; }
    size_t rounds() const override { return grp->log_K + grp->log_T; }
    size_t degree() const override { return 2; }

    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "onehot_opening: round out of order");
        coeffs.assign(3, H::zero());
        const size_t log_K = grp->log_K;
        if (round < log_K) {
            const std::vector<H::Fr>& G = grp->G[row];
            const size_t unbound = log_K - round, K = (size_t)1 << log_K, half = B.size() / 2;
            H::Fr e0 = H::zero(), e2 = H::zero();
            for (size_t kp = 0; kp < half; kp++) {
                const H::Fr b0 = B[kp], b2 = H::add(B[kp + half], H::sub(B[kp + half], b0));
                H::Fr s0 = H::zero(), s2 = H::zero();
                for (size_t k = kp; k < K; k += half) {
                    const H::Fr gf = H::mul(G[k], F[k >> unbound]);
                    if (((k >> (unbound - 1)) & 1) == 0) { s0 = H::add(s0, gf); s2 = H::sub(s2, gf); }
                    else s2 = H::add(s2, H::add(gf, gf));
                }
                e0 = H::add(e0, H::mul(b0, s0)); e2 = H::add(e2, H::mul(b2, s2));
            }
            const H::Fr ev[2] = {e0, e2};
            H::unipoly_from_evals_and_hint(claim, ev, 2, coeffs.data());
            return ATLAS_OK;
        }
        {
            std::lock_guard<atlas_rt::Mutex> lk(g.mu);
            const int rc = grp->fold_all(round - log_K);
            if (rc) return rc;
        }
        const H::Fr eqa = B[0];
        H::gruen_deg2(grp->q0_scalar, grp->q0_w, grp->q0[row], H::mul(claim, eqa_inv()), coeffs.data());
        for (auto& c : coeffs) c = H::mul(c, eqa);
        H::trim(coeffs);
        return ATLAS_OK;
    }

    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "onehot_opening: round out of order");
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, g.challenge_mode);
        const size_t log_K = grp->log_K;
        if (round < log_K) {
            const size_t half = B.size() / 2;
            for (size_t i = 0; i < half; i++) B[i] = H::add(B[i], H::mul(rf, H::sub(B[i + half], B[i])));
            B.resize(half);
            std::vector<H::Fr> nf(2 * F.size());
            for (size_t i = 0; i < F.size(); i++) { nf[2 * i + 1] = H::mul(rf, F[i]); nf[2 * i] = H::sub(F[i], nf[2 * i + 1]); }
            F.swap(nf);
            if (round == log_K - 1) {                                // this row's H = F[idx]
                std::lock_guard<atlas_rt::Mutex> lk(g.mu);
                const size_t T = grp->T;
                DevBuf Fb;
                HIP_TRY(Fb.alloc(F.size() * sizeof(Fr)));
                HIP_TRY(hipMemcpyAsync(Fb.p, F.data(), F.size() * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
                k_onehot_gather<<<grid_for(T, 4096), OP_THREADS, 0, g.stream>>>(grp->d_idx + row * T, Fb.as<Fr>(), T, grp->d_H + row * T);
                hipError_t e = hipStreamSynchronize(g.stream);
                if (e != hipSuccess) return fail(ATLAS_ENODEV, "onehot_opening: gather", e);
                grp->G[row].clear();
            }
        } else {
            std::lock_guard<atlas_rt::Mutex> lk(g.mu);
            const int rc = grp->bind_all(round - log_K, rf);
            if (rc) return rc;
        }
        round_next++;
        return ATLAS_OK;
    }

    int finals(std::vector<H::Fr>& out) override {
        if (round_next != rounds()) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        std::lock_guard<atlas_rt::Mutex> lk(g.mu);
        out.resize(1);
        HIP_TRY(hipMemcpyAsync(g.h_pinned, grp->d_H + row * grp->T, sizeof(Fr), hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        std::memcpy(out.data(), g.h_pinned, sizeof(Fr));
        return ATLAS_OK;
    }
};

}  // namespace

extern "C" {

int atlas_dense_opening_new(atlas_poly_t poly, const atlas_fr_t* opening_point, size_t n, atlas_instance_t* out) {
    NEED_INIT();
    if (!poly || (!opening_point && n) || !out) return fail(ATLAS_EINVAL, "dense_opening_new: null argument");
    if (n == 0 || poly->len != ((size_t)1 << n)) return fail(ATLAS_EINVAL, "dense_opening_new: polynomial length != 2^n, n >= 1");
    std::lock_guard<atlas_rt::Mutex> lk(g.mu);
    DenseOpening* P = new DenseOpening();
    P->n = n;
    int rc = P->D.init(reinterpret_cast<const H::Fr*>(opening_point), n);
    if (rc) { delete P; return rc; }
    P->P = poly;                                                     // ownership moves (bound in place)
    *out = P;
    return ATLAS_OK;
}

int atlas_onehot_opening_new(const int32_t* nonzero_indices, size_t log_K, size_t log_T, const atlas_fr_t* r_address,
                             const atlas_fr_t* r_cycle, atlas_instance_t* out) {
    NEED_INIT();
    if (!nonzero_indices || !r_address || !r_cycle || !out) return fail(ATLAS_EINVAL, "onehot_opening_new: null argument");
    if (log_K == 0 || log_K > 16 || log_T == 0 || log_T > 26) return fail(ATLAS_EINVAL, "onehot_opening_new: 1 <= log_K <= 16, 1 <= log_T <= 26");
    const size_t K = (size_t)1 << log_K, T = (size_t)1 << log_T;
    for (size_t j = 0; j < T; j++)          // the reference indexes F[k] / G[k] with bounds checks (opening_reduction.rs:532-560)
        if (nonzero_indices[j] >= (int32_t)K) return fail(ATLAS_EINVAL, "onehot_opening_new: index >= K");
    // D.merge() before any bind = EqPolynomial::evals(r_cycle) (scalar 1): device table for the histogram
    atlas_poly_t E = nullptr;
    int rc = atlas_eq_evals(r_cycle, log_T, nullptr, &E);
    if (rc) return rc;
    std::lock_guard<atlas_rt::Mutex> lk(g.mu);
    OneHotOpening* P = new OneHotOpening();
    P->log_K = log_K; P->log_T = log_T;
    P->B = H::eq_evals(reinterpret_cast<const H::Fr*>(r_address), log_K);   // EqAddressState::new
    P->F = {H::one()};
    P->G.resize(K);
    Fr* d_G = nullptr;
    hipError_t e = hipMalloc(&P->d_idx, T * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&P->d_H, T * sizeof(Fr));
    // slices of T so that about 2048 workgroups are in flight; the per-slice partial tables are summed on the host
    size_t slices = 2048 / K; if (slices < 1) slices = 1; if (slices > (T + 4095) / 4096) slices = (T + 4095) / 4096;
    std::vector<H::Fr> hG(slices * K);
    if (e == hipSuccess) e = hipMalloc(&d_G, slices * K * sizeof(Fr));
    if (e == hipSuccess) e = hipMemcpyAsync(P->d_idx, nonzero_indices, T * sizeof(int32_t), hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess) {
        k_onehot_G<<<dim3((unsigned)K, (unsigned)slices), OP_THREADS, 0, g.stream>>>(P->d_idx, (const Fr*)E->d, T, d_G);
        e = hipMemcpyAsync(hG.data(), d_G, slices * K * sizeof(Fr), hipMemcpyDeviceToHost, g.stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (e == hipSuccess)
        for (size_t k = 0; k < K; k++) {
            H::Fr acc = H::zero();
            for (size_t sl = 0; sl < slices; sl++) acc = H::add(acc, hG[sl * K + k]);
            P->G[k] = acc;
        }
    if (d_G) hipFree(d_G);
    if (e != hipSuccess) { delete P; hipFree(E->d); delete E; return fail(ATLAS_ENODEV, "onehot_opening_new", e); }
    hipFree(E->d); delete E;
    rc = P->D.init(reinterpret_cast<const H::Fr*>(r_cycle), log_T);
    if (rc) { delete P; return rc; }
    *out = P;
    return ATLAS_OK;
}

int atlas_onehot_opening_group_new(const int32_t* const* nonzero_indices, size_t R, size_t log_K, size_t log_T,
                                   const atlas_fr_t* r_addresses, const atlas_fr_t* r_cycle, atlas_instance_t* out) {
    NEED_INIT();
    if (!nonzero_indices || !r_addresses || !r_cycle || !out || R == 0) return fail(ATLAS_EINVAL, "onehot_opening_group_new: null argument");
    if (log_K == 0 || log_K > 16 || log_T == 0 || log_T > 26) return fail(ATLAS_EINVAL, "onehot_opening_group_new: 1 <= log_K <= 16, 1 <= log_T <= 26");
    const size_t K = (size_t)1 << log_K, T = (size_t)1 << log_T;
    for (size_t r = 0; r < R; r++) {
        if (!nonzero_indices[r]) return fail(ATLAS_EINVAL, "onehot_opening_group_new: null index row");
        for (size_t j = 0; j < T; j++)
            if (nonzero_indices[r][j] >= (int32_t)K) return fail(ATLAS_EINVAL, "onehot_opening_group_new: index >= K");
    }
    atlas_poly_t E = nullptr;
    int rc = atlas_eq_evals(r_cycle, log_T, nullptr, &E);
    if (rc) return rc;
    std::lock_guard<atlas_rt::Mutex> lk(g.mu);
    OneHotGroup* Gp = new OneHotGroup();
    Gp->R = R; Gp->log_K = log_K; Gp->log_T = log_T; Gp->T = T; Gp->H_len = T;
    Fr* d_G = nullptr;
    hipError_t e = hipMalloc(&Gp->d_idx, R * T * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&Gp->d_H, R * T * sizeof(Fr));
    if (e == hipSuccess) e = hipMalloc(&Gp->d_part, R * 512 * sizeof(Fr));
    if (e == hipSuccess) e = hipMalloc(&Gp->d_q0, R * sizeof(Fr));
    size_t slices = 2048 / (K * R); if (slices < 1) slices = 1; if (slices > (T + 4095) / 4096) slices = (T + 4095) / 4096;
    if (e == hipSuccess) e = hipMalloc(&d_G, slices * R * K * sizeof(Fr));
    for (size_t r = 0; r < R && e == hipSuccess; r++)
        e = hipMemcpyAsync(Gp->d_idx + r * T, nonzero_indices[r], T * sizeof(int32_t), hipMemcpyHostToDevice, g.stream);
    std::vector<H::Fr> hGs(slices * R * K), hG(R * K, H::zero());
    if (e == hipSuccess) {
        k_onehot_G_rows<<<dim3((unsigned)K, (unsigned)R, (unsigned)slices), OP_THREADS, 0, g.stream>>>(Gp->d_idx, (const Fr*)E->d, T, (uint32_t)K, d_G);
        e = hipMemcpyAsync(hGs.data(), d_G, slices * R * K * sizeof(Fr), hipMemcpyDeviceToHost, g.stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    for (size_t sl = 0; sl < slices && e == hipSuccess; sl++)
        for (size_t q = 0; q < R * K; q++) hG[q] = H::add(hG[q], hGs[sl * R * K + q]);
    if (d_G) hipFree(d_G);
    hipFree(E->d); delete E;
    if (e != hipSuccess) { delete Gp; return fail(ATLAS_ENODEV, "onehot_opening_group_new", e); }
    rc = Gp->D.init(reinterpret_cast<const H::Fr*>(r_cycle), log_T);
    if (rc) { delete Gp; return rc; }
    Gp->G.resize(R);
    Gp->refs = R;
    const H::Fr* ra = reinterpret_cast<const H::Fr*>(r_addresses);
    for (size_t r = 0; r < R; r++) {
        Gp->G[r].assign(hG.begin() + r * K, hG.begin() + (r + 1) * K);
        OneHotRow* P = new OneHotRow();
        P->grp = Gp; P->row = r;
        P->B = H::eq_evals(ra + r * log_K, log_K);
        P->F = {H::one()};
        out[r] = P;
    }
    return ATLAS_OK;
}

}  // extern "C"
