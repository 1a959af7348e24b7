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

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_poly.hpp"
#include "host_threads.hpp"
#include "instance.hpp"
#include "runtime.hpp"
#include "sc_consts.hpp"
#include "spliteq_kernels.hip.h"

using namespace atlas;
namespace H = atlas_host;
using atlas_rt::fail;
using atlas_rt::rt;

int atlas_rt_eq_evals_into(const H::Fr* r, size_t n, Fr* ev);      // spliteq.hip

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
        if (n) HIP_TRY(hipMemcpyAsync(d_w, w, n * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
        k_eq_cached_rev<<<1, 1024, 0, rt().stream>>>(d_ein, d_w + (n ? 1 : 0), (uint32_t)st.k_in);
        k_eq_cached_rev<<<1, 1024, 0, rt().stream>>>(d_eout, d_w + (n ? 1 : 0) + st.k_in, (uint32_t)st.k_out);
        HIP_TRY(hipStreamSynchronize(rt().stream));
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
    // the same without the wait: the sum lands in `host_slot` (pinned, device-visible) once the stream has been synchronised by the caller
    template <class T>
    int q0_async(const T* P, size_t half, Fr* host_slot) {
        const unsigned grid = grid_for(half);
        k_open_fold<T><<<grid, OP_THREADS, 0, rt().stream>>>(P, half, view(), d_part, make_consts());
        k_open_reduce<<<1, OP_THREADS, 0, rt().stream>>>(d_part, grid, host_slot);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? ATLAS_OK : fail(ATLAS_ENODEV, "dense_opening: fold", e);
    }
    template <class T>
    int q0(const T* P, size_t half, H::Fr* out) {
        const unsigned grid = grid_for(half);
        k_open_fold<T><<<grid, OP_THREADS, 0, rt().stream>>>(P, half, view(), d_part, make_consts());
        if (grid > 1) k_open_reduce<<<1, OP_THREADS, 0, rt().stream>>>(d_part, grid, d_sum);
        HIP_TRY(hipMemcpyAsync(rt().h_pinned, grid > 1 ? d_sum : d_part, sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));   // one workgroup: its partial is the sum
        HIP_TRY(hipStreamSynchronize(rt().stream));
        std::memcpy(out, rt().h_pinned, sizeof(Fr));
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
    // In a batch of thousands (the opening reduction of a graph holds ~50 dense members beside the one-hot rows) a synchronisation per dense
    // member and round was 1.3 ms per round: the driver calls shared_message_step of every member first — this one launches its fold, the sum
    // going to a pinned slot — synchronises ONCE (rt().pending_async), and message() finds the sum there.
    atlas::Chunk* slot = nullptr;
    size_t slot_round = (size_t)-1;
    bool host_parallel() const override { return true; }
    int shared_message_step(size_t round) override {
        if (round != round_next || round >= n) return ATLAS_OK;
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        if (!slot) slot = rt().chan.alloc_long(2);
        const int rc = P->is_i32 ? D.q0_async<int32_t>((const int32_t*)P->d, P->len / 2, reinterpret_cast<Fr*>(slot)) : D.q0_async<Fr>((const Fr*)P->d, P->len / 2, reinterpret_cast<Fr*>(slot));
        if (rc) return rc;
        slot_round = round;
        rt().pending_async++;
        return ATLAS_OK;
    }
    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= n) return fail(ATLAS_ESTATE, "dense_opening: round out of order");
        H::Fr q0;
        if (slot_round == round) {                                  // launched by shared_message_step; the driver has synchronised since
            std::memcpy(&q0, slot, sizeof(q0));
            slot_round = (size_t)-1;
        } else {
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            const int rc = P->is_i32 ? D.q0<int32_t>((const int32_t*)P->d, P->len / 2, &q0) : D.q0<Fr>((const Fr*)P->d, P->len / 2, &q0);
            if (rc) return rc;
        }
        coeffs.resize(3);
        H::gruen_deg2(D.st.scalar, D.st.w_cur(), q0, claim, coeffs.data());
        return ATLAS_OK;
    }
    // the device half of ingest_challenge (the driver's thread calls it through shared_ingest_step before the workers run ingest())
    size_t bound_round = (size_t)-1;
    int bind_device(const atlas_u128_t& r, size_t round) {
        if (bound_round == round) return ATLAS_OK;
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode);
        if (P->is_i32) {                                            // CompactPolynomial first bind: promotes to Fr (and syncs)
            int rc = atlas_poly_bind(P, &r, ATLAS_HIGH_TO_LOW);
            if (rc) return rc;
        } else {                                                    // in place, stream-ordered, no host wait
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            const size_t half = P->len / 2;
            k_open_bind_hi<<<grid_for(half, 4096), OP_THREADS, 0, rt().stream>>>((Fr*)P->d, half, to_dev(rf), rt().challenge_mode == 0 ? 1 : 0);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(ATLAS_ENODEV, "dense_opening: bind", e);
            P->len = half;
        }
        bound_round = round;
        return ATLAS_OK;
    }
    int shared_ingest_step(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= n) return ATLAS_OK;
        return bind_device(r, round);
    }
    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= n) return fail(ATLAS_ESTATE, "dense_opening: round out of order");
        const int rc = bind_device(r, round);
        if (rc) return rc;
        D.st.bind(H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode));
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override {
        out.resize(1);
        return atlas_poly_final_claim(P, (atlas_fr_t*)out.data());
    }
};

// The address-phase message of a one-hot opening (opening_reduction.rs:583-633): e(0), e(2) of sum_k B(k) G[k] F[k >> unbound] in the round's variable.
// G[k] = sum of eq(r_cycle, j) over the cycles j that read k has at most T nonzero entries out of K — GatherSmall's GatherRa: 16 of 2^14 — so the
// sum runs over those (`nz`, built once) instead of the whole address space: the zero terms of the reference's loop add nothing, the field sums are
// exact, and 14 rounds of K multiplications on the transcript's thread were 17 ms of the GPT-2-shaped reduction (r06x).
inline void onehot_address_message(const std::vector<H::Fr>& B, const std::vector<H::Fr>& F, const std::vector<H::Fr>& G, std::vector<uint32_t>& nz, bool& have_nz,
                                   size_t unbound, const H::Fr& claim, H::Fr* coeffs) {
    if (!have_nz) {
        for (size_t k = 0; k < G.size(); k++) if (!(G[k] == H::zero())) nz.push_back((uint32_t)k);
        have_nz = true;
    }
    const size_t half = B.size() / 2;
    H::Fr e0 = H::zero(), e2 = H::zero();
    for (const uint32_t k : nz) {
        const size_t kp = k & (half - 1);
        const H::Fr gf = H::mul(G[k], F[k >> unbound]);
        const H::Fr b0 = B[kp], b2 = H::add(B[kp + half], H::sub(B[kp + half], b0));
        if (((k >> (unbound - 1)) & 1) == 0) { e0 = H::add(e0, H::mul(b0, gf)); e2 = H::sub(e2, H::mul(b2, gf)); }
        else e2 = H::add(e2, H::mul(b2, H::add(gf, gf)));
    }
    const H::Fr ev[2] = {e0, e2};
    H::unipoly_from_evals_and_hint(claim, ev, 2, coeffs);
}

// ---------------------------------------------------------------- OneHotPolynomialProverOpening
struct OneHotOpening : atlas_instance {
    size_t log_K = 0, log_T = 0, round_next = 0;
    H::Fr eqa_inv_ = H::zero(); bool have_inv_ = false;
    const H::Fr& eqa_inv() { if (!have_inv_) { eqa_inv_ = H::inv(B[0]); have_inv_ = true; } return eqa_inv_; }   // B is fully bound
    std::vector<H::Fr> B, F, G;          // eq(r_address, .) bound HighToLow; expanding table; histogram
    std::vector<uint32_t> nz; bool have_nz = false;   // the nonzero entries of G
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
            onehot_address_message(B, F, G, nz, have_nz, log_K - round, claim, coeffs.data());
            return ATLAS_OK;
        }
        H::Fr q0;                                                    // :634-676
        {
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
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
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode);
        if (round < log_K) {
            const size_t half = B.size() / 2;
            for (size_t i = 0; i < half; i++) B[i] = H::add(B[i], H::mul(rf, H::sub(B[i + half], B[i])));
            B.resize(half);
            std::vector<H::Fr> nf(2 * F.size());                     // ExpandingTable::update, HighToLow
            for (size_t i = 0; i < F.size(); i++) { nf[2 * i + 1] = H::mul(rf, F[i]); nf[2 * i] = H::sub(F[i], nf[2 * i + 1]); }
            F.swap(nf);
            if (round == log_K - 1) {
                std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
                const size_t T = (size_t)1 << log_T;
                DevBuf Fb;
                HIP_TRY(Fb.alloc(F.size() * sizeof(Fr)));
                HIP_TRY(hipMemcpyAsync(Fb.p, F.data(), F.size() * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
                k_onehot_gather<<<grid_for(T, 4096), OP_THREADS, 0, rt().stream>>>(d_idx, Fb.as<Fr>(), T, d_H);
                hipError_t e = hipStreamSynchronize(rt().stream);
                if (e != hipSuccess) return fail(ATLAS_ENODEV, "onehot_opening: gather", e);
                H_len = T;
                G.clear();
            }
        } else {
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            const size_t half = H_len / 2;
            k_open_bind_hi<<<grid_for(half, 4096), OP_THREADS, 0, rt().stream>>>(d_H, half, to_dev(rf), rt().challenge_mode == 0 ? 1 : 0);
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
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        out.resize(1);
        HIP_TRY(hipMemcpyAsync(rt().h_pinned, d_H, sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
        std::memcpy(out.data(), rt().h_pinned, sizeof(Fr));
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
        k_open_fold_rows<<<dim3(grid, (unsigned)R), OP_THREADS, 0, rt().stream>>>(d_H, T, half, D.view(), d_part);
        k_open_reduce_rows<<<(unsigned)R, OP_THREADS, 0, rt().stream>>>(d_part, grid, d_q0);
        q0.resize(R);
        HIP_TRY(hipMemcpyAsync(q0.data(), d_q0, R * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
        q0_round = cycle_round; q0_scalar = D.st.scalar; q0_w = D.st.w_cur();
        return ATLAS_OK;
    }
    int bind_all(size_t cycle_round, const H::Fr& rf) {   // "if num_variables_bound <= round" (opening_reduction.rs:686-698)
        if (bound_rounds > cycle_round) return ATLAS_OK;
        const size_t half = H_len / 2;
        k_open_bind_hi_rows<<<dim3(grid_for(half, 1024), (unsigned)R), OP_THREADS, 0, rt().stream>>>(d_H, T, half, to_dev(rf), rt().challenge_mode == 0 ? 1 : 0);
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
    std::vector<uint32_t> nz; bool have_nz = false;   // the nonzero entries of the row's G
    ~OneHotRow() override { if (grp && --grp->refs == 0) delete grp; }
    size_t rounds() const override { return grp->log_K + grp->log_T; }
    size_t degree() const override { return 2; }

    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "onehot_opening: round out of order");
        coeffs.assign(3, H::zero());
        const size_t log_K = grp->log_K;
        if (round < log_K) {
            onehot_address_message(B, F, grp->G[row], nz, have_nz, log_K - round, claim, coeffs.data());
            return ATLAS_OK;
        }
        {
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
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
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode);
        const size_t log_K = grp->log_K;
        if (round < log_K) {
            const size_t half = B.size() / 2;
            for (size_t i = 0; i < half; i++) B[i] = H::add(B[i], H::mul(rf, H::sub(B[i + half], B[i])));
            B.resize(half);
            std::vector<H::Fr> nf(2 * F.size());
            for (size_t i = 0; i < F.size(); i++) { nf[2 * i + 1] = H::mul(rf, F[i]); nf[2 * i] = H::sub(F[i], nf[2 * i + 1]); }
            F.swap(nf);
            if (round == log_K - 1) {                                // this row's H = F[idx]
                std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
                const size_t T = grp->T;
                DevBuf Fb;
                HIP_TRY(Fb.alloc(F.size() * sizeof(Fr)));
                HIP_TRY(hipMemcpyAsync(Fb.p, F.data(), F.size() * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
                k_onehot_gather<<<grid_for(T, 4096), OP_THREADS, 0, rt().stream>>>(grp->d_idx + row * T, Fb.as<Fr>(), T, grp->d_H + row * T);
                hipError_t e = hipStreamSynchronize(rt().stream);
                if (e != hipSuccess) return fail(ATLAS_ENODEV, "onehot_opening: gather", e);
                grp->G[row].clear();
            }
        } else {
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            const int rc = grp->bind_all(round - log_K, rf);
            if (rc) return rc;
        }
        round_next++;
        return ATLAS_OK;
    }

    int finals(std::vector<H::Fr>& out) override {
        if (round_next != rounds()) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        out.resize(1);
        HIP_TRY(hipMemcpyAsync(rt().h_pinned, grp->d_H + row * grp->T, sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
        std::memcpy(out.data(), rt().h_pinned, sizeof(Fr));
        return ATLAS_OK;
    }
};

// ---------------------------------------------------------------- every one-hot opening of a proof in one set of launches
// prove_reduced_openings of a model graph registers thousands of one-hot chunk polynomials (nanoGPT: ~3000, in ~400 groups of
// equal r_cycle); stepping the groups one by one is ~400 fold + reduce + copy + synchronise per round (207 ms of a 1.25 s proof).
// OneHotPool steps ALL of them together: per global round ONE fold launch over every row in its cycle phase (a row descriptor
// carries its current half length and its group's split-eq tables), one reduction, one copy and one synchronisation; one bind
// launch; one gather launch for the rows whose address phase ends.  The rows come straight from the device-resident lookup
// indices of the node witnesses (nonzero index = (lookup >> shift) & (K - 1)): no host rows, no uploads.
// Rows of fewer rounds start later (front-loaded batching, sumcheck.rs:30-184): row r takes part in global rounds
// [max_rounds - rounds(r), max_rounds).
// H = F[idx] (T entries of K distinct values) is never materialised: the row's first cycle round folds and binds THROUGH the index row (4 bytes an
// entry instead of 32, the F table in cache) and the bind writes the first H there is, T / 2 entries — half the pool's vectors (9 -> 4.5 GB for the
// GPT-2-shaped reduction), no gather launch, and the round that moves the most bytes moves an eighth of them.
struct PoolRowDev {
    Fr* H; const int32_t* idx; const Fr* aux;        // idx / aux: set in a row's FIRST cycle round only: its index row and its F table (K Fr)
    const Fr *e_out, *e_in;
    uint32_t half, in_bits, slot, T;
};
constexpr unsigned POOL_GX = 32;                     // workgroups along a row (a row of 2^14 entries: 8192 pairs = 32 x 256)
__global__ __launch_bounds__(OP_THREADS) void k_pool_chunk_rows(const uint64_t* const* __restrict__ lookups, const uint32_t* __restrict__ shift,
                                                                const uint64_t* __restrict__ off, const uint32_t* __restrict__ Ts, uint32_t mask,
                                                                int32_t* __restrict__ idx) {
    const size_t r = blockIdx.y;
    const uint64_t* lk = lookups[r];
    int32_t* o = idx + off[r];
    const uint32_t sh = shift[r], T = Ts[r];
    for (size_t j = (size_t)blockIdx.x * OP_THREADS + threadIdx.x; j < T; j += (size_t)gridDim.x * OP_THREADS)
        o[j] = (int32_t)(sh >= 64 ? 0 : ((lk[j] >> sh) & mask));
}
// G[r][k] = sum_{j : idx_r[j] = k} E_g(r)[j] as 64-bit sums of the eight 32-bit words of the (Montgomery) residues per bin, K <= 16.
// The LDS counters are kept in 32 copies, one per eight threads: with ONE copy the 256 threads of a workgroup met sixteen deep on each of the
// 16 x 8 counters (8.7 ms for the 8292 rows of the GPT-2-shaped reduction, r06s).  Layout [copy][word][bin], a copy 144 counters apart, so that the
// lanes of one instruction (one word, sixteen bins, eight copies) spread over the banks.  (Measured and dropped: bin-owner threads that pick
// their entries out of the index stream without atomics — the scattered 32-byte loads of the matching lanes made it 5 x slower, r06t.)
constexpr uint32_t HIST_COPIES = 32, HIST_STRIDE = 144;
__global__ __launch_bounds__(OP_THREADS) void k_pool_hist(const int32_t* __restrict__ idx, const uint64_t* __restrict__ off, const uint32_t* __restrict__ Ts,
                                                          const Fr* const* __restrict__ E, uint32_t K, uint32_t n_rows, unsigned long long* __restrict__ out /* [rows][16][8] */) {
    __shared__ unsigned long long acc[HIST_COPIES * HIST_STRIDE];
    // workgroups are dealt to the 8 XCDs round-robin, each with its own L2: row = xcd * ceil(rows / 8) + k keeps the rows of a group (consecutive, and
    // all reading the group's eq table E) on ONE XCD, so that the table comes from HBM once instead of once per XCD
    const size_t per = (gridDim.x + 7) / 8;
    const size_t r = (size_t)(blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (r >= n_rows) return;
    for (uint32_t i = threadIdx.x; i < HIST_COPIES * HIST_STRIDE; i += OP_THREADS) acc[i] = 0;
    __syncthreads();
    const int32_t* ix = idx + off[r];
    const Fr* e = E[r];
    const uint32_t T = Ts[r];
    unsigned long long* mine = acc + (threadIdx.x >> 3) * HIST_STRIDE;
    for (uint32_t j = threadIdx.x; j < T; j += OP_THREADS) {
        const Fr v = fe_load(e + j);
        const int32_t k = ix[j];
#pragma unroll
        for (int w = 0; w < 8; w++) atomicAdd(&mine[w * 16 + k], (unsigned long long)v.v[w]);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < K * 8; i += OP_THREADS) {
        const uint32_t k = i >> 3, w = i & 7u;
        unsigned long long s = 0;
        for (uint32_t c = 0; c < HIST_COPIES; c++) s += acc[c * HIST_STRIDE + w * 16 + k];
        out[r * 16 * 8 + i] = s;
    }
}
__global__ __launch_bounds__(OP_THREADS) void k_pool_fold(const PoolRowDev* __restrict__ rows, Fr* __restrict__ partials /* [rows][POOL_GX] */) {
    const PoolRowDev R = rows[blockIdx.y];
    Fr acc[1];
    acc[0] = fe_zero();
    const size_t mask = ((size_t)1 << R.in_bits) - 1;
    for (size_t j = (size_t)blockIdx.x * OP_THREADS + threadIdx.x; j < R.half; j += (size_t)gridDim.x * OP_THREADS) {
        const Fr w = fr_mul(fe_load(R.e_out + (j >> R.in_bits)), fe_load(R.e_in + (j & mask)));
        const Fr h = R.idx ? fe_load(R.aux + R.idx[j]) : fe_load(R.H + j);      // a row's FIRST cycle round reads H = F[idx] through its index row (below)
        acc[0] = fr_add(acc[0], fr_mul(w, h));
    }
    block_reduce_store<1>(acc, partials + (size_t)R.slot * gridDim.x);
}
__global__ __launch_bounds__(64) void k_pool_reduce(const Fr* __restrict__ partials, uint32_t gx, Fr* __restrict__ out) {
    Fr acc = fe_zero();
    for (uint32_t i = threadIdx.x; i < gx; i += 64) acc = fr_add(acc, fe_load(partials + (size_t)blockIdx.x * gx + i));
    acc = fr_wave_sum(acc);
    if (threadIdx.x == 0) fe_store(out + blockIdx.x, acc);
}
__global__ __launch_bounds__(OP_THREADS) void k_pool_bind(const PoolRowDev* __restrict__ rows, Fr r, int r_hi_only) {
    const PoolRowDev R = rows[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * OP_THREADS + threadIdx.x; i < R.half; i += (size_t)gridDim.x * OP_THREADS) {
        const Fr a = R.idx ? fe_load(R.aux + R.idx[i]) : fe_load(R.H + i);
        const Fr b = R.idx ? fe_load(R.aux + R.idx[i + R.half]) : fe_load(R.H + i + R.half);
        fe_store(R.H + i, bind_pair(a, b, r, r_hi_only != 0));
    }
}
__global__ __launch_bounds__(OP_THREADS) void k_pool_heads(const Fr* __restrict__ H, const uint64_t* __restrict__ off, uint32_t n, Fr* __restrict__ out) {
    for (uint32_t r = blockIdx.x * OP_THREADS + threadIdx.x; r < n; r += gridDim.x * OP_THREADS) fe_store(out + r, fe_load(H + off[r]));
}

// The eq tables of ALL groups of a pool in two launches (a group = one lookup family's r_cycle; a graph has hundreds): per group the
// constructor used to enqueue k_store_point + k_eq_head + the doubling steps + two k_eq_cached_rev — ~8 launches of one or a few
// workgroups each, 30-50 us per group on the host thread and as many tiny kernels in a row on the device.
struct PoolEqJob { Fr* out; const Fr* w; uint32_t n; uint32_t pad; };
// job = (tabs, r, k) of k_eq_cached_rev; one workgroup per job
__global__ __launch_bounds__(1024) void k_pool_eq_cached_rev(const PoolEqJob* __restrict__ jobs) {
    const PoolEqJob J = jobs[blockIdx.x];
    Fr* tabs = J.out; const Fr* r = J.w; const uint32_t k = J.n;
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
// EqPolynomial::evals(w[0..n)) (big-endian index: w[0] <-> the most significant bit, eq_poly.rs:77-101) of group blockIdx.y: the table over the
// low L = min(n, 8) variables grows in LDS, a workgroup takes 64 values of the high index bits — their factors by direct products — and writes
// 64 x 2^L entries, one multiplication each.  Exact field arithmetic: the same residues as the doubling passes of atlas_rt_eq_evals_into.
constexpr uint32_t POOL_EQ_HI = 64;
__global__ __launch_bounds__(256) void k_pool_eq_full(const PoolEqJob* __restrict__ jobs) {
    __shared__ Fr tab[256];
    __shared__ Fr hif[POOL_EQ_HI];
    const PoolEqJob J = jobs[blockIdx.y];
    const uint32_t n = J.n, L = n < 8 ? n : 8, nh = n - L;
    const uint64_t n_hi = (uint64_t)1 << nh, hi0 = (uint64_t)blockIdx.x * POOL_EQ_HI;
    if (hi0 >= n_hi) return;
    if (threadIdx.x == 0) tab[0] = fr_one();
    __syncthreads();
    for (uint32_t s = 0; s < L; s++) {
        const Fr v = fe_load(J.w + nh + s);
        const bool act = threadIdx.x < (1u << s);
        Fr f = fe_zero(), hi = fe_zero();
        if (act) { f = tab[threadIdx.x]; hi = fr_mul(f, v); }
        __syncthreads();
        if (act) { tab[2 * threadIdx.x + 1] = hi; tab[2 * threadIdx.x] = fr_sub(f, hi); }
        __syncthreads();
    }
    const uint32_t cnt = (uint32_t)(n_hi - hi0 < POOL_EQ_HI ? n_hi - hi0 : POOL_EQ_HI);
    if (threadIdx.x < cnt) {
        const uint64_t hi = hi0 + threadIdx.x;
        Fr f = fr_one();
        const Fr one = fr_one();
        for (uint32_t i = 0; i < nh; i++) {
            const Fr wv = fe_load(J.w + i);
            f = fr_mul(f, ((hi >> (nh - 1 - i)) & 1) ? wv : fr_sub(one, wv));
        }
        hif[threadIdx.x] = f;
    }
    __syncthreads();
    const uint32_t width = 1u << L;
    for (uint32_t e = threadIdx.x; e < cnt * width; e += 256) {
        const uint32_t h = e >> L, t = e & (width - 1);
        fe_store(J.out + ((hi0 + h) << L) + t, fr_mul(hif[h], tab[t]));
    }
}

struct OneHotPool {
    size_t log_K = 0, K = 0, max_rounds = 0, refs = 0;
    struct Group { size_t log_T = 0, T = 0, off = 0; H::GseStateH st; Fr *d_ein = nullptr, *d_eout = nullptr; std::vector<size_t> rows; H::Fr inv_eq1; size_t inv_round = (size_t)-1; };
    struct Row { size_t group = 0, fslot = 0; uint64_t off = 0; std::vector<H::Fr> B, F, G; H::Fr eqa_inv, q0, fin; bool gathered = false; };      // off: into d_idx (T a row); H lives at off / 2
    std::vector<Group> groups;
    std::vector<Row> rows;
    // device: the rows' index vectors and H vectors back to back, the groups' split-eq tables, scratch
    int32_t* d_idx = nullptr;
    Fr *d_H = nullptr, *d_tabs = nullptr, *d_part = nullptr, *d_q0 = nullptr, *d_F = nullptr;
    uint64_t *d_off = nullptr, *d_hoff = nullptr;          // the rows' offsets into d_idx, and into d_H (half of them)
    PoolRowDev *d_desc = nullptr, *h_desc = nullptr;       // h_desc: pinned staging, three regions of rows.size() descriptors (fold, bind, spare)
    Fr* h_q0 = nullptr;                                     // pinned
    // global rounds already folded / bound.  Written LAST by fold_all / bind_all (under rt().mu): a row that reads the current round here without
    // the lock (worker threads of a large batch, host_parallel) sees everything those calls wrote
    std::atomic<size_t> folded{(size_t)-1}, bound{(size_t)-1};
    bool have_finals = false;
    double t_fold = 0, t_bind = 0, t_gather = 0;          // ATLAS_TRACE: wall clock of the shared steps (the driver's thread)
    static bool pool_trace() { static const bool on = getenv("ATLAS_TRACE") != nullptr; return on; }
    ~OneHotPool() {
        if (pool_trace()) fprintf(stderr, "[atlas trace] onehot pool (%zu rows, %zu groups): gather %.3f ms, fold + reduce + copy %.3f ms, bind %.3f ms\n", rows.size(), groups.size(), t_gather, t_fold, t_bind);
        for (void* p : {(void*)d_idx, (void*)d_H, (void*)d_tabs, (void*)d_part, (void*)d_q0, (void*)d_F, (void*)d_off, (void*)d_hoff, (void*)d_desc}) if (p) hipFree(p);
        if (h_desc) (void)hipHostFree(h_desc);
        if (h_q0) (void)hipHostFree(h_q0);
    }
    size_t row_rounds(const Row& r) const { return log_K + groups[r.group].log_T; }
    size_t row_off(const Row& r) const { return max_rounds - row_rounds(r); }
    // cycle round of group g at global round R, or -1
    long cycle_of(const Group& G, size_t R) const { const long c = (long)R - (long)(max_rounds - (log_K + G.log_T)) - (long)log_K; return c >= 0 && c < (long)G.log_T ? c : -1; }
    SplitEqView view(const Group& G) const {
        SplitEqView E;
        E.e_out = G.d_ein + (((size_t)1 << G.st.in_top) - 1);
        E.e_in = G.d_eout + (((size_t)1 << G.st.out_top) - 1);
        E.in_bits = (uint32_t)G.st.out_top;
        return E;
    }
    // the caller holds rt().mu.  The F tables of the rows whose address phase ended in the round before R (their ingests have all run) go to the device:
    // the round's fold and bind read H = F[idx] through them (PoolRowDev)
    int gather_pending(size_t R) {
        std::vector<size_t> todo;
        for (auto& G : groups) if (cycle_of(G, R) == 0) for (size_t r : G.rows) if (!rows[r].gathered) todo.push_back(r);
        if (todo.empty()) return ATLAS_OK;
        std::vector<H::Fr> Fh(todo.size() * K), b0(todo.size());
        for (size_t q = 0; q < todo.size(); q++) {
            Row& Rw = rows[todo[q]];
            std::memcpy(&Fh[q * K], Rw.F.data(), K * sizeof(H::Fr));
            b0[q] = Rw.B[0];
            Rw.fslot = q;                                                // the round's fold and bind read d_F + q K through the row's indices
            Rw.gathered = true; Rw.G.clear();
        }
        {   // eq(r_address, rho)^-1 of every row at once (Montgomery's trick: one inversion)
            std::vector<H::Fr> pre(todo.size());
            H::Fr acc = H::one();
            for (size_t q = 0; q < todo.size(); q++) { pre[q] = acc; acc = H::mul(acc, b0[q]); }
            H::Fr inv = H::inv(acc);
            for (size_t q = todo.size(); q-- > 0;) { rows[todo[q]].eqa_inv = H::mul(inv, pre[q]); inv = H::mul(inv, b0[q]); }
        }
        HIP_TRY(hipMemcpyAsync(d_F, Fh.data(), Fh.size() * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));      // (behind the previous round's bind, the last reader of d_F)
        HIP_TRY(hipStreamSynchronize(rt().stream));                 // Fh leaves scope (pageable source)
        return ATLAS_OK;
    }
    int fold_all(size_t R) {
        if (folded.load(std::memory_order_acquire) == R) return ATLAS_OK;
        const auto tq0 = std::chrono::steady_clock::now();
        int rc = gather_pending(R);
        if (rc) return rc;
        const auto tq1 = std::chrono::steady_clock::now();
        t_gather += std::chrono::duration<double, std::milli>(tq1 - tq0).count();
        struct Acc { double& t; std::chrono::steady_clock::time_point a; ~Acc() { t += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); } } acc_{t_fold, tq1};
        size_t n = 0, max_half = 0;
        for (auto& G : groups) {
            const long c = cycle_of(G, R);
            if (c < 0) continue;
            const size_t half = G.T >> (c + 1);
            const SplitEqView E = view(G);
            for (size_t r : G.rows) {
                const Row& Rw = rows[r];
                h_desc[n] = PoolRowDev{d_H + Rw.off / 2, c == 0 ? d_idx + Rw.off : nullptr, c == 0 ? d_F + Rw.fslot * K : nullptr, E.e_out, E.e_in, (uint32_t)half, E.in_bits, (uint32_t)n, (uint32_t)G.T};
                n++;
            }
            max_half = half > max_half ? half : max_half;
        }
        if (n == 0) { folded.store(R, std::memory_order_release); return ATLAS_OK; }
        const unsigned gx = grid_for(max_half, POOL_GX);
        HIP_TRY(hipMemcpyAsync(d_desc, h_desc, n * sizeof(PoolRowDev), hipMemcpyHostToDevice, rt().stream));
        k_pool_fold<<<dim3(gx, (unsigned)n), OP_THREADS, 0, rt().stream>>>(d_desc, d_part);
        k_pool_reduce<<<(unsigned)n, 64, 0, rt().stream>>>(d_part, gx, d_q0);
        HIP_TRY(hipMemcpyAsync(h_q0, d_q0, n * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
        {   // gruen_poly_deg_2's division by eq(1), shared by the rows of a group: ONE inversion for all the groups of the round (Montgomery's trick),
            // and under the device's fold — an inversion per group after the wait was 14 of the 20 ms this function took in the GPT-2-shaped reduction
            std::vector<Group*> act;
            for (auto& G : groups) if (cycle_of(G, R) >= 0) act.push_back(&G);
            std::vector<H::Fr> val(act.size()), pre(act.size());
            H::Fr run = H::one();
            for (size_t q = 0; q < act.size(); q++) {
                val[q] = H::mul(act[q]->st.scalar, act[q]->st.w_cur());
                pre[q] = run;
                if (!(val[q] == H::zero())) run = H::mul(run, val[q]);           // (a zero — r_cycle coordinate 0 — keeps inv(0) = 0 as H::inv does)
            }
            H::Fr inv = H::inv(run);
            for (size_t q = act.size(); q-- > 0;) {
                if (val[q] == H::zero()) { act[q]->inv_eq1 = H::zero(); }
                else { act[q]->inv_eq1 = H::mul(inv, pre[q]); inv = H::mul(inv, val[q]); }
                act[q]->inv_round = R;
            }
        }
        HIP_TRY(hipStreamSynchronize(rt().stream));
        n = 0;
        for (auto& G : groups) {
            if (cycle_of(G, R) < 0) continue;
            for (size_t r : G.rows) std::memcpy(&rows[r].q0, &h_q0[n++], sizeof(Fr));
        }
        folded.store(R, std::memory_order_release);
        return ATLAS_OK;
    }
    int bind_all(size_t R, const H::Fr& rf) {
        if (bound.load(std::memory_order_acquire) == R) return ATLAS_OK;
        struct Acc { double& t; std::chrono::steady_clock::time_point a; ~Acc() { t += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); } } acc_{t_bind, std::chrono::steady_clock::now()};
        size_t n = 0, max_half = 0;
        PoolRowDev* hd = h_desc + rows.size();
        for (auto& G : groups) {
            const long c = cycle_of(G, R);
            if (c < 0) continue;
            const size_t half = G.T >> (c + 1);
            for (size_t r : G.rows) { const Row& Rw = rows[r]; hd[n++] = PoolRowDev{d_H + Rw.off / 2, c == 0 ? d_idx + Rw.off : nullptr, c == 0 ? d_F + Rw.fslot * K : nullptr, nullptr, nullptr, (uint32_t)half, 0, 0, (uint32_t)G.T}; }
            max_half = half > max_half ? half : max_half;
            G.st.bind(rf);
        }
        if (n == 0) { bound.store(R, std::memory_order_release); return ATLAS_OK; }
        HIP_TRY(hipMemcpyAsync(d_desc + rows.size(), hd, n * sizeof(PoolRowDev), hipMemcpyHostToDevice, rt().stream));
        k_pool_bind<<<dim3(grid_for(max_half, POOL_GX), (unsigned)n), OP_THREADS, 0, rt().stream>>>(d_desc + rows.size(), to_dev(rf), rt().challenge_mode == 0 ? 1 : 0);
        hipError_t e = hipGetLastError();
        bound.store(R, std::memory_order_release);
        return e == hipSuccess ? ATLAS_OK : fail(ATLAS_ENODEV, "onehot pool bind", e);
    }
    int fetch_finals() {
        if (have_finals) return ATLAS_OK;
        const size_t n = rows.size();
        k_pool_heads<<<grid_for(n, 64), OP_THREADS, 0, rt().stream>>>(d_H, d_hoff, (uint32_t)n, d_q0);
        std::vector<H::Fr> f(n);
        HIP_TRY(hipMemcpyAsync(f.data(), d_q0, n * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
        for (size_t r = 0; r < n; r++) rows[r].fin = f[r];
        have_finals = true;
        return ATLAS_OK;
    }
};

struct OneHotPoolRow : atlas_instance {
    OneHotPool* P = nullptr;
    size_t row = 0, round_next = 0;
    ~OneHotPoolRow() override { if (P && --P->refs == 0) delete P; }
    size_t rounds() const override { return P->row_rounds(P->rows[row]); }
    size_t degree() const override { return 2; }
    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "onehot_opening: round out of order");
        OneHotPool::Row& Rw = P->rows[row];
        coeffs.assign(3, H::zero());
        const size_t log_K = P->log_K, K = P->K;
        if (round < log_K) {                                         // opening_reduction.rs:583-633
            const size_t unbound = log_K - round, half = Rw.B.size() / 2;
            H::Fr e0 = H::zero(), e2 = H::zero();
            for (size_t kp = 0; kp < half; kp++) {
                const H::Fr b0 = Rw.B[kp], b2 = H::add(Rw.B[kp + half], H::sub(Rw.B[kp + half], b0));
                H::Fr s0 = H::zero(), s2 = H::zero();
                for (size_t k = kp; k < K; k += half) {
                    const H::Fr gf = H::mul(Rw.G[k], Rw.F[k >> unbound]);
                    if (((k >> (unbound - 1)) & 1) == 0) { s0 = H::add(s0, gf); s2 = H::sub(s2, gf); }
                    else s2 = H::add(s2, H::add(gf, gf));
                }
                e0 = H::add(e0, H::mul(b0, s0)); e2 = H::add(e2, H::mul(b2, s2));
            }
            const H::Fr ev[2] = {e0, e2};
            H::unipoly_from_evals_and_hint(claim, ev, 2, coeffs.data());
            return ATLAS_OK;
        }
        const size_t R = round + P->row_off(Rw);
        if (P->folded.load(std::memory_order_acquire) != R) {        // the first row to ask does the shared fold; the others (worker threads: host_parallel) find it done
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            const int rc = P->fold_all(R);
            if (rc) return rc;
        }
        // gruen_poly_deg_2 (split_eq_poly.rs:379-428) on claim / eq(r_address, rho), the division by eq(1) through the group's shared inverse
        const OneHotPool::Group& G = P->groups[Rw.group];
        const H::Fr eqa = Rw.B[0], cl = H::mul(claim, Rw.eqa_inv);
        const H::Fr eq1 = H::mul(G.st.scalar, G.st.w_cur()), eq0 = H::sub(G.st.scalar, eq1), eqm = H::sub(eq1, eq0), eq2 = H::add(eq1, eqm);
        const H::Fr c0 = H::mul(eq0, Rw.q0), c1 = H::sub(cl, c0);
        const H::Fr l1 = H::mul(c1, G.inv_eq1), l2 = H::sub(H::add(l1, l1), Rw.q0);
        const H::Fr ev[2] = {c0, H::mul(eq2, l2)};
        H::unipoly_from_evals_and_hint(H::add(c0, c1), ev, 2, coeffs.data());
        for (auto& c : coeffs) c = H::mul(c, eqa);                   // UniPoly * F -> from_coeff
        H::trim(coeffs);
        return ATLAS_OK;
    }
    int ingest(const atlas_u128_t& r, size_t round) override {      // :679-718
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "onehot_opening: round out of order");
        OneHotPool::Row& Rw = P->rows[row];
        if (round < P->log_K) {
            const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode);
            const size_t half = Rw.B.size() / 2;
            for (size_t i = 0; i < half; i++) Rw.B[i] = H::add(Rw.B[i], H::mul(rf, H::sub(Rw.B[i + half], Rw.B[i])));
            Rw.B.resize(half);
            std::vector<H::Fr> nf(2 * Rw.F.size());                  // ExpandingTable::update, HighToLow
            for (size_t i = 0; i < Rw.F.size(); i++) { nf[2 * i + 1] = H::mul(rf, Rw.F[i]); nf[2 * i] = H::sub(Rw.F[i], nf[2 * i + 1]); }
            Rw.F.swap(nf);                                           // the gather H = F[idx] runs with the next round's fold (gather_pending)
        } else if (P->bound.load(std::memory_order_acquire) != round + P->row_off(Rw)) {
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            const int rc = P->bind_all(round + P->row_off(Rw), H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode));
            if (rc) return rc;
        }
        round_next++;
        return ATLAS_OK;
    }
    bool host_parallel() const override { return true; }
    const void* shared_key() const override { return P; }
    // the pool's launches of a global round, on the caller's thread (the driver's): the rows' own calls then find them done
    int shared_message_step(size_t round) override {
        if (round != round_next || round >= rounds() || round < P->log_K) return ATLAS_OK;
        const size_t R = round + P->row_off(P->rows[row]);
        if (P->folded.load(std::memory_order_acquire) == R) return ATLAS_OK;
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        return P->fold_all(R);
    }
    int shared_ingest_step(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= rounds() || round < P->log_K) return ATLAS_OK;
        const size_t R = round + P->row_off(P->rows[row]);
        if (P->bound.load(std::memory_order_acquire) == R) return ATLAS_OK;
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        return P->bind_all(R, H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode));
    }
    int finals(std::vector<H::Fr>& out) override {
        if (round_next != rounds()) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        const int rc = P->fetch_finals();
        if (rc) return rc;
        out.assign(1, P->rows[row].fin);
        return ATLAS_OK;
    }
};

// ---------------------------------------------------------------- the dense members of a reduction, stepped together
// A graph's reduction holds ~50 dense members beside its thousands of one-hot rows (the i32 quotient / remainder / auxiliary polynomials).  As
// DenseOpening instances each of them brought its own clone, five allocations, two table launches and a synchronisation at construction, two
// launches per message and one per bind: ~100 tiny kernels in a row per round and as many launches on the thread that runs the transcript —
// a third of the GPT-2-shaped reduction's batched sumcheck (r06v).  DensePool keeps their coefficients (as Fr) and split-eq tables in one
// allocation each and steps them with ONE fold / reduce / bind launch per global round, exactly like OneHotPool; the per-member host state is the
// HighToLow split-eq scalar.  Member r takes part in global rounds [max_rounds - n_r, max_rounds) (front-loaded batching).
struct DenseRowDev { Fr* P; const Fr *e_out, *e_in; uint32_t half, in_bits, slot, pad; };
struct DenseImportJob { const void* src; Fr* dst; uint32_t len, is_i32; };
__global__ __launch_bounds__(OP_THREADS) void k_dpool_import(const DenseImportJob* __restrict__ jobs, ScConsts K) {
    const DenseImportJob J = jobs[blockIdx.y];
    for (size_t j = (size_t)blockIdx.x * OP_THREADS + threadIdx.x; j < J.len; j += (size_t)gridDim.x * OP_THREADS)
        fe_store(J.dst + j, J.is_i32 ? Src<int32_t>::get((const int32_t*)J.src, j, K) : fe_load((const Fr*)J.src + j));
}
__global__ __launch_bounds__(OP_THREADS) void k_dpool_fold(const DenseRowDev* __restrict__ rows, Fr* __restrict__ partials /* [rows][gridDim.x] */) {
    const DenseRowDev R = rows[blockIdx.y];
    Fr acc[1];
    acc[0] = fe_zero();
    const size_t mask = ((size_t)1 << R.in_bits) - 1;
    for (size_t j = (size_t)blockIdx.x * OP_THREADS + threadIdx.x; j < R.half; j += (size_t)gridDim.x * OP_THREADS) {
        const Fr w = fr_mul(fe_load(R.e_out + (j >> R.in_bits)), fe_load(R.e_in + (j & mask)));
        acc[0] = fr_add(acc[0], fr_mul(w, fe_load(R.P + j)));
    }
    block_reduce_store<1>(acc, partials + (size_t)R.slot * gridDim.x);
}
__global__ __launch_bounds__(OP_THREADS) void k_dpool_bind(const DenseRowDev* __restrict__ rows, Fr r, int r_hi_only) {
    const DenseRowDev R = rows[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * OP_THREADS + threadIdx.x; i < R.half; i += (size_t)gridDim.x * OP_THREADS)
        fe_store(R.P + i, bind_pair(fe_load(R.P + i), fe_load(R.P + i + R.half), r, r_hi_only != 0));
}

struct DensePool {
    size_t max_rounds = 0, refs = 0;
    struct Row { size_t n = 0, slot = 0; uint64_t off = 0; H::GseStateH st; Fr *d_ein = nullptr, *d_eout = nullptr; H::Fr fin; };
    std::vector<Row> rows;
    Fr *d_P = nullptr, *d_tabs = nullptr, *d_part = nullptr, *d_q0 = nullptr;
    uint64_t* d_off = nullptr;
    DenseRowDev *d_desc = nullptr, *h_desc = nullptr;      // h_desc: pinned staging, two regions of rows.size() descriptors (fold, bind)
    Fr* h_q0 = nullptr;                                     // pinned
    std::atomic<size_t> folded{(size_t)-1}, bound{(size_t)-1};   // written last by fold_all / bind_all (under rt().mu), read without the lock by the rows
    bool have_finals = false;
    static constexpr unsigned GX = 64;                      // workgroups along a member
    ~DensePool() {
        for (void* p : {(void*)d_P, (void*)d_tabs, (void*)d_part, (void*)d_q0, (void*)d_off, (void*)d_desc}) if (p) hipFree(p);
        if (h_desc) (void)hipHostFree(h_desc);
        if (h_q0) (void)hipHostFree(h_q0);
    }
    long local_round(const Row& r, size_t R) const { const long c = (long)R - (long)(max_rounds - r.n); return c >= 0 && c < (long)r.n ? c : -1; }
    // the caller holds rt().mu.  wait = false: the sums are in h_q0 once the caller's thread has synchronised the stream (rt().pending_async)
    int fold_all(size_t R, bool wait) {
        if (folded.load(std::memory_order_acquire) == R) return ATLAS_OK;
        size_t n = 0, max_half = 0;
        for (auto& Rw : rows) {
            const long c = local_round(Rw, R);
            if (c < 0) continue;
            const size_t half = ((size_t)1 << Rw.n) >> (c + 1);
            h_desc[n] = DenseRowDev{d_P + Rw.off, Rw.d_ein + (((size_t)1 << Rw.st.in_top) - 1), Rw.d_eout + (((size_t)1 << Rw.st.out_top) - 1), (uint32_t)half, (uint32_t)Rw.st.out_top, (uint32_t)n, 0};
            Rw.slot = n++;
            max_half = half > max_half ? half : max_half;
        }
        if (n == 0) { folded.store(R, std::memory_order_release); return ATLAS_OK; }
        const unsigned gx = grid_for(max_half, GX);
        HIP_TRY(hipMemcpyAsync(d_desc, h_desc, n * sizeof(DenseRowDev), hipMemcpyHostToDevice, rt().stream));
        k_dpool_fold<<<dim3(gx, (unsigned)n), OP_THREADS, 0, rt().stream>>>(d_desc, d_part);
        k_pool_reduce<<<(unsigned)n, 64, 0, rt().stream>>>(d_part, gx, d_q0);
        HIP_TRY(hipMemcpyAsync(h_q0, d_q0, n * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
        if (wait) HIP_TRY(hipStreamSynchronize(rt().stream));
        else rt().pending_async++;
        folded.store(R, std::memory_order_release);
        return ATLAS_OK;
    }
    int bind_all(size_t R, const H::Fr& rf) {
        if (bound.load(std::memory_order_acquire) == R) return ATLAS_OK;
        size_t n = 0, max_half = 0;
        DenseRowDev* hd = h_desc + rows.size();
        for (auto& Rw : rows) {
            const long c = local_round(Rw, R);
            if (c < 0) continue;
            const size_t half = ((size_t)1 << Rw.n) >> (c + 1);
            hd[n++] = DenseRowDev{d_P + Rw.off, nullptr, nullptr, (uint32_t)half, 0, 0, 0};
            max_half = half > max_half ? half : max_half;
        }
        if (n == 0) { bound.store(R, std::memory_order_release); return ATLAS_OK; }
        HIP_TRY(hipMemcpyAsync(d_desc + rows.size(), hd, n * sizeof(DenseRowDev), hipMemcpyHostToDevice, rt().stream));
        k_dpool_bind<<<dim3(grid_for(max_half, GX), (unsigned)n), OP_THREADS, 0, rt().stream>>>(d_desc + rows.size(), to_dev(rf), rt().challenge_mode == 0 ? 1 : 0);
        hipError_t e = hipGetLastError();
        bound.store(R, std::memory_order_release);
        return e == hipSuccess ? ATLAS_OK : fail(ATLAS_ENODEV, "dense pool bind", e);
    }
    int fetch_finals() {
        if (have_finals) return ATLAS_OK;
        const size_t n = rows.size();
        k_pool_heads<<<grid_for(n, 64), OP_THREADS, 0, rt().stream>>>(d_P, d_off, (uint32_t)n, d_q0);
        std::vector<H::Fr> f(n);
        HIP_TRY(hipMemcpyAsync(f.data(), d_q0, n * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
        for (size_t r = 0; r < n; r++) rows[r].fin = f[r];
        have_finals = true;
        return ATLAS_OK;
    }
};

struct DensePoolRow : atlas_instance {
    DensePool* P = nullptr;
    size_t row = 0, round_next = 0;
    ~DensePoolRow() override { if (P && --P->refs == 0) delete P; }
    size_t rounds() const override { return P->rows[row].n; }
    size_t degree() const override { return 2; }
    size_t global_round(size_t round) const { return round + (P->max_rounds - P->rows[row].n); }
    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "dense_opening: round out of order");
        const size_t R = global_round(round);
        if (P->folded.load(std::memory_order_acquire) != R) {        // (a driver without shared steps: the first member to ask folds for all and waits)
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            const int rc = P->fold_all(R, true);
            if (rc) return rc;
        }
        const DensePool::Row& Rw = P->rows[row];
        H::Fr q0;
        std::memcpy(&q0, &P->h_q0[Rw.slot], sizeof(q0));
        coeffs.resize(3);
        H::gruen_deg2(Rw.st.scalar, Rw.st.w_cur(), q0, claim, coeffs.data());
        return ATLAS_OK;
    }
    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "dense_opening: round out of order");
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode);
        const size_t R = global_round(round);
        if (P->bound.load(std::memory_order_acquire) != R) {
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            const int rc = P->bind_all(R, rf);
            if (rc) return rc;
        }
        P->rows[row].st.bind(rf);
        round_next++;
        return ATLAS_OK;
    }
    bool host_parallel() const override { return true; }
    const void* shared_key() const override { return P; }
    int shared_message_step(size_t round) override {
        if (round != round_next || round >= rounds()) return ATLAS_OK;
        const size_t R = global_round(round);
        if (P->folded.load(std::memory_order_acquire) == R) return ATLAS_OK;
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        return P->fold_all(R, false);
    }
    int shared_ingest_step(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= rounds()) return ATLAS_OK;
        const size_t R = global_round(round);
        if (P->bound.load(std::memory_order_acquire) == R) return ATLAS_OK;
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        return P->bind_all(R, H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode));
    }
    int finals(std::vector<H::Fr>& out) override {
        if (round_next != rounds()) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        const int rc = P->fetch_finals();
        if (rc) return rc;
        out.assign(1, P->rows[row].fin);
        return ATLAS_OK;
    }
};

}  // namespace

// Dense opening instances stepped together (DensePool): polys[i] (Fr or i32, 2^ns[i] coefficients, ns[i] >= 1, not consumed: the pool works on an
// Fr copy) opened at points[i]; batch_max_rounds: the round count of the BatchedSumcheck they will sit in.  Internal (reduced_openings.hip).
int atlas_rt_dense_pool_new(const atlas_poly_t* polys, const atlas_fr_t* const* points, const size_t* ns, size_t count, size_t batch_max_rounds, atlas_instance_t* out) {
    if (!polys || !points || !ns || !out || count == 0) return fail(ATLAS_EINVAL, "dense_pool_new");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    std::unique_ptr<DensePool> P(new DensePool());
    P->max_rounds = batch_max_rounds;
    P->rows.resize(count);
    uint64_t total = 0;
    size_t tab_total = 0, w_total = 0, max_len = 0;
    for (size_t i = 0; i < count; i++) {
        if (!polys[i] || !points[i] || ns[i] == 0 || ns[i] > 26 || ns[i] > batch_max_rounds || polys[i]->len != ((size_t)1 << ns[i])) return fail(ATLAS_EINVAL, "dense_pool_new: member");
        DensePool::Row& R = P->rows[i];
        R.n = ns[i]; R.off = total; total += polys[i]->len;
        R.st.init(reinterpret_cast<const H::Fr*>(points[i]), ns[i]);
        if (R.st.k_in > 13 || R.st.k_out > 13) return fail(ATLAS_EINVAL, "dense_pool_new: more than 26 variables");
        tab_total += ((size_t)2 << R.st.k_in) + ((size_t)2 << R.st.k_out); w_total += ns[i];
        max_len = polys[i]->len > max_len ? polys[i]->len : max_len;
    }
    Fr* d_w = nullptr; DenseImportJob* d_imp = nullptr; PoolEqJob* d_jobs = nullptr;
    struct Tmp { std::vector<void*> v; ~Tmp() { for (void* p : v) if (p) hipFree(p); } } tmp;
    auto tmalloc = [&](void** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes); if (e == hipSuccess) tmp.v.push_back(*p); return e; };
    HIP_TRY(hipMalloc(&P->d_P, total * sizeof(Fr)));
    HIP_TRY(hipMalloc(&P->d_tabs, tab_total * sizeof(Fr)));
    HIP_TRY(hipMalloc(&P->d_part, count * DensePool::GX * sizeof(Fr)));
    HIP_TRY(hipMalloc(&P->d_q0, count * sizeof(Fr)));
    HIP_TRY(hipMalloc(&P->d_off, count * sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&P->d_desc, 2 * count * sizeof(DenseRowDev)));
    HIP_TRY(hipHostMalloc(&P->h_desc, 2 * count * sizeof(DenseRowDev), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&P->h_q0, count * sizeof(Fr), hipHostMallocDefault));
    HIP_TRY(tmalloc((void**)&d_w, w_total * sizeof(Fr)));
    HIP_TRY(tmalloc((void**)&d_imp, count * sizeof(DenseImportJob)));
    HIP_TRY(tmalloc((void**)&d_jobs, 2 * count * sizeof(PoolEqJob)));
    std::vector<H::Fr> w_all; w_all.reserve(w_total);
    std::vector<DenseImportJob> imp(count); std::vector<PoolEqJob> jobs(2 * count); std::vector<uint64_t> off(count);
    size_t wo = 0, to = 0;
    for (size_t i = 0; i < count; i++) {
        DensePool::Row& R = P->rows[i];
        w_all.insert(w_all.end(), R.st.w.begin(), R.st.w.end());
        R.d_ein = P->d_tabs + to; to += (size_t)2 << R.st.k_in;
        R.d_eout = P->d_tabs + to; to += (size_t)2 << R.st.k_out;
        jobs[2 * i] = PoolEqJob{R.d_ein, d_w + wo + 1, (uint32_t)R.st.k_in, 0};            // the suffix tables of GseDevH::init
        jobs[2 * i + 1] = PoolEqJob{R.d_eout, d_w + wo + 1 + R.st.k_in, (uint32_t)R.st.k_out, 0};
        imp[i] = DenseImportJob{polys[i]->d, P->d_P + R.off, (uint32_t)polys[i]->len, polys[i]->is_i32 ? 1u : 0u};
        off[i] = R.off;
        wo += R.n;
    }
    HIP_TRY(hipMemcpyAsync(d_w, w_all.data(), w_all.size() * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(d_imp, imp.data(), count * sizeof(DenseImportJob), hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(PoolEqJob), hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(P->d_off, off.data(), count * sizeof(uint64_t), hipMemcpyHostToDevice, rt().stream));
    k_dpool_import<<<dim3(grid_for(max_len, DensePool::GX), (unsigned)count), OP_THREADS, 0, rt().stream>>>(d_imp, make_consts());
    k_pool_eq_cached_rev<<<(unsigned)(2 * count), 1024, 0, rt().stream>>>(d_jobs);
    HIP_TRY(hipStreamSynchronize(rt().stream));                    // (the staging vectors are pageable and leave scope)
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ATLAS_ENODEV, "dense_pool_new", e);
    P->refs = count;
    DensePool* raw = P.release();
    for (size_t i = 0; i < count; i++) { DensePoolRow* I = new DensePoolRow(); I->P = raw; I->row = i; out[i] = I; }
    return ATLAS_OK;
}

// One-hot opening instances over device-resident lookup indices, stepped together (OneHotPool).  rows[i]: the T = 2^log_T lookups, the
// chunk's shift, the opening point (log_K address coordinates, then log_T cycle ones).  batch_max_rounds: the round count of the whole
// BatchedSumcheck the rows will sit in (>= log_K + the largest log_T).  d_idx_rows[i] (optional) receives the device int32 index row of
// polynomial i, alive as long as any of the instances (for build_materialized_rlc).  Internal to the library (reduced_openings.hip).
struct atlas_rt_pool_row { const uint64_t* d_lookups; size_t shift, log_T; const atlas_fr_t* point; };
// The int32 index rows of n one-hot polynomials cut from their device lookups, WITHOUT instances: what build_materialized_rlc needs of the
// rows another rank of a sharded reduction steps (reduced_openings.hip).  One allocation (*d_buf, the caller's to hipFree), rows[i] inside it.
int atlas_rt_chunk_index_rows(const atlas_rt_pool_row* in, size_t n, size_t log_K, int32_t** d_buf, const int32_t** rows) {
    if (!in || !n || !d_buf || !rows || log_K == 0 || log_K > 4) return fail(ATLAS_EINVAL, "chunk_index_rows");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    std::vector<const uint64_t*> lkp(n); std::vector<uint32_t> sh(n), Ts(n); std::vector<uint64_t> off(n);
    uint64_t total = 0; size_t maxT = 0;
    for (size_t i = 0; i < n; i++) {
        if (!in[i].d_lookups || in[i].log_T > 26) return fail(ATLAS_EINVAL, "chunk_index_rows: row");
        const size_t T = (size_t)1 << in[i].log_T;
        lkp[i] = in[i].d_lookups; sh[i] = (uint32_t)in[i].shift; Ts[i] = (uint32_t)T; off[i] = total; total += T; maxT = T > maxT ? T : maxT;
    }
    DevBuf bufh;                                    // (released to the caller on success; an early return gives it back)
    HIP_TRY(bufh.alloc(total * sizeof(int32_t)));
    int32_t* buf = bufh.as<int32_t>();
    DevBuf d_lk, d_sh, d_Ts, d_off;
    HIP_TRY(d_lk.alloc(n * sizeof(void*))); HIP_TRY(d_sh.alloc(n * 4)); HIP_TRY(d_Ts.alloc(n * 4)); HIP_TRY(d_off.alloc(n * 8));
    HIP_TRY(hipMemcpyAsync(d_lk.p, lkp.data(), n * sizeof(void*), hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(d_sh.p, sh.data(), n * 4, hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(d_Ts.p, Ts.data(), n * 4, hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(d_off.p, off.data(), n * 8, hipMemcpyHostToDevice, rt().stream));
    k_pool_chunk_rows<<<dim3(grid_for(maxT, POOL_GX), (unsigned)n), OP_THREADS, 0, rt().stream>>>(d_lk.as<const uint64_t*>(), d_sh.as<uint32_t>(), d_off.as<uint64_t>(), d_Ts.as<uint32_t>(),
                                                                                       (uint32_t)(((size_t)1 << log_K) - 1), buf);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(rt().stream);            // (the host vectors above are the sources of the copies)
    if (e != hipSuccess) return fail(ATLAS_ENODEV, "chunk_index_rows", e);
    for (size_t i = 0; i < n; i++) rows[i] = buf + off[i];
    *d_buf = static_cast<int32_t*>(bufh.release());
    return ATLAS_OK;
}

int atlas_rt_onehot_pool_new(const atlas_rt_pool_row* in, size_t n, size_t log_K, size_t batch_max_rounds, atlas_instance_t* out, const int32_t** d_idx_rows) {
    if (!in || !out || n == 0 || log_K == 0 || log_K > 4) return fail(ATLAS_EINVAL, "onehot_pool_new: 1 <= log_K <= 4");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    std::unique_ptr<OneHotPool> P(new OneHotPool());
    P->log_K = log_K; P->K = (size_t)1 << log_K; P->max_rounds = batch_max_rounds;
    P->rows.resize(n);
    uint64_t total = 0;
    size_t tab_total = 0, w_total = 0;
    for (size_t i = 0; i < n; i++) {
        if (!in[i].d_lookups || !in[i].point || in[i].log_T == 0 || in[i].log_T > 26 || log_K + in[i].log_T > batch_max_rounds) return fail(ATLAS_EINVAL, "onehot_pool_new: row");
        const atlas_fr_t* rc_ = in[i].point + log_K;
        size_t gi = P->groups.size();
        for (size_t q = P->groups.size(); q-- > 0;) {             // same r_cycle -> same group (the rows of one lookup family arrive together)
            OneHotPool::Group& G = P->groups[q];
            if (G.log_T == in[i].log_T && std::memcmp(G.st.w.data(), rc_, G.log_T * 32) == 0) { gi = q; break; }
            if (P->groups.size() - q > 8) break;
        }
        if (gi == P->groups.size()) {
            P->groups.emplace_back();
            OneHotPool::Group& G = P->groups.back();
            G.log_T = in[i].log_T; G.T = (size_t)1 << G.log_T;
            G.st.init(reinterpret_cast<const H::Fr*>(rc_), G.log_T);
            if (G.st.k_in > 13 || G.st.k_out > 13) return fail(ATLAS_EINVAL, "onehot_pool_new: more than 26 cycle variables");
            tab_total += ((size_t)2 << G.st.k_in) + ((size_t)2 << G.st.k_out); w_total += G.log_T;
        }
        P->groups[gi].rows.push_back(i);
        OneHotPool::Row& R = P->rows[i];
        R.group = gi; R.off = total; total += P->groups[gi].T;
    }
    // the rows' host tables on the host threads (thousands of rows: EqAddressState::new here and the histogram sums below were ~10 ms on one)
    atlas_host::HostThreads::get().parallel_for(n, [&](size_t lo, size_t hi, size_t) {
        for (size_t i = lo; i < hi; i++) {
            P->rows[i].B = H::eq_evals(reinterpret_cast<const H::Fr*>(in[i].point), log_K);     // EqAddressState::new
            P->rows[i].F = {H::one()};
        }
    });
    const size_t NG = P->groups.size();
    // device state
    Fr *d_w = nullptr, *d_E = nullptr;
    const uint64_t** d_lk = nullptr; uint32_t *d_shift = nullptr, *d_Ts = nullptr; const Fr** d_Eptr = nullptr; unsigned long long* d_hist = nullptr;
    struct Tmp { std::vector<void*> v; ~Tmp() { for (void* p : v) if (p) hipFree(p); } } tmp;
    auto tmalloc = [&](void** p, size_t bytes) { hipError_t e = hipMalloc(p, bytes); if (e == hipSuccess) tmp.v.push_back(*p); return e; };
    size_t E_total = 0;
    for (auto& G : P->groups) E_total += G.T;
    HIP_TRY(hipMalloc(&P->d_idx, total * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&P->d_H, (total / 2 + 1) * sizeof(Fr)));            // (the first H of a row is the bound one: T / 2 entries at off / 2; every T is a power of two >= 2)
    HIP_TRY(hipMalloc(&P->d_hoff, n * sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&P->d_tabs, tab_total * sizeof(Fr)));
    HIP_TRY(hipMalloc(&P->d_part, n * POOL_GX * sizeof(Fr)));
    HIP_TRY(hipMalloc(&P->d_q0, n * sizeof(Fr)));
    HIP_TRY(hipMalloc(&P->d_F, n * P->K * sizeof(Fr)));
    HIP_TRY(hipMalloc(&P->d_off, n * sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&P->d_desc, 3 * n * sizeof(PoolRowDev)));
    HIP_TRY(hipHostMalloc(&P->h_desc, 3 * n * sizeof(PoolRowDev), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&P->h_q0, n * sizeof(Fr), hipHostMallocDefault));
    HIP_TRY(tmalloc((void**)&d_w, (w_total ? w_total : 1) * sizeof(Fr)));
    HIP_TRY(tmalloc((void**)&d_E, E_total * sizeof(Fr)));
    HIP_TRY(tmalloc((void**)&d_lk, n * sizeof(void*)));
    HIP_TRY(tmalloc((void**)&d_shift, n * 4));
    HIP_TRY(tmalloc((void**)&d_Ts, n * 4));
    HIP_TRY(tmalloc((void**)&d_Eptr, n * sizeof(void*)));
    HIP_TRY(tmalloc((void**)&d_hist, n * 16 * 8 * 8));
    std::vector<H::Fr> w_all; w_all.reserve(w_total);
    std::vector<const uint64_t*> lkp(n); std::vector<uint32_t> sh(n), Ts(n); std::vector<uint64_t> off(n); std::vector<const Fr*> Eptr(n);
    std::vector<size_t> g_w(NG), g_E(NG);
    {
        size_t wo = 0, to = 0, eo = 0;
        for (size_t q = 0; q < NG; q++) {
            OneHotPool::Group& G = P->groups[q];
            g_w[q] = wo; g_E[q] = eo;
            w_all.insert(w_all.end(), G.st.w.begin(), G.st.w.end());
            G.d_ein = P->d_tabs + to; to += (size_t)2 << G.st.k_in;
            G.d_eout = P->d_tabs + to; to += (size_t)2 << G.st.k_out;
            wo += G.log_T; eo += G.T;
        }
    }
    size_t maxT = 0;
    for (size_t i = 0; i < n; i++) {
        const OneHotPool::Group& G = P->groups[P->rows[i].group];
        lkp[i] = in[i].d_lookups; sh[i] = (uint32_t)in[i].shift; Ts[i] = (uint32_t)G.T; off[i] = P->rows[i].off; Eptr[i] = d_E + g_E[P->rows[i].group];
        maxT = G.T > maxT ? G.T : maxT;
    }
    HIP_TRY(hipMemcpyAsync(d_w, w_all.data(), w_all.size() * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(d_lk, lkp.data(), n * sizeof(void*), hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(d_shift, sh.data(), n * 4, hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(d_Ts, Ts.data(), n * 4, hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(P->d_off, off.data(), n * 8, hipMemcpyHostToDevice, rt().stream));
    std::vector<uint64_t> hoff(n);
    for (size_t i = 0; i < n; i++) hoff[i] = off[i] / 2;
    HIP_TRY(hipMemcpyAsync(P->d_hoff, hoff.data(), n * 8, hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(d_Eptr, Eptr.data(), n * sizeof(void*), hipMemcpyHostToDevice, rt().stream));
    k_pool_chunk_rows<<<dim3(grid_for(maxT, POOL_GX), (unsigned)n), OP_THREADS, 0, rt().stream>>>(d_lk, d_shift, P->d_off, d_Ts, (uint32_t)(P->K - 1), P->d_idx);
    // the split-eq suffix tables (GseDevH::init) and D.merge() before any bind = EqPolynomial::evals(r_cycle) for the histogram: all groups at once
    std::vector<PoolEqJob> jobs(3 * NG);
    PoolEqJob* d_jobs = nullptr;
    HIP_TRY(tmalloc((void**)&d_jobs, jobs.size() * sizeof(PoolEqJob)));
    size_t max_hi_blocks = 1;
    for (size_t q = 0; q < NG; q++) {
        OneHotPool::Group& G = P->groups[q];
        jobs[2 * q] = PoolEqJob{G.d_ein, d_w + g_w[q] + 1, (uint32_t)G.st.k_in, 0};
        jobs[2 * q + 1] = PoolEqJob{G.d_eout, d_w + g_w[q] + 1 + G.st.k_in, (uint32_t)G.st.k_out, 0};
        jobs[2 * NG + q] = PoolEqJob{d_E + g_E[q], d_w + g_w[q], (uint32_t)G.log_T, 0};
        const size_t nh = G.log_T < 8 ? 0 : G.log_T - 8, hb = (((size_t)1 << nh) + POOL_EQ_HI - 1) / POOL_EQ_HI;
        max_hi_blocks = hb > max_hi_blocks ? hb : max_hi_blocks;
    }
    HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(PoolEqJob), hipMemcpyHostToDevice, rt().stream));      // (jobs lives until the synchronisation below)
    k_pool_eq_cached_rev<<<(unsigned)(2 * NG), 1024, 0, rt().stream>>>(d_jobs);
    k_pool_eq_full<<<dim3((unsigned)max_hi_blocks, (unsigned)NG), 256, 0, rt().stream>>>(d_jobs + 2 * NG);
    k_pool_hist<<<(unsigned)(((n + 7) / 8) * 8), OP_THREADS, 0, rt().stream>>>(P->d_idx, P->d_off, d_Ts, d_Eptr, (uint32_t)P->K, (uint32_t)n, d_hist);
    std::vector<unsigned long long> hist(n * 16 * 8);
    HIP_TRY(hipMemcpyAsync(hist.data(), d_hist, hist.size() * 8, hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    atlas_host::HostThreads::get().parallel_for(n, [&](size_t lo, size_t hi, size_t) {
        for (size_t i = lo; i < hi; i++) {
            OneHotPool::Row& R = P->rows[i];
            R.G.resize(P->K);
            for (size_t k = 0; k < P->K; k++) {
                uint64_t acc[9];
                for (int w = 0; w < 8; w++) acc[w] = hist[(i * 16 + k) * 8 + w];
                acc[8] = 0;
                R.G[k] = atlas_rt::sum_to_fr(acc, 32, 0);
            }
        }
    });
    for (size_t i = 0; i < n && d_idx_rows; i++) d_idx_rows[i] = P->d_idx + P->rows[i].off;
    P->refs = n;
    OneHotPool* raw = P.release();
    for (size_t i = 0; i < n; i++) { OneHotPoolRow* I = new OneHotPoolRow(); I->P = raw; I->row = i; out[i] = I; }
    return ATLAS_OK;
}

extern "C" {

int atlas_dense_opening_new(atlas_poly_t poly, const atlas_fr_t* opening_point, size_t n, atlas_instance_t* out) {
    PROF("atlas_dense_opening_new");
    NEED_INIT();
    if (!poly || (!opening_point && n) || !out) return fail(ATLAS_EINVAL, "dense_opening_new: null argument");
    if (poly->len != ((size_t)1 << n)) return fail(ATLAS_EINVAL, "dense_opening_new: polynomial length != 2^n");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    DenseOpening* P = new DenseOpening();
    P->n = n;
    // n == 0: a one-coefficient polynomial (the committed remainder / quotient of a one-element ScalarConstDiv / Div) — a member of the
    // batch without rounds of its own (opening_proof.rs:447-532: its claim rides scaled by 2^max_rounds, its final claim is the coefficient)
    int rc = n ? P->D.init(reinterpret_cast<const H::Fr*>(opening_point), n) : ATLAS_OK;
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
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
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
    if (e == hipSuccess) e = hipMemcpyAsync(P->d_idx, nonzero_indices, T * sizeof(int32_t), hipMemcpyHostToDevice, rt().stream);
    if (e == hipSuccess) {
        k_onehot_G<<<dim3((unsigned)K, (unsigned)slices), OP_THREADS, 0, rt().stream>>>(P->d_idx, (const Fr*)E->d, T, d_G);
        e = hipMemcpyAsync(hG.data(), d_G, slices * K * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(rt().stream);
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
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
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
        e = hipMemcpyAsync(Gp->d_idx + r * T, nonzero_indices[r], T * sizeof(int32_t), hipMemcpyHostToDevice, rt().stream);
    std::vector<H::Fr> hGs(slices * R * K), hG(R * K, H::zero());
    if (e == hipSuccess) {
        k_onehot_G_rows<<<dim3((unsigned)K, (unsigned)R, (unsigned)slices), OP_THREADS, 0, rt().stream>>>(Gp->d_idx, (const Fr*)E->d, T, (uint32_t)K, d_G);
        e = hipMemcpyAsync(hGs.data(), d_G, slices * R * K * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(rt().stream);
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
