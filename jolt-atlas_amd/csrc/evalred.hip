// N-to-1 evaluation reduction: EvalReductionInstance::prove, compute_h, eval_on_l
// (joltworks/src/subprotocols/evaluation_reduction.rs:91-147, 213-249).
//
// The reference builds h(t) = P(l(t)) by folding 2^n univariate polynomials on one thread
// (O(2^n * n * N) coefficient operations).  Here h is recovered from its values: deg h <= n (N-1),
// so D + 1 = n (N-1) + 1 multilinear evaluations P(l(0)), ..., P(l(D)) determine it — each one is
// a split-eq evaluation over the device-resident coefficients (eight points per pass, k_er_eval), and the
// O(D^2) Newton interpolation back to monomial coefficients is host arithmetic.  The coefficient
// vector is the exact polynomial with trailing zeros trimmed, which is what the reference's
// Add / Mul / from_coeff chain yields (unipoly.rs:415-476).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_poly.hpp"
#include "runtime.hpp"
#include "sc_consts.hpp"
#include "spliteq_kernels.hip.h"

using namespace atlas;
namespace H = atlas_host;
using atlas_rt::fail;
using atlas_rt::rt;

namespace {

constexpr int ER_THREADS = 256;
constexpr int ER_GROUP = 8;        // evaluation points per pass over the coefficients

// eq tables of every evaluation point: blockIdx.x = point, blockIdx.y = half (0: r[0..m) outer, 1: r[m..n) inner);
// big-endian like EqPolynomial::evals (pass p of the doubling uses the p-th variable from the end)
__global__ __launch_bounds__(1024) void k_er_tables(const Fr* __restrict__ points, uint32_t n, uint32_t m, Fr* __restrict__ tab1,
                                                    Fr* __restrict__ tab2) {
    const uint32_t p = blockIdx.x, half = blockIdx.y;
    const uint32_t k = half ? n - m : m;
    const Fr* r = points + (size_t)p * n + (half ? m : 0);
    Fr* ev = half ? tab2 + ((size_t)p << (n - m)) : tab1 + ((size_t)p << m);
    if (threadIdx.x == 0) fe_store(ev, fr_one());
    __syncthreads();
    for (uint32_t q = 0; q < k; q++) {
        const uint32_t size = 1u << q;
        const Fr rq = fe_load(r + (k - 1 - q));
        for (uint32_t i = threadIdx.x; i < size; i += 1024) {
            const Fr x = fe_load(ev + i), y = fr_mul(x, rq);
            fe_store(ev + i + size, y);
            fe_store(ev + i, fr_sub(x, y));
        }
        __threadfence_block();
        __syncthreads();
    }
}

// out[t0 + q] += sum_{x_hi} E1_q[x_hi] * sum_{x_lo} E2_q[x_lo] * Z[x_hi, x_lo] for q < G: one pass over Z serves G
// evaluation points (DensePolynomial::evaluate's split, dense_mlpoly.rs:265-305).  A workgroup takes whole rows.
template <class T, int G>
__global__ __launch_bounds__(ER_THREADS) void k_er_eval(const T* __restrict__ Z, uint32_t n, uint32_t m, const Fr* __restrict__ tab1,
                                                        const Fr* __restrict__ tab2, uint32_t t0, uint32_t n_pts,
                                                        Fr* __restrict__ partials /* [gridDim.x][G] */, ScConsts K) {
    __shared__ Fr red[ER_THREADS / 64][G];
    const uint32_t n2 = n - m;
    const size_t rows = (size_t)1 << m, cols = (size_t)1 << n2;
    Fr total = fe_zero();                                     // thread q < G keeps point q's running sum
    for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
        Fr acc[G];
#pragma unroll
        for (int q = 0; q < G; q++) acc[q] = fe_zero();
        for (size_t c = threadIdx.x; c < cols; c += ER_THREADS) {
            const Fr z = Src<T>::get(Z, (row << n2) | c, K);
#pragma unroll
            for (int q = 0; q < G; q++)
                if (t0 + q < n_pts) acc[q] = fr_add(acc[q], fr_mul(z, fe_load(tab2 + ((size_t)(t0 + q) << n2) + c)));
        }
#pragma unroll
        for (int q = 0; q < G; q++) {
            const Fr sres = fr_wave_sum(acc[q]);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = sres;
        }
        __syncthreads();
        if (threadIdx.x < G && t0 + threadIdx.x < n_pts) {
            Fr sres = red[0][threadIdx.x];
            for (int w = 1; w < ER_THREADS / 64; w++) sres = fr_add(sres, red[w][threadIdx.x]);
            total = fr_add(total, fr_mul(sres, fe_load(tab1 + ((size_t)(t0 + threadIdx.x) << m) + row)));
        }
        __syncthreads();
    }
    if (threadIdx.x < G) fe_store(partials + (size_t)blockIdx.x * G + threadIdx.x, total);
}

__global__ __launch_bounds__(ER_THREADS) void k_er_reduce(const Fr* __restrict__ partials, uint32_t n_blocks, uint32_t G, Fr* out) {
    __shared__ Fr red[ER_THREADS / 64];
    const uint32_t q = blockIdx.x;
    Fr acc = fe_zero();
    for (uint32_t b = threadIdx.x; b < n_blocks; b += ER_THREADS) acc = fr_add(acc, fe_load(partials + (size_t)b * G + q));
    acc = fr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Fr sres = red[0];
        for (int w = 1; w < ER_THREADS / 64; w++) sres = fr_add(sres, red[w]);
        fe_store(out + q, sres);
    }
}

// P(point_t) for all t at once; points = n_pts rows of n Fr (host)
int evaluate_many(atlas_poly_t mle, const std::vector<H::Fr>& points, size_t n_pts, size_t n, std::vector<H::Fr>& out) {
    const uint32_t m = (uint32_t)(n / 2), n2 = (uint32_t)(n - m);
    if (m > 13 || n2 > 13) return fail(ATLAS_EINVAL, "eval_reduction: more than 26 variables not supported");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    Fr *d_pts = nullptr, *tab1 = nullptr, *tab2 = nullptr, *part = nullptr, *d_out = nullptr;
    const size_t rows = (size_t)1 << m;
    const unsigned grid = (unsigned)(rows < 1024 ? rows : 1024);
    auto cleanup = [&]() { for (Fr* p : {d_pts, tab1, tab2, part, d_out}) if (p) hipFree(p); };
    hipError_t e = hipMalloc(&d_pts, (n_pts * n + 1) * sizeof(Fr));
    if (e == hipSuccess) e = hipMalloc(&tab1, (n_pts << m) * sizeof(Fr));
    if (e == hipSuccess) e = hipMalloc(&tab2, (n_pts << n2) * sizeof(Fr));
    if (e == hipSuccess) e = hipMalloc(&part, (size_t)grid * ER_GROUP * sizeof(Fr));
    if (e == hipSuccess) e = hipMalloc(&d_out, (n_pts + ER_GROUP) * sizeof(Fr));
    if (e == hipSuccess && n) e = hipMemcpyAsync(d_pts, points.data(), n_pts * n * sizeof(Fr), hipMemcpyHostToDevice, rt().stream);
    if (e != hipSuccess) { cleanup(); return fail(ATLAS_ENOMEM, "eval_reduction workspace", e); }
    k_er_tables<<<dim3((unsigned)n_pts, 2), 1024, 0, rt().stream>>>(d_pts, (uint32_t)n, m, tab1, tab2);
    const ScConsts K = make_consts();
    for (size_t t0 = 0; t0 < n_pts; t0 += ER_GROUP) {
        if (mle->is_i32) k_er_eval<int32_t, ER_GROUP><<<grid, ER_THREADS, 0, rt().stream>>>((const int32_t*)mle->d, (uint32_t)n, m, tab1, tab2, (uint32_t)t0, (uint32_t)n_pts, part, K);
        else k_er_eval<Fr, ER_GROUP><<<grid, ER_THREADS, 0, rt().stream>>>((const Fr*)mle->d, (uint32_t)n, m, tab1, tab2, (uint32_t)t0, (uint32_t)n_pts, part, K);
        k_er_reduce<<<ER_GROUP, ER_THREADS, 0, rt().stream>>>(part, grid, ER_GROUP, d_out + t0);
    }
    out.resize(n_pts + ER_GROUP);
    e = hipMemcpyAsync(out.data(), d_out, (n_pts + ER_GROUP) * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream);
    if (e == hipSuccess) e = hipStreamSynchronize(rt().stream);
    if (e == hipSuccess) e = hipGetLastError();
    cleanup();
    if (e != hipSuccess) return fail(ATLAS_ENODEV, "eval_reduction", e);
    out.resize(n_pts);
    return ATLAS_OK;
}

// coefficients of the polynomial through (0, e0), ..., (D, eD): Newton forward differences, then
// expansion of the Newton basis prod_{j<k} (x - j)
std::vector<H::Fr> interpolate_consecutive(const std::vector<H::Fr>& ev) {
    const size_t n = ev.size();
    std::vector<H::Fr> dd = ev;                       // dd[k] -> Delta^k f(0) / k!
    for (size_t k = 1; k < n; k++) {
        for (size_t i = n - 1; i >= k; i--) dd[i] = H::sub(dd[i], dd[i - 1]);
    }
    H::Fr fact = H::one();
    for (size_t k = 1; k < n; k++) { fact = H::mul(fact, H::from_u64(k)); dd[k] = H::mul(dd[k], H::inv(fact)); }
    std::vector<H::Fr> c = {dd[n - 1]};
    for (size_t k = n - 1; k-- > 0;) {                // c <- c * (x - k) + dd[k]
        const H::Fr mk = H::from_u64(k);
        std::vector<H::Fr> nx(c.size() + 1);
        for (size_t i = 0; i <= c.size(); i++) {
            const H::Fr up = i > 0 ? c[i - 1] : H::zero();
            const H::Fr dn = i < c.size() ? H::mul(mk, c[i]) : H::zero();
            nx[i] = H::sub(up, dn);
        }
        nx[0] = H::add(nx[0], dd[k]);
        c.swap(nx);
    }
    return c;
}

H::Fr horner(const std::vector<H::Fr>& c, const H::Fr& x) {
    H::Fr acc = H::zero();
    for (size_t i = c.size(); i-- > 0;) acc = H::add(H::mul(acc, x), c[i]);
    return acc;
}

}  // namespace

extern "C" int atlas_eval_reduction_prove(atlas_poly_t mle, const atlas_fr_t* points, const atlas_fr_t* claims, size_t N,
                                          size_t n, atlas_transcript_t* transcript, atlas_fr_t* h_out, size_t h_cap,
                                          size_t* h_len, atlas_fr_t* r_out, atlas_fr_t* claim_out) {
    PROF("atlas_eval_reduction_prove");
    NEED_INIT();
    if (!mle || (!points && n) || !claims || !transcript || !h_out || !h_len || (!r_out && n) || !claim_out) return fail(ATLAS_EINVAL, "eval_reduction: null argument");
    if (N == 0) return fail(ATLAS_EINVAL, "eval_reduction: EmptyInput");
    size_t len = 0;
    atlas_poly_len(mle, &len);
    if (len != ((size_t)1 << n)) return fail(ATLAS_EINVAL, "eval_reduction: InvalidInputLength (point length != num_vars)");
    const H::Fr* pts = reinterpret_cast<const H::Fr*>(points);
    if (N == 1) {                                                   // short path, evaluation_reduction.rs:113-127
        if (h_cap < 1) return fail(ATLAS_EINVAL, "eval_reduction: h buffer too small");
        std::memcpy(h_out, claims, 32); *h_len = 1;
        if (n) std::memcpy(r_out, points, n * 32);
        std::memcpy(claim_out, claims, 32);
        return ATLAS_OK;
    }
    const size_t D = n * (N - 1);
    if (h_cap < D + 1) return fail(ATLAS_EINVAL, "eval_reduction: h buffer too small");
    // l: per-variable interpolant through (j, points[j][i])  (group_by_variable + UniPoly::from_evals)
    std::vector<std::vector<H::Fr>> var(n);
    for (size_t i = 0; i < n; i++) {
        std::vector<H::Fr> ev(N);
        for (size_t j = 0; j < N; j++) ev[j] = pts[j * n + i];
        var[i] = interpolate_consecutive(ev);
    }
    // h on 0..D by multilinear evaluation on the device: all D + 1 points share the passes over the coefficients
    std::vector<H::Fr> hev, lpts((D + 1) * n);
    for (size_t t = 0; t <= D; t++) {
        const H::Fr x = H::from_u64(t);
        for (size_t i = 0; i < n; i++) lpts[t * n + i] = horner(var[i], x);
    }
    if (n == 0) {                                                   // a scalar node: h is the constant polynomial of its one value
        atlas_fr_t c0;
        int rc = atlas_poly_download(mle, &c0, 1);
        if (rc) return rc;
        hev.assign(1, *reinterpret_cast<H::Fr*>(&c0));
    } else {
        int rc = evaluate_many(mle, lpts, D + 1, n, hev);
        if (rc) return rc;
    }
    std::vector<H::Fr> h = interpolate_consecutive(hev);
    H::trim(h);
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(transcript);
    H::tr_append_message(T, "UncompressedUniPoly_begin");           // unipoly.rs:540-548
    for (auto& c : h) H::tr_append_scalar(T, c);
    H::tr_append_message(T, "UncompressedUniPoly_end");
    uint64_t lo, hi;
    H::tr_challenge_u128(T, lo, hi);
    const H::Fr xp = H::challenge_to_fr(lo, hi, rt().challenge_mode);
    for (size_t i = 0; i < n; i++) { const H::Fr v = horner(var[i], xp); std::memcpy(&r_out[i], &v, 32); }   // eval_on_l
    const H::Fr v = horner(h, xp);
    std::memcpy(claim_out, &v, 32);
    std::memcpy(h_out, h.data(), h.size() * 32);
    *h_len = h.size();
    return ATLAS_OK;
}

// EvalReductionInstance::verify (evaluation_reduction.rs:150-210): h(i) = claim_i for every opening, the degree bound, then the
// reduced instance (l(x'), h(x')) at the transcript's challenge.  One opening: the short path (nothing enters the transcript).
extern "C" int atlas_eval_reduction_verify(const atlas_fr_t* points, const atlas_fr_t* claims, size_t N, size_t n, const atlas_fr_t* h_, size_t h_len,
                                           atlas_transcript_t* transcript, atlas_fr_t* r_out, atlas_fr_t* claim_out) {
    if ((!points && n) || !claims || !transcript || (!r_out && n) || !claim_out || (!h_ && h_len)) return fail(ATLAS_EINVAL, "eval_reduction_verify: null argument");
    if (N == 0) return fail(ATLAS_EVERIFY, "eval_reduction_verify: EmptyInput");
    const H::Fr* pts = reinterpret_cast<const H::Fr*>(points);
    const H::Fr* cl = reinterpret_cast<const H::Fr*>(claims);
    if (N == 1) { if (n) std::memcpy(r_out, points, n * 32); std::memcpy(claim_out, claims, 32); return ATLAS_OK; }
    std::vector<H::Fr> h(reinterpret_cast<const H::Fr*>(h_), reinterpret_cast<const H::Fr*>(h_) + h_len);
    if (h.empty() || h.size() - 1 > n * (N - 1)) return fail(ATLAS_EVERIFY, "eval_reduction_verify: InvalidOpeningProof (degree of h)");
    for (size_t i = 0; i < N; i++) {
        const H::Fr v = horner(h, H::from_u64(i));
        if (std::memcmp(&v, &cl[i], 32) != 0) return fail(ATLAS_EVERIFY, "eval_reduction_verify: InvalidOpeningProof (h does not match an opening claim)");
    }
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(transcript);
    H::tr_append_message(T, "UncompressedUniPoly_begin");
    for (auto& c : h) H::tr_append_scalar(T, c);
    H::tr_append_message(T, "UncompressedUniPoly_end");
    uint64_t lo, hi;
    H::tr_challenge_u128(T, lo, hi);
    const H::Fr xp = H::challenge_to_fr(lo, hi, rt().challenge_mode);
    for (size_t i = 0; i < n; i++) {                                  // eval_on_l: the per-variable interpolants through (j, points[j][i])
        std::vector<H::Fr> ev(N);
        for (size_t j = 0; j < N; j++) ev[j] = pts[j * n + i];
        const H::Fr v = horner(interpolate_consecutive(ev), xp);
        std::memcpy(&r_out[i], &v, 32);
    }
    const H::Fr v = horner(h, xp);
    std::memcpy(claim_out, &v, 32);
    return ATLAS_OK;
}
