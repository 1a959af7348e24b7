// Shout lookup argument, prover-side table builds.
//
// Device counterpart of (paths under the jolt-atlas tree):
//   ReadRafProver::initialize   joltworks/src/subprotocols/shout.rs:193-225
//        G[k] = sum_{j : idx_j = k} E[j],  E = eq(r_cycle, .)     (weighted histogram)
//   ReadRafProver::compute_message / ingest_challenge   shout.rs:233-262
//        sum_k G[k] * (val[k] + gamma * int[k]),  HighToLow, degree 2
//        = the dot-product instance over (G, W), W = val + gamma * int (both multilinear, so
//          binding W equals binding val and int separately): reuses the fused dot kernels.
//   compute_ra_evals            shout.rs:550-598 : the same histogram per 4-bit chunk of the
//        lookup index (OneHotParams::lookup_index_chunk, config.rs:73-75), d chunks at once.
//
// Fr has no atomic add, so the histogram is a counting sort by bucket (u32 atomics: count ->
// exclusive scan -> scatter of j) followed by one workgroup (or one thread, for large tables)
// per bucket summing E over its list with the lazy 29-bit-limb adds.  Any summation order
// gives the same residue.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/atlas_hip.h"
#include "channel.hpp"
#include "f9.hip.h"
#include "host_field.hpp"
#include "internal.hpp"
#include "scan.hip.h"
#include "runtime.hpp"

using namespace atlas;
namespace H = atlas_host;
using atlas_rt::fail;
using atlas_rt::rt;

namespace {

constexpr int SH_THREADS = 256;

struct KeySpec {          // key(j, i) = i * k_chunk + ((idx_j >> shift_i) & (k_chunk - 1)), i < d
    uint32_t d;
    uint32_t log_k_chunk;
};

__device__ __forceinline__ uint32_t key_of(uint64_t idx, uint32_t i, KeySpec S) {
    const uint32_t shift = S.log_k_chunk * (S.d - 1 - i);
    const uint32_t chunk = S.log_k_chunk >= 64 ? 0u : (uint32_t)((idx >> shift) & ((1ull << S.log_k_chunk) - 1));
    return (i << S.log_k_chunk) + chunk;
}

// a lookup index of log_K bits or fewer?  ORs 1 into *flag otherwise (key_of masks the index, so an out-of-range one would wrap into the table)
__global__ __launch_bounds__(SH_THREADS) void k_sh_range_flag(const uint64_t* __restrict__ idx, size_t T, uint32_t log_K, uint32_t* flag) {
    uint32_t bad = 0;
    for (size_t j = (size_t)blockIdx.x * SH_THREADS + threadIdx.x; j < T; j += (size_t)gridDim.x * SH_THREADS) bad |= (log_K < 64 && (idx[j] >> log_K)) ? 1u : 0u;
    if (bad) atomicOr(flag, 1u);                                          // (rare: every offending lane reports, whatever the wavefront width)
}

__global__ __launch_bounds__(SH_THREADS) void k_sh_hist(const uint64_t* __restrict__ idx, size_t T, KeySpec S, uint32_t* counts) {
    for (size_t j = (size_t)blockIdx.x * SH_THREADS + threadIdx.x; j < T; j += (size_t)gridDim.x * SH_THREADS) {
        const uint64_t v = idx[j];
        for (uint32_t i = 0; i < S.d; i++) atomicAdd(&counts[key_of(v, i, S)], 1u);
    }
}

__global__ __launch_bounds__(SH_THREADS) void k_sh_scatter(const uint64_t* __restrict__ idx, size_t T, KeySpec S,
                                                           uint32_t* cursor, uint32_t* __restrict__ sorted) {
    for (size_t j = (size_t)blockIdx.x * SH_THREADS + threadIdx.x; j < T; j += (size_t)gridDim.x * SH_THREADS) {
        const uint64_t v = idx[j];
        for (uint32_t i = 0; i < S.d; i++) sorted[atomicAdd(&cursor[key_of(v, i, S)], 1u)] = (uint32_t)j;
    }
}

// one workgroup per bucket
__global__ __launch_bounds__(SH_THREADS) void k_sh_bucket_sum_wg(const Fe* __restrict__ E, const uint32_t* __restrict__ sorted,
                                                                 const uint32_t* __restrict__ offsets, Fe* __restrict__ G) {
    using P9 = Fr9Params;
    const uint32_t b = blockIdx.x, lo = offsets[b], hi = offsets[b + 1];
    F9 acc = f9_zero();
    for (uint32_t t = lo + threadIdx.x; t < hi; t += SH_THREADS) acc = f9_norm_red<P9>(f9_add(acc, f9_load(E + sorted[t])));
    acc = f9_wave_sum<P9>(acc);
    __shared__ F9 red[SH_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        F9 s = red[0];
        for (int w = 1; w < SH_THREADS / 64; w++) s = f9_norm_red<P9>(f9_add(s, red[w]));
        fe_store(G + b, f9_canon<P9>(s));
    }
}

// few buckets (the 4-bit chunks of compute_ra_evals: d * 16 <= 256 of them): global counters would take T * d atomics
// on a handful of addresses (measured 5.4 ms at T = 2^16, d = 16), so each workgroup sums the 32-bit words of E[j]
// into 64-bit LDS accumulators per bucket, adds them to a global accumulator once, and the host reduces the
// n_buckets x 8 word sums mod p (sum_to_fr).  No sort, one pass over the indices.
constexpr uint32_t SH_SMALL_BUCKETS = 512;
__global__ __launch_bounds__(SH_THREADS) void k_sh_hist_small(const uint64_t* __restrict__ idx, size_t T, KeySpec S,
                                                              const Fe* __restrict__ E, unsigned long long* __restrict__ acc /* [n_buckets][8] */) {
    __shared__ unsigned long long sm[SH_SMALL_BUCKETS * 8];
    const uint32_t n_buckets = S.d << S.log_k_chunk;
    for (uint32_t w = threadIdx.x; w < n_buckets * 8; w += SH_THREADS) sm[w] = 0;
    __syncthreads();
    for (size_t j = (size_t)blockIdx.x * SH_THREADS + threadIdx.x; j < T; j += (size_t)gridDim.x * SH_THREADS) {
        const uint64_t v = idx[j];
        const Fe e = fe_load(E + j);
        for (uint32_t i = 0; i < S.d; i++) {
            unsigned long long* b = sm + (size_t)key_of(v, i, S) * 8;
#pragma unroll
            for (int w = 0; w < 8; w++) atomicAdd(&b[w], (unsigned long long)e.v[w]);
        }
    }
    __syncthreads();
    for (uint32_t w = threadIdx.x; w < n_buckets * 8; w += SH_THREADS)
        if (sm[w]) atomicAdd(&acc[w], sm[w]);
}

// medium tables (the 2^16 .. 2^18-entry activation / exp tables of Tanh and SoftmaxLastAxis, T <= 2^20 lookups): no sort either.  Every lookup
// adds the eight 32-bit words of E[j] to 64-bit accumulators of its bucket in HBM (distinct buckets mostly: the atomics spread over the L2
// channels), one thread per bucket turns the word sums back into a residue (lo R R^-1 + hi R^2 R^-1, as k_ps_q_final does).  The counting
// sort below spent ~1.6 ms per call in its one-workgroup scan over 2^18 counters — 66 ms of a GPT-2-shaped proof.
__global__ __launch_bounds__(SH_THREADS) void k_sh_hist_words(const uint64_t* __restrict__ idx, size_t T, KeySpec S, const Fe* __restrict__ E,
                                                              unsigned long long* __restrict__ acc /* [n_buckets][8], zeroed */) {
    for (size_t j = (size_t)blockIdx.x * SH_THREADS + threadIdx.x; j < T; j += (size_t)gridDim.x * SH_THREADS) {
        const uint64_t v = idx[j];
        const Fe e = fe_load(E + j);
        for (uint32_t i = 0; i < S.d; i++) {
            unsigned long long* b = acc + (size_t)key_of(v, i, S) * 8;
#pragma unroll
            for (int w = 0; w < 8; w++) atomicAdd(&b[w], (unsigned long long)e.v[w]);
        }
    }
}
__global__ __launch_bounds__(SH_THREADS) void k_sh_words_final(const unsigned long long* __restrict__ acc, uint32_t n_buckets, Fe* __restrict__ G) {
    const uint32_t b = blockIdx.x * SH_THREADS + threadIdx.x;
    if (b >= n_buckets) return;
    Fe lo, hi, r2;
    unsigned long long c = 0, any = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const unsigned long long a = acc[(size_t)b * 8 + w];
        any |= a;
        const unsigned long long s2 = c + (a & 0xffffffffull);
        lo.v[w] = (uint32_t)s2;
        c = (s2 >> 32) + (a >> 32);
    }
    if (!any) { fe_store(G + b, fe_zero()); return; }
#pragma unroll
    for (int w = 0; w < 8; w++) { hi.v[w] = 0; r2.v[w] = FrParams::r2(w); }
    hi.v[0] = (uint32_t)c; hi.v[1] = (uint32_t)(c >> 32);
    fe_store(G + b, fr_add(fr_mul(lo, fr_one()), fr_mul(hi, r2)));
}

// one thread per bucket (large tables, short lists)
__global__ __launch_bounds__(SH_THREADS) void k_sh_bucket_sum_thread(const Fe* __restrict__ E, const uint32_t* __restrict__ sorted,
                                                                     const uint32_t* __restrict__ offsets, uint32_t n_buckets,
                                                                     Fe* __restrict__ G) {
    using P9 = Fr9Params;
    const uint32_t b = blockIdx.x * SH_THREADS + threadIdx.x;
    if (b >= n_buckets) return;
    F9 acc = f9_zero();
    for (uint32_t t = offsets[b]; t < offsets[b + 1]; t++) acc = f9_norm_red<P9>(f9_add(acc, f9_load(E + sorted[t])));
    fe_store(G + b, f9_canon<P9>(acc));
}

// W[k] = from_i32(table[k]) + gamma * k   (val + gamma * IdentityPolynomial)
__global__ __launch_bounds__(SH_THREADS) void k_sh_build_w(const int32_t* __restrict__ table, size_t K, Fe gamma, Fe* __restrict__ W) {
    for (size_t k = (size_t)blockIdx.x * SH_THREADS + threadIdx.x; k < K; k += (size_t)gridDim.x * SH_THREADS) {
        const Fe v = fr_from_i64((int64_t)table[k]);
        const Fe kk = fr_from_i64((int64_t)k);
        fe_store(W + k, fr_add(v, fr_mul(gamma, kk)));
    }
}

inline int grid_for(size_t work) {
    size_t b = (work + SH_THREADS - 1) / SH_THREADS;
    return (int)(b < 1 ? 1 : b > 2048 ? 2048 : b);
}

// the few-bucket histogram with its result on the host (n_buckets <= SH_SMALL_BUCKETS): word sums in LDS, one D2H of
// n_buckets x 8 words into the pinned staging area, reduction mod p on the host
int histogram_small_host(const uint64_t* h_idx, size_t T, KeySpec S, const atlas_poly* E, std::vector<H::Fr>& out) {
    const uint32_t n_buckets = S.d << S.log_k_chunk;
    const size_t nw = (size_t)n_buckets * 8;
    if (nw * 8 > atlas_rt::PINNED_BYTES) return fail(ATLAS_EINVAL, "shout: staging area too small");
    DevBuf acc_b, ix_b;
    HIP_TRY(acc_b.alloc(nw * 8));
    hipPointerAttribute_t attr;
    const bool on_device = T && hipPointerGetAttributes(&attr, h_idx) == hipSuccess && attr.type == hipMemoryTypeDevice;      // (read in place: no copy of a device vector)
    (void)hipGetLastError();
    HIP_TRY(hipMemsetAsync(acc_b.p, 0, nw * 8, rt().stream));
    if (!on_device) { HIP_TRY(ix_b.alloc((T ? T : 1) * 8)); HIP_TRY(hipMemcpyAsync(ix_b.p, h_idx, T * 8, hipMemcpyHostToDevice, rt().stream)); }
    size_t blocks = (T + SH_THREADS - 1) / SH_THREADS; if (blocks < 1) blocks = 1; if (blocks > 512) blocks = 512;
    k_sh_hist_small<<<(unsigned)blocks, SH_THREADS, 0, rt().stream>>>(on_device ? h_idx : ix_b.as<uint64_t>(), T, S, (const Fe*)E->d, acc_b.as<unsigned long long>());
    HIP_TRY(hipMemcpyAsync(rt().h_pinned, acc_b.p, nw * 8, hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    const unsigned long long* h_acc = reinterpret_cast<const unsigned long long*>(rt().h_pinned);
    out.resize(n_buckets);
    for (uint32_t b = 0; b < n_buckets; b++) {
        uint64_t a9[9];
        for (int w = 0; w < 8; w++) a9[w] = h_acc[(size_t)b * 8 + w];       // words of canonical residues < 2^32, < 2^31 of them: 64 bits suffice
        a9[8] = 0;
        out[b] = atlas_rt::sum_to_fr(a9, 32, 0);
    }
    return ATLAS_OK;
}

// weighted histogram: G[key] = sum E[j] over the (j, i) pairs with that key; n_buckets = d * k_chunk
int histogram(const uint64_t* h_idx, size_t T, KeySpec S, const atlas_poly* E, atlas_poly_t* out) {
    const uint32_t n_buckets = S.d << S.log_k_chunk;
    if (E->is_i32 || E->len < T) return fail(ATLAS_EINVAL, "shout: eq table shorter than the index list");
    if (T >= (1ull << 31) || (uint64_t)T * S.d >= (1ull << 32)) return fail(ATLAS_EINVAL, "shout: too many lookups");
    if (n_buckets <= SH_SMALL_BUCKETS && T < ((size_t)1 << 31)) {
        std::vector<H::Fr> h_G;
        int rc = histogram_small_host(h_idx, T, S, E, h_G);
        if (rc) return rc;
        Fe* Gs = nullptr;
        hipError_t e2 = hipMalloc(&Gs, (size_t)n_buckets * sizeof(Fe));
        if (e2 != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(G)", e2);
        e2 = hipMemcpyAsync(Gs, h_G.data(), (size_t)n_buckets * sizeof(Fe), hipMemcpyHostToDevice, rt().stream);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(rt().stream);
        if (e2 != hipSuccess) { hipFree(Gs); return fail(ATLAS_ENODEV, "shout histogram", e2); }
        atlas_poly* p = new atlas_poly();
        p->d = Gs; p->len = n_buckets; p->cap_bytes = (size_t)n_buckets * sizeof(Fe); p->is_i32 = false; p->owned = true;
        *out = p;
        return ATLAS_OK;
    }
    static const bool no_words = getenv("ATLAS_SH_NO_WORDS") != nullptr;      // A-B
    if (n_buckets <= (1u << 20) && !no_words) {
        hipPointerAttribute_t attr;
        const bool on_device = T && hipPointerGetAttributes(&attr, h_idx) == hipSuccess && attr.type == hipMemoryTypeDevice;
        (void)hipGetLastError();
        DevBuf acc_b, ix_b;
        HIP_TRY(acc_b.alloc((size_t)n_buckets * 64));
        if (!on_device) { HIP_TRY(ix_b.alloc((T ? T : 1) * 8)); HIP_TRY(hipMemcpyAsync(ix_b.p, h_idx, T * 8, hipMemcpyHostToDevice, rt().stream)); }
        Fe* Gw = nullptr;
        hipError_t e3 = hipMalloc(&Gw, (size_t)n_buckets * sizeof(Fe));
        if (e3 != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(G)", e3);
        HIP_TRY(hipMemsetAsync(acc_b.p, 0, (size_t)n_buckets * 64, rt().stream));
        k_sh_hist_words<<<grid_for(T), SH_THREADS, 0, rt().stream>>>(on_device ? h_idx : ix_b.as<uint64_t>(), T, S, (const Fe*)E->d, acc_b.as<unsigned long long>());
        k_sh_words_final<<<(n_buckets + SH_THREADS - 1) / SH_THREADS, SH_THREADS, 0, rt().stream>>>(acc_b.as<unsigned long long>(), n_buckets, Gw);
        // (stream order is all the caller needs: G is consumed by launches on this stream; the scratch returns to the pool under this stream's tag)
        e3 = hipGetLastError();
        if (e3 != hipSuccess) { hipFree(Gw); return fail(ATLAS_ENODEV, "shout histogram", e3); }
        atlas_poly* p = new atlas_poly();
        p->d = Gw; p->len = n_buckets; p->cap_bytes = (size_t)n_buckets * sizeof(Fe); p->is_i32 = false; p->owned = true;
        *out = p;
        return ATLAS_OK;
    }
    uint64_t* d_idx = nullptr; uint32_t *counts = nullptr, *offsets = nullptr, *cursor = nullptr, *sorted = nullptr;
    Fe* G = nullptr;
    auto cleanup = [&]() { hipFree(d_idx); hipFree(counts); hipFree(offsets); hipFree(cursor); hipFree(sorted); };
    HIP_TRY(hipMalloc(&d_idx, (T ? T : 1) * 8));
    HIP_TRY(hipMalloc(&counts, (size_t)(n_buckets + 1) * 4));
    HIP_TRY(hipMalloc(&offsets, (size_t)(n_buckets + 1) * 4));
    HIP_TRY(hipMalloc(&cursor, (size_t)(n_buckets + 1) * 4));
    HIP_TRY(hipMalloc(&sorted, (T * S.d ? T * S.d : 1) * 4));
    hipError_t e = hipMalloc(&G, (size_t)n_buckets * sizeof(Fe));
    if (e != hipSuccess) { cleanup(); return fail(ATLAS_ENOMEM, "hipMalloc(G)", e); }
    HIP_TRY(hipMemcpyAsync(d_idx, h_idx, T * 8, hipMemcpyDefault, rt().stream));   // host or device source
    HIP_TRY(hipMemsetAsync(counts, 0, (size_t)(n_buckets + 1) * 4, rt().stream));
    k_sh_hist<<<grid_for(T), SH_THREADS, 0, rt().stream>>>(d_idx, T, S, counts);
    k_exclusive_scan<<<1, 1024, 0, rt().stream>>>(counts, n_buckets, offsets, cursor);
    k_sh_scatter<<<grid_for(T), SH_THREADS, 0, rt().stream>>>(d_idx, T, S, cursor, sorted);
    if (n_buckets <= 8192)
        k_sh_bucket_sum_wg<<<n_buckets, SH_THREADS, 0, rt().stream>>>((const Fe*)E->d, sorted, offsets, G);
    else
        k_sh_bucket_sum_thread<<<(n_buckets + SH_THREADS - 1) / SH_THREADS, SH_THREADS, 0, rt().stream>>>((const Fe*)E->d, sorted, offsets, n_buckets, G);
    e = hipStreamSynchronize(rt().stream);
    cleanup();
    if (e != hipSuccess) { hipFree(G); return fail(ATLAS_ENODEV, "shout histogram", e); }
    atlas_poly* p = new atlas_poly();
    p->d = G; p->len = n_buckets; p->cap_bytes = (size_t)n_buckets * sizeof(Fe); p->is_i32 = false; p->owned = true;
    *out = p;
    return ATLAS_OK;
}

}  // namespace

extern "C" {

int atlas_shout_read_raf_G(const uint64_t* lookup_indices, size_t T, size_t log_K, atlas_poly_t eq_r, atlas_poly_t* out) {
    PROF("atlas_shout_read_raf_G");
    NEED_INIT();
    if ((!lookup_indices && T) || !eq_r || !out || log_K > 24) return fail(ATLAS_EINVAL, "shout_read_raf_G");
    // the reference indexes the table with bounds checks.  Host indices are range-checked here; device-resident ones by a kernel that
    // runs ahead of the histogram on the same stream (reading them through the host mapping of HBM costs ~1.3 us per element: 21 ms of a
    // Tanh node), its flag read back after the histogram's own synchronisation
    hipPointerAttribute_t attr;
    const bool on_device = T && hipPointerGetAttributes(&attr, lookup_indices) == hipSuccess && attr.type == hipMemoryTypeDevice;
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    DevBuf flag;
    if (!on_device) {
        (void)hipGetLastError();
        for (size_t j = 0; j < T; j++)
            if (lookup_indices[j] >> log_K) return fail(ATLAS_EINVAL, "shout_read_raf_G: lookup index outside the table");
    } else {
        HIP_TRY(flag.alloc(4));
        HIP_TRY(hipMemsetAsync(flag.p, 0, 4, rt().stream));
        k_sh_range_flag<<<grid_for(T), SH_THREADS, 0, rt().stream>>>(lookup_indices, T, (uint32_t)log_K, flag.as<uint32_t>());
    }
    int rc = histogram(lookup_indices, T, KeySpec{1u, (uint32_t)log_K}, eq_r, out);
    if (!rc && on_device) {
        uint32_t bad = 0;
        HIP_TRY(hipMemcpyAsync(&bad, flag.p, 4, hipMemcpyDeviceToHost, rt().stream));      // on the library's (non-blocking) stream, behind the flag kernel
        HIP_TRY(hipStreamSynchronize(rt().stream));
        if (bad) { atlas_poly_free(*out); *out = nullptr; return fail(ATLAS_EINVAL, "shout_read_raf_G: lookup index outside the table"); }
    }
    return rc;
}

int atlas_shout_ra_evals(const uint64_t* lookup_indices, size_t T, size_t log_K, size_t log_k_chunk, atlas_poly_t eq_r_cycle,
                         atlas_poly_t* out) {
    NEED_INIT();
    if ((!lookup_indices && T) || !eq_r_cycle || !out || log_k_chunk == 0 || log_k_chunk > 16 || log_K == 0 || log_K > 64)
        return fail(ATLAS_EINVAL, "shout_ra_evals");
    const uint32_t d = (uint32_t)((log_K + log_k_chunk - 1) / log_k_chunk);     // instruction_d (config.rs:45)
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    return histogram(lookup_indices, T, KeySpec{d, (uint32_t)log_k_chunk}, eq_r_cycle, out);
}

}  // extern "C"

// compute_ra_evals with the tables on the host (the node provers hand them to the HammingWeight / Booleanity constructors):
// no device polynomial in between.  Not part of the C-ABI (internal.hpp).
int atlas_rt_shout_ra_evals_host(const uint64_t* lookup_indices, size_t T, size_t log_K, size_t log_k_chunk, atlas_poly_t eq_r_cycle,
                                 std::vector<atlas_host::Fr>& G) {
    PROF("atlas_rt_shout_ra_evals_host");
    const uint32_t d = (uint32_t)((log_K + log_k_chunk - 1) / log_k_chunk);
    if (!lookup_indices || !eq_r_cycle || log_k_chunk == 0 || log_k_chunk > 16 || ((size_t)d << log_k_chunk) > SH_SMALL_BUCKETS)
        return fail(ATLAS_EINVAL, "shout_ra_evals_host");
    if (eq_r_cycle->is_i32 || eq_r_cycle->len < T) return fail(ATLAS_EINVAL, "shout: eq table shorter than the index list");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    return histogram_small_host(lookup_indices, T, KeySpec{d, (uint32_t)log_k_chunk}, eq_r_cycle, G);
}

// ReadRafProver (shout.rs:193-262) over a SMALL table on the host: the same dot-product instance sum_k G[k] W[k], W = val + gamma * int,
// HighToLow, degree 2 — as host arithmetic (2^log_K <= 4096 entries: a few microseconds per round), stepped over the round channel without
// a launch.  The exp-digit lookups of SoftmaxLastAxis read 512-entry tables; as device dot provers they were the one member that kept the
// whole stage-3 batch (a clamp lookup and a one-hot family beside them) on the host-stepped driver, a synchronisation per member and round.
#include "instance.hpp"
namespace {
struct HostDot : atlas_instance {
    std::vector<H::Fr> L, R;
    size_t n = 0, round_next = 0;
    size_t rounds() const override { return n; }
    size_t degree() const override { return 2; }
    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= n) return fail(ATLAS_ESTATE, "host read-raf: round out of order");
        const size_t h = L.size() / 2;
        H::Fr e0 = H::zero(), e2 = H::zero();
        for (size_t i = 0; i < h; i++) {
            if (H::detail::is_zero4(L[i].l) && H::detail::is_zero4(L[i + h].l)) continue;          // (G is a histogram: mostly empty)
            e0 = H::add(e0, H::mul(L[i], R[i]));
            const H::Fr l2 = H::sub(H::add(L[i + h], L[i + h]), L[i]), r2 = H::sub(H::add(R[i + h], R[i + h]), R[i]);
            e2 = H::add(e2, H::mul(l2, r2));
        }
        const H::Fr ev[2] = {e0, e2};
        coeffs.assign(3, H::zero());
        H::unipoly_from_evals_and_hint(claim, ev, 2, coeffs.data());
        return ATLAS_OK;
    }
    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= n) return fail(ATLAS_ESTATE, "host read-raf: round out of order");
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode);
        const size_t h = L.size() / 2;
        for (size_t i = 0; i < h; i++) {
            if (!(H::detail::is_zero4(L[i].l) && H::detail::is_zero4(L[i + h].l))) L[i] = H::add(L[i], H::mul(rf, H::sub(L[i + h], L[i])));
            R[i] = H::add(R[i], H::mul(rf, H::sub(R[i + h], R[i])));
        }
        L.resize(h); R.resize(h);
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override {
        if (round_next != n) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        out = {L[0], R[0], H::one()};
        return ATLAS_OK;
    }
    bool pipelined() const override { return true; }
    bool wide_wait(size_t) const override { return false; }
    bool silent_round(size_t) const override { return true; }
    int enqueue(size_t, const atlas::RoundIo& io, bool, atlas_mail_ref& mail) override { mail.base = io.mail; mail.blocks = 0; mail.n_vals = 0; return ATLAS_OK; }
    int finish(size_t round, const H::Fr& claim, const H::Fr*, std::vector<H::Fr>& coeffs) override { return message(round, claim, coeffs); }
    int host_ingest(const atlas_u128_t& r, size_t round) override { return ingest(r, round); }
    int enqueue_finals(const atlas::RoundIo& io, atlas_mail_ref& mail) override { mail.base = io.mail; mail.blocks = 0; mail.n_vals = 0; return ATLAS_OK; }
    int set_finals(const H::Fr*, size_t) override { return ATLAS_OK; }
};
}  // namespace
int atlas_rt_shout_read_raf_host_new(atlas_poly_t G, const int32_t* table, size_t log_K, const atlas_fr_t* gamma, atlas_instance_t* out) {
    PROF("atlas_rt_shout_read_raf_host_new");
    if (!G || !table || !gamma || !out || log_K == 0 || log_K > 12) return fail(ATLAS_EINVAL, "shout_read_raf_host_new");
    const size_t K = (size_t)1 << log_K;
    if (G->is_i32 || G->len != K) return fail(ATLAS_EINVAL, "shout_read_raf_host_new: G length != table size");
    std::unique_ptr<HostDot> P(new HostDot());
    P->n = log_K;
    P->L.resize(K); P->R.resize(K);
    int rc = atlas_poly_download(G, reinterpret_cast<atlas_fr_t*>(P->L.data()), K);
    if (rc) return rc;
    const H::Fr gm = *reinterpret_cast<const H::Fr*>(gamma);
    for (size_t k = 0; k < K; k++) {
        const int64_t v = table[k];
        const H::Fr fv = v >= 0 ? H::from_u64((uint64_t)v) : H::neg(H::from_u64((uint64_t)(-v)));
        P->R[k] = H::add(fv, H::mul(gm, H::from_u64((uint64_t)k)));
    }
    *out = P.release();
    return ATLAS_OK;
}

struct atlas_rt_ra_ticket { DevBuf acc; const unsigned long long* box = nullptr; uint32_t n_buckets = 0; };
int atlas_rt_shout_ra_evals_launch(const uint64_t* lookup_indices, size_t T, size_t log_K, size_t log_k_chunk, atlas_poly_t eq_r_cycle, atlas_rt_ra_ticket** out) {
    PROF("atlas_rt_shout_ra_evals_launch");
    const uint32_t d = (uint32_t)((log_K + log_k_chunk - 1) / log_k_chunk);
    if (!lookup_indices || !eq_r_cycle || !out || log_k_chunk == 0 || log_k_chunk > 16 || ((size_t)d << log_k_chunk) > SH_SMALL_BUCKETS)
        return fail(ATLAS_EINVAL, "shout_ra_evals_launch");
    if (eq_r_cycle->is_i32 || eq_r_cycle->len < T) return fail(ATLAS_EINVAL, "shout: eq table shorter than the index list");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, lookup_indices) != hipSuccess || attr.type != hipMemoryTypeDevice) { (void)hipGetLastError(); return fail(ATLAS_EINVAL, "shout_ra_evals_launch: device-resident indices expected"); }
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    std::unique_ptr<atlas_rt_ra_ticket> tk(new atlas_rt_ra_ticket());
    const KeySpec S{d, (uint32_t)log_k_chunk};
    tk->n_buckets = d << log_k_chunk;
    const size_t nw = (size_t)tk->n_buckets * 8;
    HIP_TRY(tk->acc.alloc(nw * 8));
    HIP_TRY(hipMemsetAsync(tk->acc.p, 0, nw * 8, rt().stream));
    size_t blocks = (T + SH_THREADS - 1) / SH_THREADS; if (blocks < 1) blocks = 1; if (blocks > 512) blocks = 512;
    k_sh_hist_small<<<(unsigned)blocks, SH_THREADS, 0, rt().stream>>>(lookup_indices, T, S, (const Fe*)eq_r_cycle->d, tk->acc.as<unsigned long long>());
    atlas::Chunk* box = rt().chan.alloc_long((nw * 8 + 15) / 16);          // pinned: the copy lands there in stream order
    HIP_TRY(hipMemcpyAsync(box, tk->acc.p, nw * 8, hipMemcpyDeviceToHost, rt().stream));
    tk->box = reinterpret_cast<const unsigned long long*>(box);
    *out = tk.release();
    return ATLAS_OK;
}
int atlas_rt_shout_ra_evals_finish(atlas_rt_ra_ticket* tk, bool wait, std::vector<atlas_host::Fr>& G) {
    if (!tk) return fail(ATLAS_EINVAL, "shout_ra_evals_finish");
    if (wait) { std::lock_guard<atlas_rt::Mutex> lk(rt().mu); HIP_TRY(hipStreamSynchronize(rt().stream)); }
    G.resize(tk->n_buckets);
    for (uint32_t b = 0; b < tk->n_buckets; b++) {
        uint64_t a9[9];
        for (int w = 0; w < 8; w++) a9[w] = tk->box[(size_t)b * 8 + w];
        a9[8] = 0;
        G[b] = atlas_rt::sum_to_fr(a9, 32, 0);
    }
    return ATLAS_OK;
}
void atlas_rt_shout_ra_evals_drop(atlas_rt_ra_ticket* tk) { delete tk; }

extern "C" {

int atlas_shout_read_raf_prover_new(atlas_poly_t G, const int32_t* table, size_t log_K, const atlas_fr_t* gamma,
                                    atlas_dot_prover_t* out) {
    PROF("atlas_shout_read_raf_prover_new");
    NEED_INIT();
    if (!G || !table || !gamma || !out) return fail(ATLAS_EINVAL, "shout_read_raf_prover_new");
    const size_t K = (size_t)1 << log_K;
    if (G->is_i32 || G->len != K) return fail(ATLAS_EINVAL, "shout_read_raf_prover_new: G length != table size");
    int32_t* d_tab = nullptr; Fe* W = nullptr;
    {
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        HIP_TRY(hipMalloc(&d_tab, K * 4));
        hipError_t e = hipMalloc(&W, K * sizeof(Fe));
        if (e != hipSuccess) { hipFree(d_tab); return fail(ATLAS_ENOMEM, "hipMalloc(W)", e); }
        HIP_TRY(hipMemcpyAsync(d_tab, table, K * 4, hipMemcpyHostToDevice, rt().stream));
        Fe gm; std::memcpy(gm.v, gamma, 32);
        k_sh_build_w<<<grid_for(K), SH_THREADS, 0, rt().stream>>>(d_tab, K, gm, W);
        e = hipStreamSynchronize(rt().stream);
        hipFree(d_tab);
        if (e != hipSuccess) { hipFree(W); return fail(ATLAS_ENODEV, "shout build W", e); }
    }
    atlas_poly* w = new atlas_poly();
    w->d = W; w->len = K; w->cap_bytes = K * sizeof(Fe); w->is_i32 = false; w->owned = true;
    int rc = atlas_dot_prover_new(G, w, nullptr, ATLAS_EQ_NONE, 0, 0, out);
    if (rc) atlas_poly_free(w);
    return rc;
}

}  // extern "C"
