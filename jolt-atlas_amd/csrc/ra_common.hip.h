// Shared device plumbing of the one-hot "ra" instances (ra.hip) and the prefix-suffix Shout read-raf
// instances (psshout.hip): the LowToHigh split-eq device tables, the d-row ping-pong buffers of an
// instance (gather from one-hot indices, bind, final claims) and the column reduction of per-block
// partials.  Everything here has internal linkage; include it from one .hip translation unit at a time.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
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
using atlas_rt::rt;

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

// The same gather straight from the T lookup indices (chunk i = (index >> log_k (d - 1 - i)) & (2^log_k - 1), i = 0 most significant:
// OneHotParams::lookup_index_chunk, config.rs:73-75): no chunk-index rows in between (a launch and d T int32 per instance).
__global__ __launch_bounds__(RA_THREADS) void k_ra_gather_lk(const uint64_t* __restrict__ lookups, const Fr* __restrict__ F, uint32_t f_stride, size_t T, uint32_t d,
                                                             uint32_t log_k, Fr* __restrict__ out) {
    const uint32_t i = blockIdx.y, shift = log_k * (d - 1 - i);
    const uint64_t mask = ((uint64_t)1 << log_k) - 1;
    for (size_t j = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; j < T; j += (size_t)gridDim.x * RA_THREADS) {
        const uint64_t k = shift >= 64 ? 0 : ((lookups[j] >> shift) & mask);
        fe_store(out + (size_t)i * T + j, fe_load(F + (size_t)i * f_stride + k));
    }
}

// ---- lazy rounds of RaPolynomial (poly/ra_poly.rs:21-110: the reference keeps (index, table) for the first binds) --------------------------
// With log_k = 4 and d <= 16 the d chunk indices of a cycle are the nibbles of ONE 64-bit word (chunk i = nibble d - 1 - i): the word is all a
// round needs while the row values come from a table — F_i (16 entries) in round 0, T1_i[a, b] = F_i[a] + r0 (F_i[b] - F_i[a]) (256 entries) in
// round 1 — so rounds 0 and 1 read 8 bytes per cycle instead of 32 d, and the rows are materialised bound through r1 (T / 4 long) for round 2.
// d rows of T int32 chunk indices -> the packed words; *bad is raised by an index outside 0..15 (a None lookup: the caller gathers rows instead)
__global__ __launch_bounds__(RA_THREADS) void k_ra_pack_nibbles(const int32_t* __restrict__ idx /* [d][T] */, size_t T, uint32_t d, uint64_t* __restrict__ out, uint32_t* bad) {
    for (size_t j = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; j < T; j += (size_t)gridDim.x * RA_THREADS) {
        uint64_t w = 0;
        bool b = false;
        for (uint32_t i = 0; i < d; i++) {
            const int32_t v = idx[(size_t)i * T + j];
            b |= v < 0 || v > 15;
            w |= (uint64_t)(v & 15) << (4 * (d - 1 - i));
        }
        out[j] = w;
        if (b) atomicOr(bad, 1u);
    }
}
// T1[i][16 a + b] = bind(F_i[a], F_i[b]; r0): one workgroup, waits for round 0's challenge
__global__ __launch_bounds__(RA_THREADS) void k_ra_lazy_t1(const Fr* __restrict__ F, uint32_t f_stride, uint32_t d, Fr* __restrict__ T1, ChanIo io, int r_hi_only) {
    Fr r;
    if (!io.challenge(r)) return;
    for (uint32_t t = threadIdx.x; t < d * 256u; t += RA_THREADS) {
        const uint32_t i = t >> 8, a = (t >> 4) & 15u, b = t & 15u;
        fe_store(T1 + t, bind_pair(fe_load(F + (size_t)i * f_stride + a), fe_load(F + (size_t)i * f_stride + b), r, r_hi_only != 0));
    }
}
// rows of round 2: out[i][j] = bind(T1_i[cycles 4j, 4j + 1], T1_i[cycles 4j + 2, 4j + 3]; r1); grid (x, d), waits for round 1's challenge
__global__ __launch_bounds__(RA_THREADS) void k_ra_lazy_rows(const uint64_t* __restrict__ lookups, const Fr* __restrict__ T1, uint32_t d, size_t quarter,
                                                             Fr* __restrict__ out, size_t out_stride, ChanIo io, int r_hi_only) {
    Fr r;
    if (!io.challenge(r)) return;
    const uint32_t i = blockIdx.y, sh = 4 * (d - 1 - i);
    const Fr* t1 = T1 + (size_t)i * 256;
    for (size_t j = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; j < quarter; j += (size_t)gridDim.x * RA_THREADS) {
        const uint64_t w0 = lookups[4 * j], w1 = lookups[4 * j + 1], w2 = lookups[4 * j + 2], w3 = lookups[4 * j + 3];
        const uint32_t p0 = (uint32_t)((w0 >> sh) & 15u) * 16u + (uint32_t)((w1 >> sh) & 15u), p1 = (uint32_t)((w2 >> sh) & 15u) * 16u + (uint32_t)((w3 >> sh) & 15u);
        fe_store(out + (size_t)i * out_stride + j, bind_pair(fe_load(t1 + p0), fe_load(t1 + p1), r, r_hi_only != 0));
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

// ---- round-channel versions (channel.hip.h): the challenge comes from the round's slot, the sums go to the host by mail
__global__ __launch_bounds__(RA_THREADS) void k_ra_bind_ch(const Fr* __restrict__ src, size_t src_stride, Fr* __restrict__ dst,
                                                           size_t dst_stride, size_t half, ChanIo io, int r_hi_only) {
    Fr r;
    if (!io.challenge(r)) return;
    const Fr* s = src + (size_t)blockIdx.y * src_stride;
    Fr* d = dst + (size_t)blockIdx.y * dst_stride;
    for (size_t j = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; j < half; j += (size_t)gridDim.x * RA_THREADS)
        fe_store(d + j, bind_pair(fe_load(s + 2 * j), fe_load(s + 2 * j + 1), r, r_hi_only != 0));
}

// column sums of a [n_partials][K] matrix of partial sums (K <= 16), mailed as ONE record of K values.  One workgroup:
// thread t adds every (256 / K')-th row of column t % K' (K' = K rounded up to a power of two); the row groups meet in LDS.
__device__ __forceinline__ void col_reduce_mail_body(const Fr* partials, uint32_t n_partials, uint32_t K, const RoundIo& io) {
    __shared__ Fr red[RA_THREADS];
    __shared__ uint32_t stage[9 * 16];
    const uint32_t kp = K <= 2 ? 2u : K <= 4 ? 4u : K <= 8 ? 8u : 16u;          // columns padded to a power of two
    const uint32_t n_grp = RA_THREADS / kp;
    const uint32_t col = threadIdx.x % kp, grp = threadIdx.x / kp;
    Fr acc = fe_zero();
    if (col < K)
        for (uint32_t p = grp; p < n_partials; p += n_grp) acc = fr_add(acc, fe_load(partials + (size_t)p * K + col));
    red[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t st = n_grp / 2; st >= 1; st >>= 1) {                          // tree over the row groups
        if (grp < st) red[threadIdx.x] = fr_add(red[threadIdx.x], red[threadIdx.x + st * kp]);
        __syncthreads();
    }
    if (threadIdx.x < 64) {
        const Fr s = threadIdx.x < kp ? red[threadIdx.x] : fe_zero();
        ch_mail_wave_fe(io, 0, K, s, stage);
    }
}
__global__ __launch_bounds__(RA_THREADS) void k_col_reduce_mail(const Fr* __restrict__ partials, uint32_t n_partials, uint32_t K, RoundIo io) {
    col_reduce_mail_body(partials, n_partials, K, io);
}

// The same as the tail of the kernel that produced the partial rows (saves a launch, ~6 us per round): every workgroup
// calls this after it stored its row; the last one to arrive (agent-scope counter, fences on both sides) adds the rows
// and mails the sums.  `tail.counter` is zero on entry and is left zero.  tail.counter == nullptr: no mail (host-stepped).
constexpr size_t MAIL_TAIL_COUNTER_BYTES = 256;
struct MailTail {
    RoundIo io;
    uint32_t* counter;
    uint32_t n_rows, K;
    Chunk* tagged = nullptr;     // != null: the tagged-row protocol below (tail_put / tail_reduce) instead of the arrival counter
};

// The same hand-over WITHOUT the arrival counter.  Behind the counter every workgroup pays an agent-scope release fence (the dirty lines of
// its XCD's L2 written back) and a read-modify-write of one address: ~0.2 us per workgroup, one after the other — 120 us for the 512
// workgroups of a one-hot round of 2^12 pairs, the whole round.  Here a partial value travels like mail does to the host: three 16-byte
// chunks {payload words, tag}, each ONE device-scope store (a reader that sees the tag sees the payload: no fence), into a row area of HBM;
// workgroup (0, 0) — once its own values are out — polls the n_rows x K values with device-scope loads, adds them word by word (32-bit words
// of canonical residues in 64-bit sums), reduces once per column and mails the K sums.  The tag is the launch's mail tag: it never repeats,
// so the area needs no clearing.  A value costs 48 bytes instead of 32.
__device__ __forceinline__ void tail_put(const MailTail& tail, uint32_t row, uint32_t k, const Fr& v) {
    if (!tail.tagged) return;
    Chunk* c = tail.tagged + ((size_t)row * tail.K + k) * 3;
    const uint32_t tag = tail.io.tag_mail;
    ch_store_dev(c, ch_u32x4{v.v[0], v.v[1], v.v[2], tag});
    ch_store_dev(c + 1, ch_u32x4{v.v[3], v.v[4], v.v[5], tag});
    ch_store_dev(c + 2, ch_u32x4{v.v[6], v.v[7], 0u, tag});
}
// every thread of every workgroup calls this last; blockDim.x = RA_THREADS
__device__ __forceinline__ void tail_reduce(const MailTail& tail) {
    if (!tail.tagged || blockIdx.x != 0 || blockIdx.y != 0) return;
    __shared__ unsigned long long tr_sm[RA_THREADS][3];
    __shared__ uint32_t tr_bad;
    __shared__ uint32_t tr_stage[9 * 16];
    const uint32_t nchunk = 3 * tail.K, n_grp = RA_THREADS / nchunk;           // K <= 16: at least 5 row groups
    const uint32_t c = threadIdx.x % nchunk, grp = threadIdx.x / nchunk, tag = tail.io.tag_mail;
    if (threadIdx.x == 0) { tr_bad = 0; ch_stamp(tail.io.abort_flag, CH_EV_REDUCE_BEGIN, tag); }
    __syncthreads();
    unsigned long long s0 = 0, s1 = 0, s2 = 0;
    if (grp < n_grp) {
        const uint64_t t0 = wall_clock64(), t_max = ch_timeout_ticks(tail.io.abort_flag);
        // four rows per trip, their loads in flight together (a device-scope load is ~0.7 us of latency; one at a time the 96 chunks a thread
        // owns in a 512-row launch were the round: 65 us); a chunk whose tag has not arrived is polled on its own
        for (uint32_t row = grp; row < tail.n_rows; row += 4 * n_grp) {
            const Chunk* p[4];
            ch_u32x4 x[4];
            bool live[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t ru = row + (uint32_t)u * n_grp;
                live[u] = ru < tail.n_rows;
                p[u] = tail.tagged + (size_t)(live[u] ? ru : row) * nchunk + c;
            }
            asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
            bool bad = false;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!live[u]) continue;
                uint32_t spins = 0;
                while (x[u].w != tag) {
                    if ((++spins & 63u) == 0 && (__hip_atomic_load(tail.io.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || wall_clock64() - t0 > t_max)) { tr_bad = 1; bad = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                    x[u] = ch_load_dev(p[u]);
                }
                if (bad) break;
                s0 += x[u].x; s1 += x[u].y; s2 += x[u].z;
            }
            if (bad) break;
        }
    }
    tr_sm[threadIdx.x][0] = s0; tr_sm[threadIdx.x][1] = s1; tr_sm[threadIdx.x][2] = s2;
    __syncthreads();
    if (threadIdx.x == 0) ch_stamp(tail.io.abort_flag, CH_EV_ROWS_IN, tag);
    if (tr_bad) return;                                                       // a workgroup of this launch gave up (abort): nothing to mail
    // The K column sums leave as the INTEGERS they are — eight 32-bit words and the carry word, V = sum_w word_w 2^(32 w) < 2^288 — and the
    // host reduces them (Channel::collect adds records word by word and sum_to_fr takes V mod p: the value a canonical residue would have
    // given).  The reduction on this side — two Montgomery products and an addition per column behind a serial walk over the row groups —
    // was 3-6 us of every round of every lane, after the last partial row had arrived.  Stage 1: thread (value k, word w) adds its word
    // over the row groups; stage 2: lane k carries its eight sums into nine words and the wavefront mails them.
    __shared__ unsigned long long tr_col[16][8];
    if (threadIdx.x < 8 * tail.K) {
        const uint32_t k = threadIdx.x >> 3, w = threadIdx.x & 7u;
        unsigned long long a = 0;
        for (uint32_t g2 = 0; g2 < n_grp; g2++) a += tr_sm[g2 * nchunk + 3 * k + w / 3][w % 3];
        tr_col[k][w] = a;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t w9[9];
#pragma unroll
        for (int w = 0; w < 9; w++) w9[w] = 0;
        if (threadIdx.x < tail.K) {
            unsigned long long carry = 0;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                const unsigned long long a = tr_col[threadIdx.x][w], t = carry + (a & 0xffffffffull);
                w9[w] = (uint32_t)t;
                carry = (t >> 32) + (a >> 32);
            }
            w9[8] = (uint32_t)carry;                                          // rows x 2^32 at most: one word
        }
        ch_mail_wave(tail.io, 0, tail.K, w9, tr_stage);
        if (threadIdx.x == 0) ch_stamp(tail.io.abort_flag, CH_EV_MAILED, tag);
    }
}
__device__ __forceinline__ void mail_tail(const Fr* partials, const MailTail& tail) {
    if (tail.tagged) { tail_reduce(tail); return; }     // (the values went out through tail_put)
    if (!tail.counter) return;
    __shared__ uint32_t s_last;
    __threadfence();                                   // this workgroup's row is visible before it is counted
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = gridDim.x * gridDim.y;
        // (Two levels of counters — sixteen on cache lines of their own, the last arrival of each moving on to this one — changed nothing: 8.5 -> 8.8 us
        // for the tail of a 256-workgroup launch, profiles/r04g_split_phases.txt / r04h_subset.txt.  The time is the release fence above: at
        // agent scope it writes the dirty lines of this XCD's L2 back, once per workgroup.)
        const uint32_t t = atomicAdd(tail.counter, 1u);
        s_last = t == total - 1;
        if (t == total - 1) *tail.counter = 0;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                   // ... and the rows of the others before they are read
    col_reduce_mail_body(partials, tail.n_rows, tail.K, tail.io);
}

// block_reduce_store<DEG> (sumcheck_kernels.hip.h) with the workgroup's row also handed to the tagged-row protocol (row = blockIdx.x)
template <int DEG>
__device__ __forceinline__ void block_reduce_put(Fr acc[DEG], Fr* partials, const MailTail& tail) {
    __shared__ Fr red[RA_THREADS / 64][DEG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        Fr s = fr_wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < DEG) {
        Fr s = red[0][threadIdx.x];
        for (int w = 1; w < RA_THREADS / 64; w++) s = fr_add(s, red[w][threadIdx.x]);
        if (tail.tagged) tail_put(tail, blockIdx.x, threadIdx.x, s);
        else fe_store(partials + (size_t)blockIdx.x * DEG + threadIdx.x, s);
    }
}

// the last bind (rows of two coefficients -> the final claims), mailed one value per record
__global__ __launch_bounds__(64) void k_rows_final_ch(const Fr* src, size_t sstride, uint32_t n_rows, ChanIo io, int hi_only) {
    __shared__ uint32_t stage[9 * 16];
    Fr r;
    if (!io.challenge(r)) return;
    Fr v = fe_zero();
    if (threadIdx.x < n_rows) v = bind_pair(fe_load(src + (size_t)threadIdx.x * sstride), fe_load(src + (size_t)threadIdx.x * sstride + 1), r, hi_only != 0);
    ch_mail_wave_fe(io.io, 0, n_rows, v, stage);
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
// out[k] = sum_p partials[p * K + k * col_stride]; one workgroup per column (col_stride = 1: rows of K sums)
__global__ __launch_bounds__(RA_THREADS) void k_col_reduce(const Fr* __restrict__ partials, uint32_t n_partials, uint32_t K,
                                                           Fr* __restrict__ out, uint32_t col_stride = 1) {
    __shared__ Fr red[RA_THREADS / 64];
    const uint32_t k = blockIdx.x;
    Fr acc = fe_zero();
    for (uint32_t p = threadIdx.x; p < n_partials; p += RA_THREADS) acc = fr_add(acc, fe_load(partials + (size_t)p * K + (size_t)k * col_stride));
    acc = fr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Fr s = red[0];
        for (int w = 1; w < RA_THREADS / 64; w++) s = fr_add(s, red[w]);
        fe_store(out + k, s);
    }
}

// A handful of field elements (a point, batching coefficients) reach the device as a KERNEL ARGUMENT: the launch copies
// them while it is being enqueued, so there is no staging buffer to keep alive and no synchronisation — a pageable
// hipMemcpyAsync + hipStreamSynchronize per constructor was ~20 us each, a dozen of them per lookup node.
struct FrArgs { Fr v[64]; };
__global__ void k_store_fr_args(FrArgs a, uint32_t n, Fr* __restrict__ dst) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) fe_store(dst + i, a.v[i]);
}
inline int store_small(const H::Fr* src, size_t n, Fr* dst) {      // n <= 64 per launch; longer arrays in pieces
    for (size_t o = 0; o < n; o += 64) {
        FrArgs a;
        const size_t m = n - o < 64 ? n - o : 64;
        std::memcpy(a.v, src + o, m * sizeof(Fr));
        k_store_fr_args<<<1, 64, 0, rt().stream>>>(a, (uint32_t)m, dst + o);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ATLAS_OK : fail(ATLAS_ENODEV, "store_small", e);
}

// Both cached prefix tables of a split eq in ONE launch, the point as kernel argument (at most 32 variables here; longer points take the
// three launches below): workgroup 0 builds the tables of w[0 .. k_out), workgroup 1 those of w[m .. m + k_in) — evals_cached
// (eq_poly.rs:174-192), table j at offset 2^j - 1, as k_eq_cached.  Every instance over a Gruen split eq starts with this.
struct GsePointArgs { Fr v[32]; };
__global__ __launch_bounds__(1024) void k_gse_init(GsePointArgs w, uint32_t m, uint32_t k_out, uint32_t k_in, Fr* __restrict__ t_out, Fr* __restrict__ t_in) {
    Fr* tabs = blockIdx.x ? t_in : t_out;
    const uint32_t k = blockIdx.x ? k_in : k_out, off = blockIdx.x ? m : 0;
    if (threadIdx.x == 0) fe_store(tabs, fr_one());
    __syncthreads();
    for (uint32_t j = 0; j < k; j++) {
        const Fr wj = w.v[off + j];
        const Fr* cur = tabs + ((1u << j) - 1);
        Fr* nxt = tabs + ((2u << j) - 1);
        for (uint32_t i = threadIdx.x; i < (1u << j); i += 1024) {
            const Fr s = fe_load(cur + i), hi = fr_mul(s, wj);
            fe_store(nxt + 2 * i + 1, hi);
            fe_store(nxt + 2 * i, fr_sub(s, hi));
        }
        __threadfence_block();
        __syncthreads();
    }
}

// device half of a LowToHigh GruenSplitEqPolynomial: the cached prefix tables
struct GseDev {
    H::GseState st;
    Fr *d_w = nullptr, *d_eout = nullptr, *d_ein = nullptr;
    int init(const H::Fr* w, size_t n) {
        st.init(w, n);
        if (st.k_out > 12 || st.k_in > 12) return fail(ATLAS_EINVAL, "split-eq: more than 25 variables not supported");
        HIP_TRY(hipMalloc(&d_eout, ((size_t)2 << st.k_out) * sizeof(Fr)));
        HIP_TRY(hipMalloc(&d_ein, ((size_t)2 << st.k_in) * sizeof(Fr)));
        static const bool one_launch = getenv("ATLAS_GSE_3_LAUNCHES") == nullptr;      // A-B
        if (n <= 32 && one_launch) {
            GsePointArgs a;
            if (n) std::memcpy(a.v, w, n * sizeof(Fr));
            k_gse_init<<<2, 1024, 0, rt().stream>>>(a, (uint32_t)st.m, (uint32_t)st.k_out, (uint32_t)st.k_in, d_eout, d_ein);
            hipError_t e = hipGetLastError();
            return e == hipSuccess ? ATLAS_OK : fail(ATLAS_ENODEV, "split-eq init", e);
        }
        HIP_TRY(hipMalloc(&d_w, (n ? n : 1) * sizeof(Fr)));
        if (n) { int rc = store_small(w, n, d_w); if (rc) return rc; }
        k_eq_cached<<<1, 1024, 0, rt().stream>>>(d_eout, d_w, (uint32_t)st.k_out);
        k_eq_cached<<<1, 1024, 0, rt().stream>>>(d_ein, d_w + st.m, (uint32_t)st.k_in);
        return ATLAS_OK;                      // (stream-ordered: every later use is on the library stream or behind it)
    }
    SplitEqView view() const {
        SplitEqView E;
        E.e_out = d_eout + (((size_t)1 << st.out_top) - 1);
        E.e_in = d_ein + (((size_t)1 << st.in_top) - 1);
        E.in_bits = (uint32_t)st.in_top;
        return E;
    }
    SplitEqView view_at(size_t out_top, size_t in_top) const {
        SplitEqView E;
        E.e_out = d_eout + (((size_t)1 << out_top) - 1);
        E.e_in = d_ein + (((size_t)1 << in_top) - 1);
        E.in_bits = (uint32_t)in_top;
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
    Fr* partials = nullptr;     // (ceil(T/2 / (RA_THREADS / 2)) + 1) * max(d, 2) Fr
    atlas::Chunk* tagged = nullptr;  // the tagged-row area behind them (tail_put): 48 bytes per partial value
    uint32_t* d_counter = nullptr;   // arrival counter of mail_tail (zero between launches)
    size_t K = 0;
    static bool tagged_off() { static const bool v = getenv("ATLAS_NO_TAGGED_ROWS") != nullptr; return v; }      // A-B: the arrival counter
    atlas::Chunk* tg() const { return tagged_off() ? nullptr : tagged; }

    // T_rounds: the length the ROUNDS start from when the buffers start later (lazy RaVirtual: buffers of T / 4, partial rows for T / 2 pairs)
    int alloc(size_t d_, size_t T, size_t k_min = 2, size_t T_rounds = 0) {     // K = width of a row of partial sums
        d = d_; len = T; K = d > k_min ? d : k_min;
        HIP_TRY(hipMalloc(&buf[0], d * T * sizeof(Fr)));
        HIP_TRY(hipMalloc(&buf[1], d * (T > 1 ? T / 2 : 1) * sizeof(Fr)));
        stride[0] = T; stride[1] = T > 1 ? T / 2 : 1;
        if (T_rounds > T) T = T_rounds;
        const size_t blocks = (T / 2 + RA_THREADS / 2 - 1) / (RA_THREADS / 2) + 1;       // a row per RA_THREADS / 2 pairs: the split product of d = 16 (ra.hip)
        const size_t cap = blocks * K > 8192 ? blocks * K : 8192;                                // room for the row-split launches of short instances (k_ra_bind_prod_f9: 512 rows of 16)
        HIP_TRY(hipMalloc(&partials, cap * sizeof(Fr) + cap * 3 * sizeof(atlas::Chunk)));
        tagged = reinterpret_cast<atlas::Chunk*>(partials + cap);
        HIP_TRY(hipMalloc(&d_counter, MAIL_TAIL_COUNTER_BYTES));
        HIP_TRY(hipMemsetAsync(d_counter, 0, MAIL_TAIL_COUNTER_BYTES, rt().stream));
        return ATLAS_OK;
    }
    // indices: d host rows of T int32 -> one device allocation (kept until the gather)
    int32_t* d_idx = nullptr;
    int upload_indices(const int32_t* const* H_indices) {
        HIP_TRY(hipMalloc(&d_idx, d * len * sizeof(int32_t)));
        for (size_t i = 0; i < d; i++)
            HIP_TRY(hipMemcpyAsync(d_idx + i * len, H_indices[i], len * sizeof(int32_t), hipMemcpyHostToDevice, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
        return ATLAS_OK;
    }
    // the same from the T lookup indices themselves: the d chunk rows are cut on the device (8 T bytes over PCIe
    // instead of 4 d T)
    const uint64_t* lk = nullptr;      // device-resident lookups read in place by the gather (k_ra_gather_lk): no index rows
    uint32_t lk_log = 0;
    int upload_lookups(const uint64_t* lookups, uint32_t log_k_chunk) {
        uint64_t* d_l = nullptr;
        hipPointerAttribute_t attr;
        const bool on_device = hipPointerGetAttributes(&attr, lookups) == hipSuccess && attr.type == hipMemoryTypeDevice;      // a device vector is cut in place
        (void)hipGetLastError();
        static const bool rows_always = getenv("ATLAS_RA_INDEX_ROWS") != nullptr;      // A-B
        if (on_device && !rows_always) { lk = lookups; lk_log = log_k_chunk; return ATLAS_OK; }
        HIP_TRY(hipMalloc(&d_idx, d * len * sizeof(int32_t)));
        hipError_t e = hipSuccess;
        if (!on_device) {
            HIP_TRY(hipMalloc(&d_l, len * sizeof(uint64_t)));
            e = hipMemcpyAsync(d_l, lookups, len * sizeof(uint64_t), hipMemcpyHostToDevice, rt().stream);
        }
        if (e == hipSuccess) {
            size_t gb = (len + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
            k_ra_chunk_indices<<<(unsigned)gb, RA_THREADS, 0, rt().stream>>>(on_device ? lookups : d_l, len, (uint32_t)d, log_k_chunk, d_idx);
            e = hipGetLastError();            // no synchronisation: d_l goes back to the pool, which hands it out in stream order
        }
        if (d_l) hipFree(d_l);
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "ra lookups upload", e);
        return ATLAS_OK;
    }
    // ra_i[j] = table_i[idx_i[j]] from device tables (d rows of f_stride Fr; f_stride 0 = shared table)
    int gather(const Fr* d_tables, uint32_t f_stride) {
        size_t gb = (len + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
        if (lk) {
            k_ra_gather_lk<<<dim3((unsigned)gb, (unsigned)d), RA_THREADS, 0, rt().stream>>>(lk, d_tables, f_stride, len, (uint32_t)d, lk_log, buf[0]);
            hipError_t e0 = hipGetLastError();
            if (e0 != hipSuccess) return fail(ATLAS_ENODEV, "ra gather", e0);
            cur = 0; stride[0] = len;
            return ATLAS_OK;
        }
        if (!d_idx) return fail(ATLAS_ESTATE, "ra gather: indices not uploaded");
        k_ra_gather<<<dim3((unsigned)gb, (unsigned)d), RA_THREADS, 0, rt().stream>>>(d_idx, d_tables, f_stride, len, buf[0]);
        hipError_t e = hipGetLastError();
        hipFree(d_idx); d_idx = nullptr;      // (pool: reused in stream order)
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "ra gather", e);
        cur = 0; stride[0] = len;
        return ATLAS_OK;
    }
    int bind(const atlas_u128_t& r) {
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode);
        const size_t half = len / 2;
        const int nxt = cur ^ 1;
        stride[nxt] = half;
        size_t gb = (half + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096; if (gb < 1) gb = 1;
        k_ra_bind<<<dim3((unsigned)gb, (unsigned)d), RA_THREADS, 0, rt().stream>>>(buf[cur], stride[cur], buf[nxt], stride[nxt], half,
                                                                               to_dev(rf), rt().challenge_mode == 0 ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "ra bind", e);
        cur = nxt; len = half;
        return ATLAS_OK;
    }
    // The sums and the final claims go straight into the pinned staging area (device-visible host memory): a
    // hipMemcpyAsync D2H of a few hundred bytes is a copy kernel of its own (4 us on the device, ~10 us end to end).
    int reduce_to_host(uint32_t n_partials, uint32_t k, H::Fr* out) {
        k_col_reduce<<<k, RA_THREADS, 0, rt().stream>>>(partials, n_partials, k, (Fr*)rt().h_pinned);
        HIP_TRY(hipStreamSynchronize(rt().stream));
        std::memcpy(out, rt().h_pinned, k * sizeof(Fr));
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) {
        if (len != 1) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        out.resize(d);
        k_col_reduce<<<(unsigned)d, RA_THREADS, 0, rt().stream>>>(buf[cur], 1u, 1u, (Fr*)rt().h_pinned, (uint32_t)stride[cur]);
        HIP_TRY(hipStreamSynchronize(rt().stream));
        std::memcpy(out.data(), rt().h_pinned, d * sizeof(Fr));
        return ATLAS_OK;
    }
    void release() { for (auto& b : buf) if (b) hipFree(b); if (partials) hipFree(partials); if (d_idx) hipFree(d_idx); if (d_counter) hipFree(d_counter); buf[0] = buf[1] = partials = nullptr; d_idx = nullptr; d_counter = nullptr; }
};


}  // namespace
