// Prefix-suffix Shout read-raf instances as host-stepped
// SumcheckInstanceProver objects over device-resident lookup indices.
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
// k_ps_q_lds (64-bit word sums per bin in LDS; k_ps_q for more than 256 bins); the per-round arithmetic over the 2^log_m entries —
// prefix evaluations, binding Q and the expanding table v_p — is host work.  The reference keeps the
// WordNoMSB suffix as u32, which is exact for N <= 32; N = 64 is refused here for that reason.
// spec of the suffix functions: mode 0/1 -> {1, suffix}; mode 2 (clamp, BOUND) additionally
// {HAZ_s, HAZ_s * lw_s, HAO_s, HAO_s * lw_s} with HAZ_s / HAO_s = "the suffix bits of significance >= BOUND are
// all zero / all one" and lw_s = the suffix bits below BOUND (suffixes/higher_all_zero.rs, hzero_mul_lword.rs,
// hone_mul_lword.rs)
#include <algorithm>
#include <chrono>

#include "ra_common.hip.h"

namespace {

// the NQ suffix-weighted copies of u for one lookup: val[q] = u * suffix_q(low `suffix_len` bits of k)
template <int NQ>
__device__ __forceinline__ void ps_entry_vals(uint64_t k, const Fr& u, uint32_t suffix_len, uint32_t bound, Fr val[NQ]) {
    const uint64_t smask = suffix_len >= 64 ? ~0ull : (((uint64_t)1 << suffix_len) - 1);
    const uint64_t sb = k & smask;
    val[0] = u;
    if constexpr (NQ == 4) {      // binary lookups: suffix = interleave(x, y) (utils/mod.rs:105-125 uninterleave_bits)
        uint64_t xb = (sb >> 1) & 0x5555555555555555ull, yb = sb & 0x5555555555555555ull;
        xb = (xb | (xb >> 1)) & 0x3333333333333333ull; xb = (xb | (xb >> 2)) & 0x0F0F0F0F0F0F0F0Full;
        xb = (xb | (xb >> 4)) & 0x00FF00FF00FF00FFull; xb = (xb | (xb >> 8)) & 0x0000FFFF0000FFFFull; xb = (xb | (xb >> 16)) & 0xFFFFFFFFull;
        yb = (yb | (yb >> 1)) & 0x3333333333333333ull; yb = (yb | (yb >> 2)) & 0x0F0F0F0F0F0F0F0Full;
        yb = (yb | (yb >> 4)) & 0x00FF00FF00FF00FFull; yb = (yb | (yb >> 8)) & 0x0000FFFF0000FFFFull; yb = (yb | (yb >> 16)) & 0xFFFFFFFFull;
        val[1] = xb < yb ? u : fe_zero();                                        // LessThan suffix
        val[2] = xb ? fr_mul(u, fr_from_i64((int64_t)xb)) : fe_zero();           // left operand of the suffix
        val[3] = yb ? fr_mul(u, fr_from_i64((int64_t)yb)) : fe_zero();           // right operand
    } else {
        val[1] = sb ? fr_mul(u, fr_from_i64((int64_t)sb)) : fe_zero();
        if constexpr (NQ == 3) {      // RightShiftSuffix: bits >> D, D passed in `bound` (suffixes/right_shift.rs:12-17)
            const uint64_t rs = bound >= 64 ? 0 : sb >> bound;
            val[2] = rs ? fr_mul(u, fr_from_i64((int64_t)rs)) : fe_zero();
        }
        if constexpr (NQ == 6) {
            bool haz = true, hao = true;
            uint64_t lw = sb;
            if (suffix_len > bound) {
                const uint64_t hi = sb >> bound, ones = (((uint64_t)1 << (suffix_len - bound)) - 1);
                haz = hi == 0; hao = hi == ones;
                lw = sb & (((uint64_t)1 << bound) - 1);
            }
            const Fr ul = ((haz || hao) && lw) ? fr_mul(u, fr_from_i64((int64_t)lw)) : fe_zero();
            val[2] = haz ? u : fe_zero(); val[3] = haz ? ul : fe_zero();
            val[4] = hao ? u : fe_zero(); val[5] = hao ? ul : fe_zero();
        }
    }
}

// Q tables of one phase, m = 2^log_m <= 256 bins: no binning at all.  Every lookup adds the eight 32-bit words of its NQ values into 64-bit
// LDS accumulators of its bin (ds_add_u64: words < 2^32, at most 2^25 lookups, so a sum stays below 2^57); a workgroup
// writes its accumulators to a slab of its own and k_ps_q_final adds the slabs and turns the word sums back into residues.
// Lanes that share a bin serialise on its accumulators (the high chunks of sign-extended values take two values in all),
// so the launch uses a workgroup per 256 lookups: one conflicted pass per wavefront instead of eight (164 -> ~10 us).
// (Summing such lanes across the wavefront first with 64-bit shuffles was slower than the conflicts: 48 x 6 shuffles.)  (A
// tiled variant in which every thread scanned its tile's entries for its bin took 168 us per phase at T = 2^16, NQ = 6,
// m = 256.)  The accumulators are NQ * m * 64 bytes of dynamic LDS (96 KB for the clamp lookup).
// v_prev != null: the finished phase's expanding table is folded into the products first (k_ps_scale fused: every lookup
// is visited by exactly one thread), prod[t] *= v_prev[(idx_t >> shift_prev) & (m - 1)].
// The sign classes of the shortcut below (PsSign): cls.only >= 0 keeps the lookups whose bit cls.sign_bit equals it (the class tables of the
// first mixed phase); cls.scal != null multiplies the product of a lookup by scal[its class] on the way (the factor the skipped phases owe).
struct PsClass { int only; uint32_t sign_bit; const Fr* scal; };
template <int NQ>
__global__ __launch_bounds__(RA_THREADS) void k_ps_q_lds(const uint64_t* __restrict__ idx, const Fr* __restrict__ u0,
                                                         Fr* __restrict__ prod, size_t T, uint32_t suffix_len, uint32_t m,
                                                         uint32_t bound, unsigned long long* __restrict__ acc /* [gridDim.x][m][NQ][8] */,
                                                         const Fr* __restrict__ v_prev, uint32_t shift_prev, PsClass cls) {
    extern __shared__ unsigned long long ps_sm[];
    const uint32_t n_words = m * NQ * 8;
    for (uint32_t w = threadIdx.x; w < n_words; w += RA_THREADS) ps_sm[w] = 0;
    __syncthreads();
    for (size_t t = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; t < T; t += (size_t)gridDim.x * RA_THREADS) {
        const uint64_t k = idx[t];
        const uint32_t sgn = (uint32_t)(k >> cls.sign_bit) & 1u;
        if (cls.only >= 0 && (int)sgn != cls.only) continue;
        const uint32_t b = (uint32_t)(k >> suffix_len) & (m - 1);
        Fr pr = fe_load(prod + t);
        if (v_prev) {
            pr = fr_mul(pr, fe_load(v_prev + ((uint32_t)(k >> shift_prev) & (m - 1))));
            if (cls.scal) pr = fr_mul(pr, fe_load(cls.scal + sgn));
            fe_store(prod + t, pr);
        }
        Fr val[NQ];
        ps_entry_vals<NQ>(k, fr_mul(fe_load(u0 + t), pr), suffix_len, bound, val);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if (fe_is_zero(val[q])) continue;
            unsigned long long* a = ps_sm + ((size_t)b * NQ + q) * 8;
#pragma unroll
            for (int w = 0; w < 8; w++) atomicAdd(&a[w], (unsigned long long)val[q].v[w]);
        }
    }
    __syncthreads();
    unsigned long long* slab = acc + (size_t)blockIdx.x * n_words;      // this workgroup's sums; k_ps_q_final adds the slabs
    for (uint32_t w = threadIdx.x; w < n_words; w += RA_THREADS) slab[w] = ps_sm[w];
}

// word sums -> canonical Montgomery residues: V = lo + hi 2^256, V mod p = lo * R * R^-1 + hi * R^2 * R^-1
// A workgroup takes 4 values (32 words): thread (word, part) adds every 8th slab, the 8 parts meet in LDS.
// pub.host_dst != null: the residues also go to pinned host memory and the last workgroup to arrive writes the tag the
// host waits for (pub.counter is zero on entry and left zero).
struct QPublish { Fr* host_dst; Chunk* tag_chunk; uint32_t tag; uint32_t* counter; };
__global__ __launch_bounds__(RA_THREADS) void k_ps_q_final(const unsigned long long* __restrict__ acc, uint32_t n_slabs, uint32_t n_vals, Fr* __restrict__ out,
                                                           QPublish pub) {
    __shared__ unsigned long long sm[8][32];
    const uint32_t word = threadIdx.x & 31u, part = threadIdx.x >> 5;
    const size_t n_words = (size_t)n_vals * 8, widx = (size_t)blockIdx.x * 32 + word;
    unsigned long long sum = 0;
    if (widx < n_words) {                              // four loads in flight per thread: the loop is latency-bound otherwise
        const unsigned long long* col = acc + widx;
        unsigned long long s1 = 0, s2 = 0, s3 = 0;
        uint32_t g2 = part;
        for (; g2 + 24 < n_slabs; g2 += 32) {
            const unsigned long long a0 = col[(size_t)g2 * n_words], a1 = col[(size_t)(g2 + 8) * n_words];
            const unsigned long long a2 = col[(size_t)(g2 + 16) * n_words], a3 = col[(size_t)(g2 + 24) * n_words];
            sum += a0; s1 += a1; s2 += a2; s3 += a3;
        }
        for (; g2 < n_slabs; g2 += 8) sum += col[(size_t)g2 * n_words];
        sum += s1 + s2 + s3;
    }
    sm[part][word] = sum;
    __syncthreads();
    const uint32_t i = blockIdx.x * 4 + threadIdx.x;
    if (threadIdx.x < 4 && i < n_vals) {
        Fr lo, hi, r2;
        unsigned long long c = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            unsigned long long s = c;
#pragma unroll
            for (int p2 = 0; p2 < 8; p2++) s += sm[p2][threadIdx.x * 8 + w];
            lo.v[w] = (uint32_t)s;
            c = s >> 32;
        }
#pragma unroll
        for (int w = 0; w < 8; w++) { hi.v[w] = 0; r2.v[w] = FrParams::r2(w); }
        hi.v[0] = (uint32_t)c; hi.v[1] = (uint32_t)(c >> 32);
        const Fr res = fr_add(fr_mul(lo, fr_one()), fr_mul(hi, r2));
        fe_store(out + i, res);
        if (pub.host_dst && !fe_is_zero(res)) fe_store(pub.host_dst + i, res);      // the host zeroed the box: a bin no lookup falls into costs no write
    }
    if (!pub.host_dst) return;
    // (The host zeroes the box before the launch and only non-zero residues cross the link: the phases over the sign extension of a
    // 64-bit lookup index fill 2 of 256 bins.)
    // (Publishing costs ~20 us per phase at 1536 values, whichever workgroups write: having only the last one to arrive
    // copy the residues out, with or without contiguous 16-byte stores, took 34-36 us against 30 — the host link takes
    // ~50 M device-initiated writes per second, tools/exp_channel2.hip.)
    __threadfence_system();                            // this workgroup's values are in host memory before it is counted
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(pub.counter, 1u);
        if (t == gridDim.x - 1) {
            *pub.counter = 0;
            __threadfence_system();
            ch_store_sys(pub.tag_chunk, ch_u32x4{n_vals, 0u, 0u, pub.tag});
        }
    }
}

// The same for at most 64 slabs (T <= 2^14, every lookup of the nanoGPT- and GPT-2-shaped graphs): 32 values per workgroup, a thread adds one word
// over all slabs — 48 workgroups instead of 384 behind the arrival counter of the publication.
__global__ __launch_bounds__(RA_THREADS) void k_ps_q_final_few(const unsigned long long* __restrict__ acc, uint32_t n_slabs, uint32_t n_vals, Fr* __restrict__ out,
                                                           QPublish pub) {
    __shared__ unsigned long long sm[RA_THREADS];
    const size_t n_words = (size_t)n_vals * 8, widx = (size_t)blockIdx.x * RA_THREADS + threadIdx.x;
    unsigned long long sum = 0;
    if (widx < n_words) {
        const unsigned long long* col = acc + widx;
        unsigned long long s1 = 0, s2 = 0, s3 = 0;
        uint32_t g2 = 0;
        for (; g2 + 7 < n_slabs; g2 += 8) {                // eight loads in flight: the loop is latency-bound
            const unsigned long long a0 = col[(size_t)g2 * n_words], a1 = col[(size_t)(g2 + 1) * n_words];
            const unsigned long long a2 = col[(size_t)(g2 + 2) * n_words], a3 = col[(size_t)(g2 + 3) * n_words];
            const unsigned long long a4 = col[(size_t)(g2 + 4) * n_words], a5 = col[(size_t)(g2 + 5) * n_words];
            const unsigned long long a6 = col[(size_t)(g2 + 6) * n_words], a7 = col[(size_t)(g2 + 7) * n_words];
            sum += a0 + a4; s1 += a1 + a5; s2 += a2 + a6; s3 += a3 + a7;
        }
        for (; g2 < n_slabs; g2++) sum += col[(size_t)g2 * n_words];
        sum += s1 + s2 + s3;
    }
    sm[threadIdx.x] = sum;
    __syncthreads();
    const uint32_t i = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x < 32 && i < n_vals) {
        Fr lo, hi, r2;
        unsigned long long c = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const unsigned long long s = c + sm[threadIdx.x * 8 + w];
            lo.v[w] = (uint32_t)s;
            c = s >> 32;
        }
#pragma unroll
        for (int w = 0; w < 8; w++) { hi.v[w] = 0; r2.v[w] = FrParams::r2(w); }
        hi.v[0] = (uint32_t)c; hi.v[1] = (uint32_t)(c >> 32);
        const Fr res = fr_add(fr_mul(lo, fr_one()), fr_mul(hi, r2));
        fe_store(out + i, res);
        if (pub.host_dst && !fe_is_zero(res)) fe_store(pub.host_dst + i, res);      // the host zeroed the box: a bin no lookup falls into costs no write
    }
    if (!pub.host_dst) return;
    // (The host zeroes the box before the launch and only non-zero residues cross the link: the phases over the sign extension of a
    // 64-bit lookup index fill 2 of 256 bins.)
    // (Publishing costs ~20 us per phase at 1536 values, whichever workgroups write: having only the last one to arrive
    // copy the residues out, with or without contiguous 16-byte stores, took 34-36 us against 30 — the host link takes
    // ~50 M device-initiated writes per second, tools/exp_channel2.hip.)
    __threadfence_system();                            // this workgroup's values are in host memory before it is counted
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(pub.counter, 1u);
        if (t == gridDim.x - 1) {
            *pub.counter = 0;
            __threadfence_system();
            ch_store_sys(pub.tag_chunk, ch_u32x4{n_vals, 0u, 0u, pub.tag});
        }
    }
}

// m > 256: one workgroup per (bin, slice of T), each filtering its slice for its bin
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
    for (size_t t = t0 + threadIdx.x; t < t1; t += RA_THREADS) {
        const uint64_t k = idx[t];
        if (((uint32_t)(k >> suffix_len) & m_mask) != y) continue;
        Fr val[NQ];
        ps_entry_vals<NQ>(k, fr_mul(fe_load(u0 + t), fe_load(prod + t)), suffix_len, bound, val);
#pragma unroll
        for (int q = 0; q < NQ; q++) acc[q] = fr_add(acc[q], val[q]);
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
                                                         uint32_t shift, uint32_t m_mask, Fr* __restrict__ prod, PsClass cls = PsClass{-1, 0u, nullptr}) {
    for (size_t t = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; t < T; t += (size_t)gridDim.x * RA_THREADS) {
        const uint64_t k = idx[t];
        Fr pr = fr_mul(fe_load(prod + t), fe_load(v + ((uint32_t)(k >> shift) & m_mask)));
        if (cls.scal) pr = fr_mul(pr, fe_load(cls.scal + ((uint32_t)(k >> cls.sign_bit) & 1u)));
        fe_store(prod + t, pr);
    }
}

// ---- the sign-extension shortcut ------------------------------------------------------------------------------------------
// The lookup index of a clamp / ReLU lookup is a sign-extended integer: `acc as u64` of an i64 accumulation that is a few thousand at most.
// Its leading chunks are 0x00.. or 0xFF.. for EVERY lookup, so through the first P phases ("pure" phases) the product of the expanding
// tables a lookup has collected depends on its sign class s alone:  u_t(p) = eq_t A_s(p),  A_s(p) = prod_{q < p} E_q[x_s]  (x_0 = 0,
// x_1 = m - 1; E_q[0] = prod (1 - r), E_q[m-1] = prod r over the phase's challenges).  Hence for p <= P
//     Q_p^f[y] = sum_s A_s(p) S_{p,s}^f[y],     S_{p,s}^f[y] = sum_{t in s, chunk_p(t) = y} eq_t f_p(t)
// and the S do not depend on any challenge: ONE pass at construction yields them — for a pure phase only bin x_s of class s is hit (NQ sums per
// class), for the first mixed phase P two class tables — and the host scales them by two scalars it keeps itself.  The P phase boundaries
// that would each have cost a pass over T and a hand-over of tables (~45 us) take no device work at all; the products on the device receive
// A_s(P) E_P[chunk_P] in the first pass that does run.  Same tables, hence the same round polynomials (exact arithmetic in another order).
//
// k_ps_sign_scan: per lookup the number of leading chunks that equal its sign pattern (minimum over T -> P) and, for the phases it is pure
// in, the word sums of its NQ suffix-weighted values per (phase, class).  The last workgroup turns the sums into residues and publishes
// them with the minimum.  Sums of a phase are complete iff every lookup is pure there, i.e. for p < the published minimum.
constexpr uint32_t PS_SIGN_PMAX = 7;
struct PsSignOut { unsigned long long* acc /* [PMAX][2][NQ][8], zeroed */; uint32_t* min_lz /* the COMPLEMENT of the minimum, as a maximum: zero on entry, so that one memset covers the whole scratch */; uint32_t* counter; Fr* host_dst; Chunk* tag_chunk; uint32_t tag; };
template <int NQ>
__global__ __launch_bounds__(RA_THREADS) void k_ps_sign_scan(const uint64_t* __restrict__ idx, const Fr* __restrict__ u0, size_t T, uint32_t N, uint32_t log_m,
                                                             uint32_t phases, uint32_t bound, PsSignOut O) {
    __shared__ unsigned long long sm[PS_SIGN_PMAX * 2 * NQ * 8];
    __shared__ uint32_t s_min, s_last;
    constexpr uint32_t n_words = PS_SIGN_PMAX * 2 * NQ * 8;
    for (uint32_t w = threadIdx.x; w < n_words; w += RA_THREADS) sm[w] = 0;
    if (threadIdx.x == 0) s_min = phases;
    __syncthreads();
    const uint32_t m_mask = (1u << log_m) - 1u;
    const uint32_t p_cap = phases - 1 < PS_SIGN_PMAX ? phases - 1 : PS_SIGN_PMAX;
    for (size_t t = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; t < T; t += (size_t)gridDim.x * RA_THREADS) {
        const uint64_t k = idx[t];
        const uint32_t sgn = (uint32_t)(k >> (N - 1)) & 1u, pat = sgn ? m_mask : 0u;
        uint32_t lz = 0;
        while (lz < phases && ((uint32_t)(k >> ((phases - 1 - lz) * log_m)) & m_mask) == pat) lz++;
        atomicMin(&s_min, lz);
        const Fr u = fe_load(u0 + t);
        const uint32_t pe = lz < p_cap ? lz : p_cap;
        for (uint32_t p = 0; p < pe; p++) {
            Fr val[NQ];
            ps_entry_vals<NQ>(k, u, (phases - 1 - p) * log_m, bound, val);
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                if (fe_is_zero(val[q])) continue;
                unsigned long long* a = sm + ((size_t)(p * 2 + sgn) * NQ + q) * 8;
#pragma unroll
                for (int w = 0; w < 8; w++) atomicAdd(&a[w], (unsigned long long)val[q].v[w]);
            }
        }
    }
    __syncthreads();
    for (uint32_t w = threadIdx.x; w < n_words; w += RA_THREADS) if (sm[w]) atomicAdd(&O.acc[w], sm[w]);
    if (threadIdx.x == 0) atomicMax(O.min_lz, ~s_min);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tk = atomicAdd(O.counter, 1u);
        s_last = tk == gridDim.x - 1;
        if (s_last) *O.counter = 0;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    constexpr uint32_t n_vals = PS_SIGN_PMAX * 2 * NQ;
    if (threadIdx.x < n_vals) {
        Fr lo, hi, r2;
        unsigned long long c = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const unsigned long long a = __hip_atomic_load(&O.acc[(size_t)threadIdx.x * 8 + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long s2 = c + (a & 0xffffffffull);
            lo.v[w] = (uint32_t)s2;
            c = (s2 >> 32) + (a >> 32);
        }
#pragma unroll
        for (int w = 0; w < 8; w++) { hi.v[w] = 0; r2.v[w] = FrParams::r2(w); }
        hi.v[0] = (uint32_t)c; hi.v[1] = (uint32_t)(c >> 32);
        fe_store(O.host_dst + threadIdx.x, fr_add(fr_mul(lo, fr_one()), fr_mul(hi, r2)));
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t mn = ~__hip_atomic_load(O.min_lz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ch_store_sys(O.tag_chunk, ch_u32x4{n_vals, mn, 0u, O.tag});
    }
}
// A_0 = prod (1 - r_i), A_1 = prod r_i over the n <= 64 challenges of the pure phases (slots of consecutive rounds, published long ago)
__global__ __launch_bounds__(64) void k_ps_class_scalars(const Chunk* slot0, uint32_t tag0, uint32_t n, uint32_t* abort_flag, int challenge_mode, Fr* __restrict__ out) {
    Fr x = fr_one(), y = fr_one();
    bool ok = true;
    if (threadIdx.x < n) {
        uint64_t lo = 0, hi = 0;
        ok = ch_poll_slot<true>(slot0 + (size_t)threadIdx.x * 4, tag0 + threadIdx.x, abort_flag, lo, hi);
        x = challenge_to_mont(lo, hi, challenge_mode);
        y = fr_sub(fr_one(), x);
    }
    if (__any(!ok)) return;
    for (int st = 1; st < 64; st <<= 1) { x = fr_mul(x, fe_shfl_xor(x, st)); y = fr_mul(y, fe_shfl_xor(y, st)); }
    if (threadIdx.x == 0) { fe_store(out, y); fe_store(out + 1, x); }
}

__global__ __launch_bounds__(RA_THREADS) void k_ps_fill_one(Fr* p, size_t T) {
    const Fr one = fr_one();
    for (size_t t = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; t < T; t += (size_t)gridDim.x * RA_THREADS) fe_store(p + t, one);
}

// sum_j E_out E_in ra[2 j]   (mod.rs:463-484)
__global__ __launch_bounds__(RA_THREADS) void k_ps_fold(const Fr* __restrict__ ra, SplitEqView E, size_t n_groups, Fr* __restrict__ partials, MailTail tail) {
    Fr acc[1];
    acc[0] = fe_zero();
    for (size_t j = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; j < n_groups; j += (size_t)gridDim.x * RA_THREADS)
        acc[0] = fr_add(acc[0], fr_mul(gse_weight(E, j), fe_load(ra + 2 * j)));
    block_reduce_put<1>(acc, partials, tail);
    mail_tail(partials, tail);
}
// a cycle round in ONE launch: wait for the challenge, bind ra (src of 4 n_groups values -> dst of 2 n_groups), fold the
// bound pairs, and the last workgroup mails the sum.  (Three launches per round cost the host thread ~10 us of enqueue.)
__global__ __launch_bounds__(RA_THREADS) void k_ps_bind_fold_ch(const Fr* __restrict__ src, Fr* __restrict__ dst, SplitEqView E, size_t n_groups,
                                                               Fr* __restrict__ partials, ChanIo io, int hi_only, MailTail tail) {
    Fr r;
    if (!io.challenge(r)) return;
    Fr acc[1];
    acc[0] = fe_zero();
    for (size_t j = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; j < n_groups; j += (size_t)gridDim.x * RA_THREADS) {
        const Fr b0 = bind_pair(fe_load(src + 4 * j), fe_load(src + 4 * j + 1), r, hi_only != 0);
        const Fr b1 = bind_pair(fe_load(src + 4 * j + 2), fe_load(src + 4 * j + 3), r, hi_only != 0);
        fe_store(dst + 2 * j, b0);
        fe_store(dst + 2 * j + 1, b1);
        acc[0] = fr_add(acc[0], fr_mul(gse_weight(E, j), b0));
    }
    block_reduce_put<1>(acc, partials, tail);
    mail_tail(partials, tail);
}

// ---- the cycle rounds of at most 2^PS_TAIL_LOG coefficients in ONE resident launch (what k_dot_tail2_f9 is for the dot product).
// A cycle round at these sizes is a round trip, not arithmetic: as a launch of its own it pays a kernel boundary, the fan-out of the
// challenge, the arrival counter and the second reduction stage of mail_tail — ~15 us; one workgroup that keeps the row in LDS, polls the
// round's slot itself and mails its one sum pays the two link crossings and ~3 us.  Tail round i is cycle round c0 + i: wait for the challenge
// of the round before (slot r_host + i, tag tag_r0 + i), bind the row LowToHigh (the first time from HBM), fold the even coefficients with the
// split-eq weights of that round, mail one canonical sum under tag_mail0 + i * tag_step at record i.  After the last round the final bind
// is mailed as record n_rounds (what k_rows_final_ch sends).  Same sums as k_ps_bind_fold_ch: exact arithmetic, another order.
constexpr uint32_t PS_TAIL_LOG = 11, PS_TAIL_THREADS = 1024, PS_TAIL_PER = (1u << PS_TAIL_LOG) / PS_TAIL_THREADS;      // (2^12 — 128 KB of LDS, one more resident round — measured: no change on the ReLU / Add / Mul nodes)
struct PsTailArgs {
    const Fr* src; uint32_t len_src;          // the row as the launch before left it: T >> (c0 - 1) coefficients
    const Fr* e_out; const Fr* e_in;          // the cached prefix tables (GseDev::d_eout, d_ein)
    uint8_t ot[16], it[16];                   // table tops of tail round i
    uint32_t n_rounds;
    Chunk* mail; const Chunk* r_host; uint32_t* abort_flag;
    uint32_t tag_mail0, tag_step, tag_r0;
    int challenge_mode, hi_only;
};
__global__ __launch_bounds__(PS_TAIL_THREADS) void k_ps_tail_ch(PsTailArgs A) {
    extern __shared__ __align__(16) unsigned char ps_tail_raw[];
    Fr* sv = reinterpret_cast<Fr*>(ps_tail_raw);
    __shared__ Fr red[PS_TAIL_THREADS / 64];
    __shared__ uint64_t s_ch[3];
    __shared__ uint32_t stage[9];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t len = A.len_src;
    for (uint32_t i = 0; i <= A.n_rounds; i++) {
        if (tid == 0) {
            uint64_t l = 0, h = 0;
            const bool ok = ch_poll_slot<true>(A.r_host + (size_t)i * CH_SLOT_CHUNKS, A.tag_r0 + i, A.abort_flag, l, h);
            s_ch[0] = l; s_ch[1] = h; s_ch[2] = ok ? 1 : 0;
        }
        __syncthreads();
        if (!s_ch[2]) return;
        const Fr r = challenge_to_mont(s_ch[0], s_ch[1], A.challenge_mode);
        const RoundIo io{A.mail, nullptr, nullptr, 1u, A.abort_flag, A.tag_mail0 + i * A.tag_step, 0u, 1u};
        if (i == A.n_rounds) {                                       // the final claim: the last two coefficients bound
            if (wave == 0) {
                Fr v = fe_zero();
                if (lane == 0) v = bind_pair(sv[0], sv[1], r, A.hi_only != 0);
                ch_mail_wave_fe(io, i * ch_stride(1), 1, v, stage);
            }
            return;
        }
        const uint32_t half = len / 2;                              // coefficients after this bind (<= 2^PS_TAIL_LOG)
        Fr b[PS_TAIL_PER];
#pragma unroll
        for (uint32_t u = 0; u < PS_TAIL_PER; u++) {
            const uint32_t j = tid + u * PS_TAIL_THREADS;
            if (j < half) b[u] = i == 0 ? bind_pair(fe_load(A.src + 2 * j), fe_load(A.src + 2 * j + 1), r, A.hi_only != 0)
                                        : bind_pair(sv[2 * j], sv[2 * j + 1], r, A.hi_only != 0);
        }
        __syncthreads();                                            // every pair has been read before its slot is overwritten
#pragma unroll
        for (uint32_t u = 0; u < PS_TAIL_PER; u++) { const uint32_t j = tid + u * PS_TAIL_THREADS; if (j < half) sv[j] = b[u]; }
        __syncthreads();
        const uint32_t n_groups = half / 2, in_bits = A.it[i];
        const Fr* eo = A.e_out + (((size_t)1 << A.ot[i]) - 1);
        const Fr* ei = A.e_in + (((size_t)1 << in_bits) - 1);
        Fr acc = fe_zero();
        for (uint32_t gq = tid; gq < n_groups; gq += PS_TAIL_THREADS)
            acc = fr_add(acc, fr_mul(fr_mul(fe_load(eo + (gq >> in_bits)), fe_load(ei + (gq & ((1u << in_bits) - 1u)))), sv[2 * gq]));
        acc = fr_wave_sum(acc);
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (wave == 0) {
            Fr t = lane < PS_TAIL_THREADS / 64 ? red[lane] : fe_zero();
            t = fr_wave_sum(t);
            ch_mail_wave_fe(io, i * ch_stride(1), 1, t, stage);
        }
        len = half;
    }
}

// ---- round-channel pieces (instance.hpp): the expanding table v of a phase is kept on the device, one challenge at a
// time, so that the phase boundary (scale the products by v, build the next Q) needs nothing from the host
// ExpandingTable::update, HighToLow (expanding_table.rs:62-88), for ALL challenges of a phase in one launch: the table
// starts as [1] in buf[0] and alternates between the two buffers, step k: dst[2i] = src[i] - r_k src[i], dst[2i+1] = r_k src[i].
// One workgroup; thread 0 polls the host slots of the phase's rounds in order (all but the last were published long
// ago).  A launch per round cost the host thread ~4.5 us each, on its critical path: the address rounds are host arithmetic.
struct PsSlots {
    const Chunk* host[12];
    uint32_t tag[12];
    uint32_t n;
    uint32_t* abort_flag;
    int challenge_mode;
};
__global__ __launch_bounds__(RA_THREADS) void k_ps_expand_all_ch(Fr* buf0, Fr* buf1, PsSlots S) {
    __shared__ uint64_t s_r[3];
    if (threadIdx.x == 0) fe_store(buf0, fr_one());
    __syncthreads();
    for (uint32_t k = 0; k < S.n; k++) {
        if (threadIdx.x == 0) {
            uint64_t lo = 0, hi = 0;
            const bool ok = ch_poll_slot<true>(S.host[k], S.tag[k], S.abort_flag, lo, hi);
            s_r[0] = lo; s_r[1] = hi; s_r[2] = ok ? 1 : 0;
        }
        __syncthreads();
        if (!s_r[2]) return;
        const Fr r = challenge_to_mont(s_r[0], s_r[1], S.challenge_mode);
        const Fr* src = (k & 1) ? buf1 : buf0;
        Fr* dst = (k & 1) ? buf0 : buf1;
        for (uint32_t i = threadIdx.x; i < (1u << k); i += RA_THREADS) {
            const Fr f = fe_load(src + i), hi = fr_mul(r, f);
            fe_store(dst + 2 * i + 1, hi);
            fe_store(dst + 2 * i, fr_sub(f, hi));
        }
        __threadfence_block();
        __syncthreads();
    }
}
// The same for m <= RA_THREADS (every lookup of the graph prover: 8-bit phases), where the phase boundary is on the critical path of
// the host's address rounds: lane k of the first wavefront polls slot k, so the host link is crossed once and not S.n times in turn
// (~2.5 us per crossing), and the table grows in LDS — the launch with one polling thread and the two global buffers took 25-38 us of
// the ~67 us between a phase's last challenge and the next phase's tables on the host (profiles/r03p_add_node_timeline.txt).
// The finished table lands where the stepwise launch leaves it: buf0 for an even number of challenges, buf1 for an odd one.
__global__ __launch_bounds__(RA_THREADS) void k_ps_expand_lds_ch(Fr* buf0, Fr* buf1, PsSlots S) {
    __shared__ uint64_t s_r[12][2];
    __shared__ uint32_t s_bad;
    __shared__ Fr tab[RA_THREADS];
    if (threadIdx.x == 0) { s_bad = 0; tab[0] = fr_one(); }
    __syncthreads();
    if (threadIdx.x < S.n) {
        uint64_t lo = 0, hi = 0;
        if (!ch_poll_slot<true>(S.host[threadIdx.x], S.tag[threadIdx.x], S.abort_flag, lo, hi)) atomicOr(&s_bad, 1u);
        s_r[threadIdx.x][0] = lo; s_r[threadIdx.x][1] = hi;
    }
    __syncthreads();
    if (s_bad) return;
    for (uint32_t k = 0; k < S.n; k++) {
        const Fr r = challenge_to_mont(s_r[k][0], s_r[k][1], S.challenge_mode);
        const bool act = threadIdx.x < (1u << k);
        Fr f = fe_zero(), hi = fe_zero();
        if (act) { f = tab[threadIdx.x]; hi = fr_mul(r, f); }
        __syncthreads();
        if (act) { tab[2 * threadIdx.x + 1] = hi; tab[2 * threadIdx.x] = fr_sub(f, hi); }
        __syncthreads();
    }
    Fr* dst = (S.n & 1) ? buf1 : buf0;
    if (threadIdx.x < (1u << S.n)) fe_store(dst + threadIdx.x, tab[threadIdx.x]);
}
// m > 256 (k_ps_q + k_col_reduce): the tables copied to pinned host memory by one workgroup, then the tag
__global__ __launch_bounds__(RA_THREADS) void k_ps_q_copy_out(const Fr* __restrict__ qsum, uint32_t n_vals, Fr* host_dst, Chunk* tag_chunk, uint32_t tag) {
    for (uint32_t i = threadIdx.x; i < n_vals; i += RA_THREADS) fe_store(host_dst + i, fe_load(qsum + i));
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) ch_store_sys(tag_chunk, ch_u32x4{n_vals, 0u, 0u, tag});
}

struct PsLookup : atlas_instance {
    size_t N = 0, log_m = 0, m = 0, log_T = 0, T = 0, round_next = 0, phases = 8;   // N = LOG_K
    int mode = 0;                         // 0 = ReLU + gamma * SignedIdentity (unary read-raf), 1 = Identity (range check), 2 = clamp family, 3 = UnsignedLessThan (binary), 4 = RightShift
    size_t bound = 0; bool symmetric = true;   // ClampBoundedTable<N, BOUND, SYMMETRIC> (lookup_tables/clamp.rs)
    size_t nq() const { return mode == 2 ? 6 : mode == 3 ? 4 : mode == 4 ? 3 : 2; }
    // The clamp family's four extra suffix functions (HAZ, HAZ lw, HAO, HAO lw) look at the suffix bits of significance >= BOUND: once a phase's
    // suffix lies below BOUND entirely (suffix_len <= bound: phases 4..7 of the 64-bit saturation lookup) there are none — HAZ = HAO = 1 and
    // lw = the suffix (ps_entry_vals) — so tables 2, 4 ARE table 0 and tables 3, 5 ARE table 1.  Such a phase builds, reduces, publishes, loads
    // and binds TWO tables instead of six (Q.size() == 2, Qk() maps the six names onto them): the same residues, a third of the work on
    // both sides of the link.  ATLAS_PS_NO_DUP=1 is the A-B.
    size_t nq_at(size_t phase) const {
        static const bool off = getenv("ATLAS_PS_NO_DUP") != nullptr;
        return (mode == 2 && !off && (phases - 1 - phase) * log_m <= bound) ? 2 : nq();
    }
    const std::vector<H::Fr>& Qk(size_t k) const { return Q.size() == 2 ? Q[k & 1] : Q[k]; }
    H::Fr gamma = H::zero();
    uint64_t* d_idx = nullptr;
    Fr *d_u0 = nullptr, *d_v = nullptr, *d_qpart = nullptr;
    RaRows rows;                          // row 0 = running product of the v tables = ra at the end
    GseDev eq;
    std::vector<std::vector<H::Fr>> Q;    // current phase's suffix tables (bound HighToLow): 0 = One, 1 = suffix, 2..5 clamp
    std::vector<H::Fr> v;                 // expanding table of the phase
    H::Fr haz_acc = H::one(), hao_acc = H::one(), lw_acc = H::zero();
    H::Fr rs_acc = H::zero();                                                                 // mode 4 (RightShift by `bound` bits)
    H::Fr lt_acc = H::zero(), eq_acc = H::one(), lop_acc = H::zero(), rop_acc = H::zero();   // mode 3 (UnsignedLessThan, binary)
    std::vector<H::Fr> r_addr;
    H::Fr word_acc = H::zero(), sid_acc = H::zero(), wv = H::zero();
    static constexpr unsigned SLICES = 64;
    // partial rows of a Q build with m > 256 (k_ps_q), at most 2^17 / m of them so that the rows stay a few MB; the same
    // area holds the word sums of k_ps_q_lds
    size_t q_rows_max() const { size_t r = ((size_t)1 << 17) / m; if (r > 2048) r = 2048; return r < SLICES ? SLICES : r; }

    bool idx_borrowed = false;            // d_idx is the caller's device vector (it outlives the instance: a node's witness), not a copy
    bool u0_borrowed = false;
    ~PsLookup() override { for (void* p : {(void*)(idx_borrowed ? nullptr : d_idx), (void*)(u0_borrowed ? nullptr : d_u0), (void*)d_v, (void*)d_qpart}) if (p) hipFree(p); rows.release(); eq.release(); }
    bool one_cycle = false;               // log_T == 0 held as two cycles, the second of weight zero (ps_new)
    size_t rounds() const override { return one_cycle ? N : N + log_T; }
    size_t degree() const override { return 2; }

    static H::Fr pow2(size_t k) {          // 2^k as a field element; k <= 64 from a table built once (the rounds ask for hundreds)
        static const std::vector<H::Fr> tab = [] { std::vector<H::Fr> t(65); t[0] = H::one(); for (size_t i = 1; i <= 64; i++) t[i] = H::add(t[i - 1], t[i - 1]); return t; }();
        if (k <= 64) return tab[k];
        H::Fr o = tab[64];
        for (size_t i = 64; i < k; i++) o = H::add(o, o);
        return o;
    }
    H::Fr rs_weight(size_t i) const { return (bound < N && i <= N - 1 - bound) ? pow2(N - 1 - i - bound) : H::zero(); }   // mode 4: bit i of k >> D
    H::Fr weight(size_t i) const { H::Fr w = pow2(N - 1 - i); return (i == 0 && mode != 1) ? H::sub(w, pow2(N)) : w; }   // (Signed)Identity coefficient of bit i

    Fr* qsum_ptr() const { return d_qpart + q_rows_max() * nq() * m; }
    int build_Q(size_t phase) {           // init_phase: Q tables of `phase` from the current products, fetched
        int rc = launch_Q(phase);
        if (rc) return rc;
        const size_t NQ = nq_at(phase);
        std::vector<H::Fr> q(NQ * m);
        HIP_TRY(hipMemcpyAsync(q.data(), qsum_ptr(), NQ * m * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
        load_Q(q.data(), NQ);
        v.assign(1, H::one());
        return ATLAS_OK;
    }
    // bins of the current tables that hold a non-zero value in any of them, ascending.  The phases over a sign extension (five of the
    // eight phases of a 64-bit lookup over 20-bit values) fill 1 or 2 bins of 256: the round's sums and the folds then walk this list
    // instead of the tables (same sums: exact arithmetic in another order).
    std::vector<uint32_t> nz;
    static constexpr size_t NZ_SPARSE = 8;
    void load_Q(const H::Fr* q, size_t NQ) {            // NQ = nq_at(the phase): the tables the device built
        Q.assign(NQ, std::vector<H::Fr>(m));
        nz.clear();
        for (size_t y = 0; y < m; y++) {
            bool any = false;
            for (size_t k = 0; k < NQ; k++) { Q[k][y] = q[NQ * y + k]; any = any || !H::detail::is_zero4(Q[k][y].l); }
            if (any) nz.push_back((uint32_t)y);
        }
    }
    // ---- the sign-extension shortcut (kernels above): phases 0 .. sgn_P - 1 are pure, phase sgn_P is the first the device works for
    struct QBox { const volatile atlas::Chunk* tagc = nullptr; const H::Fr* data = nullptr; uint32_t tag = 0; };
    size_t sgn_P = 0;                                   // 0: off
    std::vector<H::Fr> sgn_S;                           // [p][class][q], p < sgn_P
    H::Fr sgn_A[2] = {H::one(), H::one()};              // A_s(p) over the phases finished so far
    QBox sgn_box[2];                                    // class tables of phase sgn_P (published at construction)
    DevBuf sgn_scratch;                                 // word sums, minimum, counter; then the two class scalars
    const atlas::Chunk* sgn_slot0 = nullptr; uint32_t sgn_tag0 = 0;      // where the challenge of round 0 appears
    Fr* sgn_scal() const { return reinterpret_cast<Fr*>(static_cast<char*>(sgn_scratch.p) + 8192); }
    static bool wait_tag(const volatile atlas::Chunk* c, uint32_t tag) {
        const auto t0 = std::chrono::steady_clock::now();
        while (c->tag != tag) {
            for (int i = 0; i < 1024 && c->tag != tag; i++) __builtin_ia32_pause();
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > rt().chan.host_wait_s) return false;
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        return true;
    }
    // tables of a pure phase p < sgn_P: bin x_s of class s only
    void load_Q_pure(size_t p) {
        const size_t NQs = nq(), NQ = nq_at(p);             // the scan's sums are laid out for all nq() functions; tables 2.. of a phase below BOUND are copies
        Q.assign(NQ, std::vector<H::Fr>(m));
        nz.clear();
        for (int cl = 0; cl < 2; cl++) {
            const size_t y = cl ? m - 1 : 0;
            bool any = false;
            for (size_t k = 0; k < NQ; k++) {
                const H::Fr& sv = sgn_S[(p * 2 + cl) * NQs + k];
                if (H::detail::is_zero4(sv.l)) continue;
                Q[k][y] = H::mul(sgn_A[cl], sv); any = true;
            }
            if (any) nz.push_back((uint32_t)y);
        }
    }
    // tables of the first mixed phase: A_0 T_0 + A_1 T_1 over the published class tables
    int load_Q_mixed() {
        const size_t NQ = nq_at(sgn_P);
        for (int cl = 0; cl < 2; cl++)
            if (!wait_tag(sgn_box[cl].tagc, sgn_box[cl].tag)) { rt().chan.abort_dirty = true; return fail(ATLAS_ENODEV, "round channel: no class tables from the device"); }
        Q.assign(NQ, std::vector<H::Fr>(m));
        nz.clear();
        for (size_t y = 0; y < m; y++) {
            bool any = false;
            for (size_t k = 0; k < NQ; k++) {
                const H::Fr &a = sgn_box[0].data[NQ * y + k], &b = sgn_box[1].data[NQ * y + k];
                const bool za = H::detail::is_zero4(a.l), zb = H::detail::is_zero4(b.l);
                if (za && zb) continue;
                Q[k][y] = za ? H::mul(sgn_A[1], b) : zb ? H::mul(sgn_A[0], a) : H::add(H::mul(sgn_A[0], a), H::mul(sgn_A[1], b));
                any = true;
            }
            if (any) nz.push_back((uint32_t)y);
        }
        return ATLAS_OK;
    }
    // at construction (the caller holds rt().mu): scan, wait for the minimum, launch the class tables of phase P.  Leaves sgn_P = 0 and the
    // ordinary tables of phase 0 when no phase is pure.
    // `defer`: the launches only — the instance was built AHEAD of its use (NodePre: under the node's first wait) and finish_setup() runs when
    // its proof starts, by which time the scan has long answered; nothing here waits for the device.
    bool setup_pending = false;
    atlas::Chunk* scan_box = nullptr; uint32_t scan_tag = 0;
    QBox q0box;                                         // phase 0's tables of an instance without the shortcut, published instead of fetched
    int finish_setup() {
        if (!setup_pending) return ATLAS_OK;
        setup_pending = false;
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        if (scan_box) return sign_finish();
        if (!wait_tag(q0box.tagc, q0box.tag)) { rt().chan.abort_dirty = true; return fail(ATLAS_ENODEV, "round channel: no tables of phase 0 from the device"); }
        load_Q(q0box.data, nq_at(0));
        v.assign(1, H::one());
        return ATLAS_OK;
    }
    int deferred_Q0() {                                 // build_Q(0) without the wait
        const size_t n_vals = nq_at(0) * m;
        atlas::Chunk* box = rt().chan.alloc_long(2 * n_vals + 4);
        if (m <= RA_THREADS) std::memset(box + 4, 0, n_vals * sizeof(Fr));
        const uint32_t tg = rt().chan.tag();
        int rc = launch_Q(0, nullptr, 0, QPublish{reinterpret_cast<Fr*>(box + 4), box, tg, rows.d_counter});
        if (rc) return rc;
        q0box = QBox{box, reinterpret_cast<const H::Fr*>(box + 4), tg};
        setup_pending = true;
        return ATLAS_OK;
    }
    int sign_setup(bool defer = false) {
        const size_t NQ = nq(), n_vals = PS_SIGN_PMAX * 2 * NQ;
        hipError_t e = sgn_scratch.alloc(8192 + 2 * sizeof(Fr));
        if (e != hipSuccess) return fail(ATLAS_ENOMEM, "ps_shout: sign scratch", e);
        unsigned long long* acc = static_cast<unsigned long long*>(sgn_scratch.p);
        uint32_t* mn = reinterpret_cast<uint32_t*>(static_cast<char*>(sgn_scratch.p) + 7168);
        HIP_TRY(hipMemsetAsync(sgn_scratch.p, 0, 8192, rt().stream));
        const uint32_t ph = (uint32_t)phases;
        atlas::Chunk* box = rt().chan.alloc_long(2 * n_vals + 4);
        const uint32_t tag = rt().chan.tag();
        size_t gb = (T + RA_THREADS - 1) / RA_THREADS; if (gb > 256) gb = 256;
        const PsSignOut O{acc, mn, mn + 16, reinterpret_cast<Fr*>(box + 4), box, tag};
        if (NQ == 6) k_ps_sign_scan<6><<<(unsigned)gb, RA_THREADS, 0, rt().stream>>>(d_idx, d_u0, T, (uint32_t)N, (uint32_t)log_m, ph, (uint32_t)bound, O);
        else if (NQ == 4) k_ps_sign_scan<4><<<(unsigned)gb, RA_THREADS, 0, rt().stream>>>(d_idx, d_u0, T, (uint32_t)N, (uint32_t)log_m, ph, 0u, O);
        else k_ps_sign_scan<2><<<(unsigned)gb, RA_THREADS, 0, rt().stream>>>(d_idx, d_u0, T, (uint32_t)N, (uint32_t)log_m, ph, 0u, O);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) return fail(ATLAS_ENODEV, "ps_shout: sign scan", le);
        scan_box = box; scan_tag = tag;
        if (defer) { setup_pending = true; return ATLAS_OK; }
        return sign_finish();
    }
    int sign_finish() {
        const size_t NQ = nq();
        atlas::Chunk* box = scan_box;
        if (!wait_tag(box, scan_tag)) { rt().chan.abort_dirty = true; return fail(ATLAS_ENODEV, "round channel: no sign scan from the device"); }
        size_t P = box->d[1];
        if (P > phases - 1) P = phases - 1;
        if (P > PS_SIGN_PMAX) P = PS_SIGN_PMAX;
        if (P == 0) return build_Q(0);
        sgn_S.assign(reinterpret_cast<const H::Fr*>(box + 4), reinterpret_cast<const H::Fr*>(box + 4) + P * 2 * NQ);
        sgn_P = P;
        const size_t NQP = nq_at(P);
        for (int cl = 0; cl < 2; cl++) {                       // the class tables of phase P: needed 8 P rounds from now
            atlas::Chunk* tb = rt().chan.alloc_long(2 * NQP * m + 4);
            std::memset(tb + 4, 0, NQP * m * sizeof(Fr));
            const uint32_t tg = rt().chan.tag();
            int rc = launch_Q(P, nullptr, 0, QPublish{reinterpret_cast<Fr*>(tb + 4), tb, tg, rows.d_counter}, PsClass{cl, (uint32_t)(N - 1), nullptr});
            if (rc) return rc;
            sgn_box[cl] = QBox{tb, reinterpret_cast<const H::Fr*>(tb + 4), tg};
        }
        load_Q_pure(0);
        v.assign(1, H::one());
        return ATLAS_OK;
    }
    // a host-stepped caller (message / ingest with the device half) meets an instance built for the shortcut: the ordinary tables instead
    int sign_off() {
        if (!sgn_P) return ATLAS_OK;
        if (round_next != 0) return fail(ATLAS_ESTATE, "ps_shout: host-stepped call in the middle of a pipelined proof");
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        sgn_P = 0;
        return build_Q(0);
    }
    // the launches only: the tables end up at qsum_ptr().  v_prev: fold the finished phase's table into the products on the way
    // (m <= 256); pub: also publish the tables to the host
    int launch_Q(size_t phase, const Fr* v_prev = nullptr, uint32_t shift_prev = 0, QPublish pub = QPublish{nullptr, nullptr, 0, nullptr},
                 PsClass cls = PsClass{-1, 0u, nullptr}) {
        const uint32_t suffix_len = (uint32_t)((phases - 1 - phase) * log_m);
        const size_t NQ = nq_at(phase);
        Fr* d_qsum = qsum_ptr();
        if (m <= RA_THREADS) {
            unsigned long long* acc = (unsigned long long*)d_qpart;              // m * NQ * 8 word sums (the partial-row area is larger)
            const size_t lds = m * NQ * 64;
            size_t gb = (T + RA_THREADS - 1) / RA_THREADS; if (gb > 256) gb = 256;      // slabs: gb * lds bytes <= the partial-row area
#define PS_Q_LDS(NQv, BND)                                                                                                         \
            do {                                                                                                                   \
                static thread_local bool attr_set = false;                                                                                      \
                if (!attr_set) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ps_q_lds<NQv>), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * NQv * 64)); attr_set = true; } \
                k_ps_q_lds<NQv><<<(unsigned)gb, RA_THREADS, lds, rt().stream>>>(d_idx, d_u0, rows.buf[0], T, suffix_len, (uint32_t)m, (uint32_t)(BND), acc, v_prev, shift_prev, cls); \
            } while (0)
            if (NQ == 4) PS_Q_LDS(4, 0u);
            else if (NQ == 6) PS_Q_LDS(6, bound);
            else if (NQ == 3) PS_Q_LDS(3, bound);
            else PS_Q_LDS(2, 0u);
#undef PS_Q_LDS
            static const bool few_off = getenv("ATLAS_PS_NO_FEW") != nullptr;     // A-B
            if (gb <= 64 && !few_off) k_ps_q_final_few<<<(unsigned)((NQ * m + 31) / 32), RA_THREADS, 0, rt().stream>>>(acc, (uint32_t)gb, (uint32_t)(NQ * m), d_qsum, pub);
            else k_ps_q_final<<<(unsigned)((NQ * m + 3) / 4), RA_THREADS, 0, rt().stream>>>(acc, (uint32_t)gb, (uint32_t)(NQ * m), d_qsum, pub);
        } else {
            if (v_prev) { size_t gs = (T + RA_THREADS - 1) / RA_THREADS; if (gs > 4096) gs = 4096; k_ps_scale<<<(unsigned)gs, RA_THREADS, 0, rt().stream>>>(d_idx, v_prev, T, shift_prev, (uint32_t)(m - 1), rows.buf[0]); }
            if (NQ == 4) k_ps_q<4><<<dim3((unsigned)m, SLICES), RA_THREADS, 0, rt().stream>>>(d_idx, d_u0, rows.buf[0], T, suffix_len, (uint32_t)(m - 1), 0u, d_qpart);
            else if (NQ == 6) k_ps_q<6><<<dim3((unsigned)m, SLICES), RA_THREADS, 0, rt().stream>>>(d_idx, d_u0, rows.buf[0], T, suffix_len, (uint32_t)(m - 1), (uint32_t)bound, d_qpart);
            else if (NQ == 3) k_ps_q<3><<<dim3((unsigned)m, SLICES), RA_THREADS, 0, rt().stream>>>(d_idx, d_u0, rows.buf[0], T, suffix_len, (uint32_t)(m - 1), (uint32_t)bound, d_qpart);
            else k_ps_q<2><<<dim3((unsigned)m, SLICES), RA_THREADS, 0, rt().stream>>>(d_idx, d_u0, rows.buf[0], T, suffix_len, (uint32_t)(m - 1), 0u, d_qpart);
            k_col_reduce<<<(unsigned)(NQ * m), RA_THREADS, 0, rt().stream>>>(d_qpart, SLICES, (uint32_t)(NQ * m), d_qsum);
            if (pub.host_dst) k_ps_q_copy_out<<<1, RA_THREADS, 0, rt().stream>>>(d_qsum, (uint32_t)(NQ * m), pub.host_dst, pub.tag_chunk, pub.tag);
        }
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) return fail(ATLAS_ENODEV, "ps_shout: Q launch", le);
        return ATLAS_OK;
    }

    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "ps_shout: round out of order");
        if (setup_pending) { int rc = finish_setup(); if (rc) return rc; }
        if (sgn_P) { int rc = sign_off(); if (rc) return rc; }
        coeffs.assign(3, H::zero());
        if (round < N) return address_message(round, claim, coeffs);
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        const size_t n_groups = rows.len / 2;
        size_t blocks = (n_groups + RA_THREADS - 1) / RA_THREADS; if (blocks > 2048) blocks = 2048;
        k_ps_fold<<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(rows.buf[rows.cur], eq.view(), n_groups, rows.partials, MailTail{{}, nullptr, 0, 0});
        H::Fr s;
        int rc = rows.reduce_to_host((uint32_t)blocks, 1, &s);
        if (rc) return rc;
        H::gruen_deg2(eq.st.scalar, eq.st.w_cur(), H::mul(s, wv), claim, coeffs.data());
        return ATLAS_OK;
    }
    // prover_msg_read_checking (mod.rs:337-460): host arithmetic over the phase's 2^log_m-entry tables
    std::vector<H::Fr> sum_tmp;            // scratch of address_message's per-bit sums
    std::vector<uint32_t> pairs3;          // mode 3: the bin pairs of the round that hold something
    int address_message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) {
        {
            const size_t j = round, p = j / log_m, half = Q[0].size() / 2;
            const size_t suffix_len = N - (p + 1) * log_m;
            const size_t blen = log_m - (j % log_m) - 1;              // chunk bits still boolean after variable j
            const size_t hbits = mode == 2 ? N - bound : 0;           // variables 0..hbits-1 are the clamp's high bits
            const H::Fr sh = pow2(suffix_len), one = H::one();
            const H::Fr U = mode == 2 ? H::from_u64(((uint64_t)1 << bound) - 1) : H::zero();
            const H::Fr LC = symmetric ? H::add(H::add(U, U), one) : U;
            H::Fr ev[2];
            struct HalfSums { H::Fr s1, sb, ss, s2, sc1, z2, zl2, z3, o4, ol4, o5; };
            HalfSums hs[2];                                           // sums over the lower / upper half of the bins
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
                // Everything but the bin index b is the same for all bins of a round, and b enters through small integers that
                // are linear in its bits (b 2^suffix_len, the clamp's low-word bits, the shifted chunk).  So per table the
                // host takes plain sums over the bins, split by set bit of b (additions only), applies the bit weights once
                // (<= log_m multiplications), and — the tables being evaluated at 0 and at 2 = 2 * upper - lower — does so
                // for the lower and the upper half of the bins once instead of per evaluation point.
                if (mode != 3) {
                    if (ci == 0) {
                        size_t mhigh = 0;                               // bits of b that are clamp "high" variables
                        H::Fr w_b[16], w_lw[16], w_cv[16];              // weights of bit i of b (i = 0 least significant)
                        for (size_t i = 0; i < blen; i++) {
                            const size_t var = j + 1 + (blen - 1 - i);
                            w_b[i] = pow2(i);
                            const bool hb = mode == 2 && var < hbits;
                            if (hb) mhigh |= (size_t)1 << i;
                            w_lw[i] = (mode == 2 && !hb) ? pow2(N - 1 - var) : H::zero();
                            w_cv[i] = (mode == 4 && bound < 64 && i + suffix_len >= bound) ? pow2(i + suffix_len - bound) : H::zero();
                        }
                        // sum and weighted sum of table k over the bins [off, off + half) whose high bits are all `want`
                        auto sums = [&](size_t k, size_t off, int want, const H::Fr* w, H::Fr& tot, H::Fr& wtot) {
                            const auto keep = [&](size_t b) { return want < 0 || (want == 0 ? (b & mhigh) == 0 : (b & mhigh) == mhigh); };
                            wtot = H::zero();
                            if (nz.size() <= NZ_SPARSE) {              // a handful of non-empty bins: walk them
                                tot = H::zero();
                                H::Fr up[16]; uint32_t touched = 0;
                                for (uint32_t b : nz) {
                                    if (b < off || b >= off + half) continue;
                                    const size_t bb = b - off;
                                    if (!keep(bb)) continue;
                                    const H::Fr& val = Qk(k)[b];
                                    if (H::detail::is_zero4(val.l)) continue;
                                    tot = H::add(tot, val);
                                    if (!w) continue;
                                    for (size_t i = 0; i < blen; i++) if ((bb >> i) & 1) { up[i] = (touched >> i) & 1 ? H::add(up[i], val) : val; touched |= 1u << i; }
                                }
                                for (size_t i = 0; i < blen; i++) if ((touched >> i) & 1) wtot = H::add(wtot, H::mul(w[i], up[i]));
                                return;
                            }
                            if (!w) {
                                tot = H::zero();
                                for (size_t b = 0; b < half; b++) if (keep(b)) tot = H::add(tot, Qk(k)[off + b]);
                                return;
                            }
                            // per-bit sums by halving: the sum over the bins with bit i set is the sum of the upper half of
                            // the array once the bits above i have been folded away — 2 * half additions in all instead of
                            // half * (1 + blen / 2)
                            sum_tmp.resize(half);
                            for (size_t b = 0; b < half; b++) sum_tmp[b] = keep(b) ? Qk(k)[off + b] : H::zero();
                            for (size_t i = blen; i-- > 0;) {
                                const size_t h = (size_t)1 << i;
                                H::Fr up = H::zero();
                                for (size_t x = 0; x < h; x++) { up = H::add(up, sum_tmp[x + h]); sum_tmp[x] = H::add(sum_tmp[x], sum_tmp[x + h]); }
                                wtot = H::add(wtot, H::mul(w[i], up));
                            }
                            tot = sum_tmp[0];
                        };
                        for (int hf = 0; hf < 2; hf++) {
                            const size_t off = hf ? half : 0;
                            HalfSums& S = hs[hf];
                            H::Fr unused;
                            sums(0, off, -1, mode == 4 ? w_cv : w_b, S.s1, mode == 4 ? S.sc1 : S.sb);
                            if (mode == 4) { H::Fr t; sums(0, off, -1, w_b, t, S.sb); }
                            sums(1, off, -1, nullptr, S.ss, unused);
                            if (mode == 4) sums(2, off, -1, nullptr, S.s2, unused);
                            if (mode == 2 && Q.size() == 2 && mhigh == 0) {
                                // two tables (nq_at) and no clamp "high" variable left among the chunk's bits: every bin is kept by both filters, tables
                                // 2, 4 are table 0 and 3, 5 table 1, and the low-word weight of bin b is b 2^suffix_len — the sums above, reused
                                S.z2 = S.s1; S.zl2 = H::mul(sh, S.sb); S.z3 = S.ss;
                                if (symmetric) { S.o4 = S.s1; S.ol4 = S.zl2; S.o5 = S.ss; }
                            } else if (mode == 2) {
                                sums(2, off, 0, w_lw, S.z2, S.zl2);
                                sums(3, off, 0, nullptr, S.z3, unused);
                                if (symmetric) { sums(4, off, 1, w_lw, S.o4, S.ol4); sums(5, off, 1, nullptr, S.o5, unused); }
                            }
                        }
                    }
                    auto at = [&](H::Fr HalfSums::*f) { return ci ? H::sub(H::add(hs[1].*f, hs[1].*f), hs[0].*f) : hs[0].*f; };
                    const H::Fr s_q1 = at(&HalfSums::s1), s_qs = at(&HalfSums::ss);
                    const H::Fr bs_q1 = H::mul(sh, at(&HalfSums::sb));                     // sum (b 2^suffix_len) q1
                    const H::Fr idt = H::add(H::add(H::mul(sid_c, s_q1), bs_q1), s_qs);    // sum ((sid_c + bs) q1 + qs)
                    if (mode == 1) acc = idt;
                    else if (mode == 0) acc = H::add(H::mul(not_msb, H::add(H::add(H::mul(word_c, s_q1), bs_q1), s_qs)), H::mul(gamma, idt));
                    else if (mode == 4) {                              // RightShiftTable (right_shift.rs:54-58): (rs_c + chunk >> shift) q1 + q2
                        const H::Fr rs_c = H::add(rs_acc, H::mul(c, rs_weight(j)));
                        acc = H::add(H::add(H::add(H::mul(rs_c, s_q1), at(&HalfSums::sc1)), at(&HalfSums::s2)), H::mul(gamma, idt));
                    } else {      // clamp (clamp.rs:84-109): sum_z haz_c ((lw_c + lwb - U) q2 + q3) + sum_o hao_c ((lw_c + lwb) q4 + q5)
                        H::Fr val = H::mul(H::sub(U, H::mul(msb, LC)), s_q1);
                        val = H::add(val, H::mul(haz_c, H::add(H::add(H::mul(H::sub(lw_c, U), at(&HalfSums::z2)), at(&HalfSums::zl2)), at(&HalfSums::z3))));
                        if (symmetric) val = H::add(val, H::mul(hao_c, H::add(H::add(H::mul(lw_c, at(&HalfSums::o4)), at(&HalfSums::ol4)), at(&HalfSums::o5))));
                        acc = H::add(val, H::mul(gamma, idt));
                    }
                } else {
                // an empty pair of bins adds nothing: in the sparse phases the loop walks the handful of pairs that hold something
                const bool sparse3 = nz.size() <= 4 * NZ_SPARSE;
                if (sparse3 && ci == 0) {
                    pairs3.clear();
                    for (uint32_t x : nz) pairs3.push_back(x >= half ? x - (uint32_t)half : x);
                    std::sort(pairs3.begin(), pairs3.end());
                    pairs3.erase(std::unique(pairs3.begin(), pairs3.end()), pairs3.end());
                }
                const size_t n_iter = sparse3 ? pairs3.size() : half;
                for (size_t bi = 0; bi < n_iter; bi++) {
                    const size_t b = sparse3 ? pairs3[bi] : bi;
                    auto qv = [&](size_t k) { return ci ? H::sub(H::add(Q[k][b + half], Q[k][b + half]), Q[k][b]) : Q[k][b]; };
                    // UnsignedLessThan over interleaved (x, y) pairs + gamma * Left + gamma^2 * Right
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
                }
                }
                ev[ci] = acc;
            }
            H::unipoly_from_evals_and_hint(claim, ev, 2, coeffs.data());
            return ATLAS_OK;
        }
    }

    int ingest(const atlas_u128_t& r, size_t round) override { return ingest_impl(r, round, true); }
    // with_device = false: the host half only (round channel: the launches were enqueued ahead and take r from its slot)
    int ingest_impl(const atlas_u128_t& r, size_t round, bool with_device) {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "ps_shout: round out of order");
        if (setup_pending) { int rc = finish_setup(); if (rc) return rc; }
        if (with_device && sgn_P) { int rc = sign_off(); if (rc) return rc; }
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode);
        if (round < N) {
            const size_t j = round, p = j / log_m;
            if (p < sgn_P) { sgn_A[0] = H::mul(sgn_A[0], H::sub(H::one(), rf)); sgn_A[1] = H::mul(sgn_A[1], rf); }      // a pure phase: E_p[0], E_p[m - 1] factor by factor
            const size_t half = Q[0].size() / 2;
            if (nz.size() <= 4 * NZ_SPARSE) {                          // few non-empty bins: their pairs only
                std::vector<uint32_t> low;
                low.reserve(nz.size());
                for (uint32_t b : nz) low.push_back(b >= half ? b - (uint32_t)half : b);
                std::sort(low.begin(), low.end());
                low.erase(std::unique(low.begin(), low.end()), low.end());
                for (auto& q : Q) {
                    for (uint32_t i : low) q[i] = H::add(q[i], H::mul_challenge(rf, H::sub(q[i + half], q[i])));
                    q.resize(half);
                }
                nz.swap(low);
            } else {
                for (auto& q : Q) {                                   // suffix polys bind HighToLow
                    for (size_t i = 0; i < half; i++) {
                        if (H::detail::is_zero4(q[i].l) && H::detail::is_zero4(q[i + half].l)) continue;       // an empty pair of bins stays empty
                        q[i] = H::add(q[i], H::mul_challenge(rf, H::sub(q[i + half], q[i])));   // (two CIOS steps for a 128-bit challenge)
                    }
                    q.resize(half);
                }
                std::vector<uint32_t> low;                           // the list: indices above `half` fold onto their partners
                low.reserve(nz.size());
                for (uint32_t b : nz) low.push_back(b >= half ? b - (uint32_t)half : b);
                std::sort(low.begin(), low.end());
                low.erase(std::unique(low.begin(), low.end()), low.end());
                nz.swap(low);
            }
            if (with_device) {
                std::vector<H::Fr> nv(2 * v.size());                  // ExpandingTable::update, HighToLow
                for (size_t i = 0; i < v.size(); i++) { nv[2 * i + 1] = H::mul(rf, v[i]); nv[2 * i] = H::sub(v[i], nv[2 * i + 1]); }
                v.swap(nv);
            }
            if (j >= 1) word_acc = H::add(word_acc, H::mul(rf, pow2(N - 1 - j)));
            sid_acc = H::add(sid_acc, H::mul(rf, weight(j)));
            if (mode == 4) rs_acc = H::add(rs_acc, H::mul(rf, rs_weight(j)));
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
            if ((j + 1) % log_m == 0 && with_device) {                // phase boundary: fold v_p into the products
                std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
                HIP_TRY(hipMemcpyAsync(d_v, v.data(), m * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
                size_t gb = (T + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
                k_ps_scale<<<(unsigned)gb, RA_THREADS, 0, rt().stream>>>(d_idx, d_v, T, (uint32_t)((phases - 1 - p) * log_m), (uint32_t)(m - 1), rows.buf[0]);
                if (p != phases - 1) { int rc = build_Q(p + 1); if (rc) return rc; }       // (ends with a stream synchronize: v may change after it)
                else HIP_TRY(hipStreamSynchronize(rt().stream));
            }
            if (j + 1 == N) {
                // val = Val~(r_address), raf_val = gamma * SId~(r_address)   (mod.rs:523-548)
                const H::Fr val = H::mul(H::sub(H::one(), r_addr[0]), word_acc);
                wv = mode == 1 ? sid_acc : H::add(val, H::mul(gamma, sid_acc));            // identity_range_check.rs:377-380
                if (mode == 4) wv = H::add(rs_acc, H::mul(gamma, sid_acc));
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
        } else if (with_device) {
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            int rc = rows.bind(r);
            if (rc) return rc;
            eq.st.bind(rf);
        } else {
            eq.st.bind(rf);
            const size_t c = round - N;
            rows.cur = (int)((c + 1) & 1); rows.len = T >> (c + 1); rows.stride[rows.cur] = rows.len;
        }
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override {
        if (round_next != rounds()) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        if (have_finals) { out = mailed_finals; return ATLAS_OK; }
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        if (one_cycle) rows.len = 1;                                  // ra of the one lookup: entry 0 of the products (the second entry is the weight-zero cycle)
        return rows.finals(out);
    }

    // ---- round-channel stepping (instance.hpp).  Address rounds are host arithmetic and launch nothing, except the first
    // round of a phase: the finished phase's expanding table is rebuilt on the device from its challenge slots (d_v holds two
    // tables of m entries, used alternately), the products are scaled by it and the new phase's Q is built; Q travels to the
    // host through pinned memory (k_ps_q_final's QPublish) and is picked up by that round's finish().  Cycle round c has ra in buf[c & 1].
    std::vector<QBox> qbox;
    PsSlots slots{};
    bool have_finals = false;
    std::vector<H::Fr> mailed_finals;
    bool pipelined() const override { return !one_cycle && log_m <= 11 && N + log_T <= atlas_rt::Channel::RING / 2; }
    bool wide_wait(size_t round) const override {                 // address rounds and the first cycle round: only the one-workgroup table rebuild waits
        if (round <= N || round >= rounds()) return false;
        if (tail_c0() && round - N >= tail_c0()) return false;  // the resident tail: one workgroup
        const size_t n_groups = (T >> (round - N)) / 2;
        return (n_groups + RA_THREADS - 1) / RA_THREADS > WIDE_WAIT_WGS;
    }
    bool silent_round(size_t round) const override {
        if (round < N) return round % log_m != 0 || round == 0 || (sgn_P && round / log_m - 1 < sgn_P);
        return tail_c0() && round - N > tail_c0();
    }
    int enqueue(size_t round, const atlas::RoundIo& io, bool bind_prev, atlas_mail_ref& mail) override {
        if (round >= rounds() || (round > 0) != bind_prev) return fail(ATLAS_ESTATE, "ps_shout: enqueue out of order");
        if (setup_pending) { int rc = finish_setup(); if (rc) return rc; }      // (built ahead: the scan's answer, the class tables' launches)
        const ChanIo cio{io, rt().challenge_mode};
        mail.base = io.mail; mail.blocks = 0; mail.n_vals = 0; mail.radix = 32; mail.shl = 0;
        Fr* vt[2] = {d_v, d_v + m};
        size_t gbT = (T + RA_THREADS - 1) / RA_THREADS; if (gbT > 4096) gbT = 4096;
        if (round == 0) { qbox.assign(phases, QBox{}); n_lanes = io.tag_step; }
        if (round >= 1 && round <= N) {                        // remember where the challenge of round - 1 will appear
            const size_t pw = (round - 1) % log_m;
            slots.host[pw] = io.r_host; slots.tag[pw] = io.tag_r;
            if (round == 1) { sgn_slot0 = io.r_host; sgn_tag0 = io.tag_r; }
        }
        if (round < N && round % log_m != 0) return ATLAS_OK;  // an address round inside a phase launches nothing (56 of the 64 rounds of a 64-bit lookup: no runtime call at all)
        if (sgn_P && round >= 1 && round <= N && round % log_m == 0 && round / log_m - 1 < sgn_P) return ATLAS_OK;      // a pure phase ends: nothing for the device (PsSign)
        PsClass cls{-1, (uint32_t)(N - 1), nullptr};
        if (sgn_P && round >= 1 && round <= N && round % log_m == 0 && round / log_m - 1 == sgn_P) {
            // the first boundary the device works at: the products still miss the class factor A_s(P) of the pure phases
            k_ps_class_scalars<<<1, 64, 0, rt().stream>>>(sgn_slot0, sgn_tag0, (uint32_t)(sgn_P * log_m), io.abort_flag, rt().challenge_mode, sgn_scal());
            cls.scal = sgn_scal();
        }
        if (round >= 1 && round <= N && round % log_m == 0) {  // a phase is complete: its table, folded into the products
            const size_t p_done = round / log_m - 1;
            slots.n = (uint32_t)log_m; slots.abort_flag = io.abort_flag; slots.challenge_mode = rt().challenge_mode;
            if (m <= RA_THREADS) k_ps_expand_lds_ch<<<1, RA_THREADS, 0, rt().stream>>>(vt[0], vt[1], slots);
            else k_ps_expand_all_ch<<<1, RA_THREADS, 0, rt().stream>>>(vt[0], vt[1], slots);
            const uint32_t shift_done = (uint32_t)((phases - 1 - p_done) * log_m);
            if (round < N) {                                   // ... folded into the products while the Q of the phase that starts is built
                const size_t p = round / log_m, n_vals = nq_at(p) * m;
                atlas::Chunk* box = rt().chan.alloc(2 * n_vals + 4);
                if (m <= RA_THREADS) std::memset(box + 4, 0, n_vals * sizeof(Fr));      // k_ps_q_final publishes the non-zero residues only
                int rc = launch_Q(p, vt[log_m & 1], shift_done, QPublish{reinterpret_cast<Fr*>(box + 4), box, io.tag_mail, rows.d_counter}, cls);
                if (rc) return rc;
                qbox[p] = QBox{box, reinterpret_cast<const H::Fr*>(box + 4), io.tag_mail};
            } else {
                k_ps_scale<<<(unsigned)gbT, RA_THREADS, 0, rt().stream>>>(d_idx, vt[log_m & 1], T, shift_done, (uint32_t)(m - 1), rows.buf[0], cls);
            }
        }
        if (round >= N && tail_c0() && round - N >= tail_c0()) {        // the resident tail (k_ps_tail_ch): launched with its first round
            const size_t c0 = tail_c0(), c = round - N;
            if (c == c0) {
                PsTailArgs A;
                A.src = rows.buf[(c0 - 1) & 1]; A.len_src = (uint32_t)(T >> (c0 - 1));
                A.e_out = eq.d_eout; A.e_in = eq.d_ein;
                A.n_rounds = (uint32_t)(log_T - c0);
                for (size_t i = 0; i < A.n_rounds; i++) { size_t ot, it; eq.st.tops_after(c0 + i, ot, it); A.ot[i] = (uint8_t)ot; A.it[i] = (uint8_t)it; }
                tail_mail = rt().chan.alloc_long((A.n_rounds + 2) * atlas::ch_stride(1));       // (not io.mail: the per-round areas are recycled while the tail lives)
                A.mail = tail_mail; A.r_host = io.r_host; A.abort_flag = io.abort_flag;
                A.tag_mail0 = io.tag_mail; A.tag_step = io.tag_step; A.tag_r0 = io.tag_r;
                A.challenge_mode = rt().challenge_mode; A.hi_only = rt().challenge_mode == 0 ? 1 : 0;
                static thread_local bool attr_set = false;
                if (!attr_set) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ps_tail_ch), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(Fr) << PS_TAIL_LOG))); attr_set = true; }
                k_ps_tail_ch<<<1, PS_TAIL_THREADS, sizeof(Fr) << PS_TAIL_LOG, rt().stream>>>(A);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) return fail(ATLAS_ENODEV, "ps_shout: tail launch", e);
            }
            mail.base = tail_mail + (c - c0) * atlas::ch_stride(1); mail.blocks = 1; mail.n_vals = 1;
            return ATLAS_OK;
        }
        if (round >= N) {
            const size_t c = round - N, len = T >> c, n_groups = len / 2;
            size_t blocks = (n_groups + RA_THREADS - 1) / RA_THREADS; if (blocks > 2048) blocks = 2048;
            size_t ot, it;
            eq.st.tops_after(c, ot, it);
            const MailTail tail{io, rows.d_counter, (uint32_t)blocks, 1u, rows.tg()};
            if (c == 0) k_ps_fold<<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(rows.buf[0], eq.view_at(ot, it), n_groups, rows.partials, tail);
            else k_ps_bind_fold_ch<<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(rows.buf[(c - 1) & 1], rows.buf[c & 1], eq.view_at(ot, it), n_groups, rows.partials, cio,
                                                                              rt().challenge_mode == 0 ? 1 : 0, tail);
            mail.blocks = 1; mail.n_vals = 1;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "ps_shout: launch", e);
        return ATLAS_OK;
    }
    int finish(size_t round, const H::Fr& claim, const H::Fr* s, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "ps_shout: round out of order");
        coeffs.assign(3, H::zero());
        if (round >= N) {
            H::gruen_deg2(eq.st.scalar, eq.st.w_cur(), H::mul(s[0], wv), claim, coeffs.data());
            return ATLAS_OK;
        }
        if (sgn_P && round > 0 && round % log_m == 0 && round / log_m <= sgn_P) {      // a phase whose tables are class sums times two scalars (PsSign)
            PROF("ps_shout: tables of a sign phase");
            if (round / log_m < sgn_P) load_Q_pure(round / log_m);
            else { int rc = load_Q_mixed(); if (rc) return rc; }
        } else if (round > 0 && round % log_m == 0) {          // first round of a phase: its Q tables arrive through pinned memory
            const QBox& B = qbox[round / log_m];
            if (!B.tagc) return fail(ATLAS_ESTATE, "ps_shout: Q of the phase was not enqueued");
            const auto t0 = std::chrono::steady_clock::now();
            PROF("ps_shout: wait for the phase's Q tables + load");
            while (B.tagc->tag != B.tag) {
                for (int i = 0; i < 1024 && B.tagc->tag != B.tag; i++) __builtin_ia32_pause();
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > rt().chan.host_wait_s) { rt().chan.abort_dirty = true; return fail(ATLAS_ENODEV, "round channel: no Q tables from the device"); }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);          // the residues are read through a plain pointer: not before the tag (the device wrote them, fenced, then the tag)
            load_Q(B.data, nq_at(round / log_m));
            if (ps_trace()) t_qwait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        if (!ps_trace()) return address_message(round, claim, coeffs);
        const auto t1 = std::chrono::steady_clock::now();
        const int rc = address_message(round, claim, coeffs);
        t_addr += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        if (round + 1 == N) fprintf(stderr, "[atlas trace] ps_shout address rounds: wait for Q + load %.1f us, messages %.1f us\n", 1e6 * t_qwait, 1e6 * t_addr);
        return rc;
    }
    double t_qwait = 0, t_addr = 0;
    static bool ps_trace() { static const bool on = getenv("ATLAS_TRACE") != nullptr; return on; }
    int host_ingest(const atlas_u128_t& r, size_t round) override { return ingest_impl(r, round, false); }
    // ---- the resident tail: cycle rounds c0 .. log_T - 1 and the final bind in one launch; 0 = none (ATLAS_PS_NO_TAIL=1 is the A-B)
    atlas::Chunk* tail_mail = nullptr;
    size_t tail_c0() const {
        static const bool off = getenv("ATLAS_PS_NO_TAIL") != nullptr;
        // Only as the ONE lane of its proof: a resident launch waits for challenges that depend on the OTHER lanes' later launches, and HIP
        // multiplexes streams onto a few hardware queues — a lane whose stream shares the tail's queue would have those launches queued
        // behind it (seen as a 2 s stall in the Sin / Cos flow, whose shift lookup is batched with the table read).
        if (off || log_T < 2 || one_cycle || n_lanes != 1) return 0;
        return log_T > PS_TAIL_LOG + 1 ? log_T - PS_TAIL_LOG : 1;      // the first bind of the tail leaves at most 2^PS_TAIL_LOG coefficients
    }
    uint32_t n_lanes = 0;                                           // lanes of the proof this instance runs in (RoundIo::tag_step of its first enqueue)
    int enqueue_finals(const atlas::RoundIo& io, atlas_mail_ref& mail) override {
        if (tail_c0()) {                                            // the tail mails the final claim as its last record
            if (!tail_mail) return fail(ATLAS_ESTATE, "ps_shout: finals before the tail");
            mail.base = tail_mail + (log_T - tail_c0()) * atlas::ch_stride(1); mail.blocks = 1; mail.n_vals = 1; mail.radix = 32; mail.shl = 0;
            return ATLAS_OK;
        }
        k_rows_final_ch<<<1, 64, 0, rt().stream>>>(rows.buf[(log_T - 1) & 1], T >> (log_T - 1), 1u, ChanIo{io, rt().challenge_mode}, rt().challenge_mode == 0 ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "ps_shout: launch", e);
        mail.base = io.mail; mail.blocks = 1; mail.n_vals = 1; mail.radix = 32; mail.shl = 0;
        return ATLAS_OK;
    }
    int set_finals(const H::Fr* vals, size_t n) override {
        if (n != 1) return fail(ATLAS_EINVAL, "ps_shout: final claims");
        mailed_finals.assign(vals, vals + 1); have_finals = true;
        return ATLAS_OK;
    }
};

// compute_lookup_indices_from_operands (jolt-atlas-core/src/utils/mod.rs:43-122): `value as u32 as u64`, or
// interleave_bits(left as u32, right as u32) (joltworks/src/utils/mod.rs:146-164) for the binary lookups
__device__ __forceinline__ uint64_t ps_spread_bits(uint32_t v) {
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
__global__ __launch_bounds__(RA_THREADS) void k_lookup_indices(const int32_t* __restrict__ left, const int32_t* __restrict__ right, size_t n,
                                                               uint64_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * RA_THREADS)
        out[i] = right ? ((ps_spread_bits((uint32_t)left[i]) << 1) | ps_spread_bits((uint32_t)right[i])) : (uint64_t)(uint32_t)left[i];
}

}  // namespace

extern "C" {

static int ps_new(const uint64_t* lookup_indices, size_t log_T, size_t log_K, size_t phases, int mode, const atlas_fr_t* r_node_output,
                  const atlas_fr_t* gamma, atlas_instance_t* out, size_t bound = 0, bool symmetric = true, atlas_poly_t eq_shared = nullptr, bool defer = false) {
    // log_T == 0 — a read-raf instance WITHOUT cycle variables (ps_shout/mod.rs:419-446 at T = 1: log_K address rounds, then the ra value of the one
    // lookup).  Held as TWO cycles of which the second has weight u = 0 and index 0: every sum over cycles of the address rounds is the one-cycle
    // sum, no kernel meets a length of one; the instance reports log_K rounds, is stepped by the host (the fold of the last phase's table
    // happens in its ingest) and its final claim is entry 0 of the products.
    const bool one_cycle = log_T == 0;
    if (one_cycle) log_T = 1;
    atlas_poly_t E = nullptr;
    if (eq_shared && (one_cycle || eq_shared->is_i32 || eq_shared->len != ((size_t)1 << log_T))) return fail(ATLAS_EINVAL, "ps_shout_new: shared eq table of the wrong length");
    int rc = eq_shared ? ATLAS_OK : atlas_eq_evals(r_node_output, one_cycle ? 0 : log_T, nullptr, &E);      // u_evals = EqPolynomial::evals(r_node_output), mod.rs:234
    if (rc) return rc;
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    PsLookup* P = new PsLookup();
    P->one_cycle = one_cycle;
    P->N = log_K; P->phases = phases; P->mode = mode; P->bound = bound; P->symmetric = symmetric;
    P->log_m = log_K / phases; P->m = (size_t)1 << P->log_m; P->log_T = log_T; P->T = (size_t)1 << log_T;
    if (gamma) std::memcpy(&P->gamma, gamma, 32);
    const size_t T = P->T, m = P->m;
    hipError_t e = hipSuccess;
    if (one_cycle) {                                                 // u = (1, 0)
        e = hipMalloc(&P->d_u0, 2 * sizeof(Fr));
        if (e == hipSuccess) e = hipMemsetAsync(P->d_u0, 0, 2 * sizeof(Fr), rt().stream);
        if (e == hipSuccess) e = hipMemcpyAsync(P->d_u0, E->d, sizeof(Fr), hipMemcpyDeviceToDevice, rt().stream);
        atlas_poly_free(E);
    } else if (eq_shared) { P->d_u0 = (Fr*)eq_shared->d; P->u0_borrowed = true; }      // the caller's table (it outlives the instance)
    else { P->d_u0 = (Fr*)E->d; delete E; }                          // keep the table, drop the handle
    {   // device-resident indices (the graph prover's witness vectors, alive until the graph is freed) are read in place
        hipPointerAttribute_t attr;
        const bool on_device = !one_cycle && hipPointerGetAttributes(&attr, lookup_indices) == hipSuccess && attr.type == hipMemoryTypeDevice;
        (void)hipGetLastError();
        static const bool no_borrow = getenv("ATLAS_PS_COPY_IDX") != nullptr;
        if (on_device && !no_borrow) { P->d_idx = const_cast<uint64_t*>(lookup_indices); P->idx_borrowed = true; }
    }
    if (e == hipSuccess && !P->idx_borrowed) e = hipMalloc(&P->d_idx, T * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(&P->d_v, 2 * m * sizeof(Fr));       // two tables: the pipelined path alternates
    if (e == hipSuccess) e = hipMalloc(&P->d_qpart, (P->q_rows_max() + 1) * 6 * m * sizeof(Fr));
    if (e == hipSuccess && one_cycle) e = hipMemsetAsync(P->d_idx, 0, T * sizeof(uint64_t), rt().stream);
    if (e == hipSuccess && !P->idx_borrowed) e = hipMemcpyAsync(P->d_idx, lookup_indices, (one_cycle ? 1 : T) * sizeof(uint64_t), hipMemcpyDefault, rt().stream);   // host or device source
    if (e != hipSuccess) { delete P; return fail(ATLAS_ENOMEM, "ps_shout_new", e); }
    rc = P->rows.alloc(1, T);
    if (!rc) {
        size_t gb = (T + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
        k_ps_fill_one<<<(unsigned)gb, RA_THREADS, 0, rt().stream>>>(P->rows.buf[0], T);
        if (!one_cycle) rc = P->eq.init(reinterpret_cast<const H::Fr*>(r_node_output), log_T);       // (the split eq of the cycle rounds: none here)
    }
    // (the shortcut serves the round-channel drivers; a host-stepped caller falls back in its first call)
    static const bool no_sign = getenv("ATLAS_PS_NO_SIGN") != nullptr || getenv("ATLAS_NO_PIPELINE") != nullptr;     // A-B
    // (mode 3, the binary UnsignedLessThan range checks: the interleaved operands are small non-negative integers — one class, leading zero chunks)
    const bool try_sign = (mode == 0 || mode == 2 || mode == 3) && !one_cycle && m <= RA_THREADS && phases >= 3 && phases - 1 <= PS_SIGN_PMAX && rt().fs_mode == ATLAS_FS_HOST && !no_sign;
    // defer (NodePre: the instance is built ahead of its proof, under a wait its node makes anyway): launches only, finish_setup() at first use
    const bool can_defer = defer && !one_cycle && P->log_m <= 11 && rt().fs_mode == ATLAS_FS_HOST && getenv("ATLAS_NO_PIPELINE") == nullptr;
    if (!rc) rc = try_sign ? P->sign_setup(can_defer) : can_defer ? P->deferred_Q0() : P->build_Q(0);
    if (rc) { delete P; return rc; }
    *out = P;
    return ATLAS_OK;
}
// the batching challenge of an instance built before it was drawn (the constructors read it for nothing: it enters the host's round arithmetic)
extern "C++" int atlas_rt_ps_set_gamma(atlas_instance_t inst, const atlas_fr_t* gamma) {
    PsLookup* P = dynamic_cast<PsLookup*>(inst);
    if (!P || !gamma) return fail(ATLAS_EINVAL, "ps_set_gamma: not a prefix-suffix instance");
    std::memcpy(&P->gamma, gamma, 32);
    return ATLAS_OK;
}

int atlas_ps_shout_relu_new(const uint64_t* lookup_indices, size_t log_T, size_t xlen, const atlas_fr_t* r_node_output,
                            const atlas_fr_t* gamma, atlas_instance_t* out) {
    PROF("atlas_ps_shout_relu_new");
    NEED_INIT();
    if (!lookup_indices || (!r_node_output && log_T) || !gamma || !out) return fail(ATLAS_EINVAL, "ps_shout_relu_new: null argument");
    if (xlen != 16 && xlen != 32) return fail(ATLAS_EINVAL, "ps_shout_relu_new: X_LEN must be 16 or 32 (the reference's WordNoMSB suffix is a u32)");
    if (log_T > 25) return fail(ATLAS_EINVAL, "ps_shout_relu_new: log_T <= 25");
    return ps_new(lookup_indices, log_T, xlen, 8, 0, r_node_output, gamma, out);
}

int atlas_ps_shout_clamp_new(const uint64_t* lookup_indices, size_t log_T, size_t xlen, size_t bound, int symmetric,
                             const atlas_fr_t* r_node_output, const atlas_fr_t* gamma, atlas_instance_t* out) {
    PROF("atlas_ps_shout_clamp_new");
    NEED_INIT();
    if (!lookup_indices || (!r_node_output && log_T) || !gamma || !out) return fail(ATLAS_EINVAL, "ps_shout_clamp_new: null argument");
    if (xlen != 16 && xlen != 32 && xlen != 64) return fail(ATLAS_EINVAL, "ps_shout_clamp_new: X_LEN must be 16, 32 or 64");
    if (bound == 0 || bound + 1 >= xlen || bound > 31) return fail(ATLAS_EINVAL, "ps_shout_clamp_new: 1 <= BOUND <= 31 and BOUND < X_LEN - 1");
    if (log_T > 25) return fail(ATLAS_EINVAL, "ps_shout_clamp_new: log_T <= 25");
    return ps_new(lookup_indices, log_T, xlen, 8, 2, r_node_output, gamma, out, bound, symmetric != 0);
}

}  // extern "C"
int atlas_rt_ps_shout_clamp_new(const uint64_t* lookup_indices, size_t log_T, size_t xlen, size_t bound, int symmetric, const atlas_fr_t* r_node_output,
                                const atlas_fr_t* gamma, atlas_poly_t eq_shared, atlas_instance_t* out, bool defer) {
    PROF("atlas_ps_shout_clamp_new");
    NEED_INIT();
    if (!lookup_indices || (!r_node_output && log_T) || (!gamma && !defer) || !out) return fail(ATLAS_EINVAL, "ps_shout_clamp_new: null argument");
    if (xlen != 16 && xlen != 32 && xlen != 64) return fail(ATLAS_EINVAL, "ps_shout_clamp_new: X_LEN must be 16, 32 or 64");
    if (bound == 0 || bound + 1 >= xlen || bound > 31) return fail(ATLAS_EINVAL, "ps_shout_clamp_new: 1 <= BOUND <= 31 and BOUND < X_LEN - 1");
    if (log_T > 25) return fail(ATLAS_EINVAL, "ps_shout_clamp_new: log_T <= 25");
    return ps_new(lookup_indices, log_T, xlen, 8, 2, r_node_output, gamma, out, bound, symmetric != 0, log_T ? eq_shared : nullptr, defer);
}
int atlas_rt_ps_shout_relu_new(const uint64_t* lookup_indices, size_t log_T, size_t xlen, const atlas_fr_t* r_node_output, const atlas_fr_t* gamma,
                               atlas_poly_t eq_shared, atlas_instance_t* out, bool defer) {
    PROF("atlas_ps_shout_relu_new");
    NEED_INIT();
    if (!lookup_indices || (!r_node_output && log_T) || (!gamma && !defer) || !out) return fail(ATLAS_EINVAL, "ps_shout_relu_new: null argument");
    if (xlen != 16 && xlen != 32) return fail(ATLAS_EINVAL, "ps_shout_relu_new: X_LEN must be 16 or 32 (the reference's WordNoMSB suffix is a u32)");
    if (log_T > 25) return fail(ATLAS_EINVAL, "ps_shout_relu_new: log_T <= 25");
    return ps_new(lookup_indices, log_T, xlen, 8, 0, r_node_output, gamma, out, 0, true, log_T ? eq_shared : nullptr, defer);
}
int atlas_rt_identity_range_check_new(const uint64_t* lookup_indices, size_t log_T, size_t log_K, size_t phases, const atlas_fr_t* r_node_output,
                                      atlas_poly_t eq_shared, atlas_instance_t* out, bool defer) {
    PROF("atlas_identity_range_check_new");
    NEED_INIT();
    if (!lookup_indices || (!r_node_output && log_T) || !out) return fail(ATLAS_EINVAL, "identity_range_check_new: null argument");
    if (phases == 0 || log_K == 0 || log_K > 64 || log_K % phases || log_K / phases > 12)
        return fail(ATLAS_EINVAL, "identity_range_check_new: log_K must be a multiple of phases, chunks of at most 12 bits");
    if (log_T > 25) return fail(ATLAS_EINVAL, "identity_range_check_new: log_T <= 25");
    return ps_new(lookup_indices, log_T, log_K, phases, 1, r_node_output, nullptr, out, 0, true, log_T ? eq_shared : nullptr, defer);
}
extern "C" {

int atlas_ps_shout_rshift_new(const uint64_t* lookup_indices, size_t log_T, size_t xlen, size_t shift, const atlas_fr_t* r_node_output,
                              const atlas_fr_t* gamma, atlas_instance_t* out) {
    PROF("atlas_ps_shout_rshift_new");
    NEED_INIT();
    if (!lookup_indices || (!r_node_output && log_T) || !gamma || !out) return fail(ATLAS_EINVAL, "ps_shout_rshift_new: null argument");
    if ((xlen != 16 && xlen != 32) || shift >= xlen || log_T > 25)
        return fail(ATLAS_EINVAL, "ps_shout_rshift_new: xlen must be 16 or 32, shift < xlen, 1 <= log_T <= 25");
    return ps_new(lookup_indices, log_T, xlen, 8, 4, r_node_output, gamma, out, shift, true);
}

int atlas_ps_shout_ult_new(const uint64_t* lookup_indices, size_t log_T, const atlas_fr_t* r_node_output, const atlas_fr_t* gamma,
                           atlas_instance_t* out) {
    PROF("atlas_ps_shout_ult_new");
    NEED_INIT();
    if (!lookup_indices || (!r_node_output && log_T) || !gamma || !out) return fail(ATLAS_EINVAL, "ps_shout_ult_new: null argument");
    if (log_T > 25) return fail(ATLAS_EINVAL, "ps_shout_ult_new: log_T <= 25");
    return ps_new(lookup_indices, log_T, 64, 8, 3, r_node_output, gamma, out);
}

int atlas_identity_range_check_new(const uint64_t* lookup_indices, size_t log_T, size_t log_K, size_t phases,
                                   const atlas_fr_t* r_node_output, atlas_instance_t* out) {
    PROF("atlas_identity_range_check_new");
    NEED_INIT();
    if (!lookup_indices || (!r_node_output && log_T) || !out) return fail(ATLAS_EINVAL, "identity_range_check_new: null argument");
    if (phases == 0 || log_K == 0 || log_K > 64 || log_K % phases || log_K / phases > 12)
        return fail(ATLAS_EINVAL, "identity_range_check_new: log_K must be a multiple of phases, chunks of at most 12 bits");
    if (log_T > 25) return fail(ATLAS_EINVAL, "identity_range_check_new: log_T <= 25");
    return ps_new(lookup_indices, log_T, log_K, phases, 1, r_node_output, nullptr, out);
}

int atlas_u64_upload(const uint64_t* host, size_t n, uint64_t** d_out) {
    NEED_INIT();
    if (!host || !d_out || n == 0) return fail(ATLAS_EINVAL, "u64_upload");
    uint64_t* d = nullptr;
    hipError_t e = hipMalloc(&d, n * sizeof(uint64_t));
    if (e != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(u64)", e);
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    e = hipMemcpyAsync(d, host, n * sizeof(uint64_t), hipMemcpyHostToDevice, rt().stream);
    if (e == hipSuccess) e = hipStreamSynchronize(rt().stream);
    if (e != hipSuccess) { hipFree(d); return fail(ATLAS_ENODEV, "u64_upload", e); }
    *d_out = d;
    return ATLAS_OK;
}

int atlas_u64_free(uint64_t* d) {
    if (d) hipFree(d);
    return ATLAS_OK;
}

int atlas_lookup_indices_from_operands(const int32_t* d_left, const int32_t* d_right, size_t n, uint64_t** d_out) {
    PROF("atlas_lookup_indices_from_operands");
    NEED_INIT();
    if (!d_left || !d_out || n == 0) return fail(ATLAS_EINVAL, "lookup_indices_from_operands: null argument");
    uint64_t* d = nullptr;
    hipError_t e = hipMalloc(&d, n * sizeof(uint64_t));
    if (e != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(lookup indices)", e);
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    size_t gb = (n + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
    k_lookup_indices<<<(unsigned)gb, RA_THREADS, 0, rt().stream>>>(d_left, d_right, n, d);
    e = hipStreamSynchronize(rt().stream);
    if (e != hipSuccess) { hipFree(d); return fail(ATLAS_ENODEV, "lookup_indices_from_operands", e); }
    *d_out = d;
    return ATLAS_OK;
}

}  // extern "C"
