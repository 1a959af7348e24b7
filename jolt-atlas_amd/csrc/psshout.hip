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
// flushes its non-zero accumulators to the global ones and k_ps_q_final turns the word sums back into residues.  (A
// tiled variant in which every thread scanned its tile's entries for its bin took 168 us per phase at T = 2^16, NQ = 6,
// m = 256.)  The accumulators are NQ * m * 64 bytes of dynamic LDS (96 KB for the clamp lookup).
template <int NQ>
__global__ __launch_bounds__(RA_THREADS) void k_ps_q_lds(const uint64_t* __restrict__ idx, const Fr* __restrict__ u0,
                                                         const Fr* __restrict__ prod, size_t T, uint32_t suffix_len, uint32_t m,
                                                         uint32_t bound, unsigned long long* __restrict__ acc /* [m][NQ][8] */) {
    extern __shared__ unsigned long long ps_sm[];
    const uint32_t n_words = m * NQ * 8;
    for (uint32_t w = threadIdx.x; w < n_words; w += RA_THREADS) ps_sm[w] = 0;
    __syncthreads();
    for (size_t t = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; t < T; t += (size_t)gridDim.x * RA_THREADS) {
        const uint64_t k = idx[t];
        const uint32_t b = (uint32_t)(k >> suffix_len) & (m - 1);
        Fr val[NQ];
        ps_entry_vals<NQ>(k, fr_mul(fe_load(u0 + t), fe_load(prod + t)), suffix_len, bound, val);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if (fe_is_zero(val[q])) continue;
            unsigned long long* a = ps_sm + ((size_t)b * NQ + q) * 8;
#pragma unroll
            for (int w = 0; w < 8; w++) atomicAdd(&a[w], (unsigned long long)val[q].v[w]);
        }
    }
    __syncthreads();
    for (uint32_t w = threadIdx.x; w < n_words; w += RA_THREADS)
        if (ps_sm[w]) atomicAdd(&acc[w], ps_sm[w]);
}

// word sums -> canonical Montgomery residues: V = lo + hi 2^256, V mod p = lo * R * R^-1 + hi * R^2 * R^-1
__global__ __launch_bounds__(RA_THREADS) void k_ps_q_final(const unsigned long long* __restrict__ acc, uint32_t n_vals, Fr* __restrict__ out) {
    const uint32_t i = blockIdx.x * RA_THREADS + threadIdx.x;
    if (i >= n_vals) return;
    Fr lo, hi, r2;
    unsigned long long c = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const unsigned long long s = acc[(size_t)i * 8 + w] + c;
        lo.v[w] = (uint32_t)s;
        c = s >> 32;
    }
#pragma unroll
    for (int w = 0; w < 8; w++) { hi.v[w] = 0; r2.v[w] = FrParams::r2(w); }
    hi.v[0] = (uint32_t)c; hi.v[1] = (uint32_t)(c >> 32);
    fe_store(out + i, fr_add(fr_mul(lo, fr_one()), fr_mul(hi, r2)));
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

struct PsLookup : atlas_instance {
    size_t N = 0, log_m = 0, m = 0, log_T = 0, T = 0, round_next = 0, phases = 8;   // N = LOG_K
    int mode = 0;                         // 0 = ReLU + gamma * SignedIdentity (unary read-raf), 1 = Identity (range check), 2 = clamp family, 3 = UnsignedLessThan (binary), 4 = RightShift
    size_t bound = 0; bool symmetric = true;   // ClampBoundedTable<N, BOUND, SYMMETRIC> (lookup_tables/clamp.rs)
    size_t nq() const { return mode == 2 ? 6 : mode == 3 ? 4 : mode == 4 ? 3 : 2; }
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

    ~PsLookup() override { for (void* p : {(void*)d_idx, (void*)d_u0, (void*)d_v, (void*)d_qpart}) if (p) hipFree(p); rows.release(); eq.release(); }
    size_t rounds() const override { return N + log_T; }
    size_t degree() const override { return 2; }

    static H::Fr pow2(size_t k) { H::Fr o = H::one(); const H::Fr two = H::from_u64(2); for (size_t i = 0; i < k; i++) o = H::mul(o, two); return o; }
    H::Fr rs_weight(size_t i) const { return (bound < N && i <= N - 1 - bound) ? pow2(N - 1 - i - bound) : H::zero(); }   // mode 4: bit i of k >> D
    H::Fr weight(size_t i) const { H::Fr w = pow2(N - 1 - i); return (i == 0 && mode != 1) ? H::sub(w, pow2(N)) : w; }   // (Signed)Identity coefficient of bit i

    int build_Q(size_t phase) {           // init_phase: Q tables of `phase` from the current products
        const uint32_t suffix_len = (uint32_t)((phases - 1 - phase) * log_m);
        const size_t NQ = nq();
        Fr* d_qsum = d_qpart + q_rows_max() * NQ * m;
        if (m <= RA_THREADS) {
            unsigned long long* acc = (unsigned long long*)d_qpart;              // m * NQ * 8 word sums (the partial-row area is larger)
            const size_t lds = m * NQ * 64;
            size_t gb = T / 2048; if (gb < 16) gb = 16; if (gb > 256) gb = 256;
            HIP_TRY(hipMemsetAsync(acc, 0, lds, g.stream));
#define PS_Q_LDS(NQv, BND)                                                                                                         \
            do {                                                                                                                   \
                static bool attr_set = false;                                                                                      \
                if (!attr_set) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ps_q_lds<NQv>), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * NQv * 64)); attr_set = true; } \
                k_ps_q_lds<NQv><<<(unsigned)gb, RA_THREADS, lds, g.stream>>>(d_idx, d_u0, rows.buf[0], T, suffix_len, (uint32_t)m, (uint32_t)(BND), acc); \
            } while (0)
            if (NQ == 4) PS_Q_LDS(4, 0u);
            else if (NQ == 6) PS_Q_LDS(6, bound);
            else if (NQ == 3) PS_Q_LDS(3, bound);
            else PS_Q_LDS(2, 0u);
#undef PS_Q_LDS
            k_ps_q_final<<<(unsigned)((NQ * m + RA_THREADS - 1) / RA_THREADS), RA_THREADS, 0, g.stream>>>(acc, (uint32_t)(NQ * m), d_qsum);
        } else {
            if (NQ == 4) k_ps_q<4><<<dim3((unsigned)m, SLICES), RA_THREADS, 0, g.stream>>>(d_idx, d_u0, rows.buf[0], T, suffix_len, (uint32_t)(m - 1), 0u, d_qpart);
            else if (NQ == 6) k_ps_q<6><<<dim3((unsigned)m, SLICES), RA_THREADS, 0, g.stream>>>(d_idx, d_u0, rows.buf[0], T, suffix_len, (uint32_t)(m - 1), (uint32_t)bound, d_qpart);
            else if (NQ == 3) k_ps_q<3><<<dim3((unsigned)m, SLICES), RA_THREADS, 0, g.stream>>>(d_idx, d_u0, rows.buf[0], T, suffix_len, (uint32_t)(m - 1), (uint32_t)bound, d_qpart);
            else k_ps_q<2><<<dim3((unsigned)m, SLICES), RA_THREADS, 0, g.stream>>>(d_idx, d_u0, rows.buf[0], T, suffix_len, (uint32_t)(m - 1), 0u, d_qpart);
            k_col_reduce<<<(unsigned)(NQ * m), RA_THREADS, 0, g.stream>>>(d_qpart, SLICES, (uint32_t)(NQ * m), d_qsum);
        }
        std::vector<H::Fr> q(NQ * m);
        HIP_TRY(hipMemcpyAsync(q.data(), d_qsum, NQ * m * sizeof(Fr), hipMemcpyDeviceToHost, g.stream));
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
                // Everything but the bin index b is the same for all bins of a round, and b enters linearly (b * 2^suffix_len,
                // the clamp's low-word bits): the sums over b of Q_k[b] and of (small integer)(b) * Q_k[b] are taken first
                // (one small-integer multiplication per term) and the prefix factors are applied once.
                if (mode != 3) {
                    auto qv = [&](size_t k, size_t b) { return ci ? H::sub(H::add(Q[k][b + half], Q[k][b + half]), Q[k][b]) : Q[k][b]; };
                    H::Fr s_q1 = H::zero(), s_bq1 = H::zero(), s_qs = H::zero();          // sum q1, sum b q1, sum qs
                    for (size_t b = 0; b < half; b++) {
                        const H::Fr q1 = qv(0, b);
                        s_q1 = H::add(s_q1, q1);
                        if (b) s_bq1 = H::add(s_bq1, H::mul(H::from_u64(b), q1));
                        s_qs = H::add(s_qs, qv(1, b));
                    }
                    const H::Fr bs_q1 = H::mul(sh, s_bq1);                                 // sum (b 2^suffix_len) q1
                    const H::Fr idt = H::add(H::add(H::mul(sid_c, s_q1), bs_q1), s_qs);    // sum ((sid_c + bs) q1 + qs)
                    if (mode == 1) acc = idt;
                    else if (mode == 0) acc = H::add(H::mul(not_msb, H::add(H::add(H::mul(word_c, s_q1), bs_q1), s_qs)), H::mul(gamma, idt));
                    else if (mode == 4) {
                        const H::Fr rs_c = H::add(rs_acc, H::mul(c, rs_weight(j)));
                        H::Fr s_cq1 = H::zero(), s_q2 = H::zero();                         // sum (chunk >> shift) q1, sum q2
                        for (size_t b = 0; b < half; b++) {
                            const uint64_t cv = bound >= 64 ? 0 : ((uint64_t)b << suffix_len) >> bound;
                            if (cv) s_cq1 = H::add(s_cq1, H::mul(H::from_u64(cv), qv(0, b)));
                            s_q2 = H::add(s_q2, qv(2, b));
                        }
                        acc = H::add(H::add(H::add(H::mul(rs_c, s_q1), s_cq1), s_q2), H::mul(gamma, idt));
                    } else {      // clamp (clamp.rs:84-109): chunk bits of b at variable index j+1+q are high iff that index < hbits
                        H::Fr z_q2 = H::zero(), z_lq2 = H::zero(), z_q3 = H::zero(), o_q4 = H::zero(), o_lq4 = H::zero(), o_q5 = H::zero();
                        for (size_t b = 0; b < half; b++) {
                            bool z = true, o = true;
                            uint64_t lwb = 0;
                            for (size_t q = 0; q < blen; q++) {
                                const size_t var = j + 1 + q;
                                const uint64_t bit = (b >> (blen - 1 - q)) & 1;
                                if (var < hbits) { if (bit) z = false; else o = false; }
                                else lwb |= bit << (N - 1 - var);
                            }
                            if (z) {
                                const H::Fr q2 = qv(2, b);
                                z_q2 = H::add(z_q2, q2); z_q3 = H::add(z_q3, qv(3, b));
                                if (lwb) z_lq2 = H::add(z_lq2, H::mul(H::from_u64(lwb), q2));
                            }
                            if (o && symmetric) {
                                const H::Fr q4 = qv(4, b);
                                o_q4 = H::add(o_q4, q4); o_q5 = H::add(o_q5, qv(5, b));
                                if (lwb) o_lq4 = H::add(o_lq4, H::mul(H::from_u64(lwb), q4));
                            }
                        }
                        // sum_z haz_c ((lw_c + lwb - U) q2 + q3) + sum_o hao_c ((lw_c + lwb) q4 + q5)
                        H::Fr val = H::mul(H::sub(U, H::mul(msb, LC)), s_q1);
                        val = H::add(val, H::mul(haz_c, H::add(H::add(H::mul(H::sub(lw_c, U), z_q2), z_lq2), z_q3)));
                        if (symmetric) val = H::add(val, H::mul(hao_c, H::add(H::add(H::mul(lw_c, o_q4), o_lq4), o_q5)));
                        acc = H::add(val, H::mul(gamma, idt));
                    }
                } else
                for (size_t b = 0; b < half; b++) {
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
                ev[ci] = acc;
            }
            H::unipoly_from_evals_and_hint(claim, ev, 2, coeffs.data());
            return ATLAS_OK;
        }
        std::lock_guard<atlas_rt::Mutex> lk(g.mu);
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
            if ((j + 1) % log_m == 0) {                               // phase boundary: fold v_p into the products
                std::lock_guard<atlas_rt::Mutex> lk(g.mu);
                HIP_TRY(hipMemcpyAsync(d_v, v.data(), m * sizeof(Fr), hipMemcpyHostToDevice, g.stream));
                size_t gb = (T + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
                k_ps_scale<<<(unsigned)gb, RA_THREADS, 0, g.stream>>>(d_idx, d_v, T, (uint32_t)((phases - 1 - p) * log_m), (uint32_t)(m - 1), rows.buf[0]);
                if (p != phases - 1) { int rc = build_Q(p + 1); if (rc) return rc; }       // (ends with a stream synchronize: v may change after it)
                else HIP_TRY(hipStreamSynchronize(g.stream));
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
        } else {
            std::lock_guard<atlas_rt::Mutex> lk(g.mu);
            int rc = rows.bind(r);
            if (rc) return rc;
            eq.st.bind(rf);
        }
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override {
        if (round_next != rounds()) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        std::lock_guard<atlas_rt::Mutex> lk(g.mu);
        return rows.finals(out);
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
                  const atlas_fr_t* gamma, atlas_instance_t* out, size_t bound = 0, bool symmetric = true) {
    atlas_poly_t E = nullptr;
    int rc = atlas_eq_evals(r_node_output, log_T, nullptr, &E);      // u_evals = EqPolynomial::evals(r_node_output), mod.rs:234
    if (rc) return rc;
    std::lock_guard<atlas_rt::Mutex> lk(g.mu);
    PsLookup* P = new PsLookup();
    P->N = log_K; P->phases = phases; P->mode = mode; P->bound = bound; P->symmetric = symmetric;
    P->log_m = log_K / phases; P->m = (size_t)1 << P->log_m; P->log_T = log_T; P->T = (size_t)1 << log_T;
    if (gamma) std::memcpy(&P->gamma, gamma, 32);
    P->d_u0 = (Fr*)E->d; delete E;                                   // keep the table, drop the handle
    const size_t T = P->T, m = P->m;
    hipError_t e = hipMalloc(&P->d_idx, T * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc(&P->d_v, m * sizeof(Fr));
    if (e == hipSuccess) e = hipMalloc(&P->d_qpart, (P->q_rows_max() + 1) * 6 * m * sizeof(Fr));
    if (e == hipSuccess) e = hipMemcpyAsync(P->d_idx, lookup_indices, T * sizeof(uint64_t), hipMemcpyDefault, g.stream);   // host or device source
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

int atlas_ps_shout_rshift_new(const uint64_t* lookup_indices, size_t log_T, size_t xlen, size_t shift, const atlas_fr_t* r_node_output,
                              const atlas_fr_t* gamma, atlas_instance_t* out) {
    NEED_INIT();
    if (!lookup_indices || !r_node_output || !gamma || !out) return fail(ATLAS_EINVAL, "ps_shout_rshift_new: null argument");
    if ((xlen != 16 && xlen != 32) || shift >= xlen || log_T == 0 || log_T > 25)
        return fail(ATLAS_EINVAL, "ps_shout_rshift_new: xlen must be 16 or 32, shift < xlen, 1 <= log_T <= 25");
    return ps_new(lookup_indices, log_T, xlen, 8, 4, r_node_output, gamma, out, shift, true);
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

int atlas_u64_upload(const uint64_t* host, size_t n, uint64_t** d_out) {
    NEED_INIT();
    if (!host || !d_out || n == 0) return fail(ATLAS_EINVAL, "u64_upload");
    uint64_t* d = nullptr;
    hipError_t e = hipMalloc(&d, n * sizeof(uint64_t));
    if (e != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(u64)", e);
    std::lock_guard<atlas_rt::Mutex> lk(g.mu);
    e = hipMemcpyAsync(d, host, n * sizeof(uint64_t), hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (e != hipSuccess) { hipFree(d); return fail(ATLAS_ENODEV, "u64_upload", e); }
    *d_out = d;
    return ATLAS_OK;
}

int atlas_u64_free(uint64_t* d) {
    if (d) hipFree(d);
    return ATLAS_OK;
}

int atlas_lookup_indices_from_operands(const int32_t* d_left, const int32_t* d_right, size_t n, uint64_t** d_out) {
    NEED_INIT();
    if (!d_left || !d_out || n == 0) return fail(ATLAS_EINVAL, "lookup_indices_from_operands: null argument");
    uint64_t* d = nullptr;
    hipError_t e = hipMalloc(&d, n * sizeof(uint64_t));
    if (e != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(lookup indices)", e);
    std::lock_guard<atlas_rt::Mutex> lk(g.mu);
    size_t gb = (n + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
    k_lookup_indices<<<(unsigned)gb, RA_THREADS, 0, g.stream>>>(d_left, d_right, n, d);
    e = hipStreamSynchronize(g.stream);
    if (e != hipSuccess) { hipFree(d); return fail(ATLAS_ENODEV, "lookup_indices_from_operands", e); }
    *d_out = d;
    return ATLAS_OK;
}

}  // extern "C"
