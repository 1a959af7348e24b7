// The two-phase provers of softmax_last_axis and the axis-sum prover as host-stepped SumcheckInstanceProver objects.
// Tensors are laid out [k][j] (K = 2^log_K rows, N = 2^log_N = the last axis), bound LowToHigh: the log_N last-axis
// variables first (phase 1), then the log_K row variables (phase 2).  Device counterparts of
// (jolt-atlas-core/src/onnx_proof/ops/):
//   softmax_last_axis/exp_sum.rs:146-197     ExpSumProver        (exp_q; r0_k)
//   softmax_last_axis/max.rs:185-261         MaxIndicatorProver  (X, e; r1_k)
//   softmax_last_axis/recip_mult.rs:196-268  RecipMultProver     (exp_q, inv_sum of K entries; r over all variables)
//   sum/axis.rs:220-232                      SumAxisProver       (operand; HighToLow, no eq)
// In phase 1 the eq factor depends on the row only: a K-entry table indexed by kj >> (log_N - m) (ExpSum,
// MaxIndicator) or the Gruen split over all variables with inv_sum read at the row (RecipMult).  Messages: ExpSum
// phase 1 and SumAxis answer with from_evals_and_hint over ONE evaluation (degree 1, trimmed by from_coeff).
// HBM per round: the operands' current length * 32 B read by the fold, read again + half written by the bind.
#include "ra_common.hip.h"

namespace {

enum { SM_EXP_SUM = ATLAS_SM_EXP_SUM, SM_MAX_INDICATOR = ATLAS_SM_MAX_INDICATOR, SM_RECIP_MULT = ATLAS_SM_RECIP_MULT,
       SM_SUM_AXIS = ATLAS_SM_SUM_AXIS };

__global__ __launch_bounds__(RA_THREADS) void k_sm_from_i32(const int32_t* __restrict__ in, Fr* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * RA_THREADS)
        fe_store(out + i, fr_from_i64((int64_t)in[i]));
}

// phase 1, ExpSum: sum_kj exp_q[2 kj] eq_k[kj >> shift]
__global__ __launch_bounds__(RA_THREADS) void k_sm_expsum_p1(const Fr* __restrict__ a, const Fr* __restrict__ eq_k, uint32_t shift,
                                                             size_t half, Fr* __restrict__ partials, MailTail tail) {
    Fr acc[1]; acc[0] = fe_zero();
    for (size_t kj = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; kj < half; kj += (size_t)gridDim.x * RA_THREADS)
        acc[0] = fr_add(acc[0], fr_mul(fe_load(a + 2 * kj), fe_load(eq_k + (kj >> shift))));
    block_reduce_put<1>(acc, partials, tail);
    mail_tail(partials, tail);
}

// phase 1, MaxIndicator: eq_k[k] X e at 0, 2, 3
__global__ __launch_bounds__(RA_THREADS) void k_sm_max_p1(const Fr* __restrict__ X, const Fr* __restrict__ e, const Fr* __restrict__ eq_k,
                                                          uint32_t shift, size_t half, Fr* __restrict__ partials, MailTail tail) {
    Fr acc[3]; acc[0] = fe_zero(); acc[1] = fe_zero(); acc[2] = fe_zero();
    for (size_t kj = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; kj < half; kj += (size_t)gridDim.x * RA_THREADS) {
        const Fr w = fe_load(eq_k + (kj >> shift));
        const Fr x0 = fe_load(X + 2 * kj), x1 = fe_load(X + 2 * kj + 1), e0 = fe_load(e + 2 * kj), e1 = fe_load(e + 2 * kj + 1);
        const Fr dx = fr_sub(x1, x0), de = fr_sub(e1, e0);
        const Fr x2 = fr_add(x1, dx), e2 = fr_add(e1, de), x3 = fr_add(x2, dx), e3 = fr_add(e2, de);
        acc[0] = fr_add(acc[0], fr_mul(w, fr_mul(x0, e0)));
        acc[1] = fr_add(acc[1], fr_mul(w, fr_mul(x2, e2)));
        acc[2] = fr_add(acc[2], fr_mul(w, fr_mul(x3, e3)));
    }
    block_reduce_put<3>(acc, partials, tail);
    mail_tail(partials, tail);
}

// phase 1, RecipMult: sum_kj E_out E_in exp_q[2 kj] inv_sum[kj >> shift]
__global__ __launch_bounds__(RA_THREADS) void k_sm_recip_p1(const Fr* __restrict__ a, const Fr* __restrict__ inv_sum, uint32_t shift,
                                                            SplitEqView E, size_t n_groups, Fr* __restrict__ partials, MailTail tail) {
    Fr acc[1]; acc[0] = fe_zero();
    for (size_t kj = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; kj < n_groups; kj += (size_t)gridDim.x * RA_THREADS)
        acc[0] = fr_add(acc[0], fr_mul(gse_weight(E, kj), fr_mul(fe_load(a + 2 * kj), fe_load(inv_sum + (kj >> shift)))));
    block_reduce_put<1>(acc, partials, tail);
    mail_tail(partials, tail);
}

// phase 2: sum_g E_out E_in [a0] (ExpSum) or [a0 b0, a_inf b_inf] (MaxIndicator, RecipMult)
template <int NQ>
__global__ __launch_bounds__(RA_THREADS) void k_sm_p2(const Fr* __restrict__ a, const Fr* __restrict__ b, SplitEqView E, size_t n_groups,
                                                      Fr* __restrict__ partials, MailTail tail) {
    Fr acc[NQ];
#pragma unroll
    for (int k = 0; k < NQ; k++) acc[k] = fe_zero();
    for (size_t g = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; g < n_groups; g += (size_t)gridDim.x * RA_THREADS) {
        const Fr w = gse_weight(E, g);
        const Fr a0 = fe_load(a + 2 * g);
        if constexpr (NQ == 1) acc[0] = fr_add(acc[0], fr_mul(w, a0));
        else {
            const Fr a1 = fe_load(a + 2 * g + 1), b0 = fe_load(b + 2 * g), b1 = fe_load(b + 2 * g + 1);
            acc[0] = fr_add(acc[0], fr_mul(w, fr_mul(a0, b0)));
            acc[1] = fr_add(acc[1], fr_mul(w, fr_mul(fr_sub(a1, a0), fr_sub(b1, b0))));
        }
    }
    block_reduce_put<NQ>(acc, partials, tail);
    mail_tail(partials, tail);
}

// SumAxis: sum of the low half (HighToLow: the pair of i is (i, i + half))
__global__ __launch_bounds__(RA_THREADS) void k_sm_sum_half(const Fr* __restrict__ a, size_t half, Fr* __restrict__ partials, MailTail tail) {
    Fr acc[1]; acc[0] = fe_zero();
    for (size_t i = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; i < half; i += (size_t)gridDim.x * RA_THREADS)
        acc[0] = fr_add(acc[0], fe_load(a + i));
    block_reduce_put<1>(acc, partials, tail);
    mail_tail(partials, tail);
}
// the same bind with the challenge from the round's slot (round channel)
__global__ __launch_bounds__(RA_THREADS) void k_sm_bind_hi_ch(const Fr* __restrict__ a, size_t half, ChanIo io, int r_hi_only, Fr* __restrict__ out) {
    Fr r;
    if (!io.challenge(r)) return;
    for (size_t i = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; i < half; i += (size_t)gridDim.x * RA_THREADS) {
        const Fr lo = fe_load(a + i), d = fr_sub(fe_load(a + i + half), lo);
        fe_store(out + i, fr_add(lo, r_hi_only ? fr_mul_hi(d, r) : fr_mul(d, r)));
    }
}
// the last bind of every row (two coefficients each) and, for RecipMult, of inv_sum: the final claims, one value per record
__global__ __launch_bounds__(64) void k_sm_finals_ch(const Fr* rows, size_t stride, uint32_t n_rows, const Fr* inv, ChanIo io, int hi_only) {
    __shared__ uint32_t stage[9 * 16];
    Fr r;
    if (!io.challenge(r)) return;
    Fr v = fe_zero();
    if (threadIdx.x < n_rows) v = bind_pair(fe_load(rows + (size_t)threadIdx.x * stride), fe_load(rows + (size_t)threadIdx.x * stride + 1), r, hi_only != 0);
    else if (inv && threadIdx.x == n_rows) v = bind_pair(fe_load(inv), fe_load(inv + 1), r, hi_only != 0);
    ch_mail_wave_fe(io.io, 0, n_rows + (inv ? 1u : 0u), v, stage);
}
// out[i] = a[i] + r (a[i + half] - a[i])
__global__ __launch_bounds__(RA_THREADS) void k_sm_bind_hi(const Fr* __restrict__ a, size_t half, Fr r, int r_hi_only, Fr* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; i < half; i += (size_t)gridDim.x * RA_THREADS) {
        const Fr lo = fe_load(a + i), d = fr_sub(fe_load(a + i + half), lo);
        fe_store(out + i, fr_add(lo, r_hi_only ? fr_mul_hi(d, r) : fr_mul(d, r)));
    }
}

struct Softmax : atlas_instance {
    int kind = 0;
    size_t log_K = 0, log_N = 0, round_next = 0;
    RaRows rows;                   // a (and b for MaxIndicator), 2^(log_K + log_N) each
    RaRows inv;                    // RecipMult: inv_sum, 2^log_K, bound in phase 2 only
    Fr* d_eq_k = nullptr;          // ExpSum / MaxIndicator: EqPolynomial::evals(r_k)
    GseDev gs;                     // phase-2 split-eq over r_k (ExpSum, MaxIndicator) or over all variables (RecipMult)
    ~Softmax() override { rows.release(); inv.release(); gs.release(); if (d_eq_k) hipFree(d_eq_k); }
    size_t rounds() const override { return log_K + log_N; }
    size_t degree() const override { return kind == SM_SUM_AXIS ? 1 : kind == SM_EXP_SUM ? 2 : 3; }

    static void linear_from_eval0(const H::Fr& claim, const H::Fr& e0, std::vector<H::Fr>& coeffs) {   // from_evals_and_hint(claim, [e0])
        coeffs.assign(2, H::zero());
        coeffs[0] = e0; coeffs[1] = H::sub(H::sub(claim, e0), e0);
        H::trim(coeffs);
    }

    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "softmax: round out of order");
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        const size_t half = rows.len / 2;
        size_t blocks = (half + RA_THREADS - 1) / RA_THREADS; if (blocks > 1024) blocks = 1024;
        const Fr* a = rows.buf[rows.cur]; const Fr* b = a + rows.stride[rows.cur];
        H::Fr s[3];
        int rc;
        if (kind == SM_SUM_AXIS) {
            k_sm_sum_half<<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(a, half, rows.partials, MailTail{{}, nullptr, 0, 0});
            if ((rc = rows.reduce_to_host((uint32_t)blocks, 1, s))) return rc;
            linear_from_eval0(claim, s[0], coeffs);
            return ATLAS_OK;
        }
        if (round < log_N) {
            const uint32_t shift = (uint32_t)(log_N - (round + 1));
            if (kind == SM_EXP_SUM) {
                k_sm_expsum_p1<<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(a, d_eq_k, shift, half, rows.partials, MailTail{{}, nullptr, 0, 0});
                if ((rc = rows.reduce_to_host((uint32_t)blocks, 1, s))) return rc;
                linear_from_eval0(claim, s[0], coeffs);
            } else if (kind == SM_MAX_INDICATOR) {
                k_sm_max_p1<<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(a, b, d_eq_k, shift, half, rows.partials, MailTail{{}, nullptr, 0, 0});
                if ((rc = rows.reduce_to_host((uint32_t)blocks, 3, s))) return rc;
                coeffs.assign(4, H::zero());
                H::unipoly_from_evals_and_hint(claim, s, 3, coeffs.data());
            } else {
                k_sm_recip_p1<<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(a, inv.buf[inv.cur], shift, gs.view(), half, rows.partials, MailTail{{}, nullptr, 0, 0});
                if ((rc = rows.reduce_to_host((uint32_t)blocks, 1, s))) return rc;
                coeffs.assign(3, H::zero());
                H::gruen_deg2(gs.st.scalar, gs.st.w_cur(), s[0], claim, coeffs.data());
            }
            return ATLAS_OK;
        }
        if (kind == SM_EXP_SUM) {
            k_sm_p2<1><<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(a, nullptr, gs.view(), half, rows.partials, MailTail{{}, nullptr, 0, 0});
            if ((rc = rows.reduce_to_host((uint32_t)blocks, 1, s))) return rc;
            coeffs.assign(3, H::zero());
            H::gruen_deg2(gs.st.scalar, gs.st.w_cur(), s[0], claim, coeffs.data());
        } else {
            k_sm_p2<2><<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(a, kind == SM_RECIP_MULT ? inv.buf[inv.cur] : b, gs.view(), half, rows.partials, MailTail{{}, nullptr, 0, 0});
            if ((rc = rows.reduce_to_host((uint32_t)blocks, 2, s))) return rc;
            coeffs.assign(4, H::zero());
            H::gruen_deg3(gs.st, s[0], s[1], claim, coeffs.data());
        }
        return ATLAS_OK;
    }

    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "softmax: round out of order");
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode);
        int rc;
        if (kind == SM_SUM_AXIS) {                 // HighToLow, out of place between the two row buffers
            const size_t half = rows.len / 2; const int nxt = rows.cur ^ 1;
            size_t gb = (half + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096; if (gb < 1) gb = 1;
            k_sm_bind_hi<<<(unsigned)gb, RA_THREADS, 0, rt().stream>>>(rows.buf[rows.cur], half, to_dev(rf), rt().challenge_mode == 0 ? 1 : 0, rows.buf[nxt]);
            rows.stride[nxt] = half; rows.cur = nxt; rows.len = half;
            round_next++;
            return ATLAS_OK;
        }
        if ((rc = rows.bind(r))) return rc;
        if (kind == SM_RECIP_MULT) {
            gs.st.bind(rf);
            if (round >= log_N && (rc = inv.bind(r))) return rc;
        } else if (round >= log_N) gs.st.bind(rf);
        round_next++;
        return ATLAS_OK;
    }

    // ---- round-channel stepping (instance.hpp): ingest_challenge(round - 1) — the binds, challenge from the round's slot — then the fold of
    // compute_message(round), its sums mailed (tagged rows, ra_common.hip.h).  Rows of round k live in buf[k & 1] with stride T >> k; inv_sum of
    // RecipMult's phase-2 round p in inv.buf[p & 1].  A softmax stage is a batch of these with a range check and one-hot lanes: host-stepped
    // it paid a synchronisation per member and round (~100 us per round of a six-member batch).
    bool have_finals = false;
    std::vector<H::Fr> mailed_finals;
    size_t T0() const { return (size_t)1 << (log_K + log_N); }
    static bool pipe_off() { static const bool v = getenv("ATLAS_SM_NO_PIPELINE") != nullptr; return v; }      // A-B
    bool pipelined() const override { return !pipe_off() && log_K >= 1; }      // (ONE row: no phase 2, inv_sum is never bound — stepped by the host)
    int n_sums(size_t round) const {
        if (kind == SM_SUM_AXIS || kind == SM_EXP_SUM) return 1;
        if (kind == SM_MAX_INDICATOR) return round < log_N ? 3 : 2;
        return round < log_N ? 1 : 2;                               // RecipMult
    }
    bool wide_wait(size_t round) const override {
        if (round == 0 || round > rounds()) return false;
        const size_t len = T0() >> round;
        return ((len + RA_THREADS - 1) / RA_THREADS) * rows.d > WIDE_WAIT_WGS;
    }
    int enqueue(size_t round, const atlas::RoundIo& io, bool bind_prev, atlas_mail_ref& mail) override {
        if (round >= rounds() || (round > 0) != bind_prev) return fail(ATLAS_ESTATE, "softmax: enqueue out of order");
        const ChanIo cio{io, rt().challenge_mode};
        const int hi = rt().challenge_mode == 0 ? 1 : 0;
        const size_t len = T0() >> round, half = len / 2, K = (size_t)1 << log_K;
        Fr* cur = rows.buf[round & 1];
        if (bind_prev) {
            const Fr* prev = rows.buf[(round - 1) & 1];
            size_t gb = (len + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096; if (gb < 1) gb = 1;
            if (kind == SM_SUM_AXIS) k_sm_bind_hi_ch<<<(unsigned)gb, RA_THREADS, 0, rt().stream>>>(prev, len, cio, hi, cur);
            else k_ra_bind_ch<<<dim3((unsigned)gb, (unsigned)rows.d), RA_THREADS, 0, rt().stream>>>(prev, T0() >> (round - 1), cur, len, len, cio, hi);
            if (kind == SM_RECIP_MULT && round - 1 >= log_N) {      // inv_sum binds with the phase-2 challenges
                const size_t p = round - 1 - log_N, ilen = K >> (p + 1);
                size_t gi = (ilen + RA_THREADS - 1) / RA_THREADS; if (gi < 1) gi = 1;
                k_ra_bind_ch<<<dim3((unsigned)gi, 1u), RA_THREADS, 0, rt().stream>>>(inv.buf[p & 1], K >> p, inv.buf[(p + 1) & 1], ilen ? ilen : 1, ilen, cio, hi);
            }
        }
        size_t blocks = (half + RA_THREADS - 1) / RA_THREADS; if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
        const int ns = n_sums(round);
        const MailTail tail{io, rows.d_counter, (uint32_t)blocks, (uint32_t)ns, rows.tg()};
        const Fr* a = cur; const Fr* b = a + len;                   // (d = 2: row 1 at the current stride)
        if (kind == SM_SUM_AXIS) k_sm_sum_half<<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(a, half, rows.partials, tail);
        else if (round < log_N) {
            const uint32_t shift = (uint32_t)(log_N - (round + 1));
            if (kind == SM_EXP_SUM) k_sm_expsum_p1<<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(a, d_eq_k, shift, half, rows.partials, tail);
            else if (kind == SM_MAX_INDICATOR) k_sm_max_p1<<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(a, b, d_eq_k, shift, half, rows.partials, tail);
            else { size_t ot, it; gs.st.tops_after(round, ot, it); k_sm_recip_p1<<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(a, inv.buf[0], shift, gs.view_at(ot, it), half, rows.partials, tail); }
        } else {
            size_t ot, it;
            gs.st.tops_after(kind == SM_RECIP_MULT ? round : round - log_N, ot, it);
            const SplitEqView E = gs.view_at(ot, it);
            if (kind == SM_EXP_SUM) k_sm_p2<1><<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(a, nullptr, E, half, rows.partials, tail);
            else k_sm_p2<2><<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(a, kind == SM_RECIP_MULT ? inv.buf[(round - log_N) & 1] : b, E, half, rows.partials, tail);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "softmax: launch", e);
        mail.base = io.mail; mail.blocks = 1; mail.n_vals = ns; mail.radix = 32; mail.shl = 0;
        return ATLAS_OK;
    }
    void prepare(size_t round) override { if (round == round_next && kind != SM_SUM_AXIS && (kind == SM_RECIP_MULT || round >= log_N)) gs.st.prepare_inverses(false); }
    int finish(size_t round, const H::Fr& claim, const H::Fr* s, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "softmax: round out of order");
        if (kind == SM_SUM_AXIS || (kind == SM_EXP_SUM && round < log_N)) { linear_from_eval0(claim, s[0], coeffs); return ATLAS_OK; }
        if (kind == SM_MAX_INDICATOR && round < log_N) { coeffs.assign(4, H::zero()); H::unipoly_from_evals_and_hint(claim, s, 3, coeffs.data()); return ATLAS_OK; }
        if ((kind == SM_RECIP_MULT && round < log_N) || kind == SM_EXP_SUM) { coeffs.assign(3, H::zero()); H::gruen_deg2(gs.st.scalar, gs.st.w_cur(), s[0], claim, coeffs.data()); return ATLAS_OK; }
        coeffs.assign(4, H::zero());
        H::gruen_deg3(gs.st, s[0], s[1], claim, coeffs.data());
        return ATLAS_OK;
    }
    int host_ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= rounds()) return fail(ATLAS_ESTATE, "softmax: round out of order");
        const H::Fr rf = H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode);
        if (kind == SM_RECIP_MULT) gs.st.bind(rf);
        else if (kind != SM_SUM_AXIS && round >= log_N) gs.st.bind(rf);
        rows.cur = (int)((round + 1) & 1); rows.len = T0() >> (round + 1); rows.stride[rows.cur] = rows.len;
        if (kind == SM_RECIP_MULT && round >= log_N) { const size_t p = round - log_N; inv.cur = (int)((p + 1) & 1); inv.len = ((size_t)1 << log_K) >> (p + 1); inv.stride[inv.cur] = inv.len; }
        round_next++;
        return ATLAS_OK;
    }
    int enqueue_finals(const atlas::RoundIo& io, atlas_mail_ref& mail) override {
        const size_t n = rounds();
        const Fr* iv = kind == SM_RECIP_MULT ? inv.buf[(log_K - 1) & 1] : nullptr;
        k_sm_finals_ch<<<1, 64, 0, rt().stream>>>(rows.buf[(n - 1) & 1], T0() >> (n - 1), (uint32_t)rows.d, iv, ChanIo{io, rt().challenge_mode}, rt().challenge_mode == 0 ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "softmax: launch", e);
        mail.base = io.mail; mail.blocks = 1; mail.n_vals = (int)rows.d + (iv ? 1 : 0); mail.radix = 32; mail.shl = 0;
        return ATLAS_OK;
    }
    int set_finals(const H::Fr* vals, size_t n) override {
        if (n != rows.d + (kind == SM_RECIP_MULT ? 1 : 0)) return fail(ATLAS_EINVAL, "softmax: final claims");
        mailed_finals.assign(vals, vals + n); have_finals = true;
        return ATLAS_OK;
    }

    int finals(std::vector<H::Fr>& out) override {
        if (round_next != rounds()) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        if (have_finals) { out = mailed_finals; return ATLAS_OK; }
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        int rc = rows.finals(out);
        if (rc || kind != SM_RECIP_MULT) return rc;
        std::vector<H::Fr> o2;
        if ((rc = inv.finals(o2))) return rc;
        out.push_back(o2[0]);
        return ATLAS_OK;
    }
};

int load_row(atlas_poly_t p, Fr* dst, size_t n) {
    size_t gb = (n + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
    if (p->is_i32) { k_sm_from_i32<<<(unsigned)gb, RA_THREADS, 0, rt().stream>>>((const int32_t*)p->d, dst, n); return ATLAS_OK; }
    HIP_TRY(hipMemcpyAsync(dst, p->d, n * sizeof(Fr), hipMemcpyDeviceToDevice, rt().stream));
    return ATLAS_OK;
}

}  // namespace

extern "C" {

int atlas_softmax_instance_new(int kind, atlas_poly_t a, atlas_poly_t b, size_t log_K, size_t log_N, const atlas_fr_t* r,
                               atlas_instance_t* out) {
    PROF("atlas_softmax_instance_new");
    NEED_INIT();
    if (!a || !out) return fail(ATLAS_EINVAL, "softmax_instance_new: null argument");
    if (kind < SM_EXP_SUM || kind > SM_SUM_AXIS) return fail(ATLAS_EINVAL, "softmax_instance_new: unknown kind");
    if (kind == SM_SUM_AXIS ? (log_K != 0 || b || log_N == 0) : ((!r && (log_K || kind == SM_RECIP_MULT)) || log_N == 0))
        return fail(ATLAS_EINVAL, "softmax_instance_new: SumAxis takes one operand of 2^log_N (log_K = 0); the softmax provers need log_N >= 1 and r (ONE row, log_K = 0: r_k is the empty point)");
    if (log_K + log_N > 25) return fail(ATLAS_EINVAL, "softmax_instance_new: more than 25 variables");
    const size_t T = (size_t)1 << (log_K + log_N), K = (size_t)1 << log_K;
    if (a->len != T) return fail(ATLAS_EINVAL, "softmax_instance_new: operand length != 2^(log_K + log_N)");
    if ((kind == SM_MAX_INDICATOR && (!b || b->len != T)) || (kind == SM_RECIP_MULT && (!b || b->len != K)) || (kind == SM_EXP_SUM && b))
        return fail(ATLAS_EINVAL, "softmax_instance_new: second operand: e of 2^(log_K + log_N) for MaxIndicator, inv_sum of 2^log_K for RecipMult, none for ExpSum");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    Softmax* P = new Softmax();
    P->kind = kind; P->log_K = log_K; P->log_N = log_N;
    int rc = P->rows.alloc(kind == SM_MAX_INDICATOR ? 2 : 1, T, 3);
    if (!rc) rc = load_row(a, P->rows.buf[0], T);
    if (!rc && kind == SM_MAX_INDICATOR) rc = load_row(b, P->rows.buf[0] + T, T);
    if (!rc && kind == SM_RECIP_MULT) { rc = P->inv.alloc(1, K); if (!rc) rc = load_row(b, P->inv.buf[0], K); }
    if (!rc && (kind == SM_EXP_SUM || kind == SM_MAX_INDICATOR)) {
        const std::vector<H::Fr> ek = H::eq_evals(reinterpret_cast<const H::Fr*>(r), log_K);
        hipError_t e = hipMalloc(&P->d_eq_k, K * sizeof(Fr));
        if (e == hipSuccess) e = hipMemcpyAsync(P->d_eq_k, ek.data(), K * sizeof(Fr), hipMemcpyHostToDevice, rt().stream);
        if (e == hipSuccess) e = hipStreamSynchronize(rt().stream);
        if (e != hipSuccess) rc = fail(ATLAS_ENOMEM, "softmax_instance_new: eq table", e);
        if (!rc && log_K) rc = P->gs.init(reinterpret_cast<const H::Fr*>(r), log_K);     // used from round log_N on (ONE row: never)
    }
    if (!rc && kind == SM_RECIP_MULT) rc = P->gs.init(reinterpret_cast<const H::Fr*>(r), log_K + log_N);
    if (!rc) { hipError_t e = hipStreamSynchronize(rt().stream); if (e != hipSuccess) rc = fail(ATLAS_ENODEV, "softmax_instance_new", e); }
    if (rc) { delete P; return rc; }
    *out = P;
    return ATLAS_OK;
}

}  // extern "C"
