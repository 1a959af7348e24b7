// Element-wise operator sumchecks ("MulProver::compute_message, pattern for element-wise ops") as
// host-stepped SumcheckInstanceProver objects.  One GruenSplitEqPolynomial over r_node_output (LowToHigh), the
// operands bound LowToHigh; per round one fold  sum_g E_out E_in f(g)  on the device and Gruen's recovery on
// the host.  Device counterparts of (jolt-atlas-core/src/onnx_proof/ops/):
//   add.rs:283-304  sub.rs:267-288  neg.rs:206-222      [q_constant]            gruen_poly_deg_2
//   square.rs:163-183  mul.rs:160-199  iff.rs:189-224   [q_constant, q_quadratic] gruen_poly_deg_3
//   cube.rs:159-171    compute_mle_product_sum(3, operand)  (mles_product_sum.rs:41-55): grid [1, 2, inf]
//   div.rs:329-362 (left, right, q, R)   rsqrt.rs:390-433 (input, quotient, output, div_rem, sqrt_rem; S^3, gamma)
//   scalar_const_div.rs:227-248 is the Sub fold over (left, R)
//   joltworks/src/subprotocols/hamming_booleanity.rs:131-165 (hw_0 .. hw_{d-1}; gamma_powers)
//   neural_teleport/division.rs:231-256 (input, quotient, remainder; tau)
// and the selector-style provers WITHOUT an eq factor (sumcheck_evals at 0 and 2, UniPoly::from_evals_and_hint):
//   reshape.rs:286-315, slice.rs:254-281 (input, selector); concat.rs:290-325 (input_t, selector_t per term);
//   gather/mod.rs:232-268 (index_onehot, dictionary, identity; gamma)
// Operands are atlas_poly_t (Fr or I32Scalars; not consumed): the rows of the instance are Fr copies — the
// first bind of a CompactPolynomial (compact_polynomial.rs:272-353) yields the same field values.
// HBM per round: n_ops * len * 32 B read by the fold, the same read + half written by the bind.
#include "ra_common.hip.h"
#include "shard_group.hpp"

namespace {

enum { EW_ADD = ATLAS_EW_ADD, EW_SUB = ATLAS_EW_SUB, EW_NEG = ATLAS_EW_NEG, EW_SQUARE = ATLAS_EW_SQUARE, EW_IFF = ATLAS_EW_IFF,
       EW_MUL = ATLAS_EW_MUL, EW_CUBE = ATLAS_EW_CUBE, EW_DIV = ATLAS_EW_DIV, EW_RSQRT = ATLAS_EW_RSQRT,
       EW_DOT = ATLAS_EW_DOT, EW_GATHER = ATLAS_EW_GATHER, EW_HAMMING_BOOL = ATLAS_EW_HAMMING_BOOL,
       EW_TELEPORT_DIV = ATLAS_EW_TELEPORT_DIV };

struct EwConsts { Fr k[16]; uint32_t n_terms; };     // Rsqrt: S^3, gamma; Gather: gamma; HammingBooleanity: gamma_powers (n_terms of them);
                                                     // Dot: n_terms = number of (input, selector) pairs

constexpr bool ew_has_eq(int op) { return op != EW_DOT && op != EW_GATHER; }
constexpr int ew_outputs(int op) { return op == EW_CUBE ? 3 : (op == EW_DOT || op == EW_GATHER || op == EW_HAMMING_BOOL || op == EW_SQUARE || op == EW_MUL || op == EW_IFF || op == EW_DIV || op == EW_RSQRT) ? 2 : 1; }
constexpr int ew_operands(int op) { return (op == EW_GATHER || op == EW_TELEPORT_DIV) ? 3 : op == EW_RSQRT ? 5 : op == EW_DIV ? 4 : op == EW_IFF ? 3 : (op == EW_ADD || op == EW_SUB || op == EW_MUL) ? 2 : 1; }

__global__ __launch_bounds__(RA_THREADS) void k_ew_from_i32(const int32_t* __restrict__ in, Fr* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * RA_THREADS)
        fe_store(out + i, fr_from_i64((int64_t)in[i]));
}

// ---- how a fold reads pair g of row k ---------------------------------------------------------------
// EwDirect: the rows are bound already.  EwBind: the previous round's challenge is applied on the fly
// (LowToHigh: new[j] = old[2j] + r (old[2j+1] - old[2j])) and the bound pair is written to the other half of
// the ping-pong — ingest_challenge(round-1) fused into compute_message(round), one launch per round.
// Every row is visited exactly once per g (pair or lo), so a bind pass leaves every row bound.
struct EwDirect {
    const Fr* rows; size_t stride;
    __device__ __forceinline__ void pair(uint32_t k, size_t g, Fr& a0, Fr& a1) const {
        a0 = fe_load(rows + (size_t)k * stride + 2 * g); a1 = fe_load(rows + (size_t)k * stride + 2 * g + 1);
    }
    __device__ __forceinline__ Fr lo(uint32_t k, size_t g) const { return fe_load(rows + (size_t)k * stride + 2 * g); }
};
struct EwBind {
    const Fr* src; size_t sstride; Fr* dst; size_t dstride; Fr r; bool hi;
    __device__ __forceinline__ void pair(uint32_t k, size_t g, Fr& a0, Fr& a1) const {
        const Fr* s = src + (size_t)k * sstride + 4 * g;
        a0 = bind_pair(fe_load(s), fe_load(s + 1), r, hi); a1 = bind_pair(fe_load(s + 2), fe_load(s + 3), r, hi);
        Fr* d = dst + (size_t)k * dstride + 2 * g;
        fe_store(d, a0); fe_store(d + 1, a1);
    }
    __device__ __forceinline__ Fr lo(uint32_t k, size_t g) const { Fr a0, a1; pair(k, g, a0, a1); return a0; }
};

// the terms f_k(g) of operator OP at pair index g
template <int OP, class LD>
__device__ __forceinline__ void ew_terms(const LD& ld, size_t g, const EwConsts& C, Fr* v) {
    if constexpr (OP == EW_DOT) {                   // rows: (input_t, selector_t) pairs; values at X = 0 and X = 2
        v[0] = fe_zero(); v[1] = fe_zero();
        for (uint32_t tm = 0; tm < C.n_terms; tm++) {
            Fr i0, i1, s0, s1;
            ld.pair(2 * tm, g, i0, i1); ld.pair(2 * tm + 1, g, s0, s1);
            v[0] = fr_add(v[0], fr_mul(i0, s0));
            v[1] = fr_add(v[1], fr_mul(fr_add(i1, fr_sub(i1, i0)), fr_add(s1, fr_sub(s1, s0))));
        }
        return;
    } else if constexpr (OP == EW_HAMMING_BOOL) {   // rows hw_d: sum_d gamma_d [hw0 (hw0 - 1), (hw1 - hw0)^2]
        v[0] = fe_zero(); v[1] = fe_zero();
        for (uint32_t d = 0; d < C.n_terms; d++) {
            Fr h0, h1;
            ld.pair(d, g, h0, h1);
            const Fr a = fr_sub(h1, h0);
            v[0] = fr_add(v[0], fr_mul(C.k[d], fr_mul(h0, fr_sub(h0, fr_one()))));
            v[1] = fr_add(v[1], fr_mul(C.k[d], fr_mul(a, a)));
        }
        return;
    } else {
        Fr a0, a1;
        ld.pair(0, g, a0, a1);
        if constexpr (OP == EW_ADD) v[0] = fr_add(a0, ld.lo(1, g));
        else if constexpr (OP == EW_SUB) v[0] = fr_sub(a0, ld.lo(1, g));
        else if constexpr (OP == EW_NEG) v[0] = fr_sub(fe_zero(), a0);
        else if constexpr (OP == EW_TELEPORT_DIV) {    // rows: input, quotient, remainder: tau q0 + r0 - inp0
            const Fr q0 = ld.lo(1, g), r0 = ld.lo(2, g);
            v[0] = fr_sub(fr_add(fr_mul(C.k[0], q0), r0), a0);
        } else if constexpr (OP == EW_SQUARE) { const Fr d = fr_sub(a1, a0); v[0] = fr_mul(a0, a0); v[1] = fr_mul(d, d); }
        else if constexpr (OP == EW_MUL) {
            Fr b0, b1;
            ld.pair(1, g, b0, b1);
            v[0] = fr_mul(a0, b0); v[1] = fr_mul(fr_sub(a1, a0), fr_sub(b1, b0));
        } else if constexpr (OP == EW_IFF) {        // rows: mask, a, b.  c0 = b0 + m0 (a0 - b0);  e = m_inf (a_inf - b_inf)
            Fr x0, x1, y0, y1;
            ld.pair(1, g, x0, x1); ld.pair(2, g, y0, y1);
            v[0] = fr_add(y0, fr_mul(a0, fr_sub(x0, y0)));
            v[1] = fr_mul(fr_sub(a1, a0), fr_sub(fr_sub(x1, x0), fr_sub(y1, y0)));
        } else if constexpr (OP == EW_GATHER) {     // rows: ra, dictionary, identity: ra (dict + gamma id) at X = 0 and 2
            Fr d0, d1, j0, j1;
            ld.pair(1, g, d0, d1); ld.pair(2, g, j0, j1);
            const Fr a2 = fr_add(a1, fr_sub(a1, a0)), d2 = fr_add(d1, fr_sub(d1, d0)), j2 = fr_add(j1, fr_sub(j1, j0));
            v[0] = fr_mul(a0, fr_add(d0, fr_mul(C.k[0], j0)));
            v[1] = fr_mul(a2, fr_add(d2, fr_mul(C.k[0], j2)));
        } else if constexpr (OP == EW_DIV) {        // rows: left, right, q, R.  c0 = ro0 q0 + R0 - lo0;  e = ro_inf q_inf
            Fr r0, r1, q0, q1;
            ld.pair(1, g, r0, r1); ld.pair(2, g, q0, q1);
            v[0] = fr_sub(fr_add(fr_mul(r0, q0), ld.lo(3, g)), a0);
            v[1] = fr_mul(fr_sub(r1, r0), fr_sub(q1, q0));
        } else if constexpr (OP == EW_RSQRT) {      // rows: input, quotient, output, div_rem, sqrt_rem
            Fr q0, q1, o0, o1;
            ld.pair(1, g, q0, q1); ld.pair(2, g, o0, o1);
            const Fr div0 = fr_sub(fr_add(fr_mul(a0, q0), ld.lo(3, g)), C.k[0]);
            const Fr sqrt0 = fr_sub(fr_add(fr_mul(o0, o0), ld.lo(4, g)), q0);
            const Fr od = fr_sub(o1, o0);
            v[0] = fr_add(div0, fr_mul(C.k[1], sqrt0));
            v[1] = fr_add(fr_mul(fr_sub(a1, a0), fr_sub(q1, q0)), fr_mul(C.k[1], fr_mul(od, od)));
        } else {                                    // cube: p(1)^3, p(2)^3, p_inf^3
            const Fr d = fr_sub(a1, a0), p2 = fr_add(a1, d);
            v[0] = fr_mul(fr_mul(a1, a1), a1); v[1] = fr_mul(fr_mul(p2, p2), p2); v[2] = fr_mul(fr_mul(d, d), d);
        }
    }
}

template <int OP, class LD>
__device__ __forceinline__ void ew_fold_body(const LD& ld, const SplitEqView& E, size_t n_groups, const EwConsts& C, Fr* acc) {
    constexpr int NQ = ew_outputs(OP);
#pragma unroll
    for (int k = 0; k < NQ; k++) acc[k] = fe_zero();
    for (size_t g = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; g < n_groups; g += (size_t)gridDim.x * RA_THREADS) {
        Fr v[NQ];
        ew_terms<OP>(ld, g, C, v);
        if constexpr (ew_has_eq(OP)) {
            const Fr w = gse_weight(E, g);
#pragma unroll
            for (int k = 0; k < NQ; k++) acc[k] = fr_add(acc[k], fr_mul(w, v[k]));
        } else {
#pragma unroll
            for (int k = 0; k < NQ; k++) acc[k] = fr_add(acc[k], v[k]);
        }
    }
}

// partials[block][k], k < ew_outputs(OP): the block's share of sum_g E_out E_in f_k(g)
template <int OP>
__global__ __launch_bounds__(RA_THREADS) void k_ew_fold(const Fr* __restrict__ rows, size_t stride, SplitEqView E, size_t n_groups,
                                                        EwConsts C, Fr* __restrict__ partials) {
    constexpr int NQ = ew_outputs(OP);
    Fr acc[NQ];
    ew_fold_body<OP>(EwDirect{rows, stride}, E, n_groups, C, acc);
    block_reduce_store<NQ>(acc, partials);
}

// the same over the round channel: (bind_prev: wait for the previous round's challenge, bind every row into dst,)
// fold the bound pairs, mail the workgroup's sums to the host
template <int OP>
__global__ __launch_bounds__(RA_THREADS) void k_ew_fold_ch(const Fr* src, size_t sstride, Fr* dst, size_t dstride, SplitEqView E, size_t n_groups,
                                                           EwConsts C, ChanIo io, int bind_prev, int hi_only) {
    constexpr int NQ = ew_outputs(OP);
    Fr acc[NQ];
    if (bind_prev) {
        Fr r;
        if (!io.challenge(r)) return;
        ew_fold_body<OP>(EwBind{src, sstride, dst, dstride, r, hi_only != 0}, E, n_groups, C, acc);
    } else {
        ew_fold_body<OP>(EwDirect{src, sstride}, E, n_groups, C, acc);
    }
    io.template emit<NQ>(acc);
}

// the last bind (rows of two coefficients -> the final claims), mailed one value per record
__global__ __launch_bounds__(64) void k_ew_final_ch(const Fr* src, size_t sstride, uint32_t n_rows, ChanIo io, int hi_only) {
    __shared__ uint32_t stage[9 * RA_MAX_D];
    Fr r;
    if (!io.challenge(r)) return;
    Fr v = fe_zero();
    if (threadIdx.x < n_rows) v = bind_pair(fe_load(src + (size_t)threadIdx.x * sstride), fe_load(src + (size_t)threadIdx.x * sstride + 1), r, hi_only != 0);
    ch_mail_wave_fe(io.io, 0, n_rows, v, stage);
}

struct Elementwise : atlas_instance {
    int op = 0;
    EwConsts consts{};
    RaRows rows;
    GseDev eq;
    size_t n_vars = 0, round_next = 0;
    ~Elementwise() override { rows.release(); eq.release(); }
    size_t rounds() const override { return n_vars; }
    size_t degree() const override { return op == EW_CUBE ? 4 : !ew_has_eq(op) ? 2 : ew_outputs(op) + 1; }
    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& coeffs) override {
        H::Fr s[3];
        int rc = local_sums(round, s);
        if (rc) return rc;
        return finish(round, claim, s, coeffs);
    }
    // the round's fold over THIS instance's rows: q_constant (, q_quadratic (, ...)) before the Gruen finish — what a rank of a sharded
    // instance contributes to the round (atlas_elementwise_prove_sharded)
    int local_sums(size_t round, H::Fr* s) {
        if (round != round_next || round >= n_vars) return fail(ATLAS_ESTATE, "elementwise: round out of order");
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        const size_t n_groups = rows.len / 2;
        size_t blocks = (n_groups + RA_THREADS - 1) / RA_THREADS; if (blocks > 1024) blocks = 1024;
        const SplitEqView E = ew_has_eq(op) ? eq.view() : SplitEqView{nullptr, nullptr, 0};
        const Fr* src = rows.buf[rows.cur]; const size_t st = rows.stride[rows.cur];
        switch (op) {
#define EW_CASE(OP) case OP: k_ew_fold<OP><<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(src, st, E, n_groups, consts, rows.partials); break;
            EW_CASE(EW_ADD) EW_CASE(EW_SUB) EW_CASE(EW_NEG) EW_CASE(EW_SQUARE) EW_CASE(EW_IFF) EW_CASE(EW_MUL) EW_CASE(EW_CUBE) EW_CASE(EW_DIV) EW_CASE(EW_RSQRT) EW_CASE(EW_DOT) EW_CASE(EW_GATHER) EW_CASE(EW_HAMMING_BOOL) EW_CASE(EW_TELEPORT_DIV)
#undef EW_CASE
            default: return fail(ATLAS_EINVAL, "elementwise: unknown operator");
        }
        return rows.reduce_to_host((uint32_t)blocks, (uint32_t)ew_outputs(op), s);
    }
    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= n_vars) return fail(ATLAS_ESTATE, "elementwise: round out of order");
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        int rc = rows.bind(r);
        if (rc) return rc;
        if (ew_has_eq(op)) eq.st.bind(H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode));
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override {
        if (round_next != n_vars) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        if (have_finals) { out = mailed_finals; return ATLAS_OK; }
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        return rows.finals(out);
    }

    // ---- round-channel stepping: one fused launch per round (instance.hpp)
    bool have_finals = false;
    std::vector<H::Fr> mailed_finals;
    bool pipelined() const override { return n_vars >= 1; }
    bool wide_wait(size_t) const override { return false; }     // at most 256 workgroups per round (one mail record each), one for the finals
    // rows of round k live in buf[k & 1] with stride T >> k (RaRows::bind keeps rows compact)
    int enqueue(size_t round, const atlas::RoundIo& io, bool bind_prev, atlas_mail_ref& mail) override {
        if (round >= n_vars || (round > 0) != bind_prev) return fail(ATLAS_ESTATE, "elementwise: enqueue out of order");
        const size_t T = (size_t)1 << n_vars;
        const size_t len = T >> round, n_groups = len / 2;
        size_t blocks = (n_groups + RA_THREADS - 1) / RA_THREADS; if (blocks > 256) blocks = 256;     // one mail record per workgroup
        SplitEqView E{nullptr, nullptr, 0};
        if (ew_has_eq(op)) { size_t ot, it; eq.st.tops_after(round, ot, it); E = eq.view_at(ot, it); }
        const Fr* src = bind_prev ? rows.buf[(round - 1) & 1] : rows.buf[0];
        const size_t sst = bind_prev ? (T >> (round - 1)) : T;
        Fr* dst = rows.buf[round & 1];
        const ChanIo cio{io, rt().challenge_mode};
        const int hi = rt().challenge_mode == 0 ? 1 : 0;
        switch (op) {
#define EW_CASE(OP) case OP: k_ew_fold_ch<OP><<<(unsigned)blocks, RA_THREADS, 0, rt().stream>>>(src, sst, dst, len, E, n_groups, consts, cio, bind_prev ? 1 : 0, hi); break;
            EW_CASE(EW_ADD) EW_CASE(EW_SUB) EW_CASE(EW_NEG) EW_CASE(EW_SQUARE) EW_CASE(EW_IFF) EW_CASE(EW_MUL) EW_CASE(EW_CUBE) EW_CASE(EW_DIV) EW_CASE(EW_RSQRT) EW_CASE(EW_DOT) EW_CASE(EW_GATHER) EW_CASE(EW_HAMMING_BOOL) EW_CASE(EW_TELEPORT_DIV)
#undef EW_CASE
            default: return fail(ATLAS_EINVAL, "elementwise: unknown operator");
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "elementwise: launch", e);
        mail.base = io.mail; mail.blocks = blocks; mail.n_vals = ew_outputs(op); mail.radix = 32; mail.shl = 0;
        return ATLAS_OK;
    }
    void prepare(size_t round) override { if (ew_has_eq(op) && round == round_next) eq.st.prepare_inverses(ew_outputs(op) == 3 /* cube: 1 - w */); }
    int finish(size_t round, const H::Fr& claim, const H::Fr* s, std::vector<H::Fr>& coeffs) override {
        if (round != round_next || round >= n_vars) return fail(ATLAS_ESTATE, "elementwise: round out of order");
        const int nq = ew_outputs(op);
        if (!ew_has_eq(op)) {
            coeffs.assign(3, H::zero());
            H::unipoly_from_evals_and_hint(claim, s, 2, coeffs.data());
        } else if (op == EW_CUBE) {
            std::vector<H::Fr> sums(3);
            for (int k = 0; k < 3; k++) sums[k] = H::mul(s[k], eq.st.scalar);        // mles_product_sum.rs:131
            coeffs = H::finish_product_sum(sums, claim, eq.st);
        } else if (nq == 1) {
            coeffs.assign(3, H::zero());
            H::gruen_deg2(eq.st, s[0], claim, coeffs.data());
        } else {
            coeffs.assign(4, H::zero());
            H::gruen_deg3(eq.st, s[0], s[1], claim, coeffs.data());
        }
        return ATLAS_OK;
    }
    int host_ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= n_vars) return fail(ATLAS_ESTATE, "elementwise: round out of order");
        if (ew_has_eq(op)) eq.st.bind(H::challenge_to_fr(r.lo, r.hi, rt().challenge_mode));
        rows.cur = (int)((round + 1) & 1); rows.len = ((size_t)1 << n_vars) >> (round + 1); rows.stride[rows.cur] = rows.len;
        round_next++;
        return ATLAS_OK;
    }
    int enqueue_finals(const atlas::RoundIo& io, atlas_mail_ref& mail) override {
        const size_t T = (size_t)1 << n_vars;
        const Fr* src = rows.buf[(n_vars - 1) & 1];
        k_ew_final_ch<<<1, 64, 0, rt().stream>>>(src, n_vars == 1 ? T : (T >> (n_vars - 1)), (uint32_t)rows.d, ChanIo{io, rt().challenge_mode}, rt().challenge_mode == 0 ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "elementwise: launch", e);
        mail.base = io.mail; mail.blocks = 1; mail.n_vals = (int)rows.d; mail.radix = 32; mail.shl = 0;
        return ATLAS_OK;
    }
    int set_finals(const H::Fr* vals, size_t n) override {
        if (n != rows.d) return fail(ATLAS_EINVAL, "elementwise: final claims");
        mailed_finals.assign(vals, vals + n); have_finals = true;
        return ATLAS_OK;
    }
};

// A member of a BatchedSumcheck that has NO rounds of its own (an instance over zero variables: HammingBooleanity of a gather with ONE index):
// the batch absorbs its input claim, draws its coefficient, scales the claim by 2^max_rounds and never starts it (sumcheck.rs:620-700); what
// is left of the instance is its final claims.
struct ConstMember : atlas_instance {
    size_t deg = 1;
    std::vector<H::Fr> vals;
    size_t rounds() const override { return 0; }
    size_t degree() const override { return deg; }
    int message(size_t, const H::Fr&, std::vector<H::Fr>&) override { return fail(ATLAS_ESTATE, "a member without rounds has no round polynomial"); }
    int ingest(const atlas_u128_t&, size_t) override { return fail(ATLAS_ESTATE, "a member without rounds binds nothing"); }
    int finals(std::vector<H::Fr>& out) override { out = vals; return ATLAS_OK; }
};

}  // namespace

int atlas_rt_const_member_new(const atlas_fr_t* finals, size_t n, size_t degree, atlas_instance_t* out) {
    if (!finals || !n || !out) return fail(ATLAS_EINVAL, "const_member_new: null argument");
    ConstMember* P = new ConstMember();
    P->deg = degree;
    P->vals.assign(reinterpret_cast<const H::Fr*>(finals), reinterpret_cast<const H::Fr*>(finals) + n);
    *out = P;
    return ATLAS_OK;
}

extern "C" {

int atlas_elementwise_new(int op, const atlas_poly_t* operands, size_t n_operands, const atlas_fr_t* r_node_output, size_t n_vars,
                          const atlas_fr_t* constants, size_t n_constants, atlas_instance_t* out) {
    PROF("atlas_elementwise_new");
    NEED_INIT();
    if (!operands || !out || (!r_node_output && op != EW_DOT && op != EW_GATHER)) return fail(ATLAS_EINVAL, "elementwise_new: null argument");
    if (op < EW_ADD || op > EW_TELEPORT_DIV) return fail(ATLAS_EINVAL, "elementwise_new: unknown operator");
    if (n_constants != (op == EW_RSQRT ? 2u : (op == EW_GATHER || op == EW_TELEPORT_DIV) ? 1u : op == EW_HAMMING_BOOL ? n_operands : 0u) || (n_constants && !constants))
        return fail(ATLAS_EINVAL, "elementwise_new: constants are (S^3, gamma) for Rsqrt, (gamma) for Gather, (tau) for TeleportDivision, one gamma power per operand for HammingBooleanity");
    if (op == EW_DOT ? (n_operands < 2 || n_operands % 2 || n_operands > RA_MAX_D)
        : op == EW_HAMMING_BOOL ? (n_operands < 1 || n_operands > RA_MAX_D) : n_operands != (size_t)ew_operands(op))
        return fail(ATLAS_EINVAL, "elementwise_new: wrong operand count for the operator");
    if (n_vars == 0 || n_vars > 25) return fail(ATLAS_EINVAL, "elementwise_new: 1 <= n_vars <= 25");
    const size_t T = (size_t)1 << n_vars;
    for (size_t i = 0; i < n_operands; i++)
        if (!operands[i] || operands[i]->len != T) return fail(ATLAS_EINVAL, "elementwise_new: operand length != 2^n_vars");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    Elementwise* P = new Elementwise();
    P->op = op; P->n_vars = n_vars;
    for (size_t i = 0; i < n_constants; i++) std::memcpy(&P->consts.k[i], &constants[i], 32);
    P->consts.n_terms = (uint32_t)(op == EW_HAMMING_BOOL ? n_operands : n_operands / 2);
    int rc = P->rows.alloc(n_operands, T, 3);
    if (!rc) {
        size_t gb = (T + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
        hipError_t e = hipSuccess;
        for (size_t i = 0; i < n_operands && e == hipSuccess; i++) {
            if (operands[i]->is_i32) k_ew_from_i32<<<(unsigned)gb, RA_THREADS, 0, rt().stream>>>((const int32_t*)operands[i]->d, P->rows.buf[0] + i * T, T);
            else e = hipMemcpyAsync(P->rows.buf[0] + i * T, operands[i]->d, T * sizeof(Fr), hipMemcpyDeviceToDevice, rt().stream);
        }
        if (e != hipSuccess) rc = fail(ATLAS_ENODEV, "elementwise_new: operand copy", e);
    }
    if (!rc && ew_has_eq(op)) rc = P->eq.init(reinterpret_cast<const H::Fr*>(r_node_output), n_vars);
    if (rc) { delete P; return rc; }
    *out = P;
    return ATLAS_OK;
}

// One element-wise operator sumcheck (LowToHigh over the Gruen split-eq: Mul, Add, Sub, Square, Iff, Div, ... — mul.rs:160-199) sharded over
// the ranks of a group by CONTIGUOUS blocks: `inst` is this rank's instance over its block of 2^(n - log2 world) coefficients,
// built with the LOW coordinates of the opening point (r_node_output[log2 world ..]); r_high = the leading log2 world coordinates.  LowToHigh
// pairs (2i, 2i + 1) stay on a rank, so the first n - log2 world rounds exchange only the round's sums: a rank's sums weigh eq(r_high, rank)
// in the total (the eq table factorises), every rank adds the world records in rank order and runs the same Gruen finish and transcript step.
// Then the world final values of each operand cross the board once and the last log2 world rounds run on every rank over those (a small
// instance that inherits the split-eq scalar).  Same proof on every rank, the same bytes as the unsharded instance.
int atlas_elementwise_prove_sharded(atlas_instance_t inst, atlas_shard_group_t grp, const atlas_fr_t* r_high, const atlas_fr_t* input_claim,
                                    atlas_transcript_t* transcript, atlas_fr_t* compressed, size_t row_stride, uint32_t* n_coeffs, atlas_u128_t* challenges,
                                    atlas_fr_t* finals, size_t finals_cap, size_t* n_finals) {
    NEED_INIT();
    Elementwise* P = dynamic_cast<Elementwise*>(inst);
    if (!P || !grp || !input_claim || !transcript || !compressed || !n_coeffs || !challenges || !finals || !n_finals) return fail(ATLAS_EINVAL, "elementwise_prove_sharded: null argument / not an element-wise instance");
    if (!ew_has_eq(P->op)) return fail(ATLAS_EINVAL, "elementwise_prove_sharded: the selector-style operators (no eq factor) are not sharded");
    const size_t world = (size_t)grp->world, rank = (size_t)grp->rank;
    if (world == 0 || (world & (world - 1)) || rank >= world) return fail(ATLAS_EINVAL, "elementwise_prove_sharded: the world size must be a power of two (contiguous blocks of a hypercube)");
    size_t lw = 0; while (((size_t)1 << lw) < world) lw++;
    if (lw && !r_high) return fail(ATLAS_EINVAL, "elementwise_prove_sharded: r_high");
    const int nq = ew_outputs(P->op);
    if (row_stride < P->degree()) return fail(ATLAS_EINVAL, "elementwise_prove_sharded: row_stride below the degree");
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(transcript);
    const H::Fr* rh = reinterpret_cast<const H::Fr*>(r_high);
    H::Fr weight = H::one();                                                  // eq(r_high, bits(rank)), r_high[0] = the most significant bit
    for (size_t q = 0; q < lw; q++) weight = H::mul(weight, ((rank >> (lw - 1 - q)) & 1) ? rh[q] : H::sub(H::one(), rh[q]));
    H::Fr prev = *reinterpret_cast<const H::Fr*>(input_claim);
    H::tr_append_scalar(T, prev);
    std::vector<H::Fr> c;
    size_t out_round = 0;
    auto step = [&](Elementwise* I, size_t round, bool exchange) -> int {
        H::Fr s[3], tot[3];
        int rc = I->local_sums(round, s);
        if (rc) return rc;
        if (exchange) {
            H::Fr mine[3], all[atlas_shard_group::MAX_WORLD * 3];
            for (int k = 0; k < nq; k++) mine[k] = H::mul(s[k], weight);
            rc = atlas_shard_allgather(grp, mine, (size_t)nq * 32, all);
            if (rc) return rc;
            for (int k = 0; k < nq; k++) { tot[k] = H::zero(); for (size_t r = 0; r < world; r++) tot[k] = H::add(tot[k], all[r * nq + k]); }
        } else for (int k = 0; k < nq; k++) tot[k] = s[k];
        I->prepare(round);
        rc = I->finish(round, prev, tot, c);
        if (rc) return rc;
        std::vector<H::Fr> cc;
        if (c.size() < 2) cc = c;
        else { cc.push_back(c[0]); for (size_t k = 2; k < c.size(); k++) cc.push_back(c[k]); }
        if (cc.size() > row_stride) return fail(ATLAS_EINVAL, "elementwise_prove_sharded: row_stride below the degree");
        H::tr_append_message(T, "UniPoly_begin");
        for (auto& x : cc) H::tr_append_scalar(T, x);
        H::tr_append_message(T, "UniPoly_end");
        n_coeffs[out_round] = (uint32_t)cc.size();
        std::memcpy(&compressed[out_round * row_stride], cc.data(), cc.size() * 32);
        uint64_t lo, hi;
        H::tr_challenge_u128(T, lo, hi);
        challenges[out_round].lo = lo; challenges[out_round].hi = hi;
        const H::Fr rf = H::challenge_to_fr(lo, hi, rt().challenge_mode);
        H::Fr ev = c[0], pw = rf;                                             // UniPoly::evaluate
        for (size_t i = 1; i < c.size(); i++) { ev = H::add(ev, H::mul(pw, c[i])); pw = H::mul(pw, rf); }
        prev = ev;
        out_round++;
        return I->ingest(challenges[out_round - 1], round);
    };
    for (size_t round = 0; round < P->n_vars; round++) { int rc = step(P, round, world > 1); if (rc) return rc; }
    std::vector<H::Fr> f;
    int rc = P->finals(f);
    if (rc) return rc;
    if (world == 1) {
        if (f.size() > finals_cap) return fail(ATLAS_EINVAL, "elementwise_prove_sharded: finals buffer");
        std::memcpy(finals, f.data(), f.size() * 32); *n_finals = f.size();
        return ATLAS_OK;
    }
    // the operands' remaining world values (rank b holds index b: contiguous blocks), a small instance over them on every rank
    const size_t n_ops = f.size();
    if (n_ops * 32 > atlas_shard_group::PAYLOAD) return fail(ATLAS_EINVAL, "elementwise_prove_sharded: too many operands for one record");
    std::vector<H::Fr> all(world * n_ops);
    rc = atlas_shard_allgather(grp, f.data(), n_ops * 32, all.data());
    if (rc) return rc;
    std::vector<atlas_poly_t> polys(n_ops, nullptr);
    for (size_t o = 0; o < n_ops && !rc; o++) {
        std::vector<H::Fr> col(world);
        for (size_t r = 0; r < world; r++) col[r] = all[r * n_ops + o];
        rc = atlas_poly_upload_fr((const atlas_fr_t*)col.data(), world, &polys[o]);
    }
    const size_t n_consts = P->op == EW_RSQRT ? 2 : (P->op == EW_GATHER || P->op == EW_TELEPORT_DIV) ? 1 : P->op == EW_HAMMING_BOOL ? n_ops : 0;
    atlas_instance_t tail_h = nullptr;
    if (!rc) rc = atlas_elementwise_new(P->op, polys.data(), n_ops, r_high, lw, n_consts ? (const atlas_fr_t*)P->consts.k : nullptr, n_consts, &tail_h);
    for (atlas_poly_t q : polys) if (q) atlas_poly_free(q);
    if (rc) return rc;
    Elementwise* Tl = static_cast<Elementwise*>(tail_h);
    Tl->eq.st.scalar = P->eq.st.scalar;                                       // the eq factors of the variables bound so far
    for (size_t round = 0; round < lw && !rc; round++) rc = step(Tl, round, false);
    if (!rc) rc = Tl->finals(f);
    atlas_instance_free(tail_h);
    if (rc) return rc;
    if (f.size() > finals_cap) return fail(ATLAS_EINVAL, "elementwise_prove_sharded: finals buffer");
    std::memcpy(finals, f.data(), f.size() * 32); *n_finals = f.size();
    return ATLAS_OK;
}

}  // extern "C"
