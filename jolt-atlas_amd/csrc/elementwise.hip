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

// partials[block][k], k < ew_outputs(OP): the block's share of sum_g E_out E_in f_k(g)
template <int OP>
__global__ __launch_bounds__(RA_THREADS) void k_ew_fold(const Fr* __restrict__ rows, size_t stride, SplitEqView E, size_t n_groups,
                                                        EwConsts C, Fr* __restrict__ partials) {
    constexpr int NQ = ew_outputs(OP);
    Fr acc[NQ];
#pragma unroll
    for (int k = 0; k < NQ; k++) acc[k] = fe_zero();
    for (size_t g = (size_t)blockIdx.x * RA_THREADS + threadIdx.x; g < n_groups; g += (size_t)gridDim.x * RA_THREADS) {
        const Fr a0 = fe_load(rows + 2 * g), a1 = fe_load(rows + 2 * g + 1);
        Fr v[NQ];
        if constexpr (OP == EW_ADD) v[0] = fr_add(a0, fe_load(rows + stride + 2 * g));
        else if constexpr (OP == EW_SUB) v[0] = fr_sub(a0, fe_load(rows + stride + 2 * g));
        else if constexpr (OP == EW_NEG) v[0] = fr_sub(fe_zero(), a0);
        else if constexpr (OP == EW_TELEPORT_DIV)      // rows: input, quotient, remainder: tau q0 + r0 - inp0
            v[0] = fr_sub(fr_add(fr_mul(C.k[0], fe_load(rows + stride + 2 * g)), fe_load(rows + 2 * stride + 2 * g)), a0);
        else if constexpr (OP == EW_SQUARE) { const Fr d = fr_sub(a1, a0); v[0] = fr_mul(a0, a0); v[1] = fr_mul(d, d); }
        else if constexpr (OP == EW_MUL) {
            const Fr b0 = fe_load(rows + stride + 2 * g), b1 = fe_load(rows + stride + 2 * g + 1);
            v[0] = fr_mul(a0, b0); v[1] = fr_mul(fr_sub(a1, a0), fr_sub(b1, b0));
        } else if constexpr (OP == EW_IFF) {        // rows: mask, a, b.  c0 = b0 + m0 (a0 - b0);  e = m_inf (a_inf - b_inf)
            const Fr x0 = fe_load(rows + stride + 2 * g), x1 = fe_load(rows + stride + 2 * g + 1);
            const Fr y0 = fe_load(rows + 2 * stride + 2 * g), y1 = fe_load(rows + 2 * stride + 2 * g + 1);
            v[0] = fr_add(y0, fr_mul(a0, fr_sub(x0, y0)));
            v[1] = fr_mul(fr_sub(a1, a0), fr_sub(fr_sub(x1, x0), fr_sub(y1, y0)));
        } else if constexpr (OP == EW_DOT) {        // rows: (input_t, selector_t) pairs; values at X = 0 and X = 2
            v[0] = fe_zero(); v[1] = fe_zero();
            for (uint32_t tm = 0; tm < C.n_terms; tm++) {
                const Fr* in = rows + (size_t)(2 * tm) * stride; const Fr* sel = in + stride;
                const Fr i0 = fe_load(in + 2 * g), i1 = fe_load(in + 2 * g + 1), s0 = fe_load(sel + 2 * g), s1 = fe_load(sel + 2 * g + 1);
                v[0] = fr_add(v[0], fr_mul(i0, s0));
                v[1] = fr_add(v[1], fr_mul(fr_add(i1, fr_sub(i1, i0)), fr_add(s1, fr_sub(s1, s0))));
            }
        } else if constexpr (OP == EW_GATHER) {     // rows: ra, dictionary, identity: ra (dict + gamma id) at X = 0 and 2
            const Fr d0 = fe_load(rows + stride + 2 * g), d1 = fe_load(rows + stride + 2 * g + 1);
            const Fr j0 = fe_load(rows + 2 * stride + 2 * g), j1 = fe_load(rows + 2 * stride + 2 * g + 1);
            const Fr a2 = fr_add(a1, fr_sub(a1, a0)), d2 = fr_add(d1, fr_sub(d1, d0)), j2 = fr_add(j1, fr_sub(j1, j0));
            v[0] = fr_mul(a0, fr_add(d0, fr_mul(C.k[0], j0)));
            v[1] = fr_mul(a2, fr_add(d2, fr_mul(C.k[0], j2)));
        } else if constexpr (OP == EW_HAMMING_BOOL) {   // rows hw_d: sum_d gamma_d [hw0 (hw0 - 1), (hw1 - hw0)^2]
            v[0] = fe_zero(); v[1] = fe_zero();
            for (uint32_t d = 0; d < C.n_terms; d++) {
                const Fr h0 = fe_load(rows + (size_t)d * stride + 2 * g), h1 = fe_load(rows + (size_t)d * stride + 2 * g + 1);
                const Fr a = fr_sub(h1, h0);
                v[0] = fr_add(v[0], fr_mul(C.k[d], fr_mul(h0, fr_sub(h0, fr_one()))));
                v[1] = fr_add(v[1], fr_mul(C.k[d], fr_mul(a, a)));
            }
        } else if constexpr (OP == EW_DIV) {        // rows: left, right, q, R.  c0 = ro0 q0 + R0 - lo0;  e = ro_inf q_inf
            const Fr r0 = fe_load(rows + stride + 2 * g), r1 = fe_load(rows + stride + 2 * g + 1);
            const Fr q0 = fe_load(rows + 2 * stride + 2 * g), q1 = fe_load(rows + 2 * stride + 2 * g + 1);
            v[0] = fr_sub(fr_add(fr_mul(r0, q0), fe_load(rows + 3 * stride + 2 * g)), a0);
            v[1] = fr_mul(fr_sub(r1, r0), fr_sub(q1, q0));
        } else if constexpr (OP == EW_RSQRT) {      // rows: input, quotient, output, div_rem, sqrt_rem
            const Fr q0 = fe_load(rows + stride + 2 * g), q1 = fe_load(rows + stride + 2 * g + 1);
            const Fr o0 = fe_load(rows + 2 * stride + 2 * g), o1 = fe_load(rows + 2 * stride + 2 * g + 1);
            const Fr div0 = fr_sub(fr_add(fr_mul(a0, q0), fe_load(rows + 3 * stride + 2 * g)), C.k[0]);
            const Fr sqrt0 = fr_sub(fr_add(fr_mul(o0, o0), fe_load(rows + 4 * stride + 2 * g)), q0);
            const Fr od = fr_sub(o1, o0);
            v[0] = fr_add(div0, fr_mul(C.k[1], sqrt0));
            v[1] = fr_add(fr_mul(fr_sub(a1, a0), fr_sub(q1, q0)), fr_mul(C.k[1], fr_mul(od, od)));
        } else {                                    // cube: p(1)^3, p(2)^3, p_inf^3
            const Fr d = fr_sub(a1, a0), p2 = fr_add(a1, d);
            v[0] = fr_mul(fr_mul(a1, a1), a1); v[1] = fr_mul(fr_mul(p2, p2), p2); v[2] = fr_mul(fr_mul(d, d), d);
        }
        if constexpr (ew_has_eq(OP)) {
            const Fr w = gse_weight(E, g);
#pragma unroll
            for (int k = 0; k < NQ; k++) acc[k] = fr_add(acc[k], fr_mul(w, v[k]));
        } else {
#pragma unroll
            for (int k = 0; k < NQ; k++) acc[k] = fr_add(acc[k], v[k]);
        }
    }
    block_reduce_store<NQ>(acc, partials);
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
        if (round != round_next || round >= n_vars) return fail(ATLAS_ESTATE, "elementwise: round out of order");
        std::lock_guard<std::mutex> lk(g.mu);
        const size_t n_groups = rows.len / 2;
        size_t blocks = (n_groups + RA_THREADS - 1) / RA_THREADS; if (blocks > 1024) blocks = 1024;
        const SplitEqView E = ew_has_eq(op) ? eq.view() : SplitEqView{nullptr, nullptr, 0};
        const Fr* src = rows.buf[rows.cur]; const size_t st = rows.stride[rows.cur];
        switch (op) {
#define EW_CASE(OP) case OP: k_ew_fold<OP><<<(unsigned)blocks, RA_THREADS, 0, g.stream>>>(src, st, E, n_groups, consts, rows.partials); break;
            EW_CASE(EW_ADD) EW_CASE(EW_SUB) EW_CASE(EW_NEG) EW_CASE(EW_SQUARE) EW_CASE(EW_IFF) EW_CASE(EW_MUL) EW_CASE(EW_CUBE) EW_CASE(EW_DIV) EW_CASE(EW_RSQRT) EW_CASE(EW_DOT) EW_CASE(EW_GATHER) EW_CASE(EW_HAMMING_BOOL) EW_CASE(EW_TELEPORT_DIV)
#undef EW_CASE
            default: return fail(ATLAS_EINVAL, "elementwise: unknown operator");
        }
        const int nq = ew_outputs(op);
        H::Fr s[3];
        int rc = rows.reduce_to_host((uint32_t)blocks, (uint32_t)nq, s);
        if (rc) return rc;
        if (!ew_has_eq(op)) {
            coeffs.assign(3, H::zero());
            H::unipoly_from_evals_and_hint(claim, s, 2, coeffs.data());
        } else if (op == EW_CUBE) {
            std::vector<H::Fr> sums(3);
            for (int k = 0; k < 3; k++) sums[k] = H::mul(s[k], eq.st.scalar);        // mles_product_sum.rs:131
            coeffs = H::finish_product_sum(sums, claim, eq.st);
        } else if (nq == 1) {
            coeffs.assign(3, H::zero());
            H::gruen_deg2(eq.st.scalar, eq.st.w_cur(), s[0], claim, coeffs.data());
        } else {
            coeffs.assign(4, H::zero());
            H::gruen_deg3(eq.st, s[0], s[1], claim, coeffs.data());
        }
        return ATLAS_OK;
    }
    int ingest(const atlas_u128_t& r, size_t round) override {
        if (round != round_next || round >= n_vars) return fail(ATLAS_ESTATE, "elementwise: round out of order");
        std::lock_guard<std::mutex> lk(g.mu);
        int rc = rows.bind(r);
        if (rc) return rc;
        if (ew_has_eq(op)) eq.st.bind(H::challenge_to_fr(r.lo, r.hi, g.challenge_mode));
        round_next++;
        return ATLAS_OK;
    }
    int finals(std::vector<H::Fr>& out) override {
        if (round_next != n_vars) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
        std::lock_guard<std::mutex> lk(g.mu);
        return rows.finals(out);
    }
};

}  // namespace

extern "C" {

int atlas_elementwise_new(int op, const atlas_poly_t* operands, size_t n_operands, const atlas_fr_t* r_node_output, size_t n_vars,
                          const atlas_fr_t* constants, size_t n_constants, atlas_instance_t* out) {
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
    std::lock_guard<std::mutex> lk(g.mu);
    Elementwise* P = new Elementwise();
    P->op = op; P->n_vars = n_vars;
    for (size_t i = 0; i < n_constants; i++) std::memcpy(&P->consts.k[i], &constants[i], 32);
    P->consts.n_terms = (uint32_t)(op == EW_HAMMING_BOOL ? n_operands : n_operands / 2);
    int rc = P->rows.alloc(n_operands, T, 3);
    if (!rc) {
        size_t gb = (T + RA_THREADS - 1) / RA_THREADS; if (gb > 4096) gb = 4096;
        hipError_t e = hipSuccess;
        for (size_t i = 0; i < n_operands && e == hipSuccess; i++) {
            if (operands[i]->is_i32) k_ew_from_i32<<<(unsigned)gb, RA_THREADS, 0, g.stream>>>((const int32_t*)operands[i]->d, P->rows.buf[0] + i * T, T);
            else e = hipMemcpyAsync(P->rows.buf[0] + i * T, operands[i]->d, T * sizeof(Fr), hipMemcpyDeviceToDevice, g.stream);
        }
        if (e != hipSuccess) rc = fail(ATLAS_ENODEV, "elementwise_new: operand copy", e);
    }
    if (!rc && ew_has_eq(op)) rc = P->eq.init(reinterpret_cast<const H::Fr*>(r_node_output), n_vars);
    if (rc) { delete P; return rc; }
    *out = P;
    return ATLAS_OK;
}

}  // extern "C"
