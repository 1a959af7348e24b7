// One real operator node through the C-ABI: the fused-rescale Einsum (mk,kn->mn) exactly as Einsum::prove composes it
// (jolt-atlas-core/src/onnx_proof/ops/einsum/mod.rs:71-115 with fused_rebase.rs:215-279):
//   witness   acc = A B in i64, rescaled = acc >> S, R = acc mod 2^S, output = SatClamp_i32(rescaled)
//             (try_rebase_intermediates, fused_rebase.rs:110-128 -> einsum_intermediate_and_remainder) — on the device
//   prove_pre cache_remainder_prove: append R(r0) as advice (:350-360)
//             prove_clamp_lookup (clamp_lookups/mod.rs:264-309): append acc(r0); gamma; PS-Shout read-raf over
//             SaturationTable = ClampBoundedTable<64, 31, symmetric> (ProofType::Execution); ra_onehot_provers
//             (shout.rs:399-466: gamma_powers, gammas, r_address) and BatchedSumcheck [RaVirtual, HammingWeight,
//             Booleanity] (ProofType::RaOneHotChecks)
//   matmul    EinsumLayout::fold at r0 = (r_m | r_n), the degree-2 dot-product sumcheck with input claim
//             rescaled(r0) 2^S + R(r0) (fused_input_claim, :168-184), operand openings appended (dot.rs:377-400)
//   prove_remainder_rc (:253-285): identity range check R in [0, 2^S) (ProofType::RangeCheck), then the same three
//             one-hot checks over the remainder's S-bit addresses (ProofType::RescaleRemainderRaChecks)
// Every scalar the reference appends to the accumulator (and hence the transcript) is appended here in the same order;
// the five SumcheckInstanceProofs come back ark-serialized in the order Einsum::prove returns them.  Host glue over the
// library's own entry points; the only kernels here are the witness ones.  This is the first stack of the bricks
// toward ONNXProof::prove: bench.py reports its wall clock as `node_einsum_ms`.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <vector>

#include "../../include/atlas_hip.h"
#include "field.hip.h"
#include "host_field.hpp"
#include "internal.hpp"
#include "runtime.hpp"

using namespace atlas;
namespace H = atlas_host;
using atlas_rt::fail;
using atlas_rt::g;

namespace {

// acc[i][j] = sum_l A[i][l] B[l][j] (i64).  A thread owns column j for a tile of EB_ROWS rows of A and a slice of the
// contraction: B[l][j] is read once per row tile (coalesced across j), A[i][l] is the same address for the whole
// wavefront; the k-slices meet through 64-bit atomic adds.  (One thread per output element re-read B once per row of A:
// 275 us for 16 x 1024 . 1024 x 4096.)
constexpr int EB_ROWS = 8;
__global__ __launch_bounds__(256) void k_einsum_acc_mk_kn(const int32_t* __restrict__ A, const int32_t* __restrict__ B, uint32_t m, uint32_t k,
                                                          uint32_t n, uint32_t k_slice, unsigned long long* __restrict__ acc /* [m][n], zeroed */) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    const uint32_t i0 = blockIdx.y * EB_ROWS, l0 = blockIdx.z * k_slice;
    if (j >= n) return;
    const uint32_t l1 = l0 + k_slice < k ? l0 + k_slice : k;
    int64_t s[EB_ROWS];
#pragma unroll
    for (int r = 0; r < EB_ROWS; r++) s[r] = 0;
    for (uint32_t l = l0; l < l1; l++) {
        const int64_t b = (int64_t)B[(size_t)l * n + j];
#pragma unroll
        for (int r = 0; r < EB_ROWS; r++)
            if (i0 + r < m) s[r] += (int64_t)A[(size_t)(i0 + r) * k + l] * b;
    }
#pragma unroll
    for (int r = 0; r < EB_ROWS; r++)
        if (i0 + r < m && s[r]) atomicAdd(&acc[(size_t)(i0 + r) * n + j], (unsigned long long)s[r]);      // two's complement: wraps like i64
}
// try_rebase_intermediates on the accumulators: quotient / remainder by 2^S, the clamped i32 output, the lookup indices
__global__ __launch_bounds__(256) void k_einsum_rebase(int64_t* __restrict__ quot /* in: accumulators */, size_t T, uint32_t S, int32_t* __restrict__ rem,
                                                       int32_t* __restrict__ outp, uint64_t* __restrict__ clamp_idx, uint64_t* __restrict__ rem_idx) {
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < T; o += (size_t)gridDim.x * 256) {
        const int64_t acc = quot[o];
        const int64_t q = acc >> S;                                  // floor division by 2^S
        const int64_t r = acc - (q << S);                            // in [0, 2^S)
        quot[o] = q; rem[o] = (int32_t)r;
        const int64_t c = q > 2147483647ll ? 2147483647ll : q < -2147483648ll ? -2147483648ll : q;
        if (outp) outp[o] = (int32_t)c;
        clamp_idx[o] = (uint64_t)q;                                  // LookupBits::new(v as u64, 64)
        rem_idx[o] = (uint64_t)r;
    }
}
__global__ __launch_bounds__(256) void k_i64_to_fr(const int64_t* __restrict__ in, Fr* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) fe_store(out + i, fr_from_i64(in[i]));
}

// element-wise product accumulators of the fused-rescale Mul node
__global__ __launch_bounds__(256) void k_mul_acc(const int32_t* __restrict__ l, const int32_t* __restrict__ r, size_t n, int64_t* __restrict__ acc) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc[i] = (int64_t)l[i] * (int64_t)r[i];
}

// Add / Sub accumulators, their saturated i32 output and the clamp lookup indices (`acc as u64`, clamp_lookups/mod.rs:243-251)
__global__ __launch_bounds__(256) void k_addsub_witness(const int32_t* __restrict__ l, const int32_t* __restrict__ r, size_t n, int subtract,
                                                        int64_t* __restrict__ acc, int32_t* __restrict__ out, uint64_t* __restrict__ idx) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int64_t a = subtract ? (int64_t)l[i] - (int64_t)r[i] : (int64_t)l[i] + (int64_t)r[i];
        acc[i] = a;
        if (out) out[i] = (int32_t)(a > 2147483647ll ? 2147483647ll : a < -2147483648ll ? -2147483648ll : a);
        idx[i] = (uint64_t)a;
    }
}

// ReLU witness: the output tensor and the lookup indices `x as u32 as u64` (compute_lookup_indices_from_operands, utils/mod.rs:43-122)
__global__ __launch_bounds__(256) void k_relu_witness(const int32_t* __restrict__ x, size_t n, int32_t* __restrict__ out, uint64_t* __restrict__ idx) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int32_t v = x[i];
        if (out) out[i] = v > 0 ? v : 0;
        idx[i] = (uint64_t)(uint32_t)v;
    }
}

unsigned ilog2(size_t x) { unsigned n = 0; while (x > 1) { x >>= 1; n++; } return n; }
bool pow2(size_t x) { return x && !(x & (x - 1)); }

struct Out {
    uint8_t* proofs; size_t cap, len = 0; size_t* lens; int n_proofs = 0;
    atlas_fr_t* claims; size_t claims_cap, n_claims = 0;
    int put_claim(const H::Fr& c) {
        if (n_claims >= claims_cap) return fail(ATLAS_EINVAL, "prove_einsum_node: claims buffer too small");
        std::memcpy(&claims[n_claims++], &c, 32);
        return ATLAS_OK;
    }
    int put_proof(const std::vector<atlas_fr_t>& rows, size_t stride, const std::vector<uint32_t>& nco, size_t n_rounds) {
        size_t l = 0;
        int rc = atlas_sumcheck_proof_serialize(rows.data(), stride, nco.data(), n_rounds, proofs + len, cap - len, &l);
        if (rc) return rc;
        lens[n_proofs++] = l; len += l;
        return ATLAS_OK;
    }
};

// Sumcheck::prove of one instance + its single cache_openings claim (final claim 0)
int prove_single(atlas_instance_t inst, const H::Fr& input_claim, atlas_transcript_t* t, Out& O, std::vector<atlas_u128_t>& ch, H::Fr* final0) {
    const size_t n = atlas_instance_num_rounds(inst), stride = atlas_instance_degree(inst) + 1;
    std::vector<atlas_fr_t> rows(n * stride);
    std::vector<uint32_t> nco(n);
    ch.assign(n, atlas_u128_t{0, 0});
    int rc = atlas_instance_prove(inst, (const atlas_fr_t*)&input_claim, t, rows.data(), stride, nco.data(), ch.data());
    if (rc) return rc;
    atlas_fr_t fin[64]; size_t nf = 0;
    rc = atlas_instance_final_claims(inst, fin, 64, &nf);
    if (rc) return rc;
    std::memcpy(final0, &fin[0], 32);
    rc = atlas_transcript_append_scalar(t, &fin[0]);                 // cache_openings: append_virtual(.., ra.final_claim())
    if (!rc) rc = O.put_claim(*final0);
    if (!rc) rc = O.put_proof(rows, stride, nco, n);
    return rc;
}

// ra_onehot_provers (shout.rs:399-466) + BatchedSumcheck::prove over [RaVirtual, HammingWeight, Booleanity] and their
// cache_openings (ra_virtual.rs:157-184, hamming_weight.rs:149-168, booleanity.rs:350-368): d claims each
int prove_onehot_checks(const uint64_t* d_lookups, size_t log_T, size_t log_K, const atlas_fr_t* r_cycle, const std::vector<atlas_fr_t>& ra_point,
                        const H::Fr& ra_claim, atlas_transcript_t* t, Out& O) {
    const size_t lkc = 4, d = (log_K + lkc - 1) / lkc;               // OneHotParams::new: LOG_K_CHUNK = 4 (common/src/consts/general.rs:2)
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(t);
    const bool trace = getenv("ATLAS_TRACE") != nullptr;
    auto tr0 = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!trace) return;
        atlas_sync();
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[atlas trace] onehot_checks d=%zu %-24s %8.3f ms\n", d, what, std::chrono::duration<double, std::milli>(t1 - tr0).count());
        tr0 = t1;
    };
    std::vector<H::Fr> gamma_powers(d);                             // challenge_scalar_powers(d)
    { const H::Fr q = H::tr_challenge_scalar(T); gamma_powers[0] = H::one(); for (size_t i = 1; i < d; i++) gamma_powers[i] = H::mul(gamma_powers[i - 1], q); }
    std::vector<H::Fr> gammas(d), r_addr(lkc);                       // challenge_vector_optimized
    for (size_t i = 0; i < d; i++) { uint64_t lo, hi; H::tr_challenge_u128(T, lo, hi); gammas[i] = H::challenge_to_fr(lo, hi, g.challenge_mode); }
    for (size_t i = 0; i < lkc; i++) { uint64_t lo, hi; H::tr_challenge_u128(T, lo, hi); r_addr[i] = H::challenge_to_fr(lo, hi, g.challenge_mode); }
    // G = compute_ra_evals(lookup_indices, params, r_cycle) (shout.rs:550-598)
    atlas_poly_t eq_rc = nullptr;
    int rc = atlas_eq_evals(r_cycle, log_T, nullptr, &eq_rc);
    std::vector<H::Fr> Gh;
    if (!rc) rc = atlas_rt_shout_ra_evals_host(d_lookups, (size_t)1 << log_T, log_K, lkc, eq_rc, Gh);
    if (eq_rc) atlas_poly_free(eq_rc);
    if (rc) return rc;
    std::vector<atlas_fr_t> G(d << lkc);
    std::memcpy(G.data(), Gh.data(), G.size() * sizeof(atlas_fr_t));
    mark("ra_evals G");
    // RaVirtual: (r_address, r_cycle) = the ra opening point split at log_K
    atlas_instance_t ra = nullptr, hw = nullptr, bo = nullptr;
    rc = atlas_ra_virtual_from_lookups_new(d_lookups, log_T, log_K, lkc, ra_point.data(), ra_point.data() + log_K, &ra);
    if (!rc) rc = atlas_hamming_weight_new(G.data(), d, lkc, (const atlas_fr_t*)gamma_powers.data(), &hw);
    if (!rc) rc = atlas_booleanity_from_lookups_new(G.data(), d_lookups, log_T, log_K, lkc, (const atlas_fr_t*)gammas.data(), (const atlas_fr_t*)r_addr.data(), r_cycle, &bo);
    mark("booleanity_new");
    atlas_batched_t b = nullptr;
    H::Fr hw_claim = H::zero();
    for (auto& x : gamma_powers) hw_claim = H::add(hw_claim, x);    // hamming_weight.rs:49-57
    const H::Fr zero = H::zero();
    if (!rc) rc = atlas_batched_new(&b);
    if (!rc) rc = atlas_batched_add_instance(b, ra, (const atlas_fr_t*)&ra_claim);
    if (!rc) rc = atlas_batched_add_instance(b, hw, (const atlas_fr_t*)&hw_claim);
    if (!rc) rc = atlas_batched_add_instance(b, bo, (const atlas_fr_t*)&zero);
    size_t stride = d + 2, max_rounds = lkc + log_T;
    std::vector<atlas_fr_t> rows(max_rounds * stride);
    std::vector<uint32_t> nco(max_rounds);
    std::vector<atlas_u128_t> ch(max_rounds);
    size_t mr = 0;
    if (!rc) rc = atlas_batched_prove(b, t, rows.data(), stride, nco.data(), ch.data(), &mr);
    mark("batched_prove");
    for (atlas_instance_t inst : {ra, hw, bo}) {
        atlas_fr_t fin[64]; size_t nf = 0;
        if (!rc) rc = atlas_instance_final_claims(inst, fin, 64, &nf);
        for (size_t i = 0; i < d && !rc; i++) {                      // append_sparse: one transcript append per claim (:335-339)
            rc = atlas_transcript_append_scalar(t, &fin[i]);
            if (!rc) rc = O.put_claim(*reinterpret_cast<H::Fr*>(&fin[i]));
        }
    }
    if (!rc) rc = O.put_proof(rows, stride, nco, mr);
    if (b) atlas_batched_free(b);
    for (atlas_instance_t inst : {ra, hw, bo}) if (inst) atlas_instance_free(inst);
    mark("finals + free");
    return rc;
}

}  // namespace

namespace {

// The fused-rescale operator flow shared by Einsum, Mul, ... (impl_fused_rescale_proof_api, ops/mod.rs:569-612):
//   fused_rebase::prove_pre (fused_rebase.rs:215-250): remainder advice + rescaled-accumulator claim, saturating clamp lookup
//   (PS-Shout over ClampBoundedTable<64, 31, true>) and its one-hot checks;
//   the operator's own sumcheck over the i64 accumulator, input claim fused_input_claim = rescaled(r) 2^S + R(r);
//   fused_rebase::prove_remainder_rc (:252-285): identity range check of the remainder in [0, 2^S) and its one-hot checks.
// fill_acc launches the kernels that leave the T i64 accumulators in d_acc (library stream); inner(in_claim) runs the
// operator's sumcheck, appends its openings and stores its proof.  stage_ms[6]: witness, clamp, one-hot checks, inner,
// range check, remainder one-hot checks.
template <class FillAcc, class Inner>
int prove_fused_rescale(size_t T, size_t S, FillAcc&& fill_acc, Inner&& inner, const atlas_fr_t* r_node_output, const atlas_fr_t* output_claim,
                        atlas_transcript_t* t, Out& O, int32_t* d_output, double* stage_ms) {
    const size_t log_T = ilog2(T);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point a) { atlas_sync(); return std::chrono::duration<double, std::milli>(now() - a).count(); };
    H::Transcript& Tr = *reinterpret_cast<H::Transcript*>(t);

    // ---- witness (f1): try_rebase_intermediates on the device
    auto t0 = now();
    int64_t* d_quot = nullptr; int32_t* d_rem = nullptr; uint64_t *d_cidx = nullptr, *d_ridx = nullptr; Fr* d_qfr = nullptr; int32_t* d_out_own = nullptr;
    auto cleanup = [&] { for (void* p : {(void*)d_quot, (void*)d_rem, (void*)d_cidx, (void*)d_ridx, (void*)d_qfr, (void*)d_out_own}) if (p) (void)hipFree(p); };
    hipError_t e = hipMalloc(&d_quot, T * 8);
    if (e == hipSuccess) e = hipMalloc(&d_rem, T * 4);
    if (e == hipSuccess) e = hipMalloc(&d_cidx, T * 8);
    if (e == hipSuccess) e = hipMalloc(&d_ridx, T * 8);
    if (e == hipSuccess) e = hipMalloc(&d_qfr, T * sizeof(Fr));
    if (e == hipSuccess && !d_output) { e = hipMalloc(&d_out_own, T * 4); d_output = d_out_own; }
    if (e != hipSuccess) { cleanup(); return fail(ATLAS_ENOMEM, "hipMalloc(fused-rescale witness)", e); }
    int rc = ATLAS_OK;
    {
        std::lock_guard<atlas_rt::Mutex> lk(g.mu);
        size_t gb = (T + 255) / 256; if (gb > 4096) gb = 4096;
        rc = fill_acc(d_quot);
        if (rc) { cleanup(); return rc; }
        k_einsum_rebase<<<(unsigned)gb, 256, 0, g.stream>>>(d_quot, T, (uint32_t)S, d_rem, d_output, d_cidx, d_ridx);
        k_i64_to_fr<<<(unsigned)gb, 256, 0, g.stream>>>(d_quot, d_qfr, T);
    }
    atlas_poly_t p_rem = nullptr, p_quot = nullptr, p_out = nullptr;
    {   // borrowed views for evaluate
        rc = atlas_poly_wrap_device_fr(d_qfr, T, &p_quot);
        if (!rc) rc = atlas_poly_wrap_device_i32(d_rem, T, &p_rem);
        if (!rc) rc = atlas_poly_wrap_device_i32(d_output, T, &p_out);
    }
    H::Fr eval_R, acc_claim, out_claim;
    if (!rc) {   // the three openings at r_node_output share their eq tables and one synchronisation
        const atlas_poly_t ps[3] = {p_rem, p_quot, p_out};
        H::Fr ev[3];
        rc = atlas_poly_evaluate_many(ps, output_claim ? 2 : 3, r_node_output, log_T, (atlas_fr_t*)ev);
        eval_R = ev[0]; acc_claim = ev[1];
        if (output_claim) std::memcpy(&out_claim, output_claim, 32); else out_claim = ev[2];
    }
    for (atlas_poly_t p : {p_rem, p_quot, p_out}) if (p) atlas_poly_free(p);
    if (stage_ms) stage_ms[0] = ms_since(t0);
    if (rc) { cleanup(); return rc; }

    // ---- prove_pre: remainder advice, clamp lookup
    t0 = now();
    H::tr_append_scalar(Tr, eval_R); rc = O.put_claim(eval_R);               // cache_remainder_prove
    H::tr_append_scalar(Tr, acc_claim); if (!rc) rc = O.put_claim(acc_claim); // append_raf_claims_prover
    const H::Fr gamma = H::tr_challenge_scalar(Tr);                          // ps_read_raf_prover (unary.rs:112)
    atlas_instance_t exec = nullptr;
    if (!rc) rc = atlas_ps_shout_clamp_new(d_cidx, log_T, 64, 31, 1, r_node_output, (const atlas_fr_t*)&gamma, &exec);
    const H::Fr exec_claim = H::add(out_claim, H::mul(gamma, acc_claim));     // rv_claim + gamma * operand_claim (ps_shout/mod.rs:142-144)
    std::vector<atlas_u128_t> ch;
    H::Fr ra_claim;
    if (!rc) rc = prove_single(exec, exec_claim, t, O, ch, &ra_claim);
    if (exec) atlas_instance_free(exec);
    // ClampRa opening point = normalize_opening_point: address challenges as drawn, cycle challenges reversed (ps_shout/mod.rs:150-158)
    std::vector<atlas_fr_t> ra_point(64 + log_T);
    if (!rc) {
        for (size_t i = 0; i < 64; i++) { const H::Fr f = H::challenge_to_fr(ch[i].lo, ch[i].hi, g.challenge_mode); std::memcpy(&ra_point[i], &f, 32); }
        for (size_t i = 0; i < log_T; i++) { const H::Fr f = H::challenge_to_fr(ch[64 + log_T - 1 - i].lo, ch[64 + log_T - 1 - i].hi, g.challenge_mode); std::memcpy(&ra_point[64 + i], &f, 32); }
    }
    if (stage_ms) stage_ms[1] = ms_since(t0);
    t0 = now();
    if (!rc) rc = prove_onehot_checks(d_cidx, log_T, 64, r_node_output, ra_point, ra_claim, t, O);
    if (stage_ms) stage_ms[2] = ms_since(t0);

    // ---- the operator's sumcheck over the accumulator
    t0 = now();
    if (!rc) {
        // fused_input_claim: rescaled(r0) 2^S + R(r0)
        const H::Fr in_claim = H::add(H::mul(acc_claim, H::from_u64((uint64_t)1 << S)), eval_R);
        rc = inner(in_claim);
    }
    if (stage_ms) stage_ms[3] = ms_since(t0);

    // ---- prove_remainder_rc
    t0 = now();
    std::vector<atlas_fr_t> rr_point(S + log_T);
    H::Fr rr_claim;
    if (!rc) {
        size_t phases = S <= 2 ? 1 : S % 4 == 0 ? S / 4 : S % 2 == 0 ? S / 2 : S;      // IdentityRCProvider::phases (identity_range_check.rs:506-518)
        atlas_instance_t rcq = nullptr;
        rc = atlas_identity_range_check_new(d_ridx, log_T, S, phases, r_node_output, &rcq);
        if (!rc) rc = prove_single(rcq, eval_R, t, O, ch, &rr_claim);
        if (rcq) atlas_instance_free(rcq);
        if (!rc) {
            for (size_t i = 0; i < S; i++) { const H::Fr f = H::challenge_to_fr(ch[i].lo, ch[i].hi, g.challenge_mode); std::memcpy(&rr_point[i], &f, 32); }
            for (size_t i = 0; i < log_T; i++) { const H::Fr f = H::challenge_to_fr(ch[S + log_T - 1 - i].lo, ch[S + log_T - 1 - i].hi, g.challenge_mode); std::memcpy(&rr_point[S + i], &f, 32); }
        }
    }
    if (stage_ms) stage_ms[4] = ms_since(t0);
    t0 = now();
    if (!rc) rc = prove_onehot_checks(d_ridx, log_T, S, r_node_output, rr_point, rr_claim, t, O);
    if (stage_ms) stage_ms[5] = ms_since(t0);
    cleanup();
    return rc;
}



}  // namespace

extern "C" int atlas_prove_einsum_node(const atlas_einsum_node_t* node, const int32_t* d_A, const int32_t* d_B, const atlas_fr_t* r_node_output,
                                       const atlas_fr_t* output_claim, atlas_transcript_t* t, uint8_t* proofs, size_t cap, size_t* proofs_len,
                                       size_t proof_lens[5], atlas_fr_t* claims, size_t claims_cap, size_t* n_claims, int32_t* d_output,
                                       double* stage_ms) {
    NEED_INIT();
    if (!node || !d_A || !d_B || !r_node_output || !t || !proofs || !proofs_len || !proof_lens || !claims || !n_claims)
        return fail(ATLAS_EINVAL, "prove_einsum_node: null argument");
    const size_t m = node->m, k = node->k, n = node->n, S = node->scale_bits;
    if (!pow2(m) || !pow2(k) || !pow2(n) || m * n < 2 || k < 2 || S == 0 || S > 30 || m * n > ((size_t)1 << 26) || k > ((size_t)1 << 26))
        return fail(ATLAS_EINVAL, "prove_einsum_node: m, k, n powers of two (m*n >= 2, k >= 2), 1 <= scale_bits <= 30");
    const size_t T = m * n, log_T = ilog2(T), log_m = ilog2(m);
    Out O{proofs, cap, 0, proof_lens, 0, claims, claims_cap, 0};
    auto fill_acc = [&](int64_t* d_acc) -> int {
        // enough k-slices for ~2^16 threads
        uint32_t slices = 1;
        while (slices < 64 && (n * ((m + EB_ROWS - 1) / EB_ROWS)) * slices < ((size_t)1 << 16) && k / (slices * 2) >= 32) slices *= 2;
        const uint32_t k_slice = (uint32_t)((k + slices - 1) / slices);
        HIP_TRY(hipMemsetAsync(d_acc, 0, T * 8, g.stream));
        k_einsum_acc_mk_kn<<<dim3((unsigned)((n + 255) / 256), (unsigned)((m + EB_ROWS - 1) / EB_ROWS), slices), 256, 0, g.stream>>>(
            d_A, d_B, (uint32_t)m, (uint32_t)k, (uint32_t)n, k_slice, (unsigned long long*)d_acc);
        return ATLAS_OK;
    };
    auto inner = [&](const H::Fr& in_claim) -> int {               // EinsumMatmul (ops/einsum/mod.rs:71-115)
        int rc = ATLAS_OK;
        atlas_poly_t eq_m = nullptr, eq_n = nullptr, left = nullptr, right = nullptr;
        rc = atlas_eq_evals(r_node_output, log_m, nullptr, &eq_m);
        if (!rc) rc = atlas_eq_evals(r_node_output + log_m, log_T - log_m, nullptr, &eq_n);
        const size_t dims[3] = {m, k, n};
        if (!rc) rc = atlas_einsum_fold(ATLAS_EINSUM_MK_KN_MN, dims, 3, d_A, d_B, eq_m, eq_n, &left, &right);
        if (eq_m) atlas_poly_free(eq_m);
        if (eq_n) atlas_poly_free(eq_n);
        atlas_dot_prover_t dp = nullptr;
        if (!rc) rc = atlas_dot_prover_new(left, right, nullptr, ATLAS_EQ_NONE, 0, 0, &dp);
        if (rc) { if (left) atlas_poly_free(left); if (right) atlas_poly_free(right); }
        const size_t nk = ilog2(k);
        std::vector<atlas_fr_t> rows(nk * 2); std::vector<atlas_u128_t> chm(nk); atlas_fr_t fin[3];
        if (!rc) rc = atlas_sumcheck_prove_dot(dp, (const atlas_fr_t*)&in_claim, t, rows.data(), chm.data(), fin);
        if (!rc) {                                                            // dot.rs:377-400: left then right operand opening
            for (int q = 0; q < 2 && !rc; q++) { rc = atlas_transcript_append_scalar(t, &fin[q]); if (!rc) rc = O.put_claim(*reinterpret_cast<H::Fr*>(&fin[q])); }
            std::vector<uint32_t> nco(nk, 2);
            if (!rc) rc = O.put_proof(rows, 2, nco, nk);
        }
        if (dp) atlas_dot_prover_free(dp);
        return rc;
    };
    int rc = prove_fused_rescale(T, S, fill_acc, inner, r_node_output, output_claim, t, O, d_output, stage_ms);
    if (rc) return rc;
    *proofs_len = O.len; *n_claims = O.n_claims;
    return ATLAS_OK;
}

// Mul::prove / Square::prove with fused rescaling (ops/mul.rs, ops/square.rs via impl_fused_rescale_proof_api, ops/mod.rs:569-612):
// out = (left * right) >> S element-wise (Square: right = left, one operand).  The operator's own sumcheck is MulProver /
// SquareProver over the Gruen split-eq of r_node_output, input claim rescaled(r) 2^S + R(r); cache_openings appends the
// operand claim(s) in input order.  Five proofs like the Einsum node.
static int prove_ew_fused_node(int op, const int32_t* d_left, const int32_t* d_right, size_t log_T, uint32_t scale_bits, const atlas_fr_t* r_node_output,
                               const atlas_fr_t* output_claim, atlas_transcript_t* t, uint8_t* proofs, size_t cap, size_t* proofs_len,
                               size_t proof_lens[5], atlas_fr_t* claims, size_t claims_cap, size_t* n_claims, int32_t* d_output, double* stage_ms) {
    const size_t S = scale_bits;
    if (log_T == 0 || log_T > 25 || S == 0 || S > 30) return fail(ATLAS_EINVAL, "prove_mul_node: 1 <= log_T <= 25, 1 <= scale_bits <= 30");
    const size_t T = (size_t)1 << log_T, n_ops = op == ATLAS_EW_SQUARE ? 1 : 2;
    Out O{proofs, cap, 0, proof_lens, 0, claims, claims_cap, 0};
    auto fill_acc = [&](int64_t* d_acc) -> int {
        size_t gb = (T + 255) / 256; if (gb > 4096) gb = 4096;
        k_mul_acc<<<(unsigned)gb, 256, 0, g.stream>>>(d_left, d_right, T, d_acc);
        return ATLAS_OK;
    };
    auto inner = [&](const H::Fr& in_claim) -> int {
        atlas_poly_t ops[2] = {nullptr, nullptr};
        int rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(d_left), T, &ops[0]);          // read-only views
        if (!rc && n_ops == 2) rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(d_right), T, &ops[1]);
        atlas_instance_t inst = nullptr;
        if (!rc) rc = atlas_elementwise_new(op, ops, n_ops, r_node_output, log_T, nullptr, 0, &inst);
        for (atlas_poly_t p : ops) if (p) atlas_poly_free(p);
        const size_t stride = 4;
        std::vector<atlas_fr_t> rows(log_T * stride);
        std::vector<uint32_t> nco(log_T);
        std::vector<atlas_u128_t> ch(log_T);
        if (!rc) rc = atlas_instance_prove(inst, (const atlas_fr_t*)&in_claim, t, rows.data(), stride, nco.data(), ch.data());
        atlas_fr_t fin[8]; size_t nf = 0;
        if (!rc) rc = atlas_instance_final_claims(inst, fin, 8, &nf);
        for (size_t q = 0; q < n_ops && !rc; q++) {                              // append_nodeio(Input(q))  (mul.rs:188-199, square.rs:186-197)
            rc = atlas_transcript_append_scalar(t, &fin[q]);
            if (!rc) rc = O.put_claim(*reinterpret_cast<H::Fr*>(&fin[q]));
        }
        if (!rc) rc = O.put_proof(rows, stride, nco, log_T);
        if (inst) atlas_instance_free(inst);
        return rc;
    };
    int rc = prove_fused_rescale(T, S, fill_acc, inner, r_node_output, output_claim, t, O, d_output, stage_ms);
    if (rc) return rc;
    *proofs_len = O.len; *n_claims = O.n_claims;
    return ATLAS_OK;
}

extern "C" int atlas_prove_mul_node(const int32_t* d_left, const int32_t* d_right, size_t log_T, uint32_t scale_bits, const atlas_fr_t* r_node_output,
                                    const atlas_fr_t* output_claim, atlas_transcript_t* t, uint8_t* proofs, size_t cap, size_t* proofs_len,
                                    size_t proof_lens[5], atlas_fr_t* claims, size_t claims_cap, size_t* n_claims, int32_t* d_output, double* stage_ms) {
    NEED_INIT();
    if (!d_left || !d_right || !r_node_output || !t || !proofs || !proofs_len || !proof_lens || !claims || !n_claims)
        return fail(ATLAS_EINVAL, "prove_mul_node: null argument");
    return prove_ew_fused_node(ATLAS_EW_MUL, d_left, d_right, log_T, scale_bits, r_node_output, output_claim, t, proofs, cap, proofs_len, proof_lens, claims,
                               claims_cap, n_claims, d_output, stage_ms);
}

extern "C" int atlas_prove_square_node(const int32_t* d_input, size_t log_T, uint32_t scale_bits, const atlas_fr_t* r_node_output,
                                       const atlas_fr_t* output_claim, atlas_transcript_t* t, uint8_t* proofs, size_t cap, size_t* proofs_len,
                                       size_t proof_lens[5], atlas_fr_t* claims, size_t claims_cap, size_t* n_claims, int32_t* d_output, double* stage_ms) {
    NEED_INIT();
    if (!d_input || !r_node_output || !t || !proofs || !proofs_len || !proof_lens || !claims || !n_claims)
        return fail(ATLAS_EINVAL, "prove_square_node: null argument");
    return prove_ew_fused_node(ATLAS_EW_SQUARE, d_input, d_input, log_T, scale_bits, r_node_output, output_claim, t, proofs, cap, proofs_len, proof_lens, claims,
                               claims_cap, n_claims, d_output, stage_ms);
}

// The operators proved by one sumcheck and nothing else (impl_standard_sumcheck_proof_api, ops/mod.rs:505-560: And = MulProver
// without rescaling, Iff): Sumcheck::prove over the operator's element-wise prover with the node-output opening claim as the
// input claim, then cache_openings = the operand claims in input order.  op: ATLAS_EW_MUL (And) or ATLAS_EW_IFF; operands:
// n_operands device tensors of 2^log_T i32.  One proof (Execution); claims: one per operand.
extern "C" int atlas_prove_standard_node(int op, const int32_t* const* d_operands, size_t n_operands, size_t log_T, const atlas_fr_t* r_node_output,
                                         const atlas_fr_t* output_claim, atlas_transcript_t* t, uint8_t* proofs, size_t cap, size_t* proofs_len,
                                         size_t proof_lens[1], atlas_fr_t* claims, size_t claims_cap, size_t* n_claims) {
    NEED_INIT();
    if (!d_operands || !r_node_output || !output_claim || !t || !proofs || !proofs_len || !proof_lens || !claims || !n_claims)
        return fail(ATLAS_EINVAL, "prove_standard_node: null argument");
    if ((op != ATLAS_EW_MUL && op != ATLAS_EW_IFF) || n_operands != (op == ATLAS_EW_IFF ? 3u : 2u))
        return fail(ATLAS_EINVAL, "prove_standard_node: op must be ATLAS_EW_MUL (And: 2 operands) or ATLAS_EW_IFF (3 operands)");
    if (log_T == 0 || log_T > 25) return fail(ATLAS_EINVAL, "prove_standard_node: 1 <= log_T <= 25");
    const size_t T = (size_t)1 << log_T;
    Out O{proofs, cap, 0, proof_lens, 0, claims, claims_cap, 0};
    atlas_poly_t ops[3] = {nullptr, nullptr, nullptr};
    int rc = ATLAS_OK;
    for (size_t i = 0; i < n_operands && !rc; i++) {
        if (!d_operands[i]) rc = fail(ATLAS_EINVAL, "prove_standard_node: null operand");
        else rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(d_operands[i]), T, &ops[i]);          // read-only views
    }
    atlas_instance_t inst = nullptr;
    if (!rc) rc = atlas_elementwise_new(op, ops, n_operands, r_node_output, log_T, nullptr, 0, &inst);
    for (atlas_poly_t p : ops) if (p) atlas_poly_free(p);
    const size_t stride = 4;
    std::vector<atlas_fr_t> rows(log_T * stride);
    std::vector<uint32_t> nco(log_T);
    std::vector<atlas_u128_t> ch(log_T);
    if (!rc) rc = atlas_instance_prove(inst, output_claim, t, rows.data(), stride, nco.data(), ch.data());
    atlas_fr_t fin[8]; size_t nf = 0;
    if (!rc) rc = atlas_instance_final_claims(inst, fin, 8, &nf);
    for (size_t q = 0; q < n_operands && !rc; q++) {                              // append_nodeio(Input(q))  (mul.rs:188-199, iff.rs:228-241)
        rc = atlas_transcript_append_scalar(t, &fin[q]);
        if (!rc) rc = O.put_claim(*reinterpret_cast<H::Fr*>(&fin[q]));
    }
    if (!rc) rc = O.put_proof(rows, stride, nco, log_T);
    if (inst) atlas_instance_free(inst);
    if (rc) return rc;
    *proofs_len = O.len; *n_claims = O.n_claims;
    return ATLAS_OK;
}

// Add::prove / Sub::prove (ops/add.rs:70-105, ops/sub.rs): no sumcheck of their own — prove_clamp_lookup (clamp_lookups/mod.rs:264-309:
// the i64 accumulation left +- right appended as the lookup's raf, gamma, PS-Shout over SaturationTable, its ra opening, the
// one-hot checks), then the operand tie: left(r), right(r) appended in input order.  Two proofs (Execution, RaOneHotChecks);
// claims: acc, ra, 3 d one-hot claims, left, right.  stage_ms[3]: witness, clamp lookup, one-hot checks (+ operand claims).
extern "C" int atlas_prove_addsub_node(const int32_t* d_left, const int32_t* d_right, size_t log_T, int subtract, const atlas_fr_t* r_node_output,
                                       const atlas_fr_t* output_claim, atlas_transcript_t* t, uint8_t* proofs, size_t cap, size_t* proofs_len,
                                       size_t proof_lens[2], atlas_fr_t* claims, size_t claims_cap, size_t* n_claims, int32_t* d_output, double* stage_ms) {
    NEED_INIT();
    if (!d_left || !d_right || !r_node_output || !t || !proofs || !proofs_len || !proof_lens || !claims || !n_claims)
        return fail(ATLAS_EINVAL, "prove_addsub_node: null argument");
    if (log_T == 0 || log_T > 25) return fail(ATLAS_EINVAL, "prove_addsub_node: 1 <= log_T <= 25");
    const size_t T = (size_t)1 << log_T;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point a) { atlas_sync(); return std::chrono::duration<double, std::milli>(now() - a).count(); };
    Out O{proofs, cap, 0, proof_lens, 0, claims, claims_cap, 0};
    H::Transcript& Tr = *reinterpret_cast<H::Transcript*>(t);
    auto t0 = now();
    DevBuf acc_b, idx_b, out_b, fr_b;
    HIP_TRY(acc_b.alloc(T * 8)); HIP_TRY(idx_b.alloc(T * 8)); HIP_TRY(fr_b.alloc(T * sizeof(Fr)));
    if (!d_output) { HIP_TRY(out_b.alloc(T * 4)); d_output = out_b.as<int32_t>(); }
    {
        std::lock_guard<atlas_rt::Mutex> lk(g.mu);
        size_t gb = (T + 255) / 256; if (gb > 4096) gb = 4096;
        k_addsub_witness<<<(unsigned)gb, 256, 0, g.stream>>>(d_left, d_right, T, subtract ? 1 : 0, acc_b.as<int64_t>(), d_output, idx_b.as<uint64_t>());
        k_i64_to_fr<<<(unsigned)gb, 256, 0, g.stream>>>(acc_b.as<int64_t>(), fr_b.as<Fr>(), T);
    }
    int rc = ATLAS_OK;
    atlas_poly_t p_acc = nullptr, p_out = nullptr, p_l = nullptr, p_r = nullptr;
    H::Fr acc_claim, out_claim, l_claim, r_claim;
    rc = atlas_poly_wrap_device_fr(fr_b.p, T, &p_acc);
    if (!rc) rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(d_left), T, &p_l);
    if (!rc) rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(d_right), T, &p_r);
    if (!rc) rc = atlas_poly_wrap_device_i32(d_output, T, &p_out);
    if (!rc) {
        const atlas_poly_t ps[4] = {p_acc, p_l, p_r, p_out};
        H::Fr ev[4];
        rc = atlas_poly_evaluate_many(ps, output_claim ? 3 : 4, r_node_output, log_T, (atlas_fr_t*)ev);
        acc_claim = ev[0]; l_claim = ev[1]; r_claim = ev[2];
        if (output_claim) std::memcpy(&out_claim, output_claim, 32); else out_claim = ev[3];
    }
    for (atlas_poly_t p : {p_acc, p_out, p_l, p_r}) if (p) atlas_poly_free(p);
    if (stage_ms) stage_ms[0] = ms_since(t0);
    if (rc) return rc;

    t0 = now();
    H::tr_append_scalar(Tr, acc_claim); rc = O.put_claim(acc_claim);             // append_raf_claims_prover
    const H::Fr gamma = H::tr_challenge_scalar(Tr);                              // ps_read_raf_prover (unary.rs:112)
    atlas_instance_t exec = nullptr;
    if (!rc) rc = atlas_ps_shout_clamp_new(idx_b.as<uint64_t>(), log_T, 64, 31, 1, r_node_output, (const atlas_fr_t*)&gamma, &exec);
    const H::Fr exec_claim = H::add(out_claim, H::mul(gamma, acc_claim));
    std::vector<atlas_u128_t> ch;
    H::Fr ra_claim;
    if (!rc) rc = prove_single(exec, exec_claim, t, O, ch, &ra_claim);
    if (exec) atlas_instance_free(exec);
    std::vector<atlas_fr_t> ra_point(64 + log_T);
    if (!rc) {
        for (size_t i = 0; i < 64; i++) { const H::Fr f = H::challenge_to_fr(ch[i].lo, ch[i].hi, g.challenge_mode); std::memcpy(&ra_point[i], &f, 32); }
        for (size_t i = 0; i < log_T; i++) { const H::Fr f = H::challenge_to_fr(ch[64 + log_T - 1 - i].lo, ch[64 + log_T - 1 - i].hi, g.challenge_mode); std::memcpy(&ra_point[64 + i], &f, 32); }
    }
    if (stage_ms) stage_ms[1] = ms_since(t0);
    t0 = now();
    if (!rc) rc = prove_onehot_checks(idx_b.as<uint64_t>(), log_T, 64, r_node_output, ra_point, ra_claim, t, O);
    if (!rc) { H::tr_append_scalar(Tr, l_claim); rc = O.put_claim(l_claim); }    // operand tie (add.rs:95-102)
    if (!rc) { H::tr_append_scalar(Tr, r_claim); rc = O.put_claim(r_claim); }
    if (stage_ms) stage_ms[2] = ms_since(t0);
    if (rc) return rc;
    *proofs_len = O.len; *n_claims = O.n_claims;
    return ATLAS_OK;
}

// ReLU::prove (jolt-atlas-core/src/onnx_proof/ops/relu.rs:22-70): OpLookupProvider::read_raf_prove over ReluTable<XLEN>
// (op_lookups/mod.rs:250-267: the operand's claim at r_cycle appended, gamma drawn, UnaryReadRafSumcheckProver), Sumcheck::prove
// + its ra opening, then ra_onehot_provers + BatchedSumcheck::prove (shout.rs:399-466).  d_input: the node's operand, 2^log_T
// i32 on the device (padded by the caller); r_node_output: the node-output opening point (log_T challenges as field
// elements); output_claim: relu(x)~(r) if the caller has it, else evaluated here.  Two proofs (Execution, RaOneHotChecks),
// ark-serialized back to back; claims in accumulator order: operand, ra, then the 3 d one-hot claims.
extern "C" int atlas_prove_relu_node(const int32_t* d_input, size_t log_T, const atlas_fr_t* r_node_output, const atlas_fr_t* output_claim,
                                     atlas_transcript_t* t, uint8_t* proofs, size_t cap, size_t* proofs_len, size_t proof_lens[2],
                                     atlas_fr_t* claims, size_t claims_cap, size_t* n_claims, int32_t* d_output, double* stage_ms) {
    NEED_INIT();
    if (!d_input || !r_node_output || !t || !proofs || !proofs_len || !proof_lens || !claims || !n_claims)
        return fail(ATLAS_EINVAL, "prove_relu_node: null argument");
    if (log_T == 0 || log_T > 25) return fail(ATLAS_EINVAL, "prove_relu_node: 1 <= log_T <= 25");
    const size_t T = (size_t)1 << log_T, XLEN = 32;                  // common/src/consts/general.rs:1
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point a) { atlas_sync(); return std::chrono::duration<double, std::milli>(now() - a).count(); };
    Out O{proofs, cap, 0, proof_lens, 0, claims, claims_cap, 0};
    H::Transcript& Tr = *reinterpret_cast<H::Transcript*>(t);
    auto t0 = now();
    DevBuf idx_b, out_b;
    HIP_TRY(idx_b.alloc(T * 8));
    if (!d_output) { HIP_TRY(out_b.alloc(T * 4)); d_output = out_b.as<int32_t>(); }
    {
        std::lock_guard<atlas_rt::Mutex> lk(g.mu);
        size_t gb = (T + 255) / 256; if (gb > 4096) gb = 4096;
        k_relu_witness<<<(unsigned)gb, 256, 0, g.stream>>>(d_input, T, d_output, idx_b.as<uint64_t>());
    }
    int rc = ATLAS_OK;
    atlas_poly_t p_in = nullptr, p_out = nullptr;
    H::Fr operand_claim, out_claim;
    rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(d_input), T, &p_in);          // a read-only view: evaluate does not write
    if (!rc) rc = atlas_poly_wrap_device_i32(d_output, T, &p_out);
    if (!rc) {
        const atlas_poly_t ps[2] = {p_in, p_out};
        H::Fr ev[2];
        rc = atlas_poly_evaluate_many(ps, output_claim ? 1 : 2, r_node_output, log_T, (atlas_fr_t*)ev);
        operand_claim = ev[0];
        if (output_claim) std::memcpy(&out_claim, output_claim, 32); else out_claim = ev[1];
    }
    for (atlas_poly_t p : {p_in, p_out}) if (p) atlas_poly_free(p);
    if (stage_ms) stage_ms[0] = ms_since(t0);
    if (rc) return rc;

    t0 = now();
    H::tr_append_scalar(Tr, operand_claim); rc = O.put_claim(operand_claim);      // append_raf_claims_prover (op_lookups/mod.rs:404-418)
    const H::Fr gamma = H::tr_challenge_scalar(Tr);                               // ps_read_raf_prover (unary.rs:112)
    atlas_instance_t exec = nullptr;
    if (!rc) rc = atlas_ps_shout_relu_new(idx_b.as<uint64_t>(), log_T, XLEN, r_node_output, (const atlas_fr_t*)&gamma, &exec);
    const H::Fr exec_claim = H::add(out_claim, H::mul(gamma, operand_claim));     // rv_claim + gamma * operand_claim (ps_shout/mod.rs:142-144)
    std::vector<atlas_u128_t> ch;
    H::Fr ra_claim;
    if (!rc) rc = prove_single(exec, exec_claim, t, O, ch, &ra_claim);
    if (exec) atlas_instance_free(exec);
    std::vector<atlas_fr_t> ra_point(XLEN + log_T);                               // normalize_opening_point (ps_shout/mod.rs:150-158)
    if (!rc) {
        for (size_t i = 0; i < XLEN; i++) { const H::Fr f = H::challenge_to_fr(ch[i].lo, ch[i].hi, g.challenge_mode); std::memcpy(&ra_point[i], &f, 32); }
        for (size_t i = 0; i < log_T; i++) { const H::Fr f = H::challenge_to_fr(ch[XLEN + log_T - 1 - i].lo, ch[XLEN + log_T - 1 - i].hi, g.challenge_mode); std::memcpy(&ra_point[XLEN + i], &f, 32); }
    }
    if (stage_ms) stage_ms[1] = ms_since(t0);
    t0 = now();
    if (!rc) rc = prove_onehot_checks(idx_b.as<uint64_t>(), log_T, XLEN, r_node_output, ra_point, ra_claim, t, O);
    if (stage_ms) stage_ms[2] = ms_since(t0);
    if (rc) return rc;
    *proofs_len = O.len; *n_claims = O.n_claims;
    return ATLAS_OK;
}
