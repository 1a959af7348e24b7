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
#include "node_flow.hip.h"

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
        HIP_TRY(hipMemsetAsync(d_acc, 0, T * 8, rt().stream));
        k_einsum_acc_mk_kn<<<dim3((unsigned)((n + 255) / 256), (unsigned)((m + EB_ROWS - 1) / EB_ROWS), slices), 256, 0, rt().stream>>>(
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
    auto tw = std::chrono::steady_clock::now();
    RescaleWitness W;
    int rc = make_rescale_witness(T, S, fill_acc, d_output, W);
    double wms = 0;
    if (!rc && stage_ms) { atlas_sync(); wms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count(); }
    if (!rc) rc = prove_fused_rescale(W, inner, r_node_output, output_claim, t, O, stage_ms);
    if (stage_ms) stage_ms[0] += wms;
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
        k_mul_acc<<<(unsigned)gb, 256, 0, rt().stream>>>(d_left, d_right, T, d_acc);
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
    auto tw = std::chrono::steady_clock::now();
    RescaleWitness W;
    int rc = make_rescale_witness(T, S, fill_acc, d_output, W);
    double wms = 0;
    if (!rc && stage_ms) { atlas_sync(); wms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count(); }
    if (!rc) rc = prove_fused_rescale(W, inner, r_node_output, output_claim, t, O, stage_ms);
    if (stage_ms) stage_ms[0] += wms;
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
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        size_t gb = (T + 255) / 256; if (gb > 4096) gb = 4096;
        k_addsub_witness<<<(unsigned)gb, 256, 0, rt().stream>>>(d_left, d_right, T, subtract ? 1 : 0, acc_b.as<int64_t>(), d_output, idx_b.as<uint64_t>());
        k_i64_to_fr<<<(unsigned)gb, 256, 0, rt().stream>>>(acc_b.as<int64_t>(), fr_b.as<Fr>(), T);
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
        for (size_t i = 0; i < 64; i++) { const H::Fr f = H::challenge_to_fr(ch[i].lo, ch[i].hi, rt().challenge_mode); std::memcpy(&ra_point[i], &f, 32); }
        for (size_t i = 0; i < log_T; i++) { const H::Fr f = H::challenge_to_fr(ch[64 + log_T - 1 - i].lo, ch[64 + log_T - 1 - i].hi, rt().challenge_mode); std::memcpy(&ra_point[64 + i], &f, 32); }
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
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        size_t gb = (T + 255) / 256; if (gb > 4096) gb = 4096;
        k_relu_witness<<<(unsigned)gb, 256, 0, rt().stream>>>(d_input, T, d_output, idx_b.as<uint64_t>());
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
        for (size_t i = 0; i < XLEN; i++) { const H::Fr f = H::challenge_to_fr(ch[i].lo, ch[i].hi, rt().challenge_mode); std::memcpy(&ra_point[i], &f, 32); }
        for (size_t i = 0; i < log_T; i++) { const H::Fr f = H::challenge_to_fr(ch[XLEN + log_T - 1 - i].lo, ch[XLEN + log_T - 1 - i].hi, rt().challenge_mode); std::memcpy(&ra_point[XLEN + i], &f, 32); }
    }
    if (stage_ms) stage_ms[1] = ms_since(t0);
    t0 = now();
    if (!rc) rc = prove_onehot_checks(idx_b.as<uint64_t>(), log_T, XLEN, r_node_output, ra_point, ra_claim, t, O);
    if (stage_ms) stage_ms[2] = ms_since(t0);
    if (rc) return rc;
    *proofs_len = O.len; *n_claims = O.n_claims;
    return ATLAS_OK;
}
