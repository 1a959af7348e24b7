// ONNXProof::verify for a graph resident in the library (jolt-atlas-core/src/onnx_proof/mod.rs:207-241, verifier.rs):
//   populate_accumulator        verifier.rs:66-80      opening claims from the proof, commitments into the transcript
//   verify_output_claim         verifier.rs:83-119     the output tensor's MLE at the transcript's point
//   verify_iop                  verifier.rs:123-137    nodes in reverse: NodeEvalReduction::verify, then the operator's verifier instances
//   verify_reduced_openings     verifier.rs:140-187    the batched opening-reduction sumcheck, the joint commitment, HyperKZG::verify
// Host arithmetic, like the reference's verifier; the device only evaluates the PUBLIC tensors (inputs, constants, output) at the
// verifier's points and adds up the joint commitment.  Nothing of the prover's trace is read.
// Operators with a verifier composition: Input, Constant, Identity, Add, Sub, Mul, Square, Cube, Einsum, And, Iff, ReLU, Reshape, MoveAxis,
// Broadcast (the fused-rescale family, the clamp / ReLU lookups with their one-hot checks, the shape operators).  The other operators of the
// graph prover (Sum, ScalarConstDiv, Div, MeanOfSquares, Rsqrt, Slice, Tanh, GatherLarge, SoftmaxLastAxis) return ATLAS_EINVAL.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "graph_state.hip.h"
#include "host_curve.hpp"

using gr::Node;
using gr::OpeningId;
using gr::Point;
using gr::PolyId;

namespace {

struct Rows { std::vector<atlas_fr_t> c; std::vector<uint32_t> n; size_t stride = 0, rounds = 0; };
struct Reader {
    const uint8_t* p; size_t left; bool ok = true;
    uint64_t u64() { if (left < 8) { ok = false; return 0; } uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i); p += 8; left -= 8; return v; }
    uint8_t u8() { if (left < 1) { ok = false; return 0; } uint8_t v = *p; p++; left--; return v; }
    bool fr(H::Fr* o) { if (left < 32) { ok = false; return false; } if (atlas_fr_from_bytes(p, (atlas_fr_t*)o)) ok = false; p += 32; left -= 32; return ok; }
    bool g1(atlas_g1_affine_t* o) { if (left < 32) { ok = false; return false; } if (atlas_g1_from_bytes_compressed(p, o)) ok = false; p += 32; left -= 32; return ok; }
    bool rows(Rows& R) {                                   // SumcheckInstanceProof { compressed_polys: Vec<CompressedUniPoly> }
        const uint64_t nr = u64();
        if (!ok || nr > 4096) return ok = false;
        std::vector<std::vector<H::Fr>> tmp(nr);
        size_t st = 1;
        for (auto& r : tmp) { const uint64_t k = u64(); if (!ok || k > 64) return ok = false; r.resize(k); for (auto& x : r) if (!fr(&x)) return false; st = k > st ? k : st; }
        R.stride = st; R.rounds = nr; R.c.assign(nr * st, atlas_fr_t{}); R.n.resize(nr);
        for (size_t i = 0; i < nr; i++) { R.n[i] = (uint32_t)tmp[i].size(); std::memcpy(&R.c[i * st], tmp[i].data(), tmp[i].size() * 32); }
        return true;
    }
};

H::Fr eq_mle(const H::Fr* a, const H::Fr* b, size_t n) {     // EqPolynomial::mle: prod (a_i b_i + (1 - a_i)(1 - b_i))
    H::Fr acc = H::one();
    const H::Fr one = H::one();
    for (size_t i = 0; i < n; i++) acc = H::mul(acc, H::add(H::mul(a[i], b[i]), H::mul(H::sub(one, a[i]), H::sub(one, b[i]))));
    return acc;
}
H::Fr identity_mle(const H::Fr* r, size_t n) { H::Fr y = H::zero(); for (size_t i = 0; i < n; i++) y = H::add(y, H::mul(r[n - 1 - i], H::from_canonical((const uint64_t[4]){i < 64 ? (uint64_t)1 << i : 0, 0, 0, 0}))); return y; }
H::Fr pow2_fr(size_t e) { uint64_t c[4] = {0, 0, 0, 0}; c[e >> 6] = (uint64_t)1 << (e & 63); return H::from_canonical(c); }
// SignedIdentityPoly::evaluate (signed_identity_poly.rs:56-75): the two's-complement value of the n address bits
H::Fr signed_identity_mle(const H::Fr* r, size_t n) {
    H::Fr y = H::zero();
    for (size_t i = 0; i < n; i++) y = H::add(y, H::mul(r[n - 1 - i], pow2_fr(i)));
    return H::sub(y, H::mul(pow2_fr(n), r[0]));
}
// ReluTable<XLEN>::evaluate_mle (relu.rs:29-43)
H::Fr relu_mle(const H::Fr* r, size_t n) {
    H::Fr res = H::zero();
    for (size_t i = 0; i + 1 < n; i++) res = H::add(res, H::mul(r[n - 1 - i], pow2_fr(i)));
    return H::mul(res, H::sub(H::one(), r[0]));
}
// ClampBoundedTable<XLEN, BOUND, SYMMETRIC>::evaluate_mle (clamp.rs:139-193)
H::Fr clamp_mle(const H::Fr* r, size_t xlen, size_t bound, bool symmetric) {
    const size_t ub = xlen - bound - 1;
    const H::Fr one = H::one(), msb = r[0];
    H::Fr haz = one, hao = symmetric ? one : H::zero(), lw = H::zero();
    for (size_t i = 0; i <= ub; i++) { haz = H::mul(haz, H::sub(one, r[i])); if (symmetric) hao = H::mul(hao, r[i]); }
    for (size_t i = ub + 1; i < xlen; i++) lw = H::add(lw, H::mul(r[i], pow2_fr(xlen - i - 1)));
    const H::Fr cu = H::sub(pow2_fr(bound), one), lc = symmetric ? H::add(H::add(cu, cu), one) : cu;
    return H::add(H::add(H::sub(cu, H::mul(msb, lc)), H::mul(haz, H::sub(lw, cu))), H::mul(hao, lw));
}

struct Verifier {
    atlas_graph& G;
    const atlas_hyperkzg_vk_t* vk;
    atlas_transcript_t t;
    H::Transcript& Tr;
    std::map<OpeningId, H::Fr> claims;                    // the proof's opening claims
    std::map<OpeningId, Point> points;                    // filled in as the verifier replays the appends
    std::map<gr::ProofId, Rows> proofs;
    std::map<size_t, std::vector<H::Fr>> evalred;
    std::map<size_t, gr::Opening> reduced;
    struct CP { size_t log_T; atlas_g1_affine_t commitment; bool opened = false; Point point; H::Fr claim; size_t rounds() const { return point.size(); } };
    std::map<PolyId, CP> committed;                       // BTreeMap<CommittedPoly, _>
    Rows ro_rows; std::vector<H::Fr> ro_claims; std::vector<atlas_g1_affine_t> ro_com, ro_w; std::vector<atlas_fr_t> ro_v; bool ro_present = false;
    uint64_t cur = 0;
    const int32_t* const* inputs; const int32_t* output; size_t output_len;

    Verifier(atlas_graph& g_, const atlas_hyperkzg_vk_t* v) : G(g_), vk(v), Tr(*reinterpret_cast<H::Transcript*>(&t)) {}
    static int bad(const char* what) { return fail(ATLAS_EVERIFY, what); }
    int mode() const { return g.challenge_mode; }

    // VerifierOpeningAccumulator::append_virtual: the claim comes from the proof
    int append_virtual(const OpeningId& id, const Point& pt) {
        auto it = claims.find(id);
        if (it == claims.end()) return bad("verify_graph: an opening claim the verifier needs is not in the proof");
        H::tr_append_scalar(Tr, it->second);
        points[id] = pt;
        return ATLAS_OK;
    }
    int append_nodeio(const Node& nd, size_t pos, const Point& pt) { return append_virtual(gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.inputs[pos]), nd.idx), pt); }
    int append_advice(const Node& nd, uint8_t vp, const Point& pt) { return append_virtual(gr::node_exec(gr::virt(vp, nd.idx), nd.idx), pt); }
    const H::Fr& claim_of(const OpeningId& id) const { return claims.at(id); }
    H::Fr nodeio_claim(const Node& nd, size_t pos) const { return claims.at(gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.inputs[pos]), nd.idx)); }
    H::Fr advice_claim(const Node& nd, uint8_t vp) const { return claims.at(gr::node_exec(gr::virt(vp, nd.idx), nd.idx)); }
    int append_sparse(uint8_t cp, const Node& nd, size_t chunk, uint8_t sc, const Point& pt) {
        const PolyId p = gr::comm(cp, nd.idx, chunk);
        const OpeningId id = gr::oid(p, sc);
        auto it = claims.find(id);
        if (it == claims.end()) return bad("verify_graph: a one-hot opening claim is not in the proof");
        H::tr_append_scalar(Tr, it->second);
        points[id] = pt;
        auto c = committed.find(p);
        if (c == committed.end()) return bad("verify_graph: an opening of a polynomial that is not committed");
        c->second.opened = true; c->second.point = pt; c->second.claim = it->second;
        return ATLAS_OK;
    }
    H::Fr ch_fr(const atlas_u128_t& c) const { return H::challenge_to_fr(c.lo, c.hi, mode()); }
    Point challenge_point(size_t n) { Point r(n); for (auto& x : r) { uint64_t lo, hi; H::tr_challenge_u128(Tr, lo, hi); x = H::challenge_to_fr(lo, hi, mode()); } return r; }

    // MultilinearPolynomial::from(tensor).evaluate(point) of a PUBLIC tensor
    int eval_public(const int32_t* host, size_t T, const Point& pt, H::Fr* out) {
        if (((size_t)1 << pt.size()) != T) return bad("verify_graph: an opening point of the wrong length");
        atlas_poly_t p = nullptr;
        int rc = atlas_poly_upload_i32(host, T, &p);
        if (!rc) rc = atlas_poly_evaluate(p, (const atlas_fr_t*)pt.data(), pt.size(), (atlas_fr_t*)out);
        if (p) atlas_poly_free(p);
        return rc;
    }

    // Sumcheck::verify up to the final comparison: returns the final claim and the challenges as field elements
    int single(uint8_t proof_type, size_t n_rounds, size_t degree, const H::Fr& input_claim, H::Fr* e, std::vector<H::Fr>& rs) {
        auto it = proofs.find(gr::ProofId{cur, proof_type});
        if (it == proofs.end()) return bad("verify_graph: MissingProof");
        const Rows& R = it->second;
        if (R.rounds != n_rounds) return bad("verify_graph: a sumcheck proof with the wrong number of rounds");
        H::tr_append_scalar(Tr, input_claim);
        std::vector<atlas_u128_t> ch(n_rounds ? n_rounds : 1);
        int rc = atlas_sumcheck_proof_verify(R.c.data(), R.stride, R.n.data(), n_rounds, degree, (const atlas_fr_t*)&input_claim, &t, (atlas_fr_t*)e, ch.data());
        if (rc) return rc;
        rs.resize(n_rounds);
        for (size_t i = 0; i < n_rounds; i++) rs[i] = ch_fr(ch[i]);
        return ATLAS_OK;
    }
    static Point reversed(const std::vector<H::Fr>& v, size_t from = 0) { return Point(v.rbegin(), v.rend() - from); }
    static bool same(const H::Fr& a, const H::Fr& b) { return std::memcmp(&a, &b, 32) == 0; }

    // ---- a unary prefix-suffix lookup: read_raf_verify (op_lookups/mod.rs:270-283) + Sumcheck::verify (ps_shout/mod.rs:612-643)
    enum Table { T_RELU, T_CLAMP_SYM };
    int ps_unary(const Node& nd, Table tab, size_t xlen, size_t bound, const OpeningId& witness_id, const H::Fr& rv_claim, const Point& r_cycle, uint8_t ra_vp,
                 uint8_t proof_type, Point* ra_point) {
        int rc = append_virtual(witness_id, r_cycle);                        // append_raf_claims_verifier
        if (rc) return rc;
        const H::Fr gamma = H::tr_challenge_scalar(Tr);                      // ps_read_raf_verifier
        const H::Fr operand = claim_of(witness_id);
        const size_t log_T = r_cycle.size();
        H::Fr e; std::vector<H::Fr> rs;
        rc = single(proof_type, xlen + log_T, 2, H::add(rv_claim, H::mul(gamma, operand)), &e, rs);
        if (rc) return rc;
        Point pt(rs.begin(), rs.begin() + xlen);
        for (size_t q = 0; q < log_T; q++) pt.push_back(rs[xlen + log_T - 1 - q]);
        rc = append_advice(nd, ra_vp, pt);                                   // cache_openings
        if (rc) return rc;
        const H::Fr val = tab == T_RELU ? relu_mle(pt.data(), xlen) : clamp_mle(pt.data(), xlen, bound, true);
        const H::Fr expect = H::mul(H::mul(eq_mle(r_cycle.data(), pt.data() + xlen, log_T), advice_claim(nd, ra_vp)), H::add(val, H::mul(gamma, signed_identity_mle(pt.data(), xlen))));
        if (!same(e, expect)) return bad("verify_graph: SumcheckVerificationError (prefix-suffix lookup)");
        *ra_point = pt;
        return ATLAS_OK;
    }
    // IdentityRCVerifier (identity_range_check.rs:455-473)
    int identity_rc(const Node& nd, size_t log_K, const H::Fr& input_claim, const Point& r_cycle, uint8_t ra_vp, uint8_t proof_type, Point* ra_point) {
        const size_t log_T = r_cycle.size();
        H::Fr e; std::vector<H::Fr> rs;
        int rc = single(proof_type, log_K + log_T, 2, input_claim, &e, rs);
        if (rc) return rc;
        Point pt(rs.begin(), rs.begin() + log_K);
        for (size_t q = 0; q < log_T; q++) pt.push_back(rs[log_K + log_T - 1 - q]);
        rc = append_advice(nd, ra_vp, pt);
        if (rc) return rc;
        const H::Fr expect = H::mul(H::mul(eq_mle(r_cycle.data(), pt.data() + log_K, log_T), advice_claim(nd, ra_vp)), identity_mle(pt.data(), log_K));
        if (!same(e, expect)) return bad("verify_graph: SumcheckVerificationError (identity range check)");
        *ra_point = pt;
        return ATLAS_OK;
    }
    // ra_onehot_verifiers + BatchedSumcheck::verify over [RaVirtual, HammingWeight, Booleanity] (shout.rs:474-529)
    int onehot_checks(const Node& nd, size_t log_K, const Point& r_cycle, const Point& ra_point, const H::Fr& ra_claim, uint8_t rad_cp, uint8_t proof_type) {
        const size_t lkc = 4, d = (log_K + lkc - 1) / lkc, log_T = r_cycle.size(), pad = d * lkc - log_K;
        std::vector<H::Fr> gp(d);
        { const H::Fr q = H::tr_challenge_scalar(Tr); gp[0] = H::one(); for (size_t i = 1; i < d; i++) gp[i] = H::mul(gp[i - 1], q); }
        const Point gammas = challenge_point(d), r_addr = challenge_point(lkc);
        H::Fr hw_claim = H::zero();
        for (auto& x : gp) hw_claim = H::add(hw_claim, x);
        auto it = proofs.find(gr::ProofId{cur, proof_type});
        if (it == proofs.end()) return bad("verify_graph: MissingProof (one-hot checks)");
        const Rows& R = it->second;
        const H::Fr ic[3] = {ra_claim, hw_claim, H::zero()};
        const size_t nr[3] = {log_T, lkc, lkc + log_T}, deg[3] = {d + 1, 1, 3};
        H::Fr coeff[3], e;
        const size_t mr = lkc + log_T;
        if (R.rounds != mr) return bad("verify_graph: one-hot checks with the wrong number of rounds");
        std::vector<atlas_u128_t> ch(mr);
        int rc = atlas_batched_sumcheck_verify(R.c.data(), R.stride, R.n.data(), mr, (const atlas_fr_t*)ic, nr, deg, 3, &t, (atlas_fr_t*)coeff, (atlas_fr_t*)&e, ch.data());
        if (rc) return rc;
        std::vector<H::Fr> rs(mr);
        for (size_t i = 0; i < mr; i++) rs[i] = ch_fr(ch[i]);
        H::Fr expect[3];
        // RaVirtual: openings at (chunk i of the zero-padded r_address | reversed cycle challenges); eq(r_cycle_ra, .) prod ra_i
        {
            Point rc_rev(log_T);
            for (size_t q = 0; q < log_T; q++) rc_rev[q] = rs[mr - 1 - q];
            H::Fr prod = H::one();
            for (size_t i = 0; i < d; i++) {
                Point pt(lkc + log_T);
                for (size_t q = 0; q < lkc; q++) { const size_t pos = i * lkc + q; pt[q] = pos < pad ? H::zero() : ra_point[pos - pad]; }
                std::copy(rc_rev.begin(), rc_rev.end(), pt.begin() + lkc);
                rc = append_sparse(rad_cp, nd, i, gr::SC_RaVirtualization, pt);
                if (rc) return rc;
                prod = H::mul(prod, claim_of(gr::oid(gr::comm(rad_cp, nd.idx, i), gr::SC_RaVirtualization)));
            }
            expect[0] = H::mul(eq_mle(ra_point.data() + log_K, rc_rev.data(), log_T), prod);
        }
        {   // HammingWeight: (reversed address challenges | r_cycle); sum gamma^i ra_i
            Point pt(lkc + log_T);
            for (size_t q = 0; q < lkc; q++) pt[q] = rs[mr - 1 - q];
            std::copy(r_cycle.begin(), r_cycle.end(), pt.begin() + lkc);
            H::Fr s = H::zero();
            for (size_t i = 0; i < d; i++) {
                rc = append_sparse(rad_cp, nd, i, gr::SC_HammingWeight, pt);
                if (rc) return rc;
                s = H::add(s, H::mul(claim_of(gr::oid(gr::comm(rad_cp, nd.idx, i), gr::SC_HammingWeight)), gp[i]));
            }
            expect[1] = s;
        }
        {   // Booleanity: both halves reversed; eq(challenges, rev(r_address) | rev(r_cycle)) sum gamma_i (ra_i^2 - ra_i)
            Point pt(lkc + log_T), comb(lkc + log_T);
            for (size_t q = 0; q < lkc; q++) { pt[q] = rs[lkc - 1 - q]; comb[q] = r_addr[lkc - 1 - q]; }
            for (size_t q = 0; q < log_T; q++) { pt[lkc + q] = rs[mr - 1 - q]; comb[lkc + q] = r_cycle[log_T - 1 - q]; }
            H::Fr s = H::zero();
            for (size_t i = 0; i < d; i++) {
                rc = append_sparse(rad_cp, nd, i, gr::SC_Booleanity, pt);
                if (rc) return rc;
                const H::Fr ra = claim_of(gr::oid(gr::comm(rad_cp, nd.idx, i), gr::SC_Booleanity));
                s = H::add(s, H::mul(H::sub(H::mul(ra, ra), ra), gammas[i]));
            }
            expect[2] = H::mul(eq_mle(rs.data(), comb.data(), mr), s);
        }
        rc = atlas_batched_sumcheck_check((const atlas_fr_t*)coeff, (const atlas_fr_t*)expect, 3, (const atlas_fr_t*)&e);
        return rc;
    }
    // verify_clamp_lookup (clamp_lookups/mod.rs:311-346): SaturationTable = ClampBoundedTable<64, 31, true>, witness = ClampAcc
    int clamp_lookup(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        Point ra_point;
        int rc = ps_unary(nd, T_CLAMP_SYM, 64, 31, gr::node_exec(gr::virt(gr::VP_ClampAcc, nd.idx), nd.idx), R.claim, R.point, gr::VP_ClampRa, gr::PT_Execution, &ra_point);
        if (!rc) rc = onehot_checks(nd, 64, R.point, ra_point, advice_claim(nd, gr::VP_ClampRa), gr::CP_ClampRaD, gr::PT_RaOneHotChecks);
        return rc;
    }

    // ---- stages
    int output_claim() {
        const Node& nd = G.nodes.at(G.outputs[0]);
        const size_t T = gr::padded_len(nd.dims);
        if (output_len != T) return fail(ATLAS_EINVAL, "verify_graph: output tensor length");
        const Point r = challenge_point(gr::log2u(T));
        H::Fr expect;
        int rc = eval_public(output, T, r, &expect);
        if (rc) return rc;
        const OpeningId id = gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.idx), nd.idx + 1);
        rc = append_virtual(id, r);
        if (rc) return rc;
        return same(expect, claim_of(id)) ? ATLAS_OK : bad("verify_graph: InvalidOpeningProof (the output claim does not match the output tensor)");
    }
    int eval_reduction(const Node& nd) {
        const size_t n = gr::log2u(gr::padded_len(nd.dims));
        const OpeningId lo = gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.idx), nd.idx), hi = gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.idx), ~(uint64_t)0);
        std::vector<atlas_fr_t> pts, cls;
        size_t N = 0;
        for (auto it = points.lower_bound(lo); it != points.end() && !(hi < it->first); ++it) {
            if (it->second.size() != n) return bad("verify_graph: an opening of a node output with the wrong number of variables");
            const size_t o = pts.size(); pts.resize(o + n);
            if (n) std::memcpy(&pts[o], it->second.data(), n * 32);
            atlas_fr_t c; std::memcpy(&c, &claims.at(it->first), 32); cls.push_back(c);
            N++;
        }
        if (N == 0) return bad("verify_graph: EmptyInput (a node output without opening claims)");
        auto h = evalred.find(nd.idx);
        if (h == evalred.end()) return bad("verify_graph: missing evaluation reduction proof");
        gr::Opening& R = reduced[nd.idx];
        R.point.resize(n);
        return atlas_eval_reduction_verify(pts.data(), cls.data(), N, n, (const atlas_fr_t*)h->second.data(), h->second.size(), &t, (atlas_fr_t*)R.point.data(), (atlas_fr_t*)&R.claim);
    }
    // the element-wise sumcheck of an operator: eq(r0, r') f(operand claims)
    int ew_verify(const Node& nd, size_t n_ops, size_t degree, const H::Fr& input_claim, uint8_t proof_type, int kind /* 0 product, 1 iff */) {
        const gr::Opening& R = reduced.at(nd.idx);
        const size_t n = R.point.size();
        H::Fr e; std::vector<H::Fr> rs;
        int rc = single(proof_type, n, degree, input_claim, &e, rs);
        if (rc) return rc;
        const Point pt = reversed(rs);
        for (size_t q = 0; q < n_ops && !rc; q++) rc = append_nodeio(nd, q, pt);
        if (rc) return rc;
        H::Fr f;
        if (kind == 1) { const H::Fr m = nodeio_claim(nd, 0); f = H::add(H::mul(m, nodeio_claim(nd, 1)), H::mul(H::sub(H::one(), m), nodeio_claim(nd, 2))); }
        else {
            f = nodeio_claim(nd, 0);
            if (nd.op == ATLAS_OP_MUL || nd.op == ATLAS_OP_AND) f = H::mul(f, nodeio_claim(nd, 1));
            if (nd.op == ATLAS_OP_SQUARE) f = H::mul(f, f);
            if (nd.op == ATLAS_OP_CUBE) f = H::mul(H::mul(f, f), f);
        }
        return same(e, H::mul(eq_mle(R.point.data(), pt.data(), n), f)) ? ATLAS_OK : bad("verify_graph: SumcheckVerificationError (element-wise operator)");
    }
    int einsum_verify(const Node& nd, const H::Fr& input_claim) {
        const gr::Opening& R = reduced.at(nd.idx);
        const int layout = (int)nd.p[0];
        const std::vector<size_t>& d = nd.shape;
        size_t b = 1, m = 1, k = 1, n = 1;
        switch (layout) {
            case ATLAS_EINSUM_MK_KN_MN: m = d[0]; k = d[1]; n = d[2]; break;
            case ATLAS_EINSUM_K_NK_N: k = d[0]; n = d[1]; break;
            default: b = d[0]; m = d[1]; k = d[2]; n = d[3]; break;
        }
        const size_t lb = gr::log2u(b), lm = gr::log2u(m), lk = gr::log2u(k), ln = gr::log2u(n);
        const H::Fr* r = R.point.data();
        const H::Fr *r_b = nullptr, *r_m = nullptr, *r_n = nullptr;
        bool bmk = false, with_eq = false;
        switch (layout) {
            case ATLAS_EINSUM_MK_KN_MN: r_m = r; r_n = r + lm; break;
            case ATLAS_EINSUM_K_NK_N: r_n = r; break;
            case ATLAS_EINSUM_BMK_BKN_MBN: case ATLAS_EINSUM_BMK_KBN_MBN: r_m = r; r_b = r + lm; r_n = r + lm + lb; bmk = true; with_eq = true; break;
            case ATLAS_EINSUM_MBK_BNK_BMN: case ATLAS_EINSUM_MBK_NBK_BMN: r_b = r; r_m = r + lb; r_n = r + lb + lm; with_eq = true; break;
            default: return fail(ATLAS_EINVAL, "verify_graph: einsum layout without a verifier");
        }
        const size_t nr = with_eq ? lk + lb : lk;
        H::Fr e; std::vector<H::Fr> ch;
        int rc = single(gr::PT_EinsumMatmul, nr, with_eq ? 3 : 2, input_claim, &e, ch);
        if (rc) return rc;
        auto cat = [](std::initializer_list<std::pair<const H::Fr*, size_t>> parts) { Point p; for (auto& x : parts) p.insert(p.end(), x.first, x.first + x.second); return p; };
        Point lp, rp;
        H::Fr eqv = H::one();
        switch (layout) {                                                     // EinsumLayout::{operand_points, output_eq}
            case ATLAS_EINSUM_MK_KN_MN: lp = cat({{r_m, lm}, {ch.data(), lk}}); rp = cat({{ch.data(), lk}, {r_n, ln}}); break;
            case ATLAS_EINSUM_K_NK_N: lp = cat({{ch.data(), lk}}); rp = cat({{r_n, ln}, {ch.data(), lk}}); break;
            default:
                if (bmk) {
                    const H::Fr *rj = ch.data(), *rh = ch.data() + lk;
                    lp = cat({{rh, lb}, {r_m, lm}, {rj, lk}});
                    rp = layout == ATLAS_EINSUM_BMK_BKN_MBN ? cat({{rh, lb}, {rj, lk}, {r_n, ln}}) : cat({{rj, lk}, {rh, lb}, {r_n, ln}});
                    eqv = eq_mle(r_b, rh, lb);
                } else {
                    const H::Fr *rh = ch.data(), *rj = ch.data() + lb;
                    lp = cat({{r_m, lm}, {ch.data(), lb + lk}});
                    rp = layout == ATLAS_EINSUM_MBK_BNK_BMN ? cat({{rh, lb}, {r_n, ln}, {rj, lk}}) : cat({{r_n, ln}, {ch.data(), lb + lk}});
                    eqv = eq_mle(r_b, rh, lb);
                }
        }
        rc = append_nodeio(nd, 0, lp);
        if (!rc) rc = append_nodeio(nd, 1, rp);
        if (rc) return rc;
        return same(e, H::mul(H::mul(nodeio_claim(nd, 0), nodeio_claim(nd, 1)), eqv)) ? ATLAS_OK : bad("verify_graph: SumcheckVerificationError (einsum)");
    }
    // impl_fused_rescale_proof_api / Einsum::verify: verify_pre, the operator's sumcheck, verify_post (fused_rebase.rs:281-340)
    int op_fused(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        if (R.point.empty()) return fail(ATLAS_EINVAL, "verify_graph: scalar fused-rescale nodes are not composed");
        const size_t S = nd.op == ATLAS_OP_EINSUM ? (size_t)nd.p[1] : nd.op == ATLAS_OP_CUBE ? 2 * (size_t)nd.p[0] : (size_t)nd.p[0];
        int rc = append_advice(nd, gr::VP_RescaleRemainder, R.point);        // cache_remainder_verify
        if (!rc) rc = clamp_lookup(nd);
        if (rc) return rc;
        const H::Fr eval_R = advice_claim(nd, gr::VP_RescaleRemainder), acc = advice_claim(nd, gr::VP_ClampAcc);
        const H::Fr in_claim = H::add(H::mul(acc, pow2_fr(S)), eval_R);      // fused_input_claim
        if (nd.op == ATLAS_OP_EINSUM) rc = einsum_verify(nd, in_claim);
        else rc = ew_verify(nd, nd.op == ATLAS_OP_MUL ? 2 : 1, nd.op == ATLAS_OP_CUBE ? 4 : 3, in_claim, gr::PT_RescaleArith, 0);
        if (rc) return rc;
        Point rr_point;
        rc = identity_rc(nd, S, eval_R, R.point, gr::VP_RescaleRemainderRa, gr::PT_RangeCheck, &rr_point);
        if (!rc) rc = onehot_checks(nd, S, R.point, rr_point, advice_claim(nd, gr::VP_RescaleRemainderRa), gr::CP_RescaleRemainderRaD, gr::PT_RescaleRemainderRaChecks);
        return rc;
    }
    int op_addsub(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        if (R.point.empty()) return fail(ATLAS_EINVAL, "verify_graph: scalar Add / Sub nodes are not composed");
        int rc = clamp_lookup(nd);
        if (!rc) rc = append_nodeio(nd, 0, R.point);
        if (!rc) rc = append_nodeio(nd, 1, R.point);
        if (rc) return rc;
        const H::Fr l = nodeio_claim(nd, 0), r = nodeio_claim(nd, 1), acc = advice_claim(nd, gr::VP_ClampAcc);
        return same(nd.op == ATLAS_OP_ADD ? H::add(l, r) : H::sub(l, r), acc) ? ATLAS_OK : bad("verify_graph: InvalidOpeningProof (left +- right must equal the accumulation)");
    }
    int op_relu(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        Point ra_point;
        int rc = ps_unary(nd, T_RELU, 32, 0, gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.inputs[0]), nd.idx), R.claim, R.point, gr::VP_NodeOutputRa, gr::PT_Execution, &ra_point);
        if (!rc) rc = onehot_checks(nd, 32, R.point, ra_point, advice_claim(nd, gr::VP_NodeOutputRa), gr::CP_NodeOutputRaD, gr::PT_RaOneHotChecks);
        return rc;
    }
    int op_reshape(const Node& nd) {                                          // selector = the eq table of the reduced point over the flat index
        const gr::Opening& R = reduced.at(nd.idx);
        const size_t n = R.point.size();
        H::Fr e; std::vector<H::Fr> rs;
        int rc = single(gr::PT_Execution, n, 2, R.claim, &e, rs);
        if (rc) return rc;
        const Point pt = reversed(rs);
        rc = append_nodeio(nd, 0, pt);
        if (rc) return rc;
        return same(e, H::mul(nodeio_claim(nd, 0), eq_mle(R.point.data(), pt.data(), n))) ? ATLAS_OK : bad("verify_graph: SumcheckVerificationError (reshape)");
    }
    int public_tensor(const Node& nd, const int32_t* host) {                  // Input / Constant: the verifier evaluates the tensor itself
        const gr::Opening& R = reduced.at(nd.idx);
        H::Fr expect;
        int rc = eval_public(host, gr::padded_len(nd.dims), R.point, &expect);
        if (rc) return rc;
        return same(expect, R.claim) ? ATLAS_OK : bad("verify_graph: InvalidOpeningProof (input / constant claim)");
    }
    int verify_node(const Node& nd, size_t& next_input_from_end) {
        cur = nd.idx;
        int rc = eval_reduction(nd);
        if (rc) return rc;
        const gr::Opening& R = reduced.at(nd.idx);
        switch (nd.op) {
            case ATLAS_OP_INPUT: { const std::vector<size_t> in = G.input_nodes(); const size_t pos = std::find(in.begin(), in.end(), nd.idx) - in.begin(); (void)next_input_from_end; return public_tensor(nd, inputs[pos]); }
            case ATLAS_OP_CONSTANT: return public_tensor(nd, nd.constant.data());
            case ATLAS_OP_IDENTITY: case ATLAS_OP_MOVEAXIS: case ATLAS_OP_BROADCAST: {
                Point r_in = R.point;
                if (nd.op == ATLAS_OP_MOVEAXIS) {                             // permute_challenge_groups (ops/moveaxis.rs)
                    std::vector<Point> groups; size_t off = 0;
                    for (size_t dim : nd.dims) { const size_t v = gr::log2u(dim); groups.emplace_back(R.point.begin() + off, R.point.begin() + off + v); off += v; }
                    const size_t src = (size_t)nd.p[0], dst = (size_t)nd.p[1];
                    Point gdst = groups[dst]; groups.erase(groups.begin() + dst); groups.insert(groups.begin() + src, gdst);
                    r_in.clear(); for (auto& gq : groups) r_in.insert(r_in.end(), gq.begin(), gq.end());
                } else if (nd.op == ATLAS_OP_BROADCAST) {                     // the operand at the variables of the non-broadcast axes; the unit tensor's MLE is 1
                    const Node& in = G.nodes.at(nd.inputs[0]);
                    const size_t off = nd.dims.size() - in.dims.size();
                    r_in.clear(); size_t pos = 0;
                    for (size_t a = 0; a < nd.dims.size(); a++) { const size_t v = gr::log2u(nd.dims[a]); if (a >= off && in.dims[a - off] == nd.dims[a]) r_in.insert(r_in.end(), R.point.begin() + pos, R.point.begin() + pos + v); pos += v; }
                }
                rc = append_nodeio(nd, 0, r_in);
                if (rc) return rc;
                return same(nodeio_claim(nd, 0), R.claim) ? ATLAS_OK : bad("verify_graph: InvalidOpeningProof (identity / moveaxis / broadcast claim)");
            }
            case ATLAS_OP_ADD: case ATLAS_OP_SUB: return op_addsub(nd);
            case ATLAS_OP_EINSUM: case ATLAS_OP_MUL: case ATLAS_OP_SQUARE: case ATLAS_OP_CUBE: return op_fused(nd);
            case ATLAS_OP_AND: return ew_verify(nd, 2, 3, R.claim, gr::PT_Execution, 0);
            case ATLAS_OP_IFF: return ew_verify(nd, 3, 3, R.claim, gr::PT_Execution, 1);
            case ATLAS_OP_RELU: return op_relu(nd);
            case ATLAS_OP_RESHAPE: return op_reshape(nd);
            default: return fail(ATLAS_EINVAL, "verify_graph: operator without a verifier composition");
        }
    }

    // get_committed_polynomials of every node, for the operators the verifier composes
    int collect_committed(const std::vector<atlas_g1_affine_t>& coms) {
        for (auto& kv : G.nodes) {
            const Node& nd = kv.second;
            const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T);
            if (T == 1) continue;
            auto chunks = [&](uint8_t cp, size_t log_K) { for (size_t i = 0; i < (log_K + 3) / 4; i++) committed[gr::comm(cp, nd.idx, i)].log_T = log_T; };
            switch (nd.op) {
                case ATLAS_OP_ADD: case ATLAS_OP_SUB: chunks(gr::CP_ClampRaD, 64); break;
                case ATLAS_OP_EINSUM: case ATLAS_OP_MUL: case ATLAS_OP_SQUARE: case ATLAS_OP_CUBE:
                    chunks(gr::CP_RescaleRemainderRaD, nd.op == ATLAS_OP_EINSUM ? (size_t)nd.p[1] : nd.op == ATLAS_OP_CUBE ? 2 * (size_t)nd.p[0] : (size_t)nd.p[0]);
                    chunks(gr::CP_ClampRaD, 64);
                    break;
                case ATLAS_OP_RELU: chunks(gr::CP_NodeOutputRaD, 32); break;
                default: break;
            }
        }
        if (coms.size() != committed.size()) return bad("verify_graph: the proof carries a different number of commitments than the model commits to");
        size_t i = 0;
        for (auto& kv : committed) kv.second.commitment = coms[i++];
        return ATLAS_OK;
    }

    // verify_reduced_openings (verifier.rs:140-187)
    int reduced_openings() {
        if (committed.empty()) return ro_present ? bad("verify_graph: a reduced opening proof without committed polynomials") : ATLAS_OK;
        if (!ro_present) return bad("verify_graph: MissingReductionProof");
        const size_t n = committed.size();
        if (ro_claims.size() != n) return bad("verify_graph: sumcheck_claims length");
        std::vector<H::Fr> ic(n); std::vector<size_t> nr(n), deg(n, 2);
        size_t mr = 0, i = 0;
        for (auto& kv : committed) {
            if (!kv.second.opened) return bad("verify_graph: a committed polynomial was never opened");
            ic[i] = kv.second.claim; nr[i] = kv.second.rounds(); mr = nr[i] > mr ? nr[i] : mr; i++;
        }
        if (ro_rows.rounds != mr) return bad("verify_graph: the opening-reduction sumcheck has the wrong number of rounds");
        std::vector<H::Fr> coeff(n); H::Fr e;
        std::vector<atlas_u128_t> ch(mr);
        int rc = atlas_batched_sumcheck_verify(ro_rows.c.data(), ro_rows.stride, ro_rows.n.data(), mr, (const atlas_fr_t*)ic.data(), nr.data(), deg.data(), n, &t,
                                               (atlas_fr_t*)coeff.data(), (atlas_fr_t*)&e, ch.data());
        if (rc) return rc;
        std::vector<H::Fr> rs(mr);
        for (size_t q = 0; q < mr; q++) rs[q] = ch_fr(ch[q]);
        std::vector<H::Fr> expect(n);
        i = 0;
        for (auto& kv : committed) { expect[i] = H::mul(eq_mle(kv.second.point.data(), rs.data() + (mr - nr[i]), nr[i]), ro_claims[i]); i++; }     // opening_reduction.rs:297-304
        rc = atlas_batched_sumcheck_check((const atlas_fr_t*)coeff.data(), (const atlas_fr_t*)expect.data(), n, (const atlas_fr_t*)&e);
        if (rc) return rc;
        H::tr_append_scalars(Tr, ro_claims.data(), n);                        // finalize_batch_opening_sumcheck
        std::vector<H::Fr> gam(n);
        { const H::Fr q = H::tr_challenge_scalar(Tr); gam[0] = H::one(); for (size_t k = 1; k < n; k++) gam[k] = H::mul(gam[k - 1], q); }
        // PCS::combine_commitments: sum gamma_i C_i (on the device: the commitments as a throw-away set of bases)
        const size_t np2 = gr::next_pow2(n);                                  // polynomial handles have power-of-two lengths: pad with (infinity, 0)
        std::vector<atlas_g1_affine_t> bases(np2);
        std::memset(bases.data(), 0, np2 * sizeof(atlas_g1_affine_t));
        for (size_t k = n; k < np2; k++) bases[k].infinity = 1;
        std::vector<H::Fr> gpad(np2, H::zero());
        std::copy(gam.begin(), gam.end(), gpad.begin());
        i = 0;
        for (auto& kv : committed) bases[i++] = kv.second.commitment;
        atlas_srs_t tmp = nullptr; atlas_poly_t sc = nullptr;
        atlas_g1_affine_t joint;
        rc = atlas_srs_upload(bases.data(), np2, sizeof(atlas_g1_affine_t), &tmp);
        if (!rc) rc = atlas_poly_upload_fr((const atlas_fr_t*)gpad.data(), np2, &sc);
        if (!rc) rc = atlas_msm_poly(tmp, 0, sc, &joint);
        if (sc) atlas_poly_free(sc);
        if (tmp) atlas_srs_free(tmp);
        if (rc) return rc;
        // compute_joint_claim (opening_proof.rs:1016-1036): shorter polynomials sit in the low corner of the joint one
        H::Fr jc = H::zero();
        const H::Fr one = H::one();
        for (size_t k = 0; k < n; k++) {
            H::Fr lag = one;
            for (size_t q = 0; q < mr - nr[k]; q++) lag = H::mul(lag, H::sub(one, rs[q]));
            jc = H::add(jc, H::mul(H::mul(gam[k], ro_claims[k]), lag));
        }
        if (ro_com.size() + 1 != mr || ro_v.size() != 3 * mr) return bad("verify_graph: HyperKZG proof shape");
        return atlas_hyperkzg_verify(vk, &joint, ch.data(), mr, (const atlas_fr_t*)&jc, ro_com.data(), ro_w.data(), ro_v.data(), &t);
    }

    int parse(const uint8_t* proof, size_t len, std::vector<atlas_g1_affine_t>& coms) {
        Reader R{proof, len};
        const uint64_t n_open = R.u64();
        for (uint64_t i = 0; i < n_open && R.ok; i++) {                      // (OpeningId, claim): opening_proof.rs:1313-1333
            OpeningId id;
            const uint8_t tag = R.u8();
            id.poly.committed = tag == 0 ? 1 : 0; id.poly.var = R.u8();
            const int ar = id.poly.committed ? gr::cp_arity(id.poly.var) : gr::vp_arity(id.poly.var);
            if (ar >= 1) id.poly.a = R.u64();
            if (ar >= 2) id.poly.b = R.u64();
            id.sc = R.u8();
            if (id.sc == gr::SC_NodeExecution || id.sc == gr::SC_RLC) id.sc_idx = R.u64();
            H::Fr c; R.fr(&c);
            claims[id] = c;
        }
        const uint64_t n_pr = R.u64();
        for (uint64_t i = 0; i < n_pr && R.ok; i++) { gr::ProofId id; id.node = R.u64(); id.type = R.u8(); Rows rw; R.rows(rw); proofs[id] = std::move(rw); }
        const uint64_t n_c = R.u64();
        if (!R.ok || n_c > ((uint64_t)1 << 24)) return bad("verify_graph: malformed proof");
        coms.resize(n_c);
        for (auto& c : coms) R.g1(&c);
        const uint64_t n_e = R.u64();
        for (uint64_t i = 0; i < n_e && R.ok; i++) { const uint64_t node = R.u64(), l = R.u64(); if (l > 4096) { R.ok = false; break; } std::vector<H::Fr> h(l); for (auto& x : h) R.fr(&x); evalred[node] = std::move(h); }
        ro_present = R.u8() == 1;
        if (R.ok && ro_present) {
            R.rows(ro_rows);
            const uint64_t nc = R.u64();
            if (!R.ok || nc > ((uint64_t)1 << 24)) return bad("verify_graph: malformed proof");
            ro_claims.resize(nc); for (auto& x : ro_claims) R.fr(&x);
            const uint64_t ncom = R.u64();
            if (!R.ok || ncom > 64) return bad("verify_graph: malformed proof");
            ro_com.resize(ncom); for (auto& x : ro_com) R.g1(&x);
            if (R.u64() != 3) R.ok = false;
            ro_w.resize(3); for (auto& x : ro_w) R.g1(&x);
            if (R.u64() != 3) R.ok = false;
            for (int k = 0; k < 3 && R.ok; k++) { const uint64_t ell = R.u64(); if (ell != ncom + 1) { R.ok = false; break; } for (uint64_t j = 0; j < ell; j++) { H::Fr x; R.fr(&x); atlas_fr_t y; std::memcpy(&y, &x, 32); ro_v.push_back(y); } }
        }
        if (!R.ok || R.left != 0) return bad("verify_graph: malformed proof (ark deserialisation)");
        return ATLAS_OK;
    }
};

}  // namespace

extern "C" int atlas_verify_graph(atlas_graph_t G, const atlas_hyperkzg_vk_t* vk, const int32_t* const* inputs, size_t n_inputs, const int32_t* output, size_t output_len,
                                  const uint8_t* proof, size_t proof_len, atlas_transcript_t* final_transcript) {
    NEED_INIT();
    if (!G || !vk || (!inputs && n_inputs) || !output || !proof) return fail(ATLAS_EINVAL, "verify_graph: null argument");
    if (G->outputs.empty() || n_inputs != G->input_nodes().size()) return fail(ATLAS_EINVAL, "verify_graph: outputs / inputs");
    Verifier V(*G, vk);
    V.inputs = inputs; V.output = output; V.output_len = output_len;
    int rc = atlas_transcript_new(&V.t, (const uint8_t*)"ONNXProof", 9);
    if (rc) return rc;
    {   // the inputs enter the transcript as on the prover's side (onnx_proof/mod.rs:90-122)
        const std::vector<size_t> in_nodes = G->input_nodes();
        H::tr_append_message(V.Tr, "model_inputs");
        H::tr_append_u64(V.Tr, in_nodes.size());
        H::tr_append_u64(V.Tr, in_nodes.size());
        for (size_t i = 0; i < in_nodes.size(); i++) {
            const Node& nd = G->nodes.at(in_nodes[i]);
            H::tr_append_u64(V.Tr, nd.idx);
            H::tr_append_u64(V.Tr, nd.dims.size());
            for (size_t d : nd.dims) H::tr_append_u64(V.Tr, d);
            H::tr_append_bytes(V.Tr, (const uint8_t*)inputs[i], nd.numel() * 4);
        }
    }
    std::vector<atlas_g1_affine_t> coms;
    rc = V.parse(proof, proof_len, coms);
    if (!rc) rc = V.collect_committed(coms);
    if (rc) return rc;
    for (auto& c : coms) {                                                    // populate_accumulator: append_serializable(commitment)
        uint8_t b[64], rev[64];
        rc = atlas_g1_to_bytes_uncompressed(&c, b);
        if (rc) return rc;
        for (int i = 0; i < 64; i++) rev[i] = b[63 - i];
        H::tr_append_bytes(V.Tr, rev, 64);
    }
    rc = V.output_claim();
    size_t dummy = 0;
    for (auto it = G->nodes.rbegin(); it != G->nodes.rend() && !rc; ++it) rc = V.verify_node(it->second, dummy);
    if (!rc) rc = V.reduced_openings();
    if (!rc && final_transcript) *final_transcript = V.t;
    return rc;
}
