// ONNXProof::verify for a graph resident in the library (jolt-atlas-core/src/onnx_proof/mod.rs:207-241, verifier.rs):
//   populate_accumulator        verifier.rs:66-80      opening claims from the proof, commitments into the transcript
//   verify_output_claim         verifier.rs:83-119     the output tensor's MLE at the transcript's point
//   verify_iop                  verifier.rs:123-137    nodes in reverse: NodeEvalReduction::verify, then the operator's verifier instances
//   verify_reduced_openings     verifier.rs:140-187    the batched opening-reduction sumcheck, the joint commitment, HyperKZG::verify
// Host arithmetic, like the reference's verifier; the device only evaluates the PUBLIC tensors (inputs, constants, output) at the
// verifier's points and adds up the joint commitment.  Nothing of the prover's trace is read.
// Every operator the graph prover composes has its verifier composition here: Input, Constant, Identity, Add, Sub, Mul, Square, Cube, Einsum,
// And, Iff, ReLU, Clamp, Neg, IsNan, Reshape, MoveAxis, Broadcast, Sum, ScalarConstDiv, Slice, Concat, Div, MeanOfSquares, Rsqrt, Tanh, Erf, Sigmoid, Sin, Cos, GatherLarge, GatherSmall, SoftmaxLastAxis.
// A verifier instance is a VInst (input claim, rounds, degree, and a closure = cache_openings + expected_output_claim); `run_single` is
// Sumcheck::verify, `batch` is BatchedSumcheck::verify (an instance of n rounds sees the LAST n challenges, sumcheck.rs:150-170).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>

#include "graph_state.hip.h"
#include "host_curve.hpp"

int atlas_rt_sum_config(const std::vector<size_t>& idims, size_t axis, size_t& m, size_t& n, int& ax);

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

// RightShiftTable<XLEN>::evaluate_mle by D bits (lookup_tables/right_shift.rs): bit i (MSB first) weighs 2^(XLEN - 1 - i - D) while it stays above the cut
H::Fr rshift_mle(const H::Fr* r, size_t xlen, size_t D) {
    H::Fr y = H::zero();
    for (size_t i = 0; i + D < xlen; i++) y = H::add(y, H::mul(r[i], pow2_fr(xlen - 1 - i - D)));
    return y;
}
H::Fr fr_i64(int64_t v) { return v >= 0 ? H::from_u64((uint64_t)v) : H::neg(H::from_u64((uint64_t)(-v))); }
// UnsignedLessThanTable<XLEN>::evaluate_mle over interleaved operands (unsigned_less_than.rs:26-43)
H::Fr ult_mle(const H::Fr* r, size_t xlen) {
    const H::Fr one = H::one();
    H::Fr res = H::zero(), eqt = one;
    for (size_t i = 0; i < xlen; i++) {
        const H::Fr x = r[2 * i], y = r[2 * i + 1];
        res = H::add(res, H::mul(H::mul(H::sub(one, x), y), eqt));
        eqt = H::mul(eqt, H::add(H::mul(x, y), H::mul(H::sub(one, x), H::sub(one, y))));
    }
    return res;
}
// SignedOperandPoly::evaluate (signed_identity_poly.rs:255-274): the two's-complement value of one interleaved operand (shift 0 = left, 1 = right)
H::Fr signed_operand_mle(const H::Fr* r, size_t xlen, size_t shift) {
    H::Fr y = H::zero();
    for (size_t i = 0; i < xlen; i++) y = H::add(y, H::mul(r[2 * i + shift], pow2_fr(xlen - 1 - i)));
    return H::sub(y, H::mul(r[shift], pow2_fr(xlen)));
}
// EqPolynomial::evals: big-endian table of eq(r, .)
std::vector<H::Fr> eq_table(const H::Fr* r, size_t n) {
    std::vector<H::Fr> ev((size_t)1 << n);
    ev[0] = H::one();
    size_t len = 1;
    for (size_t i = 0; i < n; i++) {
        for (size_t j = len; j-- > 0;) { const H::Fr hi = H::mul(ev[j], r[i]); ev[2 * j + 1] = hi; ev[2 * j] = H::sub(ev[j], hi); }
        len <<= 1;
    }
    return ev;
}
// MultilinearPolynomial::from(Vec<i32>).evaluate(point) of a short public vector
H::Fr mle_i32(const std::vector<int32_t>& v, const H::Fr* r, size_t n) {
    const std::vector<H::Fr> ev = eq_table(r, n);
    H::Fr acc = H::zero();
    for (size_t i = 0; i < v.size() && i < ev.size(); i++) acc = H::add(acc, H::mul(ev[i], fr_i64(v[i])));
    return acc;
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
    int mode() const { return rt().challenge_mode; }

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
        if (n_rounds == 0) { *e = input_claim; rs.clear(); return ATLAS_OK; }     // a scalar node: the claim is compared with the expected claim as it is
        std::vector<atlas_u128_t> ch(n_rounds ? n_rounds : 1);
        int rc = atlas_sumcheck_proof_verify(R.c.data(), R.stride, R.n.data(), n_rounds, degree, (const atlas_fr_t*)&input_claim, &t, (atlas_fr_t*)e, ch.data());
        if (rc) return rc;
        rs.resize(n_rounds);
        for (size_t i = 0; i < n_rounds; i++) rs[i] = ch_fr(ch[i]);
        return ATLAS_OK;
    }
    static Point reversed(const std::vector<H::Fr>& v, size_t from = 0) { return Point(v.rbegin(), v.rend() - from); }
    static bool same(const H::Fr& a, const H::Fr& b) { return std::memcmp(&a, &b, 32) == 0; }

    // ---- verifier instances
    struct VInst { H::Fr claim; size_t rounds = 0, degree = 0; std::function<int(const H::Fr* ch, H::Fr* expected)> finish; };
    // Sumcheck::verify (sumcheck.rs:103-138): cache_openings, then the final claim against expected_output_claim
    int run_single(uint8_t proof_type, const VInst& I, const char* what) {
        H::Fr e, expect; std::vector<H::Fr> rs;
        int rc = single(proof_type, I.rounds, I.degree, I.claim, &e, rs);
        if (!rc) rc = I.finish(rs.data(), &expect);
        if (rc) return rc;
        return same(e, expect) ? ATLAS_OK : bad(what);
    }
    // BatchedSumcheck::verify (sumcheck.rs:252-330)
    int batch(uint8_t proof_type, const std::vector<VInst>& I) {
        auto it = proofs.find(gr::ProofId{cur, proof_type});
        if (it == proofs.end()) return bad("verify_graph: MissingProof (batched sumcheck)");
        const Rows& R = it->second;
        const size_t n = I.size();
        std::vector<H::Fr> ic(n), coeff(n), expect(n); std::vector<size_t> nr(n), deg(n);
        size_t mr = 0;
        for (size_t i = 0; i < n; i++) { ic[i] = I[i].claim; nr[i] = I[i].rounds; deg[i] = I[i].degree; mr = nr[i] > mr ? nr[i] : mr; }
        if (R.rounds != mr) return bad("verify_graph: a batched sumcheck with the wrong number of rounds");
        std::vector<atlas_u128_t> ch(mr ? mr : 1);
        H::Fr e;
        int rc = atlas_batched_sumcheck_verify(R.c.data(), R.stride, R.n.data(), mr, (const atlas_fr_t*)ic.data(), nr.data(), deg.data(), n, &t, (atlas_fr_t*)coeff.data(), (atlas_fr_t*)&e, ch.data());
        if (rc) return rc;
        std::vector<H::Fr> rs(mr);
        for (size_t i = 0; i < mr; i++) rs[i] = ch_fr(ch[i]);
        for (size_t i = 0; i < n && !rc; i++) rc = I[i].finish(rs.data() + (mr - nr[i]), &expect[i]);
        if (rc) return rc;
        return atlas_batched_sumcheck_check((const atlas_fr_t*)coeff.data(), (const atlas_fr_t*)expect.data(), n, (const atlas_fr_t*)&e);
    }
    // the opening point of a read-raf ra polynomial: address challenges as drawn, cycle challenges reversed (ps_shout/mod.rs:150-158)
    static Point ra_point_of(const H::Fr* ch, size_t log_K, size_t log_T) {
        Point pt(ch, ch + log_K);
        for (size_t q = 0; q < log_T; q++) pt.push_back(ch[log_K + log_T - 1 - q]);
        return pt;
    }

    // ---- a unary prefix-suffix lookup: read_raf_verify (op_lookups/mod.rs:270-283) + its verifier instance (ps_shout/mod.rs:612-643)
    enum Table { T_RELU, T_CLAMP_SYM, T_CLAMP, T_RSHIFT };
    int inst_ps_unary(const Node& nd, Table tab, size_t xlen, size_t bound, const OpeningId& witness_id, const H::Fr& rv_claim, const Point& r_cycle, uint8_t ra_vp, Point* ra_point, VInst* out) {
        int rc = append_virtual(witness_id, r_cycle);                        // append_raf_claims_verifier
        if (rc) return rc;
        const H::Fr gamma = H::tr_challenge_scalar(Tr);                      // ps_read_raf_verifier
        const size_t log_T = r_cycle.size();
        const Node* np = &nd;
        out->claim = H::add(rv_claim, H::mul(gamma, claim_of(witness_id))); out->rounds = xlen + log_T; out->degree = 2;
        out->finish = [=](const H::Fr* ch, H::Fr* expect) {
            const Point pt = ra_point_of(ch, xlen, log_T);
            int rc2 = append_advice(*np, ra_vp, pt);                         // cache_openings
            if (rc2) return rc2;
            const H::Fr val = tab == T_RELU ? relu_mle(pt.data(), xlen) : tab == T_RSHIFT ? rshift_mle(pt.data(), xlen, bound) : clamp_mle(pt.data(), xlen, bound, tab == T_CLAMP_SYM);
            *expect = H::mul(H::mul(eq_mle(r_cycle.data(), pt.data() + xlen, log_T), advice_claim(*np, ra_vp)), H::add(val, H::mul(gamma, signed_identity_mle(pt.data(), xlen))));
            *ra_point = pt;
            return (int)ATLAS_OK;
        };
        return ATLAS_OK;
    }
    int ps_unary(const Node& nd, Table tab, size_t xlen, size_t bound, const OpeningId& witness_id, const H::Fr& rv_claim, const Point& r_cycle, uint8_t ra_vp,
                 uint8_t proof_type, Point* ra_point) {
        VInst I;
        int rc = inst_ps_unary(nd, tab, xlen, bound, witness_id, rv_claim, r_cycle, ra_vp, ra_point, &I);
        if (!rc) rc = run_single(proof_type, I, "verify_graph: SumcheckVerificationError (prefix-suffix lookup)");
        return rc;
    }
    // the binary range check `left < right` (range_checking/mod.rs:37-100, ps_shout/binary.rs): UnsignedLessThanTable<32> on the interleaved
    // operands; every lookup's value is 1, the operands ride along as gamma left + gamma^2 right
    VInst inst_range_check(const Node& nd, const Point& r_cycle, const H::Fr& left, const H::Fr& right, uint8_t ra_vp, Point* ra_point) {
        const H::Fr gamma = H::tr_challenge_scalar(Tr);
        const size_t log_T = r_cycle.size();
        const Node* np = &nd;
        VInst I;
        I.claim = H::add(H::one(), H::add(H::mul(gamma, left), H::mul(H::mul(gamma, gamma), right))); I.rounds = 64 + log_T; I.degree = 2;
        I.finish = [=](const H::Fr* ch, H::Fr* expect) {
            const Point pt = ra_point_of(ch, 64, log_T);
            int rc2 = append_advice(*np, ra_vp, pt);
            if (rc2) return rc2;
            const H::Fr val = H::add(ult_mle(pt.data(), 32), H::mul(gamma, H::add(signed_operand_mle(pt.data(), 32, 0), H::mul(gamma, signed_operand_mle(pt.data(), 32, 1)))));
            *expect = H::mul(H::mul(eq_mle(r_cycle.data(), pt.data() + 64, log_T), advice_claim(*np, ra_vp)), val);
            *ra_point = pt;
            return (int)ATLAS_OK;
        };
        return I;
    }
    // IdentityRCVerifier (identity_range_check.rs:455-473)
    VInst inst_identity_rc(const Node& nd, size_t log_K, const H::Fr& input_claim, const Point& r_cycle, uint8_t ra_vp, Point* ra_point) {
        const size_t log_T = r_cycle.size();
        const Node* np = &nd;
        VInst I;
        I.claim = input_claim; I.rounds = log_K + log_T; I.degree = 2;
        I.finish = [=](const H::Fr* ch, H::Fr* expect) {
            const Point pt = ra_point_of(ch, log_K, log_T);
            int rc2 = append_advice(*np, ra_vp, pt);
            if (rc2) return rc2;
            *expect = H::mul(H::mul(eq_mle(r_cycle.data(), pt.data() + log_K, log_T), advice_claim(*np, ra_vp)), identity_mle(pt.data(), log_K));
            *ra_point = pt;
            return (int)ATLAS_OK;
        };
        return I;
    }
    int identity_rc(const Node& nd, size_t log_K, const H::Fr& input_claim, const Point& r_cycle, uint8_t ra_vp, uint8_t proof_type, Point* ra_point) {
        return run_single(proof_type, inst_identity_rc(nd, log_K, input_claim, r_cycle, ra_vp, ra_point), "verify_graph: SumcheckVerificationError (identity range check)");
    }
    // the dense Shout read-raf verifier (shout.rs:330-420): sum_k ra(k) (table(k) + gamma k); ra opened at (challenges | r_cycle)
    VInst inst_shout(const Node& nd, const std::vector<int32_t>& table, size_t log_K, const H::Fr& rv_claim, const H::Fr& raf_claim, const Point& r_cycle, uint8_t ra_vp, Point* ra_point) {
        const H::Fr gamma = H::tr_challenge_scalar(Tr);                      // ReadRafParams::new
        const Node* np = &nd;
        const std::vector<int32_t>* tp = &table;
        VInst I;
        I.claim = H::add(rv_claim, H::mul(gamma, raf_claim)); I.rounds = log_K; I.degree = 2;
        I.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point pt(ch, ch + log_K);
            pt.insert(pt.end(), r_cycle.begin(), r_cycle.end());
            int rc2 = append_advice(*np, ra_vp, pt);
            if (rc2) return rc2;
            *expect = H::mul(advice_claim(*np, ra_vp), H::add(mle_i32(*tp, ch, log_K), H::mul(gamma, identity_mle(ch, log_K))));
            *ra_point = pt;
            return (int)ATLAS_OK;
        };
        return I;
    }
    // ra_onehot_verifiers (shout.rs:474-529): [RaVirtual, HammingWeight, Booleanity] of one lookup family, appended to `out`
    void onehot_insts(const Node& nd, size_t log_K, const Point& r_cycle, const Point& ra_point, const H::Fr& ra_claim, uint8_t rad_cp, std::vector<VInst>& out) {
        const size_t lkc = 4, d = (log_K + lkc - 1) / lkc, log_T = r_cycle.size(), pad = d * lkc - log_K;
        std::vector<H::Fr> gp(d);
        { const H::Fr q = H::tr_challenge_scalar(Tr); gp[0] = H::one(); for (size_t i = 1; i < d; i++) gp[i] = H::mul(gp[i - 1], q); }
        const Point gammas = challenge_point(d), r_addr = challenge_point(lkc);
        H::Fr hw_claim = H::zero();
        for (auto& x : gp) hw_claim = H::add(hw_claim, x);
        const Node* np = &nd;
        VInst ra, hw, bo;
        // RaVirtual: openings at (chunk i of the zero-padded r_address | reversed cycle challenges); eq(r_cycle_ra, .) prod ra_i
        ra.claim = ra_claim; ra.rounds = log_T; ra.degree = d + 1;
        ra.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point rc_rev(log_T);
            for (size_t q = 0; q < log_T; q++) rc_rev[q] = ch[log_T - 1 - q];
            H::Fr prod = H::one();
            for (size_t i = 0; i < d; i++) {
                Point pt(lkc + log_T);
                for (size_t q = 0; q < lkc; q++) { const size_t pos = i * lkc + q; pt[q] = pos < pad ? H::zero() : ra_point[pos - pad]; }
                std::copy(rc_rev.begin(), rc_rev.end(), pt.begin() + lkc);
                int rc2 = append_sparse(rad_cp, *np, i, gr::SC_RaVirtualization, pt);
                if (rc2) return rc2;
                prod = H::mul(prod, claim_of(gr::oid(gr::comm(rad_cp, np->idx, i), gr::SC_RaVirtualization)));
            }
            *expect = H::mul(eq_mle(ra_point.data() + log_K, rc_rev.data(), log_T), prod);
            return (int)ATLAS_OK;
        };
        // HammingWeight: (reversed address challenges | r_cycle); sum gamma^i ra_i
        hw.claim = hw_claim; hw.rounds = lkc; hw.degree = 1;
        hw.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point pt(lkc + log_T);
            for (size_t q = 0; q < lkc; q++) pt[q] = ch[lkc - 1 - q];
            std::copy(r_cycle.begin(), r_cycle.end(), pt.begin() + lkc);
            H::Fr s = H::zero();
            for (size_t i = 0; i < d; i++) {
                int rc2 = append_sparse(rad_cp, *np, i, gr::SC_HammingWeight, pt);
                if (rc2) return rc2;
                s = H::add(s, H::mul(claim_of(gr::oid(gr::comm(rad_cp, np->idx, i), gr::SC_HammingWeight)), gp[i]));
            }
            *expect = s;
            return (int)ATLAS_OK;
        };
        // Booleanity: both halves reversed; eq(challenges, rev(r_address) | rev(r_cycle)) sum gamma_i (ra_i^2 - ra_i)
        bo.claim = H::zero(); bo.rounds = lkc + log_T; bo.degree = 3;
        bo.finish = [=](const H::Fr* ch, H::Fr* expect) {
            const size_t mr = lkc + log_T;
            Point pt(mr), comb(mr);
            for (size_t q = 0; q < lkc; q++) { pt[q] = ch[lkc - 1 - q]; comb[q] = r_addr[lkc - 1 - q]; }
            for (size_t q = 0; q < log_T; q++) { pt[lkc + q] = ch[mr - 1 - q]; comb[lkc + q] = r_cycle[log_T - 1 - q]; }
            H::Fr s = H::zero();
            for (size_t i = 0; i < d; i++) {
                int rc2 = append_sparse(rad_cp, *np, i, gr::SC_Booleanity, pt);
                if (rc2) return rc2;
                const H::Fr c = claim_of(gr::oid(gr::comm(rad_cp, np->idx, i), gr::SC_Booleanity));
                s = H::add(s, H::mul(H::sub(H::mul(c, c), c), gammas[i]));
            }
            *expect = H::mul(eq_mle(ch, comb.data(), mr), s);
            return (int)ATLAS_OK;
        };
        out.push_back(std::move(ra)); out.push_back(std::move(hw)); out.push_back(std::move(bo));
    }
    int onehot_checks(const Node& nd, size_t log_K, const Point& r_cycle, const Point& ra_point, const H::Fr& ra_claim, uint8_t rad_cp, uint8_t proof_type) {
        std::vector<VInst> I;
        onehot_insts(nd, log_K, r_cycle, ra_point, ra_claim, rad_cp, I);
        return batch(proof_type, I);
    }
    // verify_clamp_lookup (clamp_lookups/mod.rs:311-346): SaturationTable = ClampBoundedTable<64, 31, true>, witness = ClampAcc
    int clamp_lookup(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        Point ra_point;
        int rc = ps_unary(nd, T_CLAMP_SYM, 64, 31, gr::node_exec(gr::virt(gr::VP_ClampAcc, nd.idx), nd.idx), R.claim, R.point, gr::VP_ClampRa, gr::PT_Execution, &ra_point);
        if (!rc) rc = onehot_checks(nd, 64, R.point, ra_point, advice_claim(nd, gr::VP_ClampRa), gr::CP_ClampRaD, gr::PT_RaOneHotChecks);
        return rc;
    }
    // recover_small_int (clamp_lookups/mod.rs:78-86)
    static bool small_int(const H::Fr& x, int64_t* out) {
        uint64_t c[4], m[4];
        H::to_canonical(x, c);
        H::to_canonical(H::neg(x), m);
        if (!c[1] && !c[2] && !c[3] && c[0] < ((uint64_t)1 << 63)) { *out = (int64_t)c[0]; return true; }
        if (!m[1] && !m[2] && !m[3] && m[0] < ((uint64_t)1 << 63)) { *out = -(int64_t)m[0]; return true; }
        return false;
    }
    // verify_scalar_clamp (clamp_lookups/mod.rs:365-382): output == SatClamp_i32(the accumulation read in the clear)
    int scalar_clamp_of(const H::Fr& combined, const H::Fr& output_claim) {
        int64_t acc;
        if (!small_int(combined, &acc)) return bad("verify_graph: InvalidOpeningProof (scalar: the accumulation claim is not a small signed integer)");
        const int64_t cl = acc > INT32_MAX ? INT32_MAX : acc < INT32_MIN ? INT32_MIN : acc;
        return same(fr_i64(cl), output_claim) ? ATLAS_OK : bad("verify_graph: InvalidOpeningProof (scalar: output must equal SatClamp(input))");
    }
    // verify_append_acc + verify_scalar_clamp: a scalar node's accumulation opens in the clear
    int scalar_clamp(const Node& nd) {
        const OpeningId id = gr::node_exec(gr::virt(gr::VP_ClampAcc, nd.idx), nd.idx);
        int rc = append_virtual(id, reduced.at(nd.idx).point);
        if (rc) return rc;
        return scalar_clamp_of(claim_of(id), reduced.at(nd.idx).claim);
    }
    // a dense committed polynomial of this node (VerifierOpeningAccumulator::append_dense)
    int append_dense(const Node& nd, uint8_t cp, const Point& pt) {
        const PolyId p = gr::comm(cp, nd.idx);
        const OpeningId id = gr::node_exec(p, nd.idx);
        auto it = claims.find(id);
        if (it == claims.end()) return bad("verify_graph: a dense opening claim is not in the proof");
        H::tr_append_scalar(Tr, it->second);
        points[id] = pt;
        auto c = committed.find(p);
        if (c == committed.end()) return bad("verify_graph: an opening of a polynomial that is not committed");
        c->second.opened = true; c->second.point = pt; c->second.claim = it->second;
        return ATLAS_OK;
    }
    H::Fr dense_claim(const Node& nd, uint8_t cp) const { return claims.at(gr::node_exec(gr::comm(cp, nd.idx), nd.idx)); }
    H::Fr current_claim(const Node& nd) const { return claims.at(gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.idx), nd.idx)); }
    int append_current(const Node& nd, const Point& pt) { return append_virtual(gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.idx), nd.idx), pt); }

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
        const bool scalar = R.point.empty();                                   // is_scalar: verify_append_acc instead of the lookup (fused_rebase.rs:281-293)
        const size_t S = nd.op == ATLAS_OP_EINSUM ? (size_t)nd.p[1] : nd.op == ATLAS_OP_CUBE ? 2 * (size_t)nd.p[0] : (size_t)nd.p[0];
        int rc = append_advice(nd, gr::VP_RescaleRemainder, R.point);        // cache_remainder_verify
        if (!rc) rc = scalar ? append_advice(nd, gr::VP_ClampAcc, R.point) : clamp_lookup(nd);
        if (rc) return rc;
        const H::Fr eval_R = advice_claim(nd, gr::VP_RescaleRemainder), acc = advice_claim(nd, gr::VP_ClampAcc);
        const H::Fr in_claim = H::add(H::mul(acc, pow2_fr(S)), eval_R);      // fused_input_claim
        if (nd.op == ATLAS_OP_EINSUM) rc = einsum_verify(nd, in_claim);
        else rc = ew_verify(nd, nd.op == ATLAS_OP_MUL ? 2 : 1, nd.op == ATLAS_OP_CUBE ? 4 : 3, in_claim, gr::PT_RescaleArith, 0);
        if (rc) return rc;
        if (scalar) {                                                         // verify_post, scalar: remainder in [0, 2^S) and the clamp, in the clear
            int64_t rv;
            if (!small_int(eval_R, &rv) || rv < 0 || rv >= ((int64_t)1 << S)) return bad("verify_graph: InvalidOpeningProof (scalar fused rescale: remainder must lie in [0, 2^S))");
            return scalar_clamp_of(acc, R.claim);
        }
        Point rr_point;
        rc = identity_rc(nd, S, eval_R, R.point, gr::VP_RescaleRemainderRa, gr::PT_RangeCheck, &rr_point);
        if (!rc) rc = onehot_checks(nd, S, R.point, rr_point, advice_claim(nd, gr::VP_RescaleRemainderRa), gr::CP_RescaleRemainderRaD, gr::PT_RescaleRemainderRaChecks);
        return rc;
    }
    int op_addsub(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        const bool scalar = R.point.empty();                                   // is_scalar (ops/add.rs:107-150): no lookup, the operands open in the clear
        int rc = scalar ? (int)ATLAS_OK : clamp_lookup(nd);
        if (!rc) rc = append_nodeio(nd, 0, R.point);
        if (!rc) rc = append_nodeio(nd, 1, R.point);
        if (rc) return rc;
        if (scalar) { const H::Fr a = nodeio_claim(nd, 0), b = nodeio_claim(nd, 1); return scalar_clamp_of(nd.op == ATLAS_OP_ADD ? H::add(a, b) : H::sub(a, b), R.claim); }
        const H::Fr l = nodeio_claim(nd, 0), r = nodeio_claim(nd, 1), acc = advice_claim(nd, gr::VP_ClampAcc);
        return same(nd.op == ATLAS_OP_ADD ? H::add(l, r) : H::sub(l, r), acc) ? ATLAS_OK : bad("verify_graph: InvalidOpeningProof (left +- right must equal the accumulation)");
    }
    int op_relu(const Node& nd) {                                             // ReLU (ops/relu.rs) and Clamp (ops/clamp.rs: ClampTable<32> = ClampBoundedTable<32, 9, true>)
        const gr::Opening& R = reduced.at(nd.idx);
        const bool clamp = nd.op == ATLAS_OP_CLAMP;
        const uint8_t ra_vp = clamp ? gr::VP_SymmetricClampRa : gr::VP_NodeOutputRa, rad_cp = clamp ? gr::CP_SymmetricClampRaD : gr::CP_NodeOutputRaD;
        Point ra_point;
        int rc = ps_unary(nd, clamp ? T_CLAMP_SYM : T_RELU, 32, clamp ? gr::CLAMP_BOUND : 0, gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.inputs[0]), nd.idx), R.claim, R.point, ra_vp, gr::PT_Execution, &ra_point);
        if (!rc) rc = onehot_checks(nd, 32, R.point, ra_point, advice_claim(nd, ra_vp), rad_cp, gr::PT_RaOneHotChecks);
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
    // range_check + its one-hot checks on their own proofs (ops/div.rs verify_range_and_onehot, mean_of_squares.rs)
    int range_and_onehot(const Node& nd, const Point& r_cycle, const H::Fr& left, const H::Fr& right, uint8_t ra_vp, uint8_t rad_cp, uint8_t pt_onehot) {
        Point ra_point;
        int rc = run_single(gr::PT_RangeCheck, inst_range_check(nd, r_cycle, left, right, ra_vp, &ra_point), "verify_graph: SumcheckVerificationError (range check)");
        if (!rc) rc = onehot_checks(nd, 64, r_cycle, ra_point, advice_claim(nd, ra_vp), rad_cp, pt_onehot);
        return rc;
    }
    // Sum (ops/sum/mod.rs verify): output = SatClamp(acc), then SumAxisVerifier over the reduced axis (sum/axis.rs)
    int op_sum(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        size_t m, n; int axis;
        int rc = atlas_rt_sum_config(G.nodes.at(nd.inputs[0]).dims, nd.shape[0], m, n, axis);
        if (!rc) rc = R.point.empty() ? scalar_clamp(nd) : clamp_lookup(nd);
        if (rc) return rc;
        const Node* np = &nd;
        const Point r0 = R.point;
        const size_t nr = gr::log2u(axis == 0 ? m : n);
        VInst I;
        I.claim = advice_claim(nd, gr::VP_ClampAcc); I.rounds = nr; I.degree = 1;
        I.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point pt;
            if (axis == 0) { pt.assign(ch, ch + nr); pt.insert(pt.end(), r0.begin(), r0.end()); }
            else { pt = r0; pt.insert(pt.end(), ch, ch + nr); }
            int rc2 = append_nodeio(*np, 0, pt);
            if (!rc2) *expect = nodeio_claim(*np, 0);
            return rc2;
        };
        return run_single(gr::PT_SumReduction, I, "verify_graph: SumcheckVerificationError (sum over an axis)");
    }
    // ScalarConstDiv (ops/scalar_const_div.rs): eq(r0, r') (left - R) against q(r0) divisor
    int op_scalar_const_div(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        const Node* np = &nd;
        const Point r0 = R.point;
        VInst I;
        I.claim = H::mul(R.claim, fr_i64(nd.p[0])); I.rounds = r0.size(); I.degree = 2;
        I.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point pt(r0.size());
            for (size_t q = 0; q < pt.size(); q++) pt[q] = ch[pt.size() - 1 - q];
            int rc2 = append_nodeio(*np, 0, pt);
            if (!rc2) rc2 = append_dense(*np, gr::CP_ScalarConstDivNodeRemainder, pt);
            if (!rc2) *expect = H::mul(eq_mle(r0.data(), pt.data(), pt.size()), H::sub(nodeio_claim(*np, 0), dense_claim(*np, gr::CP_ScalarConstDivNodeRemainder)));
            return rc2;
        };
        return run_single(gr::PT_Execution, I, "verify_graph: SumcheckVerificationError (division by a constant)");
    }
    // Slice (ops/slice.rs): input(r') selector~(r'), the selector = eq(r0, o) at the input index of output cell o
    int op_slice(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        const Node& in = G.nodes.at(nd.inputs[0]);
        const size_t T_out = gr::padded_len(nd.dims), log_in = gr::log2u(gr::padded_len(in.dims));
        const Node* np = &nd; const Node* ip = &in;
        const Point r0 = R.point;
        VInst I;
        I.claim = R.claim; I.rounds = log_in; I.degree = 2;
        I.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point pt(log_in);
            for (size_t q = 0; q < log_in; q++) pt[q] = ch[log_in - 1 - q];
            int rc2 = append_nodeio(*np, 0, pt);
            if (rc2) return rc2;
            const std::vector<H::Fr> eo = eq_table(r0.data(), r0.size()), ei = eq_table(pt.data(), log_in);
            std::vector<size_t> stride(ip->dims.size());
            size_t st = 1;
            for (int a = (int)ip->dims.size() - 1; a >= 0; a--) { stride[a] = st; st *= ip->dims[a]; }
            const size_t base = (size_t)np->p[1] * stride[np->p[0]];
            H::Fr sel = H::zero();
            for (size_t o = 0; o < T_out; o++) {
                size_t rem = o, off = base;
                for (int d = (int)np->dims.size() - 1; d >= 0; d--) { off += (rem % np->dims[d]) * stride[d]; rem /= np->dims[d]; }
                sel = H::add(sel, H::mul(eo[o], ei[off]));
            }
            *expect = H::mul(nodeio_claim(*np, 0), sel);
            return (int)ATLAS_OK;
        };
        return run_single(gr::PT_Execution, I, "verify_graph: SumcheckVerificationError (slice)");
    }
    // Concat (ops/concat.rs:372-425): sum_t input_t(r'_lead) selector_t~(r')
    int op_concat(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        const size_t n_in = nd.inputs.size(), ax = (size_t)nd.p[0], r = nd.dims.size();
        size_t mx = 0;
        std::vector<size_t> nv(n_in);
        for (size_t k = 0; k < n_in; k++) { nv[k] = gr::log2u(gr::padded_len(G.nodes.at(nd.inputs[k]).dims)); mx = nv[k] > mx ? nv[k] : mx; }
        const Node* np = &nd;
        const Point r0 = R.point;
        VInst I;
        I.claim = R.claim; I.rounds = mx; I.degree = 2;
        I.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point pt(mx);
            for (size_t q = 0; q < mx; q++) pt[q] = ch[mx - 1 - q];
            for (size_t k = 0; k < n_in; k++) { int rc2 = append_nodeio(*np, k, Point(pt.begin(), pt.begin() + nv[k])); if (rc2) return rc2; }
            const std::vector<H::Fr> eo = eq_table(r0.data(), r0.size()), ei = eq_table(pt.data(), mx);
            std::vector<size_t> ostr(r);
            { size_t st = 1; for (int a = (int)r - 1; a >= 0; a--) { ostr[a] = st; st *= np->dims[a]; } }
            H::Fr acc = H::zero();
            size_t off = 0;
            for (size_t k = 0; k < n_in; k++) {
                const Node& in = G.nodes.at(np->inputs[k]);
                const size_t T_in = gr::padded_len(in.dims), shift = mx - nv[k];
                H::Fr sel = H::zero();
                for (size_t c = 0; c < T_in; c++) {
                    size_t rem = c, o = off * ostr[ax];
                    for (int d = (int)r - 1; d >= 0; d--) { o += (rem % in.dims[d]) * ostr[d]; rem /= in.dims[d]; }
                    sel = H::add(sel, H::mul(eo[o], ei[c << shift]));
                }
                acc = H::add(acc, H::mul(nodeio_claim(*np, k), sel));
                off += in.dims[ax];
            }
            *expect = acc;
            return (int)ATLAS_OK;
        };
        return run_single(gr::PT_Execution, I, "verify_graph: SumcheckVerificationError (concat)");
    }
    // Div (ops/div.rs verify_with_reduction, ReductionFlow::Custom)
    int op_div(const Node& nd) {
        const size_t log_T = gr::log2u(gr::padded_len(nd.dims));
        const Point r = challenge_point(log_T);                              // DivParams::new
        const Node* np = &nd;
        Point pt_out;
        Point* ptp = &pt_out;
        VInst I;
        I.claim = H::zero(); I.rounds = log_T; I.degree = 3;
        I.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point pt(log_T);
            for (size_t q = 0; q < log_T; q++) pt[q] = ch[log_T - 1 - q];
            int rc2 = append_nodeio(*np, 0, pt);
            if (!rc2) rc2 = append_nodeio(*np, 1, pt);
            if (!rc2) rc2 = append_current(*np, pt);
            if (!rc2) rc2 = append_advice(*np, gr::VP_DivRemainder, pt);
            if (rc2) return rc2;
            *expect = H::mul(eq_mle(r.data(), pt.data(), log_T), H::sub(H::add(H::mul(nodeio_claim(*np, 1), current_claim(*np)), advice_claim(*np, gr::VP_DivRemainder)), nodeio_claim(*np, 0)));
            *ptp = pt;
            return (int)ATLAS_OK;
        };
        int rc = run_single(gr::PT_Execution, I, "verify_graph: SumcheckVerificationError (division)");
        if (!rc) rc = eval_reduction(nd);
        if (rc) return rc;
        const gr::Opening& R = reduced.at(nd.idx);
        rc = append_dense(nd, gr::CP_DivNodeQuotient, R.point);
        if (rc) return rc;
        if (!same(dense_claim(nd, gr::CP_DivNodeQuotient), R.claim)) return bad("verify_graph: InvalidOpeningProof (Div quotient claim does not match the reduced node-output claim)");
        if (log_T == 0) return ATLAS_OK;
        return range_and_onehot(nd, pt_out, advice_claim(nd, gr::VP_DivRemainder), nodeio_claim(nd, 1), gr::VP_DivRangeCheckRa, gr::CP_DivRangeCheckRaD, gr::PT_RaOneHotChecks);
    }
    // MeanOfSquares (ops/mean_of_squares.rs): fused-rescale pre, eq(r0, retained) x^2 over the input's hypercube, the remainder's range check
    int op_mean_of_squares(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        const size_t log_T = R.point.size(), log_red = gr::log2u(gr::padded_len(G.nodes.at(nd.inputs[0]).dims)) - log_T;
        const int64_t D = ((int64_t)1 << nd.p[0]) * (int64_t)nd.p[1];
        int rc = append_advice(nd, gr::VP_RescaleRemainder, R.point);
        if (!rc) rc = log_T ? clamp_lookup(nd) : scalar_clamp(nd);
        if (rc) return rc;
        const H::Fr eval_R = advice_claim(nd, gr::VP_RescaleRemainder);
        const Node* np = &nd;
        const Point r0 = R.point;
        VInst I;
        I.claim = H::add(H::mul(advice_claim(nd, gr::VP_ClampAcc), fr_i64(D)), eval_R); I.rounds = log_T + log_red; I.degree = 3;
        I.finish = [=](const H::Fr* ch, H::Fr* expect) {
            const Point pt(ch, ch + log_T + log_red);
            int rc2 = append_nodeio(*np, 0, pt);
            if (rc2) return rc2;
            const H::Fr x = nodeio_claim(*np, 0);
            *expect = H::mul(eq_mle(r0.data(), ch, log_T), H::mul(x, x));
            return (int)ATLAS_OK;
        };
        rc = run_single(gr::PT_RescaleArith, I, "verify_graph: SumcheckVerificationError (mean of squares)");
        if (rc || log_T == 0) return rc;
        return range_and_onehot(nd, R.point, eval_R, fr_i64(D), gr::VP_MeanOfSquaresRangeCheckRa, gr::CP_MeanOfSquaresRangeCheckRaD, gr::PT_RescaleRemainderRaChecks);
    }
    // Rsqrt (ops/rsqrt.rs, ReductionFlow::Custom): x q + r_d = S^3 and out^2 + r_s = q at a fresh point, then both range checks batched
    int op_rsqrt(const Node& nd) {
        const size_t log_T = gr::log2u(gr::padded_len(nd.dims));
        const Point r = challenge_point(log_T);
        const H::Fr gamma = H::tr_challenge_scalar(Tr);
        const H::Fr s_cubed = pow2_fr(3 * (size_t)nd.p[0]);
        const Node* np = &nd;
        Point pt_out;
        Point* ptp = &pt_out;
        VInst I;
        I.claim = H::zero(); I.rounds = log_T; I.degree = 3;
        I.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point pt(log_T);
            for (size_t q = 0; q < log_T; q++) pt[q] = ch[log_T - 1 - q];
            int rc2 = append_nodeio(*np, 0, pt);
            if (!rc2) rc2 = append_dense(*np, gr::CP_RsqrtQuotient, pt);
            if (!rc2) rc2 = append_current(*np, pt);
            if (!rc2) rc2 = append_advice(*np, gr::VP_DivRemainder, pt);
            if (!rc2) rc2 = append_advice(*np, gr::VP_SqrtRemainder, pt);
            if (rc2) return rc2;
            const H::Fr x = nodeio_claim(*np, 0), q = dense_claim(*np, gr::CP_RsqrtQuotient), o = current_claim(*np), rd = advice_claim(*np, gr::VP_DivRemainder), rs_ = advice_claim(*np, gr::VP_SqrtRemainder);
            const H::Fr a = H::sub(H::add(H::mul(x, q), rd), s_cubed), b = H::sub(H::add(H::mul(o, o), rs_), q);
            *expect = H::mul(eq_mle(r.data(), pt.data(), log_T), H::add(a, H::mul(gamma, b)));
            *ptp = pt;
            return (int)ATLAS_OK;
        };
        int rc = run_single(gr::PT_Execution, I, "verify_graph: SumcheckVerificationError (rsqrt)");
        if (!rc) rc = eval_reduction(nd);
        if (rc) return rc;                                                    // (one element: no is_scalar branch in ops/rsqrt.rs — the range checks run over one cycle)
        // verify_range_and_onehot (rsqrt.rs): r_d < x and r_s < 2 out + 1
        const H::Fr o = current_claim(nd);
        Point ra0, ra1;
        std::vector<VInst> B;
        B.push_back(inst_range_check(nd, pt_out, advice_claim(nd, gr::VP_DivRemainder), nodeio_claim(nd, 0), gr::VP_DivRangeCheckRa, &ra0));
        B.push_back(inst_range_check(nd, pt_out, advice_claim(nd, gr::VP_SqrtRemainder), H::add(H::add(o, o), H::one()), gr::VP_SqrtRangeCheckRa, &ra1));
        rc = batch(gr::PT_RangeCheck, B);
        if (rc) return rc;
        std::vector<VInst> O;
        onehot_insts(nd, 64, pt_out, ra0, advice_claim(nd, gr::VP_DivRangeCheckRa), gr::CP_SqrtDivRangeCheckRaD, O);
        onehot_insts(nd, 64, pt_out, ra1, advice_claim(nd, gr::VP_SqrtRangeCheckRa), gr::CP_SqrtRangeCheckRaD, O);
        return batch(gr::PT_RaOneHotChecks, O);
    }
    // Tanh (ops/tanh.rs -> activation_clamped/mod.rs verify_clamped_activation): the small-table lookup of the clamped input, the clamp
    // lookup tying `clamped` to the raw input, both families' one-hot checks in one batch
    int op_tanh(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        const size_t LK = gr::ACTIVATION_TABLE_VARS, log_T = R.point.size();
        const H::Fr gamma = H::tr_challenge_scalar(Tr);                      // SmallTableParams::new
        int rc = append_advice(nd, gr::VP_ActivationClampedOutput, R.point);
        if (rc) return rc;
        const std::vector<int32_t>* table = nullptr;
        rc = atlas_rt_activation_table(nd.op, nullptr, &table);
        if (rc) return rc;
        const H::Fr clamped = advice_claim(nd, gr::VP_ActivationClampedOutput);
        const Node* np = &nd;
        const Point r0 = R.point;
        Point small_pt; Point* spp = &small_pt;
        VInst I;
        I.claim = H::add(R.claim, H::mul(gamma, clamped)); I.rounds = LK; I.degree = 2;
        I.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point ra(LK);
            for (size_t q = 0; q < LK; q++) ra[q] = ch[LK - 1 - q];
            Point pt = ra;
            pt.insert(pt.end(), r0.begin(), r0.end());
            int rc2 = append_advice(*np, gr::VP_ActivationSmallRa, pt);
            if (rc2) return rc2;
            H::Fr tv;
            rc2 = eval_public(table->data(), table->size(), ra, &tv);
            if (rc2) return rc2;
            *expect = H::mul(advice_claim(*np, gr::VP_ActivationSmallRa), H::add(tv, H::mul(gamma, signed_identity_mle(ra.data(), LK))));
            *spp = pt;
            return (int)ATLAS_OK;
        };
        rc = run_single(gr::PT_Execution, I, "verify_graph: SumcheckVerificationError (activation table)");
        if (rc) return rc;
        Point clamp_pt;
        rc = ps_unary(nd, T_CLAMP_SYM, 32, gr::ACTIVATION_BOUND, gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.inputs[0]), nd.idx), clamped, R.point, gr::VP_ActivationClampRa, gr::PT_NeuralTeleport, &clamp_pt);
        if (rc) return rc;
        (void)log_T;
        std::vector<VInst> O;
        onehot_insts(nd, LK, R.point, small_pt, advice_claim(nd, gr::VP_ActivationSmallRa), gr::CP_ActivationSmallRaD, O);
        onehot_insts(nd, 32, R.point, clamp_pt, advice_claim(nd, gr::VP_ActivationClampRa), gr::CP_ActivationClampRaD, O);
        return batch(gr::PT_RaOneHotChecks, O);
    }
    // Sin / Cos (ops/sin.rs:188-285 verify + verify_with_reduction; neural_teleport/division.rs, trig_downscale.rs, range_and_onehot.rs:142-181)
    int op_trig(const Node& nd) {
        const size_t log_T = gr::log2u(gr::padded_len(nd.dims)), LK = gr::TRIG_TABLE_VARS;
        const uint8_t vp_ra = nd.op == ATLAS_OP_SIN ? gr::VP_SinRa : gr::VP_CosRa, cp_rad = nd.op == ATLAS_OP_SIN ? gr::CP_SinRaD : gr::CP_CosRaD;
        const H::Fr tau = fr_i64(gr::TRIG_PERIOD_MODULUS);
        const Node* np = &nd;
        // 1a: TeleportDivisionVerifier::new_from_transcript
        const Point r = challenge_point(log_T);
        Point pt_out; Point* ptp = &pt_out;
        VInst D;
        D.claim = H::zero(); D.rounds = log_T; D.degree = 2;
        D.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point pt(log_T);
            for (size_t q = 0; q < log_T; q++) pt[q] = ch[log_T - 1 - q];
            int rc2 = append_nodeio(*np, 0, pt);
            if (!rc2) rc2 = append_advice(*np, gr::VP_TeleportQuotient, pt);
            if (!rc2) rc2 = append_advice(*np, gr::VP_TeleportRemainder, pt);
            if (rc2) return rc2;
            *expect = H::mul(eq_mle(r.data(), pt.data(), log_T), H::sub(H::add(H::mul(tau, advice_claim(*np, gr::VP_TeleportQuotient)), advice_claim(*np, gr::VP_TeleportRemainder)), nodeio_claim(*np, 0)));
            *ptp = pt;
            return (int)ATLAS_OK;
        };
        int rc = run_single(gr::PT_NeuralTeleport, D, "verify_graph: SumcheckVerificationError (teleport division)");
        if (rc) return rc;
        const Point pt = pt_out;
        // 1b: cache_downscaled_verify, the right-shift read-raf (its witness claim = the remainder, appended again), the table read-raf
        rc = append_advice(nd, gr::VP_TrigDownscaled, pt);
        if (rc) return rc;
        const H::Fr down = advice_claim(nd, gr::VP_TrigDownscaled);
        Point Dra, tab_pt; Point* tpp = &tab_pt;
        std::vector<VInst> B(1);
        rc = inst_ps_unary(nd, T_RSHIFT, 32, gr::TRIG_DOWNSCALE_BITS, gr::node_exec(gr::virt(gr::VP_TeleportRemainder, nd.idx), nd.idx), down, pt, gr::VP_TrigDownscaleRa, &Dra, &B[0]);
        if (rc) return rc;
        const H::Fr g_s = H::tr_challenge_scalar(Tr);                        // SinParams::new
        rc = append_current(nd, pt);                                         // SinVerifier::new: Target::Current
        if (rc) return rc;
        const std::vector<int32_t>* table = nullptr;
        rc = atlas_rt_trig_table(nd.op, nullptr, &table);
        if (rc) return rc;
        VInst S;
        S.claim = H::add(current_claim(nd), H::mul(g_s, down)); S.rounds = LK; S.degree = 2;
        S.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point ra(LK);
            for (size_t q = 0; q < LK; q++) ra[q] = ch[LK - 1 - q];
            Point p = ra;
            p.insert(p.end(), pt.begin(), pt.end());
            int rc2 = append_advice(*np, vp_ra, p);
            if (rc2) return rc2;
            H::Fr tv;
            rc2 = eval_public(table->data(), table->size(), ra, &tv);
            if (rc2) return rc2;
            *expect = H::mul(advice_claim(*np, vp_ra), H::add(tv, H::mul(g_s, identity_mle(ra.data(), LK))));
            *tpp = p;
            return (int)ATLAS_OK;
        };
        B.push_back(std::move(S));
        rc = batch(gr::PT_Execution, B);
        // 1c: the downscale lookup's one-hot checks
        if (!rc) rc = onehot_checks(nd, 32, pt, Dra, advice_claim(nd, gr::VP_TrigDownscaleRa), gr::CP_TrigDownscaleRaD, gr::PT_TrigDownscaleRaChecks);
        if (rc) return rc;
        // verify_with_reduction: the committed quotient at the quotient's point, equal claims; the eval reduction
        rc = append_dense(nd, gr::CP_TeleportNodeQuotient, pt);
        if (rc) return rc;
        if (!same(dense_claim(nd, gr::CP_TeleportNodeQuotient), advice_claim(nd, gr::VP_TeleportQuotient))) return bad("verify_graph: InvalidOpeningProof (teleport quotient claim does not match the committed quotient claim)");
        rc = eval_reduction(nd);
        if (rc) return rc;
        // verify_range_and_onehot: [range check R < tau, the table lookup's one-hot triple], then the range check's own triple
        Point Rra;
        std::vector<VInst> C;
        C.push_back(inst_range_check(nd, pt, advice_claim(nd, gr::VP_TeleportRemainder), tau, gr::VP_TeleportRangeCheckRa, &Rra));
        onehot_insts(nd, LK, pt, tab_pt, advice_claim(nd, vp_ra), cp_rad, C);
        rc = batch(gr::PT_RaOneHotChecks, C);
        if (!rc) rc = onehot_checks(nd, 64, pt, Rra, advice_claim(nd, gr::VP_TeleportRangeCheckRa), gr::CP_TeleportRangeCheckRaD, gr::PT_RaHammingWeight);
        return rc;
    }
    // GatherLarge (ops/gather/mod.rs + large.rs): sum_k ra(k) (dict_r(k) + gamma k), then the one-hot checks over the index cycle
    int op_gather(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        const Node& dict = G.nodes.at(nd.inputs[0]);
        const Node& idxn = G.nodes.at(nd.inputs[1]);
        const size_t V = dict.dims[0], lv = gr::log2u(V), ln = gr::log2u(gr::padded_len(idxn.dims));
        const H::Fr gamma = H::tr_challenge_scalar(Tr);                      // GatherParams::new
        const Point r_index(R.point.begin(), R.point.begin() + ln), r_word(R.point.begin() + ln, R.point.end());
        int rc = append_nodeio(nd, 1, r_index);
        if (rc) return rc;
        const Node* np = &nd;
        Point ra_pt; Point* rpp = &ra_pt;
        VInst I;
        I.claim = H::add(R.claim, H::mul(gamma, nodeio_claim(nd, 1))); I.rounds = lv; I.degree = 2;
        I.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point ra(lv);
            for (size_t q = 0; q < lv; q++) ra[q] = ch[lv - 1 - q];
            Point p_ra = ra, p_dict = ra;
            p_ra.insert(p_ra.end(), r_index.begin(), r_index.end());
            p_dict.insert(p_dict.end(), r_word.begin(), r_word.end());
            int rc2 = append_advice(*np, gr::VP_NodeOutputRa, p_ra);
            if (!rc2) rc2 = append_nodeio(*np, 0, p_dict);
            if (rc2) return rc2;
            *expect = H::mul(advice_claim(*np, gr::VP_NodeOutputRa), H::add(nodeio_claim(*np, 0), H::mul(gamma, identity_mle(ra.data(), lv))));
            *rpp = p_ra;
            return (int)ATLAS_OK;
        };
        rc = run_single(gr::PT_Execution, I, "verify_graph: SumcheckVerificationError (gather)");
        if (rc) return rc;
        if (nd.op == ATLAS_OP_GATHER_LARGE) return onehot_checks(nd, lv, r_index, ra_pt, advice_claim(nd, gr::VP_NodeOutputRa), gr::CP_GatherRaD, gr::PT_RaOneHotChecks);
        // GatherSmall (ops/gather/small.rs:64-110, 170-255, 314-377): [HammingBooleanity, Booleanity (d = 1, log_k_chunk = log dict_len)], then HammingWeight
        const Point r_address = challenge_point(lv);                         // ra_booleanity_params
        const H::Fr gamma_b = H::challenge_to_fr(1, 0, mode());              // F::Challenge::from(1)
        std::vector<VInst> B(2);
        B[0].claim = H::zero(); B[0].rounds = ln; B[0].degree = 3;           // hamming_booleanity.rs:175-192
        B[0].finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point pt(ln);
            for (size_t q = 0; q < ln; q++) pt[q] = ch[ln - 1 - q];
            const OpeningId id = gr::oid(gr::virt(gr::VP_HammingWeight), gr::SC_RamHammingBooleanity);
            int rc2 = append_virtual(id, pt);
            if (rc2) return rc2;
            const H::Fr hw = claim_of(id);
            *expect = H::mul(eq_mle(r_index.data(), pt.data(), ln), H::mul(hw, H::sub(hw, H::one())));
            return (int)ATLAS_OK;
        };
        B[1].claim = H::zero(); B[1].rounds = lv + ln; B[1].degree = 3;      // booleanity.rs:391-419 with d = 1
        B[1].finish = [=](const H::Fr* ch, H::Fr* expect) {
            const size_t mr = lv + ln;
            Point pt(mr), comb(mr);
            for (size_t q = 0; q < lv; q++) { pt[q] = ch[lv - 1 - q]; comb[q] = r_address[lv - 1 - q]; }
            for (size_t q = 0; q < ln; q++) { pt[lv + q] = ch[mr - 1 - q]; comb[lv + q] = r_index[ln - 1 - q]; }
            int rc2 = append_sparse(gr::CP_GatherRa, *np, 0, gr::SC_Booleanity, pt);
            if (rc2) return rc2;
            const H::Fr c = claim_of(gr::oid(gr::comm(gr::CP_GatherRa, np->idx), gr::SC_Booleanity));
            *expect = H::mul(eq_mle(ch, comb.data(), mr), H::mul(gamma_b, H::sub(H::mul(c, c), c)));
            return (int)ATLAS_OK;
        };
        rc = batch(gr::PT_RaOneHotChecks, B);
        if (rc) return rc;
        VInst W;
        W.claim = H::one(); W.rounds = lv; W.degree = 1;                     // hamming_weight.rs:193-207 with d = 1, gamma_powers = [1]
        W.finish = [=](const H::Fr* ch, H::Fr* expect) {
            Point pt(lv);
            for (size_t q = 0; q < lv; q++) pt[q] = ch[lv - 1 - q];
            pt.insert(pt.end(), r_index.begin(), r_index.end());
            int rc2 = append_sparse(gr::CP_GatherRa, *np, 0, gr::SC_HammingWeight, pt);
            if (!rc2) *expect = claim_of(gr::oid(gr::comm(gr::CP_GatherRa, np->idx), gr::SC_HammingWeight));
            return rc2;
        };
        return run_single(gr::PT_RaHammingWeight, W, "verify_graph: SumcheckVerificationError (gather hamming weight)");
    }
    // SoftmaxLastAxis (ops/softmax_last_axis/mod.rs:286-333, 742-1135): the auxiliary vectors, four BatchedSumcheck stages, the operand link
    int op_softmax(const Node& nd) {
        const gr::Opening& R = reduced.at(nd.idx);
        const size_t LS = gr::MODEL_SCALE, N = nd.dims.back(), T = gr::padded_len(nd.dims), F = T / gr::next_pow2(N), lf = gr::log2u(F), log_T = gr::log2u(T);
        const ExpLut* L = nullptr;
        int rc = atlas_rt_exp_lut(&L);
        if (rc) return rc;
        const size_t lk_hi = gr::log2u(L->hi.size()), lk_lo = gr::log2u(L->lo.size());
        const Node* np = &nd;
        // SoftmaxLastAxisVerifier::new: exp_sum_q[k], max_k[k], argmax_k[k] from the proof, as u64 -> i32
        std::vector<int32_t> exp_sum(F), max_k(F), inv_sum(F); std::vector<size_t> argmax(F);
        {
            const Point empty;
            const uint8_t vps[3] = {gr::VP_SoftmaxSumOutput, gr::VP_SoftmaxMaxOutput, gr::VP_SoftmaxMaxIndex};
            for (size_t k = 0; k < F; k++)
                for (int q = 0; q < 3; q++) {
                    const OpeningId id = gr::node_exec(gr::virt(vps[q], nd.idx, k), nd.idx);
                    rc = append_virtual(id, empty);
                    if (rc) return rc;
                    uint64_t c[4];
                    H::to_canonical(claim_of(id), c);
                    if (c[1] || c[2] || c[3]) return bad("verify_graph: an auxiliary softmax scalar does not fit 64 bits");
                    if (q == 0) exp_sum[k] = (int32_t)(uint32_t)c[0]; else if (q == 1) max_k[k] = (int32_t)(uint32_t)c[0]; else argmax[k] = (size_t)c[0];
                    if (q == 0) {                                           // inv_sum_evals: floor(S^2 / exp_sum_q[k])
                        if ((int64_t)c[0] == 0) return bad("verify_graph: InvalidOpeningProof (exp_sum_q is zero)");
                        inv_sum[k] = (int32_t)((((int64_t)1 << LS) * ((int64_t)1 << LS)) / (int64_t)c[0]);
                    }
                }
        }
        const Point r0 = R.point, r_lead(r0.begin(), r0.begin() + lf);
        rc = append_advice(nd, gr::VP_SoftmaxExpSum, r_lead);                // cache_exp_sum
        if (rc) return rc;
        if (!same(mle_i32(exp_sum, r_lead.data(), lf), advice_claim(nd, gr::VP_SoftmaxExpSum))) return bad("verify_graph: InvalidOpeningProof (exp_sum evaluation mismatch)");
        rc = append_advice(nd, gr::VP_SoftmaxRecipMultRemainder, r0);        // cache_R
        if (rc) return rc;
        const H::Fr S_fr = pow2_fr(LS), R_claim = advice_claim(nd, gr::VP_SoftmaxRecipMultRemainder);
        auto rev = [](const H::Fr* ch, size_t n) { Point p(n); for (size_t q = 0; q < n; q++) p[q] = ch[n - 1 - q]; return p; };
        Point r1, r2, Rra, Era, hi_pt, lo_pt, Cra;
        Point *r1p = &r1, *r2p = &r2;
        {   // ---- stage 1: RecipMult, ExpSum, IdentityRC of R
            std::vector<VInst> B(2);
            B[0].claim = H::add(H::mul(R.claim, S_fr), R_claim); B[0].rounds = log_T; B[0].degree = 3;
            B[0].finish = [=](const H::Fr* ch, H::Fr* expect) {
                const Point p = rev(ch, log_T);
                int rc2 = append_advice(*np, gr::VP_SoftmaxExpQ, p);
                if (rc2) return rc2;
                *expect = H::mul(eq_mle(r0.data(), p.data(), log_T), H::mul(advice_claim(*np, gr::VP_SoftmaxExpQ), mle_i32(inv_sum, p.data(), lf)));
                *r1p = p;
                return (int)ATLAS_OK;
            };
            B[1].claim = advice_claim(nd, gr::VP_SoftmaxExpSum); B[1].rounds = log_T; B[1].degree = 2;
            B[1].finish = [=](const H::Fr* ch, H::Fr* expect) {
                const Point p = rev(ch, log_T);
                int rc2 = append_advice(*np, gr::VP_SoftmaxExpQ, p);
                if (rc2) return rc2;
                *expect = H::mul(eq_mle(r0.data(), p.data(), lf), advice_claim(*np, gr::VP_SoftmaxExpQ));
                return (int)ATLAS_OK;
            };
            B.push_back(inst_identity_rc(nd, LS, R_claim, r0, gr::VP_SoftmaxRemainderRa, &Rra));
            rc = batch(gr::PT_SoftmaxStage1, B);
            if (rc) return rc;
        }
        {   // ---- stage 2: Mult, MaxIndicator, IdentityRC of r_exp, the one-hot checks of R
            rc = append_advice(nd, gr::VP_SoftmaxExpRemainder, r1);          // cache_r_exp
            if (rc) return rc;
            const H::Fr r_exp_claim = advice_claim(nd, gr::VP_SoftmaxExpRemainder);
            const Point r1c = r1;
            std::vector<VInst> B(2);
            B[0].claim = H::add(H::mul(advice_claim(nd, gr::VP_SoftmaxExpQ), S_fr), r_exp_claim); B[0].rounds = log_T; B[0].degree = 3;
            B[0].finish = [=](const H::Fr* ch, H::Fr* expect) {
                const Point p = rev(ch, log_T);
                int rc2 = append_advice(*np, gr::VP_SoftmaxExpHi, p);
                if (!rc2) rc2 = append_advice(*np, gr::VP_SoftmaxExpLo, p);
                if (rc2) return rc2;
                *expect = H::mul(eq_mle(r1c.data(), p.data(), log_T), H::mul(advice_claim(*np, gr::VP_SoftmaxExpHi), advice_claim(*np, gr::VP_SoftmaxExpLo)));
                *r2p = p;
                return (int)ATLAS_OK;
            };
            B[1].claim = mle_i32(max_k, r1.data(), lf); B[1].rounds = log_T; B[1].degree = 3;
            B[1].finish = [=](const H::Fr* ch, H::Fr* expect) {
                const Point p = rev(ch, log_T);
                int rc2 = append_nodeio(*np, 0, p);
                if (rc2) return rc2;
                const std::vector<H::Fr> ek = eq_table(p.data(), lf);       // e~(r_k, r_j) = sum_k eq(r_k, k) eq(r_j, bits(argmax_k))
                const size_t ln = log_T - lf;
                H::Fr e_claim = H::zero();
                for (size_t k = 0; k < F; k++) {
                    H::Fr y = ek[k];
                    for (size_t q = 0; q < ln; q++) { const bool bit = (argmax[k] >> (ln - 1 - q)) & 1; y = H::mul(y, bit ? p[lf + q] : H::sub(H::one(), p[lf + q])); }
                    e_claim = H::add(e_claim, y);
                }
                *expect = H::mul(H::mul(eq_mle(r1c.data(), p.data(), lf), e_claim), nodeio_claim(*np, 0));
                return (int)ATLAS_OK;
            };
            B.push_back(inst_identity_rc(nd, LS, r_exp_claim, r1, gr::VP_SoftmaxExpRemainderRa, &Era));
            onehot_insts(nd, LS, r0, Rra, advice_claim(nd, gr::VP_SoftmaxRemainderRa), gr::CP_SoftmaxRemainderRaD, B);
            rc = batch(gr::PT_SoftmaxStage2, B);
            if (rc) return rc;
        }
        {   // ---- stage 3: the two exp-digit Shout lookups, the significance clamp, the one-hot checks of r_exp
            rc = append_advice(nd, gr::VP_SoftmaxZHi, r2);                   // cache_z_hi_lo
            if (!rc) rc = append_advice(nd, gr::VP_SoftmaxZLo, r2);
            if (rc) return rc;
            const H::Fr zhi = advice_claim(nd, gr::VP_SoftmaxZHi), zlo = advice_claim(nd, gr::VP_SoftmaxZLo);
            std::vector<VInst> B;
            B.push_back(inst_shout(nd, L->hi, lk_hi, advice_claim(nd, gr::VP_SoftmaxExpHi), zhi, r2, gr::VP_SoftmaxZHiRa, &hi_pt));
            B.push_back(inst_shout(nd, L->lo, lk_lo, advice_claim(nd, gr::VP_SoftmaxExpLo), zlo, r2, gr::VP_SoftmaxZLoRa, &lo_pt));
            VInst C;
            const H::Fr rv = H::add(H::mul(zhi, pow2_fr(L->log2_base)), zlo);                                 // significance_clamp.rs:61-69
            rc = inst_ps_unary(nd, T_CLAMP, 32, lk_hi + L->log2_base, gr::node_exec(gr::virt(gr::VP_SoftmaxClampWitness, nd.idx), nd.idx), rv, r2, gr::VP_SoftmaxClampRa, &Cra, &C);
            if (rc) return rc;
            B.push_back(std::move(C));
            onehot_insts(nd, LS, r1, Era, advice_claim(nd, gr::VP_SoftmaxExpRemainderRa), gr::CP_SoftmaxExpRemainderRaD, B);
            rc = batch(gr::PT_SoftmaxStage3, B);
            if (rc) return rc;
        }
        // operand_link: X(r2) = max_k(r2_lead) - z(r2)
        if (!same(nodeio_claim(nd, 0), H::sub(mle_i32(max_k, r2.data(), lf), advice_claim(nd, gr::VP_SoftmaxClampWitness))))
            return bad("verify_graph: InvalidOpeningProof (operand link: X(r2) does not match max_k - z)");
        std::vector<VInst> B;                                                // ---- stage 4
        onehot_insts(nd, lk_hi, r2, hi_pt, advice_claim(nd, gr::VP_SoftmaxZHiRa), gr::CP_SoftmaxZHiRaD, B);
        onehot_insts(nd, lk_lo, r2, lo_pt, advice_claim(nd, gr::VP_SoftmaxZLoRa), gr::CP_SoftmaxZLoRaD, B);
        onehot_insts(nd, 32, r2, Cra, advice_claim(nd, gr::VP_SoftmaxClampRa), gr::CP_SoftmaxClampRaD, B);
        return batch(gr::PT_SoftmaxStage4, B);
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
        if (nd.op == ATLAS_OP_SOFTMAX && gr::padded_len(nd.dims) == 1)
            return fail(ATLAS_EINVAL, "verify_graph: SoftmaxLastAxis over ONE element is not composed");
        if (nd.op == ATLAS_OP_DIV) return op_div(nd);                          // ReductionFlow::Custom
        if (nd.op == ATLAS_OP_RSQRT) return op_rsqrt(nd);
        if (nd.op == ATLAS_OP_SIN || nd.op == ATLAS_OP_COS) return op_trig(nd);
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
            case ATLAS_OP_RELU: case ATLAS_OP_CLAMP: return op_relu(nd);
            case ATLAS_OP_NEG: case ATLAS_OP_IS_NAN: {                        // ops/neg.rs, is_nan.rs: no sumcheck
                rc = append_nodeio(nd, 0, R.point);
                if (rc) return rc;
                if (nd.op == ATLAS_OP_NEG) return same(H::neg(nodeio_claim(nd, 0)), R.claim) ? ATLAS_OK : bad("verify_graph: InvalidOpeningProof (-operand(r) must equal output(r))");
                return same(R.claim, H::zero()) ? ATLAS_OK : bad("verify_graph: InvalidOpeningProof (isNan claim should be zero)");
            }
            case ATLAS_OP_RESHAPE: return op_reshape(nd);
            case ATLAS_OP_SUM: return op_sum(nd);
            case ATLAS_OP_SCALAR_CONST_DIV: return op_scalar_const_div(nd);
            case ATLAS_OP_SLICE: return op_slice(nd);
            case ATLAS_OP_CONCAT: return op_concat(nd);
            case ATLAS_OP_MEAN_OF_SQUARES: return op_mean_of_squares(nd);
            case ATLAS_OP_TANH: case ATLAS_OP_ERF: case ATLAS_OP_SIGMOID: return op_tanh(nd);      // verify_clamped_activation<Table>
            case ATLAS_OP_GATHER_LARGE: case ATLAS_OP_GATHER_SMALL: return op_gather(nd);
            case ATLAS_OP_SOFTMAX: return op_softmax(nd);
            default: return fail(ATLAS_EINVAL, "verify_graph: operator without a verifier composition");
        }
    }

    // get_committed_polynomials of every node, for the operators the verifier composes
    int collect_committed(const std::vector<atlas_g1_affine_t>& coms) {
        for (auto& kv : G.nodes) {
            const Node& nd = kv.second;
            const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T);
            if (T == 1 && nd.op != ATLAS_OP_SCALAR_CONST_DIV && nd.op != ATLAS_OP_DIV && nd.op != ATLAS_OP_RELU && nd.op != ATLAS_OP_CLAMP && !atlas_rt_is_activation(nd.op) &&
                nd.op != ATLAS_OP_RSQRT && nd.op != ATLAS_OP_SIN && nd.op != ATLAS_OP_COS && nd.op != ATLAS_OP_GATHER_SMALL && nd.op != ATLAS_OP_GATHER_LARGE) continue;      // (the lookup operators proper and the gathers run their generic flows over one cycle)      // one element: ScalarConstDiv keeps its remainder, Div its quotient (ops/div.rs:157-160)
            auto chunks = [&](uint8_t cp, size_t log_K) { for (size_t i = 0; i < (log_K + 3) / 4; i++) committed[gr::comm(cp, nd.idx, i)].log_T = log_T; };
            auto dense = [&](uint8_t cp) { committed[gr::comm(cp, nd.idx)].log_T = log_T; };
            switch (nd.op) {
                case ATLAS_OP_ADD: case ATLAS_OP_SUB: chunks(gr::CP_ClampRaD, 64); break;
                case ATLAS_OP_EINSUM: case ATLAS_OP_MUL: case ATLAS_OP_SQUARE: case ATLAS_OP_CUBE:
                    chunks(gr::CP_RescaleRemainderRaD, nd.op == ATLAS_OP_EINSUM ? (size_t)nd.p[1] : nd.op == ATLAS_OP_CUBE ? 2 * (size_t)nd.p[0] : (size_t)nd.p[0]);
                    chunks(gr::CP_ClampRaD, 64);
                    break;
                case ATLAS_OP_RELU: chunks(gr::CP_NodeOutputRaD, 32); break;
                case ATLAS_OP_CLAMP: chunks(gr::CP_SymmetricClampRaD, 32); break;
                case ATLAS_OP_SUM: chunks(gr::CP_ClampRaD, 64); break;
                case ATLAS_OP_SCALAR_CONST_DIV: dense(gr::CP_ScalarConstDivNodeRemainder); break;
                case ATLAS_OP_DIV: dense(gr::CP_DivNodeQuotient); if (T > 1) chunks(gr::CP_DivRangeCheckRaD, 64); break;
                case ATLAS_OP_MEAN_OF_SQUARES: chunks(gr::CP_ClampRaD, 64); chunks(gr::CP_MeanOfSquaresRangeCheckRaD, 64); break;
                case ATLAS_OP_RSQRT: dense(gr::CP_RsqrtQuotient); chunks(gr::CP_SqrtDivRangeCheckRaD, 64); chunks(gr::CP_SqrtRangeCheckRaD, 64); break;
                case ATLAS_OP_TANH: case ATLAS_OP_ERF: case ATLAS_OP_SIGMOID: chunks(gr::CP_ActivationClampRaD, 32); chunks(gr::CP_ActivationSmallRaD, gr::ACTIVATION_TABLE_VARS); break;
                case ATLAS_OP_SOFTMAX: {
                    const ExpLut* L = nullptr;
                    int rc = atlas_rt_exp_lut(&L);
                    if (rc) return rc;
                    chunks(gr::CP_SoftmaxRemainderRaD, gr::MODEL_SCALE); chunks(gr::CP_SoftmaxExpRemainderRaD, gr::MODEL_SCALE); chunks(gr::CP_SoftmaxClampRaD, 32);
                    chunks(gr::CP_SoftmaxZHiRaD, gr::log2u(L->hi.size())); chunks(gr::CP_SoftmaxZLoRaD, gr::log2u(L->lo.size()));
                    break;
                }
                case ATLAS_OP_SIN: case ATLAS_OP_COS:
                    dense(gr::CP_TeleportNodeQuotient); chunks(gr::CP_TrigDownscaleRaD, 32); chunks(nd.op == ATLAS_OP_SIN ? gr::CP_SinRaD : gr::CP_CosRaD, gr::TRIG_TABLE_VARS);
                    chunks(gr::CP_TeleportRangeCheckRaD, 64);
                    break;
                case ATLAS_OP_GATHER_SMALL: committed[gr::comm(gr::CP_GatherRa, nd.idx)].log_T = gr::log2u(gr::padded_len(G.nodes.at(nd.inputs[1]).dims)); break;
                case ATLAS_OP_GATHER_LARGE: {
                    const size_t ln = gr::log2u(gr::padded_len(G.nodes.at(nd.inputs[1]).dims)), lv = gr::log2u(G.nodes.at(nd.inputs[0]).dims[0]);
                    for (size_t i = 0; i < (lv + 3) / 4; i++) committed[gr::comm(gr::CP_GatherRaD, nd.idx, i)].log_T = ln;
                    break;
                }
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
    // a verifier's graph is never traced: refuse here what the trace would refuse (operand shapes, axes, layouts, a scale or bound the
    // compiled-in tables are not for), before any operator formula reads nd.dims / nd.p / nd.shape
    if (int vrc = atlas_rt_validate_graph(*G)) return vrc;
    for (size_t i = 0; i < n_inputs; i++) if (!inputs[i]) return fail(ATLAS_EINVAL, "verify_graph: null input tensor");
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
    for (auto it = G->nodes.rbegin(); it != G->nodes.rend() && !rc; ++it) {
        rc = V.verify_node(it->second, dummy);
        if (rc && getenv("ATLAS_GRAPH_TRACE")) fprintf(stderr, "[atlas graph] verify: node %zu (operator %d) -> %d\n", it->second.idx, it->second.op, rc);
    }
    if (!rc) rc = V.reduced_openings();
    if (!rc && final_transcript) *final_transcript = V.t;
    return rc;
}
