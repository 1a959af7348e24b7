// ONNXProof::prove for a graph resident in the library (SURVEY §2 C1/C3/C4/C13, §8 x1 + B1):
//   Model::trace                         (graph_exec.hip, f3)
//   append_inputs_to_transcript          (onnx_proof/mod.rs:90-122)
//   commit_witness_polynomials           (prover.rs:71-87, witness.rs:136-200): BTreeMap<CommittedPoly> order
//   output_claim                         (prover.rs:89-121)
//   iop: nodes in reverse index order    (prover.rs:127-138): NodeEvalReduction::prove (ops/eval_reduction.rs:17-40) then the
//                                        operator's own composition (ops/*.rs)
//   prove_reduced_openings               (prover.rs:141-176)
//   finalize_proof + serialize           (prover.rs:178-205, proof_serialization.rs:200-224)
// Host glue over the library's instance provers; the O(T) work is in their kernels.  Every scalar the reference appends
// to the ProverOpeningAccumulator is appended here in the same order under the same OpeningId.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>

#include "graph_state.hip.h"
#include "internal.hpp"
#include "shard_group.hpp"
#include "host_sampler.hpp"

using gr::Node;
using gr::OpeningId;
using gr::Point;
using gr::PolyId;

int atlas_rt_einsum_strides(int layout, const std::vector<size_t>& d, std::vector<size_t>& out_dims, std::vector<size_t>& la, std::vector<size_t>& ra, size_t& K,
                            size_t& lsk, size_t& rsk);

int atlas_rt_sum_config(const std::vector<size_t>& idims, size_t axis, size_t& m, size_t& n, int& ax);

namespace {

__global__ __launch_bounds__(256) void k_i32_to_fr(const int32_t* __restrict__ in, Fr* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) fe_store(out + i, fr_from_i64((int64_t)in[i]));
}
// build_slice_selector (ops/slice.rs): selector[input index of output cell o] = eq(r_output, o), zero elsewhere
struct SliceMap { uint32_t n; uint32_t dim[6]; uint32_t stride[6]; };
__global__ __launch_bounds__(256) void k_slice_selector(const Fr* __restrict__ eq, SliceMap M, size_t base, size_t T_out, Fr* __restrict__ sel /* zeroed */) {
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < T_out; o += (size_t)gridDim.x * 256) {
        size_t rem = o, off = base;
        for (int d = (int)M.n - 1; d >= 0; d--) { off += (rem % M.dim[d]) * M.stride[d]; rem /= M.dim[d]; }
        sel[off] = eq[o];
    }
}
// Concat operands over the largest input's hypercube (ops/concat.rs:429-443, 445-487): an input of fewer variables repeated over the low
// ones, ext[i] = in[i >> shift]; its selector = eq(r_output, .) at the output index of input cell c, placed at c << shift (zero elsewhere)
__global__ __launch_bounds__(256) void k_concat_extend(const int32_t* __restrict__ in, uint32_t shift, size_t len, Fr* __restrict__ ext) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (size_t)gridDim.x * 256) fe_store(ext + i, fr_from_i64((int64_t)in[i >> shift]));
}
__global__ __launch_bounds__(256) void k_concat_selector(const Fr* __restrict__ eq, SliceMap M /* dim = operand dims, stride = output strides */, size_t base, uint32_t shift,
                                                         size_t T_in, Fr* __restrict__ sel /* zeroed */) {
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < T_in; c += (size_t)gridDim.x * 256) {
        size_t rem = c, off = base;
        for (int d = (int)M.n - 1; d >= 0; d--) { off += (rem % M.dim[d]) * M.stride[d]; rem /= M.dim[d]; }
        sel[c << shift] = eq[off];
    }
}
unsigned grid_of(size_t n) { size_t b = (n + 255) / 256; return (unsigned)(b > 4096 ? 4096 : b ? b : 1); }
H::Fr fr_from_i64_host(int64_t v) { return v >= 0 ? H::from_u64((uint64_t)v) : H::neg(H::from_u64((uint64_t)(-v))); }

double ms_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

struct Prover : FlowSink {
    atlas_graph& G;
    atlas_srs_t srs;
    atlas_transcript_t t;
    H::Transcript& Tr;
    std::map<OpeningId, gr::Opening> openings;            // ProverOpeningAccumulator::openings
    std::map<size_t, gr::Opening> reduced;                // ::reduced_evaluations
    std::map<PolyId, gr::Committed*> committed;           // polynomial_map (BTreeMap<CommittedPoly, _>)
    std::map<gr::ProofId, std::vector<uint8_t>> proofs;
    std::map<size_t, std::vector<H::Fr>> evalred;         // eval_reduction_proofs: h coefficients
    uint64_t cur = 0;
    atlas_shard_group* sh = nullptr;                      // the ranks of a sharded whole proof (atlas_prove_graph_sharded); NULL: one GPU

    Prover(atlas_graph& g_, atlas_srs_t s) : G(g_), srs(s), Tr(*reinterpret_cast<H::Transcript*>(&t)) {}

    // ---- FlowSink
    void virt(const OpeningId& id, const Point& point, const H::Fr& claim) override { openings[id] = gr::Opening{point, claim}; }
    void sparse(uint8_t cp_var, size_t chunk, uint8_t sc, const Point& point, const H::Fr& claim) override {
        const PolyId p = gr::comm(cp_var, cur, chunk);
        openings[gr::oid(p, sc)] = gr::Opening{point, claim};
        auto it = committed.find(p);
        if (it != committed.end()) { it->second->opened = true; it->second->point = point; it->second->claim = claim; }    // sumchecks.insert: the last append wins
    }
    void proof(uint8_t proof_type, const uint8_t* bytes, size_t len) override { proofs[gr::ProofId{cur, proof_type}].assign(bytes, bytes + len); }

    Out out() { Out O{nullptr, 0, 0, nullptr, 0, nullptr, 0, 0}; O.sink = this; O.node = cur; return O; }
    // ATLAS_GRAPH_TRACE=2: wall clock between the marks of an operator flow on stderr (synchronises the device: diagnosis only)
    std::chrono::steady_clock::time_point mark_t;
    void mark(const char* what) override {
        static const bool on = getenv("ATLAS_GRAPH_TRACE") && atoi(getenv("ATLAS_GRAPH_TRACE")) >= 2;
        if (!on) return;
        atlas_sync();
        const auto n = std::chrono::steady_clock::now();
        if (what) fprintf(stderr, "[atlas graph]   node %llu op %d | %s | %8.3f ms\n", (unsigned long long)cur, G.nodes.at(cur).op, what, ms_between(mark_t, n));
        mark_t = n;
    }
    // AccOpeningProvider::append_nodeio(Target::Input(pos), claim) at `point` (utils/opening_access.rs)
    int append_nodeio(const Node& nd, size_t pos, const Point& point, const H::Fr& claim) {
        Out O = out();
        return O.append_virtual(Tr, gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.inputs[pos]), nd.idx), point, claim);
    }
    // append_advice(CommittedPoly::X) -> ProverOpeningAccumulator::append_dense (opening_proof.rs:265-315): a dense committed
    // polynomial of this node opened at `point`
    int append_dense(const Node& nd, uint8_t cp_var, const Point& point, const H::Fr& claim) {
        H::tr_append_scalar(Tr, claim);
        const PolyId p = gr::comm(cp_var, nd.idx);
        openings[gr::node_exec(p, nd.idx)] = gr::Opening{point, claim};
        auto it = committed.find(p);
        if (it == committed.end()) return fail(ATLAS_ESTATE, "prove_graph: dense opening of a polynomial that was not committed");
        it->second->opened = true; it->second->point = point; it->second->claim = claim;
        return ATLAS_OK;
    }
    int append_advice(const Node& nd, uint8_t vp, const Point& point, const H::Fr& claim) {
        Out O = out();
        return O.append_virtual(Tr, gr::node_exec(gr::virt(vp, nd.idx), nd.idx), point, claim);
    }
    const gr::Opening& red(const Node& nd) const { return reduced.at(nd.idx); }

    // MultilinearPolynomial::from(tensor.padded_next_power_of_two()).evaluate(point) for device tensors
    int eval_i32(const int32_t* const* tensors, size_t count, size_t T, const Point& point, H::Fr* out_) {
        if (T == 1) { int rc1 = ATLAS_OK; for (size_t i = 0; i < count && !rc1; i++) rc1 = scalar_of(tensors[i], &out_[i]); return rc1; }
        std::vector<atlas_poly_t> ps(count, nullptr);
        int rc = ATLAS_OK;
        for (size_t i = 0; i < count && !rc; i++) rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(tensors[i]), T, &ps[i]);
        if (!rc) rc = atlas_poly_evaluate_many(ps.data(), count, (const atlas_fr_t*)point.data(), point.size(), (atlas_fr_t*)out_);
        for (auto p : ps) if (p) atlas_poly_free(p);
        return rc;
    }

    // Sumcheck::prove of one instance; returns the challenges (as field elements, in drawing order) and the final claims
    int run_single(atlas_instance_t inst, const H::Fr& claim, uint8_t proof_type, std::vector<H::Fr>& rs, std::vector<H::Fr>& fin) {
        const size_t n = atlas_instance_num_rounds(inst), stride = atlas_instance_degree(inst) + 1;
        std::vector<atlas_fr_t> rows(n * stride); std::vector<uint32_t> nco(n); std::vector<atlas_u128_t> ch(n);
        int rc = atlas_instance_prove(inst, (const atlas_fr_t*)&claim, &t, rows.data(), stride, nco.data(), ch.data());
        if (rc) return rc;
        atlas_fr_t f[64]; size_t nf = 0;
        rc = atlas_instance_final_claims(inst, f, 64, &nf);
        if (rc) return rc;
        fin.resize(nf); std::memcpy(fin.data(), f, nf * 32);
        rs.resize(n);
        for (size_t i = 0; i < n; i++) rs[i] = ch_fr(ch[i]);
        Out O = out();
        return O.put_proof(rows, stride, nco, n, proof_type);
    }
    static Point reversed(const std::vector<H::Fr>& rs) { return Point(rs.rbegin(), rs.rend()); }     // LITTLE_ENDIAN -> BIG_ENDIAN
    // Sumcheck::prove of an instance over ZERO variables (a scalar node): the input claim enters the transcript, no round follows, the proof
    // is the empty SumcheckInstanceProof; the caller's cache_openings appends the operands' single values
    int zero_rounds(const H::Fr& claim, uint8_t proof_type) {
        H::tr_append_scalar(Tr, claim);
        const uint8_t empty[8] = {0, 0, 0, 0, 0, 0, 0, 0};                     // compressed_polys: Vec of length 0
        proof(proof_type, empty, 8);
        return ATLAS_OK;
    }
    // the one value of a one-coefficient Fr polynomial in HBM
    int scalar_fr_of(const void* d_fr, H::Fr* out_) {
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        HIP_TRY(hipMemcpyAsync(out_, d_fr, sizeof(H::Fr), hipMemcpyDeviceToHost, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
        return ATLAS_OK;
    }
    int scalar_of(const int32_t* d_tensor, H::Fr* out_) {
        int32_t v = 0;
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        HIP_TRY(hipMemcpyAsync(&v, d_tensor, 4, hipMemcpyDeviceToHost, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
        *out_ = fr_from_i64_host((int64_t)v);
        return ATLAS_OK;
    }

    // ---------------------------------------------------------------- commit_witness_polynomials
    // get_committed_polynomials of every node (ops/*.rs) over the witness the trace left in HBM
    int collect_committed() {
        for (auto& kv : G.nodes) {
            const Node& nd = kv.second;
            NodeWitness& W = G.wit[nd.idx];
            W.committed.clear();
            for (auto pv : W.dense_views) if (pv) atlas_poly_free(pv);
            W.dense_views.clear();
            W.one_cycle_rows.clear();
            const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T);
            int drc = ATLAS_OK;
            // ONE cycle: a one-hot polynomial over K addresses has K x 1 coefficients, a single one at the address read — committed, opened and folded
            // into the joint polynomial as the K-coefficient dense row it is (same commitment g1[k], same round polynomials: both openings bind the
            // address variables HighToLow over eq(r_address, .)); the reference's generic flow at T = 1
            auto first_lookup = [&](const uint64_t* d_lookups, uint64_t* lk) {
                std::lock_guard<atlas_rt::Mutex> lk_(rt().mu);
                if (hipMemcpyAsync(lk, d_lookups, 8, hipMemcpyDeviceToHost, rt().stream) != hipSuccess || hipStreamSynchronize(rt().stream) != hipSuccess) drc = fail(ATLAS_ENODEV, "prove_graph: lookup index of a one-element node");
            };
            auto one_cycle_row = [&](gr::PolyId id, size_t hot, size_t width) {
                if (hot >= width) { drc = fail(ATLAS_EINVAL, "prove_graph: the lookup index of a one-element node lies outside its table"); return; }
                std::vector<int32_t> row(width, 0);
                row[hot] = 1;
                W.one_cycle_rows.emplace_back(new DevBuf());
                DevBuf& B = *W.one_cycle_rows.back();
                if (B.alloc(width * 4) != hipSuccess) { drc = fail(ATLAS_ENOMEM, "prove_graph: one-cycle chunk row"); return; }
                {
                    std::lock_guard<atlas_rt::Mutex> lk_(rt().mu);
                    if (hipMemcpyAsync(B.p, row.data(), width * 4, hipMemcpyHostToDevice, rt().stream) != hipSuccess || hipStreamSynchronize(rt().stream) != hipSuccess) { drc = fail(ATLAS_ENODEV, "prove_graph: one-cycle chunk row"); return; }
                }
                atlas_poly_t v = nullptr;
                drc = atlas_poly_wrap_device_i32(B.as<int32_t>(), width, &v);
                if (drc) return;
                W.dense_views.push_back(v);
                gr::Committed c; c.id = id; c.kind = 0; c.dense = v; c.log_T = gr::log2u(width);
                W.committed.push_back(c);
            };
            auto chunks = [&](uint8_t cp, const uint64_t* d_lookups, size_t log_K, bool one_cycle) {
                const size_t d = (log_K + 3) / 4;
                if (one_cycle) {
                    uint64_t lk = 0;
                    first_lookup(d_lookups, &lk);
                    for (size_t i = 0; i < d && !drc; i++) one_cycle_row(gr::comm(cp, nd.idx, i), (lk >> (4 * (d - 1 - i))) & 15, 16);
                    return;
                }
                for (size_t i = 0; i < d; i++) {
                    gr::Committed c; c.id = gr::comm(cp, nd.idx, i); c.kind = 1; c.d_lookups = d_lookups; c.log_T = log_T; c.log_K = log_K; c.chunk = i;
                    W.committed.push_back(c);
                }
            };
            auto dense = [&](uint8_t cp, void* d_data, bool is_i32) {
                atlas_poly_t v = nullptr;
                drc = is_i32 ? atlas_poly_wrap_device_i32((int32_t*)d_data, T, &v) : atlas_poly_wrap_device_fr(d_data, T, &v);
                if (drc) return;
                W.dense_views.push_back(v);
                gr::Committed c; c.id = gr::comm(cp, nd.idx); c.kind = 0; c.dense = v; c.log_T = log_T;
                W.committed.push_back(c);
            };
            // one element: the is_scalar operators commit nothing; ScalarConstDiv keeps its remainder and Div its quotient (ops/div.rs:157-160)
            if (T == 1 && nd.op != ATLAS_OP_SCALAR_CONST_DIV && nd.op != ATLAS_OP_DIV && nd.op != ATLAS_OP_RELU && nd.op != ATLAS_OP_CLAMP && !atlas_rt_is_activation(nd.op) &&
                nd.op != ATLAS_OP_RSQRT && nd.op != ATLAS_OP_SIN && nd.op != ATLAS_OP_COS && nd.op != ATLAS_OP_GATHER_SMALL && nd.op != ATLAS_OP_GATHER_LARGE) continue;      // (the lookup operators proper run their generic flows over one cycle)
            switch (nd.op) {
                case ATLAS_OP_ADD: case ATLAS_OP_SUB: chunks(gr::CP_ClampRaD, W.cidx.as<uint64_t>(), 64, T == 1); break;       // clamp_committed_polys
                case ATLAS_OP_EINSUM: case ATLAS_OP_MUL: case ATLAS_OP_SQUARE: case ATLAS_OP_CUBE:                     // fused_rebase::committed_polys
                    chunks(gr::CP_RescaleRemainderRaD, W.rescale->ridx.as<uint64_t>(), W.rescale->S, T == 1);
                    chunks(gr::CP_ClampRaD, W.rescale->cidx.as<uint64_t>(), 64, T == 1);
                    break;
                case ATLAS_OP_RELU: chunks(gr::CP_NodeOutputRaD, W.lookups.as<uint64_t>(), 32, T == 1); break;                 // ops/relu.rs
                case ATLAS_OP_CLAMP: chunks(gr::CP_SymmetricClampRaD, W.lookups.as<uint64_t>(), 32, T == 1); break;            // ops/clamp.rs
                case ATLAS_OP_SUM: chunks(gr::CP_ClampRaD, W.cidx.as<uint64_t>(), 64, T == 1); break;                           // ops/sum/mod.rs
                case ATLAS_OP_SCALAR_CONST_DIV: dense(gr::CP_ScalarConstDivNodeRemainder, W.rem.p, true); break;         // ops/scalar_const_div.rs
                case ATLAS_OP_DIV:                                                                                       // ops/div.rs
                    dense(gr::CP_DivNodeQuotient, G.out[nd.idx].p, true);
                    if (T > 1) chunks(gr::CP_DivRangeCheckRaD, W.lookups.as<uint64_t>(), 64, T == 1);
                    break;
                case ATLAS_OP_MEAN_OF_SQUARES:                                                                           // ops/mean_of_squares.rs
                    chunks(gr::CP_ClampRaD, W.rescale->cidx.as<uint64_t>(), 64, T == 1);
                    chunks(gr::CP_MeanOfSquaresRangeCheckRaD, W.lookups.as<uint64_t>(), 64, T == 1);
                    break;
                case ATLAS_OP_RSQRT:                                                                                     // ops/rsqrt.rs
                    dense(gr::CP_RsqrtQuotient, W.quot_fr.p, false);
                    chunks(gr::CP_SqrtDivRangeCheckRaD, W.lookups.as<uint64_t>(), 64, T == 1);
                    chunks(gr::CP_SqrtRangeCheckRaD, W.lookups2.as<uint64_t>(), 64, T == 1);
                    break;
                case ATLAS_OP_TANH: case ATLAS_OP_ERF: case ATLAS_OP_SIGMOID:                                            // clamped_activation_committed_polynomials
                    chunks(gr::CP_ActivationClampRaD, W.lookups.as<uint64_t>(), 32, T == 1);
                    chunks(gr::CP_ActivationSmallRaD, W.lookups2.as<uint64_t>(), gr::ACTIVATION_TABLE_VARS, T == 1);
                    break;
                case ATLAS_OP_SOFTMAX: {                                                                                 // ops/softmax_last_axis/mod.rs:136-159
                    SoftmaxWitness& Sm = *W.softmax;
                    chunks(gr::CP_SoftmaxRemainderRaD, Sm.idx_R.as<uint64_t>(), gr::MODEL_SCALE, T == 1);
                    chunks(gr::CP_SoftmaxExpRemainderRaD, Sm.idx_rexp.as<uint64_t>(), gr::MODEL_SCALE, T == 1);
                    chunks(gr::CP_SoftmaxClampRaD, Sm.idx_z.as<uint64_t>(), 32, T == 1);
                    chunks(gr::CP_SoftmaxZHiRaD, Sm.idx_zhi.as<uint64_t>(), Sm.lk_hi, T == 1);
                    chunks(gr::CP_SoftmaxZLoRaD, Sm.idx_zlo.as<uint64_t>(), Sm.lk_lo, T == 1);
                    break;
                }
                case ATLAS_OP_SIN: case ATLAS_OP_COS:                                                                    // ops/sin.rs:169-186
                    dense(gr::CP_TeleportNodeQuotient, W.rem2.p, true);
                    chunks(gr::CP_TrigDownscaleRaD, W.lookups.as<uint64_t>(), 32, T == 1);
                    chunks(nd.op == ATLAS_OP_SIN ? gr::CP_SinRaD : gr::CP_CosRaD, W.lookups2.as<uint64_t>(), gr::TRIG_TABLE_VARS, T == 1);
                    chunks(gr::CP_TeleportRangeCheckRaD, W.cidx.as<uint64_t>(), 64, T == 1);
                    break;
                case ATLAS_OP_GATHER_SMALL: {                                                                            // ops/gather/small.rs:119-121: ONE one-hot polynomial, dict_len addresses
                    if (gr::padded_len(G.nodes.at(nd.inputs[1]).dims) == 1) {                                               // ONE index: the dict_len x 1 coefficients as a dense row
                        uint64_t lk = 0;
                        first_lookup(W.lookups.as<uint64_t>(), &lk);
                        if (!drc) one_cycle_row(gr::comm(gr::CP_GatherRa, nd.idx), (size_t)lk, gr::next_pow2(G.nodes.at(nd.inputs[0]).dims[0]));
                        break;
                    }
                    gr::Committed c; c.id = gr::comm(gr::CP_GatherRa, nd.idx); c.kind = 1; c.d_lookups = W.lookups.as<uint64_t>();
                    c.log_T = gr::log2u(gr::padded_len(G.nodes.at(nd.inputs[1]).dims)); c.log_K = gr::log2u(G.nodes.at(nd.inputs[0]).dims[0]); c.chunk = 0; c.lkc = c.log_K;
                    W.committed.push_back(c);
                    break;
                }
                case ATLAS_OP_GATHER_LARGE: {                                                                            // ops/gather/large.rs:105-111
                    const size_t N = gr::padded_len(G.nodes.at(nd.inputs[1]).dims), V = G.nodes.at(nd.inputs[0]).dims[0], lk = gr::log2u(V), d = (lk + 3) / 4;
                    if (N == 1) { chunks(gr::CP_GatherRaD, W.lookups.as<uint64_t>(), lk, true); break; }                      // ONE index
                    for (size_t i = 0; i < d; i++) {
                        gr::Committed c; c.id = gr::comm(gr::CP_GatherRaD, nd.idx, i); c.kind = 1; c.d_lookups = W.lookups.as<uint64_t>(); c.log_T = gr::log2u(N); c.log_K = lk; c.chunk = i;
                        W.committed.push_back(c);
                    }
                    break;
                }
                default: break;
            }
            if (drc) return drc;
            for (auto& c : W.committed) committed[c.id] = &c;
        }
        return ATLAS_OK;
    }
    int commit_all() {
        // every lookup family of the graph in ONE launch (its d chunk commitments are consecutive rows); the dense advice polynomials one by one
        const bool trace = getenv("ATLAS_TRACE") != nullptr;
        auto t_prev = std::chrono::steady_clock::now();
        auto mark = [&](const char* what) {
            if (!trace) return;
            atlas_sync();
            const auto t1 = std::chrono::steady_clock::now();
            fprintf(stderr, "[atlas trace] commit_witness_polynomials %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t_prev).count());
            t_prev = t1;
        };
        std::vector<atlas_lookup_family_t> fams;
        std::vector<gr::Committed*> first;                                    // first chunk of each family
        std::vector<gr::Committed*> dense;
        for (auto& kv : G.wit) {
            auto& cs = kv.second.committed;
            for (size_t i = 0; i < cs.size();) {
                size_t j = i + 1;
                if (cs[i].kind == 1 && cs[i].lkc != 4) {                     // GatherRa: one polynomial over all log_K address bits
                    int rc = atlas_commit_lookup_chunks(srs, cs[i].d_lookups, cs[i].log_T, cs[i].log_K, cs[i].lkc, &cs[i].commitment);
                    if (rc) return rc;
                } else if (cs[i].kind == 1) {
                    while (j < cs.size() && cs[j].kind == 1 && cs[j].d_lookups == cs[i].d_lookups) j++;
                    fams.push_back(atlas_lookup_family_t{cs[i].d_lookups, cs[i].log_T, cs[i].log_K});
                    first.push_back(&cs[i]);
                } else dense.push_back(&cs[i]);
                i = j;
            }
        }
        if (!dense.empty()) {                                                 // the dense advice polynomials: one batch (atlas_commit_batch: the i32 ones share a bucket pipeline)
            std::vector<atlas_poly_t> dp(dense.size());
            std::vector<atlas_g1_affine_t> dc(dense.size());
            for (size_t q = 0; q < dense.size(); q++) dp[q] = dense[q]->dense;
            int rc = atlas_commit_batch(srs, dp.data(), dp.size(), dc.data());
            if (rc) return rc;
            for (size_t q = 0; q < dense.size(); q++) dense[q]->commitment = dc[q];
        }
        mark("dense + wide one-hot members");
        if (!fams.empty()) {
            size_t total = 0;
            for (auto& f : fams) total += (f.log_K + 3) / 4;
            std::vector<atlas_g1_affine_t> pts(total);
            int rc = ATLAS_OK;
            if (!sh || sh->world < 2) rc = atlas_commit_lookup_chunks_multi(srs, fams.data(), fams.size(), 4, pts.data());
            else {
                // commit_witness_polynomials by polynomial range over the ranks (prover.rs:71-87 commits them independently): a
                // contiguous range of families per rank, cut where the running cost d T (chunk polynomials x cycles = point additions) passes
                // rank / world of the total; every rank commits its range against its copy of the SRS, the commitments cross the board.
                const size_t world = (size_t)sh->world, rank = (size_t)sh->rank;
                std::vector<double> pre(fams.size() + 1, 0.0);
                for (size_t f = 0; f < fams.size(); f++) pre[f + 1] = pre[f] + (double)((fams[f].log_K + 3) / 4) * (double)((size_t)1 << fams[f].log_T);
                std::vector<size_t> cut(world + 1, fams.size());
                cut[0] = 0;
                for (size_t r = 1, f = 0; r < world; r++) { while (f < fams.size() && pre[f] < pre.back() * (double)r / (double)world) f++; cut[r] = f; }
                std::vector<size_t> row0(fams.size() + 1, 0);
                for (size_t f = 0; f < fams.size(); f++) row0[f + 1] = row0[f] + (fams[f].log_K + 3) / 4;
                size_t max_rows = 0;
                for (size_t r = 0; r < world; r++) max_rows = std::max(max_rows, row0[cut[r + 1]] - row0[cut[r]]);
                // record 0 of a rank's payload is its status word: a rank whose commit failed still ENTERS the exchange (the others would wait for the
                // timeout and the board's sequence numbers would part), and every rank returns the same error
                const size_t rec = max_rows + 1;
                std::vector<atlas_g1_affine_t> mine(rec), all(rec * world);
                std::memset(mine.data(), 0, mine.size() * sizeof(atlas_g1_affine_t));
                if (cut[rank + 1] > cut[rank]) rc = atlas_commit_lookup_chunks_multi(srs, fams.data() + cut[rank], cut[rank + 1] - cut[rank], 4, mine.data() + 1);
                const int32_t my_rc = (int32_t)rc;
                std::memcpy(mine.data(), &my_rc, sizeof(my_rc));
                if (!sh->allgather_bulk(mine.data(), rec * sizeof(atlas_g1_affine_t), all.data())) return fail(ATLAS_ENODEV, "prove_graph_sharded: a rank did not answer (commitments)");
                if (rc) return rc;
                for (size_t r = 0; r < world; r++) {
                    int32_t their_rc = 0;
                    std::memcpy(&their_rc, all.data() + r * rec, sizeof(their_rc));
                    if (their_rc) return fail(their_rc, "prove_graph_sharded: the witness commitments of another rank failed");
                }
                for (size_t r = 0; r < world; r++) std::memcpy(pts.data() + row0[cut[r]], all.data() + r * rec + 1, (row0[cut[r + 1]] - row0[cut[r]]) * sizeof(atlas_g1_affine_t));
            }
            if (rc) return rc;
            size_t o = 0;
            for (size_t f = 0; f < fams.size(); f++) { const size_t d = (fams[f].log_K + 3) / 4; for (size_t q = 0; q < d; q++) first[f][q].commitment = pts[o++]; }
        }
        mark("lookup families");
        for (auto& kv : committed) {                                          // transcript.append_serializable(commitment), BTreeMap order
            uint8_t b[64], rev[64];
            int rc = atlas_g1_to_bytes_uncompressed(&kv.second->commitment, b);
            if (rc) return rc;
            for (int i = 0; i < 64; i++) rev[i] = b[63 - i];
            H::tr_append_bytes(Tr, rev, 64);
        }
        mark("transcript");
        return ATLAS_OK;
    }

    // ---------------------------------------------------------------- output_claim (prover.rs:89-121)
    int output_claim() {
        const Node& nd = G.nodes.at(G.outputs[0]);
        const size_t T = gr::padded_len(nd.dims), n = gr::log2u(T);
        Point r(n);
        for (size_t i = 0; i < n; i++) { uint64_t lo, hi; H::tr_challenge_u128(Tr, lo, hi); r[i] = H::challenge_to_fr(lo, hi, rt().challenge_mode); }
        H::Fr claim;
        const int32_t* tp = G.tensor(nd.idx);
        int rc = eval_i32(&tp, 1, T, r, &claim);
        if (rc) return rc;
        cur = nd.idx;
        Out O = out();
        return O.append_virtual(Tr, gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.idx), nd.idx + 1), r, claim);
    }

    // ---------------------------------------------------------------- NodeEvalReduction::prove
    int eval_reduction(const Node& nd) {
        const size_t T = gr::padded_len(nd.dims), n = gr::log2u(T);
        const OpeningId lo = gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.idx), nd.idx), hi = gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.idx), ~(uint64_t)0);
        std::vector<atlas_fr_t> pts, cls;
        size_t N = 0;
        for (auto it = openings.lower_bound(lo); it != openings.end() && !(hi < it->first); ++it) {
            if (it->second.point.size() != n) return fail(ATLAS_ESTATE, "prove_graph: an opening of a node output has the wrong number of variables");
            const size_t o = pts.size(); pts.resize(o + n);
            if (n) std::memcpy(&pts[o], it->second.point.data(), n * 32);
            atlas_fr_t c; std::memcpy(&c, &it->second.claim, 32); cls.push_back(c);
            N++;
        }
        if (N == 0) return fail(ATLAS_ESTATE, "prove_graph: a node output without opening claims (every node must reach the output)");
        atlas_poly_t mle = nullptr;
        int rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(G.tensor(nd.idx)), T, &mle);
        if (rc) return rc;
        std::vector<atlas_fr_t> h(n * (N - 1) + 2), r(n ? n : 1);
        size_t h_len = 0; atlas_fr_t claim;
        rc = atlas_eval_reduction_prove(mle, pts.data(), cls.data(), N, n, &t, h.data(), h.size(), &h_len, r.data(), &claim);
        atlas_poly_free(mle);
        if (rc) return rc;
        std::vector<H::Fr>& hv = evalred[nd.idx];
        hv.resize(h_len); std::memcpy(hv.data(), h.data(), h_len * 32);
        gr::Opening& R = reduced[nd.idx];
        R.point.resize(n); if (n) std::memcpy(R.point.data(), r.data(), n * 32);
        std::memcpy(&R.claim, &claim, 32);
        return ATLAS_OK;
    }

    // ---------------------------------------------------------------- operators
    // Add / Sub (ops/add.rs:70-105): prove_clamp_lookup over the i64 accumulation, then the operand tie
    int op_addsub(const Node& nd) {
        const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T);
        const gr::Opening& R = red(nd);
        NodeWitness& W = G.wit[nd.idx];
        Out O = out();
        int rc = ATLAS_OK;
        H::Fr lr[2];
        bool have_lr = false;
        const int32_t* ops[2] = {G.tensor(nd.inputs[0]), G.tensor(nd.inputs[1])};
        if (T > 1) {
            atlas_poly_t p_acc = nullptr;
            H::Fr acc_claim;
            NodePre pre;                                                      // eq(r) and the clamp lookup's G table: in flight under the evaluation's wait
            rc = pre.begin((const atlas_fr_t*)R.point.data(), log_T, {{W.cidx.as<uint64_t>(), (size_t)64}});
            if (!rc) rc = pre.prebuild_clamp(W.cidx.as<uint64_t>(), log_T, (const atlas_fr_t*)R.point.data());
            if (!rc) rc = atlas_poly_wrap_device_fr(W.acc_fr.p, T, &p_acc);
            if (!rc && pre.eq) {
                // the operands' openings at the same point ride the same pass (they are appended after the lookup's proofs, but depend on nothing drawn
                // since: one evaluation and one wait per Add / Sub node instead of two — 136 nodes of the GPT-2-shaped graph)
                atlas_poly_t p0 = nullptr, p1 = nullptr;
                rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(ops[0]), T, &p0);
                if (!rc) rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(ops[1]), T, &p1);
                H::Fr ev[3];
                if (!rc) { const atlas_poly_t ps[3] = {p_acc, p0, p1}; rc = atlas_rt_evaluate_with_eq(ps, 3, pre.eq, (atlas_fr_t*)ev); }
                acc_claim = ev[0]; lr[0] = ev[1]; lr[1] = ev[2];
                have_lr = !rc;
                for (atlas_poly_t p : {p0, p1}) if (p) atlas_poly_free(p);
            } else if (!rc) rc = atlas_poly_evaluate(p_acc, (const atlas_fr_t*)R.point.data(), log_T, (atlas_fr_t*)&acc_claim);
            if (p_acc) atlas_poly_free(p_acc);
            if (!rc) rc = pre.collect(false);
            if (!rc) rc = O.append_virtual(Tr, gr::node_exec(gr::virt(gr::VP_ClampAcc, nd.idx), nd.idx), R.point, acc_claim);       // append_raf_claims_prover
            if (!rc) rc = prove_clamp_lookup_flow(W.cidx.as<uint64_t>(), log_T, (const atlas_fr_t*)R.point.data(), acc_claim, R.claim, &t, O, nullptr, &pre);
        }
        if (!rc && !have_lr) rc = eval_i32(ops, 2, T, R.point, lr);
        if (!rc) rc = append_nodeio(nd, 0, R.point, lr[0]);
        if (!rc) rc = append_nodeio(nd, 1, R.point, lr[1]);
        return rc;
    }

    // the element-wise sumcheck of an operator (MulProver / SquareProver / CubeProver / IffProver, ...): LowToHigh over the
    // Gruen split-eq of the reduced output point; cache_openings = the operand claims at the reversed challenges
    int ew_sumcheck(const Node& nd, int ew_op, size_t n_ops, const H::Fr& in_claim, uint8_t proof_type) {
        const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T);
        const gr::Opening& R = red(nd);
        atlas_poly_t ops[3] = {nullptr, nullptr, nullptr};
        int rc = ATLAS_OK;
        if (log_T == 0) {                                                     // a scalar node
            rc = zero_rounds(in_claim, proof_type);
            for (size_t q = 0; q < n_ops && !rc; q++) { H::Fr v; rc = scalar_of(G.tensor(nd.inputs[q]), &v); if (!rc) rc = append_nodeio(nd, q, Point(), v); }
            return rc;
        }
        for (size_t i = 0; i < n_ops && !rc; i++) rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(G.tensor(nd.inputs[i])), T, &ops[i]);
        atlas_instance_t inst = nullptr;
        if (!rc) rc = atlas_elementwise_new(ew_op, ops, n_ops, (const atlas_fr_t*)R.point.data(), log_T, nullptr, 0, &inst);
        for (atlas_poly_t p : ops) if (p) atlas_poly_free(p);
        std::vector<H::Fr> rs, fin;
        if (!rc) rc = run_single(inst, in_claim, proof_type, rs, fin);
        if (inst) atlas_instance_free(inst);
        const Point pt = reversed(rs);
        for (size_t q = 0; q < n_ops && !rc; q++) rc = append_nodeio(nd, q, pt, fin[q]);
        return rc;
    }

    // Mul / Square / Cube with fused rescaling (impl_fused_rescale_proof_api, ops/mod.rs:569-612) and Einsum (ops/einsum/mod.rs:57-115)
    int op_fused(const Node& nd) {
        const gr::Opening& R = red(nd);
        NodeWitness& W = G.wit[nd.idx];
        Out O = out();
        auto inner = [&](const H::Fr& in_claim) -> int {
            if (nd.op == ATLAS_OP_EINSUM) return einsum_matmul(nd, in_claim);
            const int ew = nd.op == ATLAS_OP_MUL ? ATLAS_EW_MUL : nd.op == ATLAS_OP_SQUARE ? ATLAS_EW_SQUARE : ATLAS_EW_CUBE;
            return ew_sumcheck(nd, ew, nd.op == ATLAS_OP_MUL ? 2 : 1, in_claim, gr::PT_RescaleArith);
        };
        return prove_fused_rescale(*W.rescale, inner, (const atlas_fr_t*)R.point.data(), (const atlas_fr_t*)&R.claim, &t, O, nullptr);
    }

    // EinsumProver::sumcheck + Sumcheck::prove (ProofType::EinsumMatmul): EinsumLayout::fold at the reduced point, the
    // dot-product sumcheck (HighToLow, EqSchedule of the layout), operand openings at EinsumLayout::operand_points
    int einsum_matmul(const Node& nd, const H::Fr& in_claim) {
        const gr::Opening& R = red(nd);
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
        int sched = ATLAS_EQ_NONE; size_t sa = 0, sb = 0;
        switch (layout) {
            case ATLAS_EINSUM_MK_KN_MN: r_m = r; r_n = r + lm; break;
            case ATLAS_EINSUM_K_NK_N: r_n = r; break;
            case ATLAS_EINSUM_BMK_BKN_MBN: case ATLAS_EINSUM_BMK_KBN_MBN: r_m = r; r_b = r + lm; r_n = r + lm + lb; sched = ATLAS_EQ_LOW; sa = lk; sb = lb; break;
            case ATLAS_EINSUM_MBK_BNK_BMN: case ATLAS_EINSUM_MBK_NBK_BMN: r_b = r; r_m = r + lb; r_n = r + lb + lm; sched = ATLAS_EQ_HIGH; sa = lb; sb = lk; break;
            default: return fail(ATLAS_EINVAL, "prove_graph: einsum layout not wired into the graph prover");
        }
        atlas_poly_t eq_m = nullptr, eq_n = nullptr, eq_b = nullptr, left = nullptr, right = nullptr;
        int rc = ATLAS_OK;
        if (layout != ATLAS_EINSUM_K_NK_N) rc = atlas_eq_evals((const atlas_fr_t*)r_m, lm, nullptr, &eq_m);     // (a scalar output: the empty point, the one-entry table)
        if (!rc) rc = atlas_eq_evals((const atlas_fr_t*)r_n, ln, nullptr, &eq_n);
        if (!rc) rc = atlas_einsum_fold(layout, d.data(), d.size(), G.tensor(nd.inputs[0]), G.tensor(nd.inputs[1]), eq_m, eq_n, &left, &right);
        if (eq_m) atlas_poly_free(eq_m);
        if (eq_n) atlas_poly_free(eq_n);
        if (!rc && layout == ATLAS_EINSUM_K_NK_N) {                          // left = the k-vector itself (k_nk_n.rs:46-68)
            atlas_poly_t v = nullptr;
            rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(G.tensor(nd.inputs[0])), k, &v);
            if (!rc) rc = atlas_poly_clone(v, &left);
            if (v) atlas_poly_free(v);
        }
        if (!rc && sched != ATLAS_EQ_NONE) rc = atlas_eq_evals((const atlas_fr_t*)r_b, lb, nullptr, &eq_b);
        atlas_dot_prover_t dp = nullptr;
        if (!rc) rc = atlas_dot_prover_new(left, right, eq_b, sched, sa, sb, &dp);
        if (rc) { for (atlas_poly_t p : {left, right, eq_b}) if (p) atlas_poly_free(p); return rc; }
        const size_t nr = atlas_dot_num_rounds(dp), deg = (size_t)atlas_dot_degree(dp);
        std::vector<atlas_fr_t> rows(nr * deg); std::vector<atlas_u128_t> chm(nr); atlas_fr_t fin[3];
        rc = atlas_sumcheck_prove_dot(dp, (const atlas_fr_t*)&in_claim, &t, rows.data(), chm.data(), fin);
        atlas_dot_prover_free(dp);
        if (rc) return rc;
        Out O = out();
        std::vector<uint32_t> nco(nr, (uint32_t)deg);
        rc = O.put_proof(rows, deg, nco, nr, gr::PT_EinsumMatmul);
        if (rc) return rc;
        std::vector<H::Fr> ch(nr);
        for (size_t i = 0; i < nr; i++) ch[i] = ch_fr(chm[i]);
        auto cat = [](std::initializer_list<std::pair<const H::Fr*, size_t>> parts) { Point p; for (auto& x : parts) p.insert(p.end(), x.first, x.first + x.second); return p; };
        Point lp, rp;
        switch (layout) {                                                     // EinsumLayout::operand_points
            case ATLAS_EINSUM_MK_KN_MN: lp = cat({{r_m, lm}, {ch.data(), lk}}); rp = cat({{ch.data(), lk}, {r_n, ln}}); break;
            case ATLAS_EINSUM_K_NK_N: lp = cat({{ch.data(), lk}}); rp = cat({{r_n, ln}, {ch.data(), lk}}); break;
            case ATLAS_EINSUM_BMK_BKN_MBN: case ATLAS_EINSUM_BMK_KBN_MBN: {   // (r_j, r_h) = challenges.split_at(log_k)
                const H::Fr *rj = ch.data(), *rh = ch.data() + lk;
                lp = cat({{rh, lb}, {r_m, lm}, {rj, lk}});
                rp = layout == ATLAS_EINSUM_BMK_BKN_MBN ? cat({{rh, lb}, {rj, lk}, {r_n, ln}}) : cat({{rj, lk}, {rh, lb}, {r_n, ln}});
                break;
            }
            default: {                                                        // MBK: (r_h, r_j) = challenges.split_at(log_b)
                const H::Fr *rh = ch.data(), *rj = ch.data() + lb;
                lp = cat({{r_m, lm}, {ch.data(), lb + lk}});
                rp = layout == ATLAS_EINSUM_MBK_BNK_BMN ? cat({{rh, lb}, {r_n, ln}, {rj, lk}}) : cat({{r_n, ln}, {ch.data(), lb + lk}});
                break;
            }
        }
        rc = append_nodeio(nd, 0, lp, *reinterpret_cast<H::Fr*>(&fin[0]));
        if (!rc) rc = append_nodeio(nd, 1, rp, *reinterpret_cast<H::Fr*>(&fin[1]));
        return rc;
    }

    // Neg / IsNan (ops/neg.rs, is_nan.rs): no sumcheck, the operand opened at the reduced point
    int op_operand_at_point(const Node& nd) {
        const gr::Opening& R = red(nd);
        H::Fr operand_claim;
        const int32_t* tp = G.tensor(nd.inputs[0]);
        int rc = eval_i32(&tp, 1, gr::padded_len(G.nodes.at(nd.inputs[0]).dims), R.point, &operand_claim);
        if (!rc) rc = append_nodeio(nd, 0, R.point, operand_claim);
        return rc;
    }

    // ReLU (ops/relu.rs:22-70); Clamp (ops/clamp.rs) is the same flow over ClampTable<32> = ClampBoundedTable<32, 9, true> under SymmetricClampRa(D)
    int op_relu(const Node& nd) {
        const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T), XLEN = 32;
        const gr::Opening& R = red(nd);
        NodeWitness& W = G.wit[nd.idx];
        Out O = out();
        const bool clamp = nd.op == ATLAS_OP_CLAMP;
        const uint8_t ra_vp = clamp ? gr::VP_SymmetricClampRa : gr::VP_NodeOutputRa, rad_cp = clamp ? gr::CP_SymmetricClampRaD : gr::CP_NodeOutputRaD;
        H::Fr operand_claim;
        const int32_t* tp = G.tensor(nd.inputs[0]);
        NodePre pre;                                                          // eq(r) and the lookup's G table: in flight under the operand's evaluation
        int rc = T > 1 ? pre.begin((const atlas_fr_t*)R.point.data(), log_T, {{W.lookups.as<uint64_t>(), XLEN}}) : ATLAS_OK;
        if (!rc && T > 1 && !clamp) rc = pre.prebuild_relu(W.lookups.as<uint64_t>(), log_T, XLEN, (const atlas_fr_t*)R.point.data());
        if (!rc && pre.eq) {
            atlas_poly_t pv = nullptr;
            rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(tp), T, &pv);
            if (!rc) rc = atlas_rt_evaluate_with_eq(&pv, 1, pre.eq, (atlas_fr_t*)&operand_claim);
            if (pv) atlas_poly_free(pv);
        } else if (!rc) rc = eval_i32(&tp, 1, T, R.point, &operand_claim);
        if (!rc) rc = pre.collect(false);
        if (!rc) rc = append_nodeio(nd, 0, R.point, operand_claim);           // append_raf_claims_prover: witness_opening_id = Input(0)
        if (rc) return rc;
        const H::Fr gamma = H::tr_challenge_scalar(Tr);
        atlas_instance_t exec = pre.take(pre.relu);                           // built ahead (NodePre::prebuild_relu)?
        if (exec) rc = atlas_rt_ps_set_gamma(exec, (const atlas_fr_t*)&gamma);
        else if (pre.eq && !clamp) rc = atlas_rt_ps_shout_relu_new(W.lookups.as<uint64_t>(), log_T, XLEN, (const atlas_fr_t*)R.point.data(), (const atlas_fr_t*)&gamma, pre.eq, &exec);
        else
        rc = clamp ? atlas_ps_shout_clamp_new(W.lookups.as<uint64_t>(), log_T, XLEN, gr::CLAMP_BOUND, 1, (const atlas_fr_t*)R.point.data(), (const atlas_fr_t*)&gamma, &exec)
                   : atlas_ps_shout_relu_new(W.lookups.as<uint64_t>(), log_T, XLEN, (const atlas_fr_t*)R.point.data(), (const atlas_fr_t*)&gamma, &exec);
        const H::Fr exec_claim = H::add(R.claim, H::mul(gamma, operand_claim));
        std::vector<atlas_u128_t> ch; H::Fr ra_claim; std::vector<atlas_fr_t> ra_point;
        if (!rc) rc = prove_single(exec, exec_claim, &t, O, ch, &ra_claim, XLEN, ra_vp, gr::PT_Execution, &ra_point);
        if (exec) atlas_instance_free(exec);
        if (!rc) rc = prove_onehot_checks(W.lookups.as<uint64_t>(), log_T, XLEN, (const atlas_fr_t*)R.point.data(), ra_point, ra_claim, &t, O, rad_cp, gr::PT_RaOneHotChecks, &pre);
        return rc;
    }

    // Reshape (ops/reshape.rs): sum_x input(x) selector(x), selector = the eq table of the reduced point carried over the
    // flat index (identical order when every dimension is a power of two); LowToHigh, degree 2
    int op_reshape(const Node& nd) {
        const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T);
        const gr::Opening& R = red(nd);
        atlas_poly_t ops[2] = {nullptr, nullptr};
        if (log_T == 0) {                                                     // a scalar: zero rounds, the operand's one value
            H::Fr v;
            int rc0 = zero_rounds(R.claim, gr::PT_Execution);
            if (!rc0) rc0 = scalar_of(G.tensor(nd.inputs[0]), &v);
            if (!rc0) rc0 = append_nodeio(nd, 0, Point(), v);
            return rc0;
        }
        int rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(G.tensor(nd.inputs[0])), T, &ops[0]);
        if (!rc) rc = atlas_eq_evals((const atlas_fr_t*)R.point.data(), log_T, nullptr, &ops[1]);
        atlas_instance_t inst = nullptr;
        if (!rc) rc = atlas_elementwise_new(ATLAS_EW_DOT, ops, 2, nullptr, log_T, nullptr, 0, &inst);
        for (atlas_poly_t p : ops) if (p) atlas_poly_free(p);
        std::vector<H::Fr> rs, fin;
        if (!rc) rc = run_single(inst, R.claim, gr::PT_Execution, rs, fin);
        if (inst) atlas_instance_free(inst);
        if (!rc) rc = append_nodeio(nd, 0, reversed(rs), fin[0]);
        return rc;
    }

    // MoveAxis (ops/moveaxis.rs): permute_challenge_groups, no sumcheck
    int op_moveaxis(const Node& nd) {
        const gr::Opening& R = red(nd);
        std::vector<Point> groups;
        size_t off = 0;
        for (size_t dim : nd.dims) { const size_t v = gr::log2u(dim); groups.emplace_back(R.point.begin() + off, R.point.begin() + off + v); off += v; }
        const size_t src = (size_t)nd.p[0], dst = (size_t)nd.p[1];
        Point gdst = groups[dst];
        groups.erase(groups.begin() + dst);
        groups.insert(groups.begin() + src, gdst);
        Point r_in;
        for (auto& gq : groups) r_in.insert(r_in.end(), gq.begin(), gq.end());
        return append_nodeio(nd, 0, r_in, R.claim);
    }

    // Broadcast (ops/broadcast.rs): the operand at the variables of the non-broadcast axes
    int op_broadcast(const Node& nd) {
        const gr::Opening& R = red(nd);
        const Node& in = G.nodes.at(nd.inputs[0]);
        const size_t off = nd.dims.size() - in.dims.size();
        Point r_in;
        size_t pos = 0;
        for (size_t a = 0; a < nd.dims.size(); a++) {
            const size_t v = gr::log2u(nd.dims[a]);
            const bool kept = a >= off && in.dims[a - off] == nd.dims[a];     // get_broadcast_dims: broadcast_dims[a] == 1
            if (kept) r_in.insert(r_in.end(), R.point.begin() + pos, R.point.begin() + pos + v);
            pos += v;
        }
        H::Fr claim;
        const int32_t* tp = G.tensor(in.idx);
        int rc = eval_i32(&tp, 1, gr::padded_len(in.dims), r_in, &claim);
        if (!rc) rc = append_nodeio(nd, 0, r_in, claim);
        return rc;
    }

    // the binary range check `remainder < bound` (range_checking/mod.rs:37-100, ps_shout/binary.rs:148-200): gamma, PS-Shout over
    // UnsignedLessThanTable<32> on interleave(remainder, bound), input claim 1 + gamma left + gamma^2 right
    int range_check_new(const uint64_t* d_lookups, size_t log_T, const Point& r_cycle, const H::Fr& left, const H::Fr& right, atlas_instance_t* inst, H::Fr* claim) {
        const H::Fr gamma = H::tr_challenge_scalar(Tr);
        *claim = H::add(H::one(), H::add(H::mul(gamma, left), H::mul(H::mul(gamma, gamma), right)));
        return atlas_ps_shout_ult_new(d_lookups, log_T, (const atlas_fr_t*)r_cycle.data(), (const atlas_fr_t*)&gamma, inst);
    }
    // one range check on its own Sumcheck::prove + its one-hot checks (ops/div.rs:453-497, mean_of_squares.rs:104-133)
    int range_and_onehot(const Node& nd, const uint64_t* d_lookups, const Point& r_cycle, const H::Fr& left, const H::Fr& right, uint8_t ra_vp, uint8_t rad_cp,
                         uint8_t pt_onehot) {
        const size_t log_T = r_cycle.size();
        Out O = out();
        atlas_instance_t inst = nullptr; H::Fr claim;
        int rc = range_check_new(d_lookups, log_T, r_cycle, left, right, &inst, &claim);
        std::vector<atlas_u128_t> ch; H::Fr ra_claim; std::vector<atlas_fr_t> ra_point;
        if (!rc) rc = prove_single(inst, claim, &t, O, ch, &ra_claim, 64, ra_vp, gr::PT_RangeCheck, &ra_point);
        if (inst) atlas_instance_free(inst);
        if (!rc) rc = prove_onehot_checks(d_lookups, log_T, 64, (const atlas_fr_t*)r_cycle.data(), ra_point, ra_claim, &t, O, rad_cp, pt_onehot);
        return rc;
    }
    Point challenge_point(size_t n) {
        Point r(n);
        for (size_t i = 0; i < n; i++) { uint64_t lo, hi; H::tr_challenge_u128(Tr, lo, hi); r[i] = H::challenge_to_fr(lo, hi, rt().challenge_mode); }
        return r;
    }

    // Sum (ops/sum/mod.rs:53-86): the clamp lookup over the i64 accumulation, then SumAxisProver over the reduced axis
    int op_sum(const Node& nd) {
        const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T);
        const gr::Opening& R = red(nd);
        NodeWitness& W = G.wit[nd.idx];
        Out O = out();
        size_t m, n; int axis;
        int rc = atlas_rt_sum_config(G.nodes.at(nd.inputs[0]).dims, nd.shape[0], m, n, axis);
        if (rc) return rc;
        atlas_poly_t p_acc = nullptr;
        H::Fr acc_claim;
        NodePre pre;
        if (T > 1) rc = pre.begin((const atlas_fr_t*)R.point.data(), log_T, {{W.cidx.as<uint64_t>(), (size_t)64}});
        if (!rc && T > 1) rc = pre.prebuild_clamp(W.cidx.as<uint64_t>(), log_T, (const atlas_fr_t*)R.point.data());
        if (!rc) rc = atlas_poly_wrap_device_fr(W.acc_fr.p, T, &p_acc);
        if (!rc) rc = atlas_poly_evaluate(p_acc, (const atlas_fr_t*)R.point.data(), log_T, (atlas_fr_t*)&acc_claim);
        if (p_acc) atlas_poly_free(p_acc);
        if (!rc) rc = pre.collect(T == 1);
        if (!rc) rc = O.append_virtual(Tr, gr::node_exec(gr::virt(gr::VP_ClampAcc, nd.idx), nd.idx), R.point, acc_claim);           // prove_append_acc / append_raf_claims_prover
        if (!rc && T > 1) rc = prove_clamp_lookup_flow(W.cidx.as<uint64_t>(), log_T, (const atlas_fr_t*)R.point.data(), acc_claim, R.claim, &t, O, nullptr, &pre);
        if (rc) return rc;
        atlas_poly_t eq = nullptr, v = nullptr;
        rc = atlas_eq_evals((const atlas_fr_t*)R.point.data(), log_T, nullptr, &eq);
        if (!rc) rc = axis == 0 ? atlas_fold_i32_rows(G.tensor(nd.inputs[0]), m, n, eq, &v) : atlas_fold_i32_cols(G.tensor(nd.inputs[0]), m, n, eq, &v);
        if (eq) atlas_poly_free(eq);
        atlas_instance_t inst = nullptr;
        if (!rc) rc = atlas_softmax_instance_new(ATLAS_SM_SUM_AXIS, v, nullptr, 0, gr::log2u(axis == 0 ? m : n), nullptr, &inst);
        if (v) atlas_poly_free(v);
        std::vector<H::Fr> rs, fin;
        if (!rc) rc = run_single(inst, acc_claim, gr::PT_SumReduction, rs, fin);
        if (inst) atlas_instance_free(inst);
        if (rc) return rc;
        Point pt;                                                             // axis.rs:244-262: (challenges | r) for Axis0, (r | challenges) for Axis1
        if (axis == 0) { pt = rs; pt.insert(pt.end(), R.point.begin(), R.point.end()); }
        else { pt = R.point; pt.insert(pt.end(), rs.begin(), rs.end()); }
        return append_nodeio(nd, 0, pt, fin[0]);
    }

    // ScalarConstDiv (ops/scalar_const_div.rs): sum_x eq(r, x) (left(x) - R(x)) = q(r) * divisor
    int op_scalar_const_div(const Node& nd) {
        const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T);
        const gr::Opening& R = red(nd);
        NodeWitness& W = G.wit[nd.idx];
        atlas_poly_t ops[2] = {nullptr, nullptr};
        if (log_T == 0) {                                                     // one element: zero rounds; the operand and the one-coefficient remainder open at the empty point
            int rc0 = zero_rounds(H::mul(R.claim, fr_from_i64_host(nd.p[0])), gr::PT_Execution);
            H::Fr v_in, v_rem;
            if (!rc0) rc0 = scalar_of(G.tensor(nd.inputs[0]), &v_in);
            if (!rc0) rc0 = scalar_of(W.rem.as<int32_t>(), &v_rem);
            if (!rc0) rc0 = append_nodeio(nd, 0, Point(), v_in);
            if (!rc0) rc0 = append_dense(nd, gr::CP_ScalarConstDivNodeRemainder, Point(), v_rem);
            return rc0;
        }
        int rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(G.tensor(nd.inputs[0])), T, &ops[0]);
        if (!rc) rc = atlas_poly_wrap_device_i32(W.rem.as<int32_t>(), T, &ops[1]);
        atlas_instance_t inst = nullptr;
        if (!rc) rc = atlas_elementwise_new(ATLAS_EW_SUB, ops, 2, (const atlas_fr_t*)R.point.data(), log_T, nullptr, 0, &inst);
        for (atlas_poly_t p : ops) if (p) atlas_poly_free(p);
        std::vector<H::Fr> rs, fin;
        if (!rc) rc = run_single(inst, H::mul(R.claim, fr_from_i64_host(nd.p[0])), gr::PT_Execution, rs, fin);
        if (inst) atlas_instance_free(inst);
        const Point pt = reversed(rs);
        if (!rc) rc = append_nodeio(nd, 0, pt, fin[0]);
        if (!rc) rc = append_dense(nd, gr::CP_ScalarConstDivNodeRemainder, pt, fin[1]);
        return rc;
    }

    // Slice (ops/slice.rs): sum_x input(x) selector(x) over the INPUT's hypercube
    int op_slice(const Node& nd) {
        const Node& in = G.nodes.at(nd.inputs[0]);
        const size_t T_out = gr::padded_len(nd.dims), T_in = gr::padded_len(in.dims), log_in = gr::log2u(T_in);
        const gr::Opening& R = red(nd);
        atlas_poly_t eq = nullptr, ops[2] = {nullptr, nullptr};
        DevBuf sel;
        HIP_TRY(sel.alloc(T_in * sizeof(Fr)));
        int rc = atlas_eq_evals((const atlas_fr_t*)R.point.data(), gr::log2u(T_out), nullptr, &eq);
        if (!rc) {
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            SliceMap M{}; M.n = (uint32_t)nd.dims.size();
            size_t stride = 1;
            for (int a = (int)in.dims.size() - 1; a >= 0; a--) { M.dim[a] = (uint32_t)nd.dims[a]; M.stride[a] = (uint32_t)stride; stride *= in.dims[a]; }
            const size_t base = (size_t)nd.p[1] * M.stride[nd.p[0]];
            HIP_TRY(hipMemsetAsync(sel.p, 0, T_in * sizeof(Fr), rt().stream));
            k_slice_selector<<<grid_of(T_out), 256, 0, rt().stream>>>((const Fr*)eq->d, M, base, T_out, sel.as<Fr>());
            HIP_TRY(hipStreamSynchronize(rt().stream));
        }
        if (eq) atlas_poly_free(eq);
        if (!rc) rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(G.tensor(in.idx)), T_in, &ops[0]);
        if (!rc) rc = atlas_poly_wrap_device_fr(sel.p, T_in, &ops[1]);
        atlas_instance_t inst = nullptr;
        if (!rc) rc = atlas_elementwise_new(ATLAS_EW_DOT, ops, 2, nullptr, log_in, nullptr, 0, &inst);
        for (atlas_poly_t p : ops) if (p) atlas_poly_free(p);
        std::vector<H::Fr> rs, fin;
        if (!rc) rc = run_single(inst, R.claim, gr::PT_Execution, rs, fin);
        if (inst) atlas_instance_free(inst);
        if (!rc) rc = append_nodeio(nd, 0, reversed(rs), fin[0]);
        return rc;
    }

    // Concat (ops/concat.rs): sum_t input_t(x) selector_t(x) over the largest operand's hypercube; each operand opened at the leading
    // variables of the (reversed) challenges
    int op_concat(const Node& nd) {
        const gr::Opening& R = red(nd);
        const size_t n_in = nd.inputs.size(), ax = (size_t)nd.p[0], r = nd.dims.size();
        size_t mx = 0;
        std::vector<size_t> nv(n_in);
        for (size_t k = 0; k < n_in; k++) { nv[k] = gr::log2u(gr::padded_len(G.nodes.at(nd.inputs[k]).dims)); mx = nv[k] > mx ? nv[k] : mx; }
        const size_t len = (size_t)1 << mx;
        if (mx == 0) {                                                        // every operand a scalar: zero rounds
            int rc0 = zero_rounds(R.claim, gr::PT_Execution);
            for (size_t k = 0; k < n_in && !rc0; k++) { H::Fr v; rc0 = scalar_of(G.tensor(nd.inputs[k]), &v); if (!rc0) rc0 = append_nodeio(nd, k, Point(), v); }
            return rc0;
        }
        std::vector<size_t> ostr(r);
        { size_t st = 1; for (int a = (int)r - 1; a >= 0; a--) { ostr[a] = st; st *= nd.dims[a]; } }
        atlas_poly_t eq = nullptr;
        int rc = atlas_eq_evals((const atlas_fr_t*)R.point.data(), R.point.size(), nullptr, &eq);
        std::vector<DevBuf> bufs(2 * n_in);
        std::vector<atlas_poly_t> ops(2 * n_in, nullptr);
        size_t off = 0;
        for (size_t k = 0; k < n_in && !rc; k++) {
            const Node& in = G.nodes.at(nd.inputs[k]);
            const size_t T_in = gr::padded_len(in.dims);
            const uint32_t shift = (uint32_t)(mx - nv[k]);
            HIP_TRY(bufs[2 * k].alloc(len * sizeof(Fr))); HIP_TRY(bufs[2 * k + 1].alloc(len * sizeof(Fr)));
            {
                std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
                SliceMap M{}; M.n = (uint32_t)r;
                for (size_t a = 0; a < r; a++) { M.dim[a] = (uint32_t)in.dims[a]; M.stride[a] = (uint32_t)ostr[a]; }
                k_concat_extend<<<grid_of(len), 256, 0, rt().stream>>>(G.tensor(in.idx), shift, len, bufs[2 * k].as<Fr>());
                HIP_TRY(hipMemsetAsync(bufs[2 * k + 1].p, 0, len * sizeof(Fr), rt().stream));
                k_concat_selector<<<grid_of(T_in), 256, 0, rt().stream>>>((const Fr*)eq->d, M, off * ostr[ax], shift, T_in, bufs[2 * k + 1].as<Fr>());
                HIP_TRY(hipStreamSynchronize(rt().stream));
            }
            off += in.dims[ax];
            rc = atlas_poly_wrap_device_fr(bufs[2 * k].p, len, &ops[2 * k]);
            if (!rc) rc = atlas_poly_wrap_device_fr(bufs[2 * k + 1].p, len, &ops[2 * k + 1]);
        }
        if (eq) atlas_poly_free(eq);
        atlas_instance_t inst = nullptr;
        if (!rc) rc = atlas_elementwise_new(ATLAS_EW_DOT, ops.data(), 2 * n_in, nullptr, mx, nullptr, 0, &inst);
        for (atlas_poly_t p : ops) if (p) atlas_poly_free(p);
        std::vector<H::Fr> rs, fin;
        if (!rc) rc = run_single(inst, R.claim, gr::PT_Execution, rs, fin);
        if (inst) atlas_instance_free(inst);
        const Point pt = reversed(rs);
        for (size_t k = 0; k < n_in && !rc; k++) rc = append_nodeio(nd, k, Point(pt.begin(), pt.begin() + nv[k]), fin[2 * k]);
        return rc;
    }

    // Div (ops/div.rs, ReductionFlow::Custom): the division sumcheck at a FRESH point, then the node's eval reduction, the committed
    // quotient at the reduced point, the range check R < divisor and its one-hot checks
    int op_div(const Node& nd) {
        const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T);
        NodeWitness& W = G.wit[nd.idx];
        const Point r = challenge_point(log_T);                              // DivParams::new
        atlas_poly_t ops[4] = {nullptr, nullptr, nullptr, nullptr};
        const int32_t* src[4] = {G.tensor(nd.inputs[0]), G.tensor(nd.inputs[1]), G.tensor(nd.idx), W.rem.as<int32_t>()};
        int rc = ATLAS_OK;
        std::vector<H::Fr> rs, fin;
        if (log_T == 0) {                                                     // one element (ops/div.rs:93, 146-160): zero rounds, the four single values; the quotient only, no range check
            rc = zero_rounds(H::zero(), gr::PT_Execution);
            fin.resize(4);
            for (int i = 0; i < 4 && !rc; i++) rc = scalar_of(src[i], &fin[i]);
        } else {
            for (int i = 0; i < 4 && !rc; i++) rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(src[i]), T, &ops[i]);
            atlas_instance_t inst = nullptr;
            if (!rc) rc = atlas_elementwise_new(ATLAS_EW_DIV, ops, 4, (const atlas_fr_t*)r.data(), log_T, nullptr, 0, &inst);
            for (atlas_poly_t p : ops) if (p) atlas_poly_free(p);
            if (!rc) rc = run_single(inst, H::zero(), gr::PT_Execution, rs, fin);
            if (inst) atlas_instance_free(inst);
        }
        if (rc) return rc;
        const Point pt = reversed(rs);
        rc = append_nodeio(nd, 0, pt, fin[0]);
        if (!rc) rc = append_nodeio(nd, 1, pt, fin[1]);
        if (!rc) { Out O = out(); rc = O.append_virtual(Tr, gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.idx), nd.idx), pt, fin[2]); }     // Target::Current
        if (!rc) rc = append_advice(nd, gr::VP_DivRemainder, pt, fin[3]);
        if (!rc) rc = eval_reduction(nd);
        if (rc) return rc;
        const gr::Opening& R = red(nd);
        rc = append_dense(nd, gr::CP_DivNodeQuotient, R.point, R.claim);
        if (rc || T == 1) return rc;
        return range_and_onehot(nd, W.lookups.as<uint64_t>(), pt, fin[3], fin[1], gr::VP_DivRangeCheckRa, gr::CP_DivRangeCheckRaD, gr::PT_RaOneHotChecks);
    }

    // MeanOfSquares (ops/mean_of_squares.rs:50-133)
    int op_mean_of_squares(const Node& nd) {
        const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T);
        const gr::Opening& R = red(nd);
        NodeWitness& W = G.wit[nd.idx];
        RescaleWitness& RW = *W.rescale;
        const Node& in = G.nodes.at(nd.inputs[0]);
        const size_t T_in = gr::padded_len(in.dims), log_red = gr::log2u(T_in) - log_T;
        const int64_t D = ((int64_t)1 << nd.p[0]) * (int64_t)nd.p[1];
        Out O = out();
        atlas_poly_t p_rem = nullptr, p_quot = nullptr;
        NodePre pre;
        int rc = T > 1 ? pre.begin((const atlas_fr_t*)R.point.data(), log_T, {{RW.cidx.as<uint64_t>(), (size_t)64}}) : ATLAS_OK;
        if (!rc && T > 1) rc = pre.prebuild_clamp(RW.cidx.as<uint64_t>(), log_T, (const atlas_fr_t*)R.point.data());
        if (!rc) rc = atlas_poly_wrap_device_fr(RW.qfr.p, T, &p_quot);
        if (!rc) rc = atlas_poly_wrap_device_i32(RW.rem.as<int32_t>(), T, &p_rem);
        H::Fr ev[2];
        if (!rc) { const atlas_poly_t ps[2] = {p_rem, p_quot}; rc = pre.eq ? atlas_rt_evaluate_with_eq(ps, 2, pre.eq, (atlas_fr_t*)ev) : atlas_poly_evaluate_many(ps, 2, (const atlas_fr_t*)R.point.data(), log_T, (atlas_fr_t*)ev); }
        for (atlas_poly_t p : {p_rem, p_quot}) if (p) atlas_poly_free(p);
        if (!rc) rc = pre.collect(false);
        if (rc) return rc;
        const H::Fr eval_R = ev[0], acc_claim = ev[1];
        rc = append_advice(nd, gr::VP_RescaleRemainder, R.point, eval_R);                                                // fused_rebase::prove_pre
        if (!rc) rc = append_advice(nd, gr::VP_ClampAcc, R.point, acc_claim);
        if (!rc && T > 1) rc = prove_clamp_lookup_flow(RW.cidx.as<uint64_t>(), log_T, (const atlas_fr_t*)R.point.data(), acc_claim, R.claim, &t, O, nullptr, &pre);
        if (rc) return rc;
        // MeanOfSquaresReductionProver: sum_{k,j} eq(r, k) x[k,j]^2 = rescaled(r) D + R(r); HighToLow, EqSchedule::High { log_retained, log_reduced }
        DevBuf lbuf, rbuf;
        HIP_TRY(lbuf.alloc(T_in * sizeof(Fr))); HIP_TRY(rbuf.alloc(T_in * sizeof(Fr)));
        {
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            k_i32_to_fr<<<grid_of(T_in), 256, 0, rt().stream>>>(G.tensor(in.idx), lbuf.as<Fr>(), T_in);
            HIP_TRY(hipMemcpyAsync(rbuf.p, lbuf.p, T_in * sizeof(Fr), hipMemcpyDeviceToDevice, rt().stream));
        }
        atlas_poly_t left = nullptr, right = nullptr, eq = nullptr;
        rc = atlas_poly_wrap_device_fr(lbuf.p, T_in, &left);
        if (!rc) rc = atlas_poly_wrap_device_fr(rbuf.p, T_in, &right);
        if (!rc) rc = atlas_eq_evals((const atlas_fr_t*)R.point.data(), log_T, nullptr, &eq);
        atlas_dot_prover_t dp = nullptr;
        if (!rc) rc = atlas_dot_prover_new(left, right, eq, ATLAS_EQ_HIGH, log_T, log_red, &dp);
        if (rc) { for (atlas_poly_t p : {left, right, eq}) if (p) atlas_poly_free(p); return rc; }
        const size_t nr = log_T + log_red;
        std::vector<atlas_fr_t> rows(nr * 3); std::vector<atlas_u128_t> chm(nr); atlas_fr_t fin[3];
        const H::Fr in_claim = H::add(H::mul(acc_claim, H::from_u64((uint64_t)D)), eval_R);
        rc = atlas_sumcheck_prove_dot(dp, (const atlas_fr_t*)&in_claim, &t, rows.data(), chm.data(), fin);
        atlas_dot_prover_free(dp);
        if (rc) return rc;
        std::vector<uint32_t> nco(nr, 3);
        rc = O.put_proof(rows, 3, nco, nr, gr::PT_RescaleArith);
        Point pt(nr);
        for (size_t i = 0; i < nr; i++) pt[i] = ch_fr(chm[i]);
        if (!rc) rc = append_nodeio(nd, 0, pt, *reinterpret_cast<H::Fr*>(&fin[0]));
        if (rc || T == 1) return rc;
        return range_and_onehot(nd, W.lookups.as<uint64_t>(), R.point, eval_R, fr_from_i64_host(D), gr::VP_MeanOfSquaresRangeCheckRa, gr::CP_MeanOfSquaresRangeCheckRaD,
                                gr::PT_RescaleRemainderRaChecks);
    }

    // Rsqrt (ops/rsqrt.rs, ReductionFlow::Custom): x q + r_d = S^3 and out^2 + r_s = q in one sumcheck at a fresh point, the eval
    // reduction, then BOTH range checks in one BatchedSumcheck and their six one-hot instances in another
    int op_rsqrt(const Node& nd) {
        const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T);
        NodeWitness& W = G.wit[nd.idx];
        const Point r = challenge_point(log_T);
        const H::Fr gamma = H::tr_challenge_scalar(Tr);
        atlas_poly_t ops[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        int rc = ATLAS_OK;
        std::vector<H::Fr> rs, fin;
        if (log_T == 0) {                                                     // one element: zero rounds, the five single values (the generic flow at T = 1)
            rc = zero_rounds(H::zero(), gr::PT_Execution);
            fin.resize(5);
            if (!rc) rc = scalar_of(G.tensor(nd.inputs[0]), &fin[0]);
            if (!rc) rc = scalar_fr_of(W.quot_fr.p, &fin[1]);
            if (!rc) rc = scalar_of(G.tensor(nd.idx), &fin[2]);
            if (!rc) rc = scalar_of(W.rem.as<int32_t>(), &fin[3]);
            if (!rc) rc = scalar_of(W.rem2.as<int32_t>(), &fin[4]);
        } else {
            rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(G.tensor(nd.inputs[0])), T, &ops[0]);
            if (!rc) rc = atlas_poly_wrap_device_fr(W.quot_fr.p, T, &ops[1]);
            if (!rc) rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(G.tensor(nd.idx)), T, &ops[2]);
            if (!rc) rc = atlas_poly_wrap_device_i32(W.rem.as<int32_t>(), T, &ops[3]);
            if (!rc) rc = atlas_poly_wrap_device_i32(W.rem2.as<int32_t>(), T, &ops[4]);
            const H::Fr consts[2] = {H::from_u64((uint64_t)1 << (3 * nd.p[0])), gamma};
            atlas_instance_t inst = nullptr;
            if (!rc) rc = atlas_elementwise_new(ATLAS_EW_RSQRT, ops, 5, (const atlas_fr_t*)r.data(), log_T, (const atlas_fr_t*)consts, 2, &inst);
            for (atlas_poly_t p : ops) if (p) atlas_poly_free(p);
            if (!rc) rc = run_single(inst, H::zero(), gr::PT_Execution, rs, fin);
            if (inst) atlas_instance_free(inst);
        }
        if (rc) return rc;
        const Point pt = reversed(rs);
        rc = append_nodeio(nd, 0, pt, fin[0]);
        if (!rc) rc = append_dense(nd, gr::CP_RsqrtQuotient, pt, fin[1]);
        if (!rc) { Out O = out(); rc = O.append_virtual(Tr, gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.idx), nd.idx), pt, fin[2]); }
        if (!rc) rc = append_advice(nd, gr::VP_DivRemainder, pt, fin[3]);
        if (!rc) rc = append_advice(nd, gr::VP_SqrtRemainder, pt, fin[4]);
        if (!rc) rc = eval_reduction(nd);
        if (rc) return rc;
        // prove_range_and_onehot (rsqrt.rs:527-586)
        Out O = out();
        atlas_instance_t rc_i[2] = {nullptr, nullptr}; H::Fr cl[2];
        rc = range_check_new(W.lookups.as<uint64_t>(), log_T, pt, fin[3], fin[0], &rc_i[0], &cl[0]);                      // RiRangeCheckOperands: r_d < x
        if (!rc) rc = range_check_new(W.lookups2.as<uint64_t>(), log_T, pt, fin[4], H::add(H::add(fin[2], fin[2]), H::one()), &rc_i[1], &cl[1]);     // Rs: r_s < 2 out + 1
        atlas_batched_t b = nullptr;
        if (!rc) rc = atlas_batched_new(&b);
        for (int i = 0; i < 2 && !rc; i++) rc = atlas_batched_add_instance(b, rc_i[i], (const atlas_fr_t*)&cl[i]);
        const size_t mr0 = 64 + log_T;
        std::vector<atlas_fr_t> rows(mr0 * 3); std::vector<uint32_t> nco(mr0); std::vector<atlas_u128_t> ch(mr0); size_t mr = 0;
        if (!rc) rc = atlas_batched_prove(b, &t, rows.data(), 3, nco.data(), ch.data(), &mr);
        std::vector<OneHotFamily> fams(2);
        for (int i = 0; i < 2 && !rc; i++) {                                  // cache_openings: the ra claim of each read-raf instance
            atlas_fr_t f[64]; size_t nf = 0;
            rc = atlas_instance_final_claims(rc_i[i], f, 64, &nf);
            Point rp(mr);
            for (size_t q = 0; q < mr; q++) rp[q] = ch_fr(q < 64 ? ch[q] : ch[64 + (mr - 1 - q)]);
            if (!rc) rc = O.append_virtual(Tr, gr::node_exec(gr::virt(i == 0 ? gr::VP_DivRangeCheckRa : gr::VP_SqrtRangeCheckRa, nd.idx), nd.idx), rp, *reinterpret_cast<H::Fr*>(&f[0]));
            fams[i].d_lookups = i == 0 ? W.lookups.as<uint64_t>() : W.lookups2.as<uint64_t>(); fams[i].log_K = 64; fams[i].r_cycle = (const atlas_fr_t*)pt.data();
            fams[i].ra_point.resize(mr); std::memcpy(fams[i].ra_point.data(), rp.data(), mr * 32);
            std::memcpy(&fams[i].ra_claim, &f[0], 32);
            fams[i].rad_cp = i == 0 ? gr::CP_SqrtDivRangeCheckRaD : gr::CP_SqrtRangeCheckRaD;
        }
        if (!rc) rc = O.put_proof(rows, 3, nco, mr, gr::PT_RangeCheck);
        if (b) atlas_batched_free(b);
        for (auto i : rc_i) if (i) atlas_instance_free(i);
        if (!rc) rc = prove_onehot_checks_multi(fams, log_T, &t, O, gr::PT_RaOneHotChecks);
        return rc;
    }

    // Tanh (ops/tanh.rs -> activation_clamped/mod.rs:440-527): Execution = the 2^18-entry table lookup of the clamped input
    // (gamma-batched with the signed identity), NeuralTeleport = the clamp lookup tying `clamped` to the raw input, then both
    // lookups' one-hot checks in ONE batched sumcheck
    int op_tanh(const Node& nd) {
        const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T), LK = gr::ACTIVATION_TABLE_VARS, K = (size_t)1 << LK;
        const gr::Opening& R = red(nd);
        NodeWitness& W = G.wit[nd.idx];
        Out O = out();
        mark(nullptr);
        const H::Fr gamma = H::tr_challenge_scalar(Tr);                      // SmallTableParams::new
        const int32_t* cl = W.clamped.as<int32_t>();
        H::Fr clamped_claim;
        int rc = eval_i32(&cl, 1, T, R.point, &clamped_claim);
        if (!rc) rc = append_advice(nd, gr::VP_ActivationClampedOutput, R.point, clamped_claim);
        if (rc) return rc;
        // input_onehot[k] = sum_{j : idx_j = k} eq(r, j); table and signed identity as polynomials over the 18 address bits
        atlas_poly_t eq = nullptr, ops[3] = {nullptr, nullptr, nullptr};
        rc = atlas_eq_evals((const atlas_fr_t*)R.point.data(), log_T, nullptr, &eq);
        if (!rc) rc = atlas_shout_read_raf_G(W.lookups2.as<uint64_t>(), T, LK, eq, &ops[0]);
        if (eq) atlas_poly_free(eq);
        mark("tanh: clamped claim + G");
        const int32_t* d_table = nullptr;
        if (!rc) rc = atlas_rt_activation_table(nd.op, &d_table, nullptr);
        if (!rc) rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(d_table), K, &ops[1]);
        std::vector<int32_t> ident(K);
        for (size_t i = 0; i < K; i++) ident[i] = i >= K / 2 ? (int32_t)i - (int32_t)K : (int32_t)i;      // SignedIdentityPoly (signed_identity_poly.rs)
        if (!rc) rc = atlas_poly_upload_i32(ident.data(), K, &ops[2]);
        atlas_instance_t inst = nullptr;
        if (!rc) rc = atlas_elementwise_new(ATLAS_EW_GATHER, ops, 3, nullptr, LK, (const atlas_fr_t*)&gamma, 1, &inst);
        for (atlas_poly_t p : ops) if (p) atlas_poly_free(p);
        mark("tanh: table operands + instance");
        std::vector<H::Fr> rs, fin;
        if (!rc) rc = run_single(inst, H::add(R.claim, H::mul(gamma, clamped_claim)), gr::PT_Execution, rs, fin);
        if (inst) atlas_instance_free(inst);
        mark("tanh: table sumcheck");
        if (rc) return rc;
        Point small_pt = reversed(rs);                                        // ActivationSmallRa at (r_table | r_node_output)
        small_pt.insert(small_pt.end(), R.point.begin(), R.point.end());
        rc = append_advice(nd, gr::VP_ActivationSmallRa, small_pt, fin[0]);
        // ActivationClamp: read_raf_prove over ActivationClampTable<32> = ClampBoundedTable<32, 17, true> (ProofType::NeuralTeleport)
        H::Fr operand_claim;
        const int32_t* tp = G.tensor(nd.inputs[0]);
        if (!rc) rc = eval_i32(&tp, 1, T, R.point, &operand_claim);
        if (!rc) rc = append_nodeio(nd, 0, R.point, operand_claim);           // append_raf_claims_prover
        if (rc) return rc;
        const H::Fr gamma2 = H::tr_challenge_scalar(Tr);
        atlas_instance_t cinst = nullptr;
        rc = atlas_ps_shout_clamp_new(W.lookups.as<uint64_t>(), log_T, 32, gr::ACTIVATION_BOUND, 1, (const atlas_fr_t*)R.point.data(), (const atlas_fr_t*)&gamma2, &cinst);
        std::vector<atlas_u128_t> ch; H::Fr ra_claim; std::vector<atlas_fr_t> ra_point;
        if (!rc) rc = prove_single(cinst, H::add(clamped_claim, H::mul(gamma2, operand_claim)), &t, O, ch, &ra_claim, 32, gr::VP_ActivationClampRa, gr::PT_NeuralTeleport, &ra_point);
        if (cinst) atlas_instance_free(cinst);
        mark("tanh: clamp lookup");
        if (rc) return rc;
        std::vector<OneHotFamily> fams(2);
        fams[0].d_lookups = W.lookups2.as<uint64_t>(); fams[0].log_K = LK; fams[0].r_cycle = (const atlas_fr_t*)R.point.data();
        fams[0].ra_point.resize(small_pt.size()); std::memcpy(fams[0].ra_point.data(), small_pt.data(), small_pt.size() * 32);
        fams[0].ra_claim = fin[0]; fams[0].rad_cp = gr::CP_ActivationSmallRaD;
        fams[1].d_lookups = W.lookups.as<uint64_t>(); fams[1].log_K = 32; fams[1].r_cycle = (const atlas_fr_t*)R.point.data();
        fams[1].ra_point = ra_point; fams[1].ra_claim = ra_claim; fams[1].rad_cp = gr::CP_ActivationClampRaD;
        return prove_onehot_checks_multi(fams, log_T, &t, O, gr::PT_RaOneHotChecks);
    }

    // GatherLarge (ops/gather/large.rs + mod.rs): out[j][w] = dict[idx_j][w] as sum_k ra(k) (dict_r(k) + gamma k)
    int op_gather(const Node& nd) {
        const Node& dict = G.nodes.at(nd.inputs[0]);
        const Node& idxn = G.nodes.at(nd.inputs[1]);
        const size_t V = dict.dims[0], word = gr::padded_len(dict.dims) / V, N = gr::padded_len(idxn.dims), lv = gr::log2u(V), ln = gr::log2u(N), lw = gr::log2u(word);
        const gr::Opening& R = red(nd);
        NodeWitness& W = G.wit[nd.idx];
        Out O = out();
        const H::Fr gamma = H::tr_challenge_scalar(Tr);                      // GatherParams::new
        const Point r_index(R.point.begin(), R.point.begin() + ln), r_word(R.point.begin() + ln, R.point.end());
        H::Fr index_claim;
        const int32_t* ip = G.tensor(idxn.idx);
        int rc = eval_i32(&ip, 1, N, r_index, &index_claim);
        if (!rc) rc = append_nodeio(nd, 1, r_index, index_claim);
        if (rc) return rc;
        atlas_poly_t eq_i = nullptr, eq_w = nullptr, ops[3] = {nullptr, nullptr, nullptr};
        rc = atlas_eq_evals((const atlas_fr_t*)r_index.data(), ln, nullptr, &eq_i);
        if (!rc) rc = atlas_shout_read_raf_G(W.lookups.as<uint64_t>(), N, lv, eq_i, &ops[0]);          // compute_ra_evals(r_index, indexes, num_words)
        if (!rc) rc = atlas_eq_evals((const atlas_fr_t*)r_word.data(), lw, nullptr, &eq_w);
        if (!rc) rc = atlas_fold_i32_rows(G.tensor(dict.idx), V, word, eq_w, &ops[1]);                  // fold_dictionary(r_word, dictionary)
        for (atlas_poly_t p : {eq_i, eq_w}) if (p) atlas_poly_free(p);
        std::vector<int32_t> ident(V);
        for (size_t i = 0; i < V; i++) ident[i] = (int32_t)i;                // IdentityPolynomial
        if (!rc) rc = atlas_poly_upload_i32(ident.data(), V, &ops[2]);
        atlas_instance_t inst = nullptr;
        if (!rc) rc = atlas_elementwise_new(ATLAS_EW_GATHER, ops, 3, nullptr, lv, (const atlas_fr_t*)&gamma, 1, &inst);
        for (atlas_poly_t p : ops) if (p) atlas_poly_free(p);
        std::vector<H::Fr> rs, fin;
        if (!rc) rc = run_single(inst, H::add(R.claim, H::mul(gamma, index_claim)), gr::PT_Execution, rs, fin);
        if (inst) atlas_instance_free(inst);
        if (rc) return rc;
        Point ra_pt = reversed(rs), dict_pt = ra_pt;
        ra_pt.insert(ra_pt.end(), r_index.begin(), r_index.end());
        dict_pt.insert(dict_pt.end(), r_word.begin(), r_word.end());
        rc = append_advice(nd, gr::VP_NodeOutputRa, ra_pt, fin[0]);
        if (!rc) rc = append_nodeio(nd, 0, dict_pt, fin[1]);
        if (rc) return rc;
        if (nd.op == ATLAS_OP_GATHER_SMALL) return gather_small_checks(nd, r_index, lv, ln);
        std::vector<atlas_fr_t> rap(ra_pt.size());
        std::memcpy(rap.data(), ra_pt.data(), ra_pt.size() * 32);
        return prove_onehot_checks(W.lookups.as<uint64_t>(), ln, lv, (const atlas_fr_t*)r_index.data(), rap, fin[0], &t, O, gr::CP_GatherRaD, gr::PT_RaOneHotChecks);
    }

    // GatherSmall after the execution sumcheck (ops/gather/small.rs:44-62, 124-168, 290-312): ONE committed one-hot polynomial GatherRa over all
    // dict_len addresses; BatchedSumcheck [HammingBooleanity of the all-ones hamming weight vector, Booleanity with d = 1 and log_k_chunk =
    // log dict_len, gamma = Challenge::from(1)] under RaOneHotChecks, then HammingWeight (d = 1) on its own under RaHammingWeight.  r_cycle of
    // all three = the point the index operand was opened at.
    int gather_small_checks(const Node& nd, const Point& r_index, size_t lv, size_t ln) {
        NodeWitness& W = G.wit[nd.idx];
        const size_t N = (size_t)1 << ln, K = (size_t)1 << lv;
        const uint64_t* lookups = W.lookups.as<uint64_t>();
        const Point r_address = challenge_point(lv);                          // ra_booleanity_params: challenge_vector_optimized(log num_words)
        // compute_ra_evals(r_cycle, indexes, num_words): the histogram of eq(r_index, .) over the addresses, on the host for the two instances
        atlas_poly_t eq_i = nullptr, Gp = nullptr;
        int rc = atlas_eq_evals((const atlas_fr_t*)r_index.data(), ln, nullptr, &eq_i);
        if (!rc) rc = atlas_shout_read_raf_G(lookups, N, lv, eq_i, &Gp);
        if (eq_i) atlas_poly_free(eq_i);
        std::vector<atlas_fr_t> Gh(K);
        if (!rc) rc = atlas_poly_download(Gp, Gh.data(), K);
        if (Gp) atlas_poly_free(Gp);
        if (rc) return rc;
        // hw = [1; N] (every lookup reads exactly one word)
        std::vector<H::Fr> ones(N, H::one());
        atlas_poly_t hw = nullptr;
        rc = atlas_poly_upload_fr((const atlas_fr_t*)ones.data(), N, &hw);
        const H::Fr one = H::one(), zero = H::zero();
        const H::Fr gamma_b = H::challenge_to_fr(1, 0, rt().challenge_mode);   // F::Challenge::from(1): the challenge READING of the integer 1
        atlas_instance_t i_hb = nullptr, i_bool = nullptr, i_hw = nullptr;
        const atlas_fr_t no_point{};
        const atlas_fr_t* rix = ln ? (const atlas_fr_t*)r_index.data() : &no_point;          // ONE index: r_cycle is the empty point
        if (!rc) rc = ln ? atlas_elementwise_new(ATLAS_EW_HAMMING_BOOL, &hw, 1, rix, ln, (const atlas_fr_t*)&one, 1, &i_hb)
                         : atlas_rt_const_member_new((const atlas_fr_t*)&one, 1, 3, &i_hb);  // hw over zero variables: no rounds, final claim hw[0] = 1
        if (hw) atlas_poly_free(hw);
        if (!rc) rc = atlas_booleanity_from_lookups_new(Gh.data(), lookups, ln, lv, lv, (const atlas_fr_t*)&gamma_b, (const atlas_fr_t*)r_address.data(), rix, &i_bool);
        atlas_batched_t b = nullptr;
        if (!rc) rc = atlas_batched_new(&b);
        if (!rc) rc = atlas_batched_add_instance(b, i_hb, (const atlas_fr_t*)&zero);
        if (!rc) rc = atlas_batched_add_instance(b, i_bool, (const atlas_fr_t*)&zero);
        std::vector<H::Fr> rs;
        if (!rc) rc = run_batch(b, 8, lv + ln, gr::PT_RaOneHotChecks, rs);
        Out O = out();
        if (!rc) {
            const size_t mr = rs.size();                                      // = lv + ln; HammingBooleanity sees the last ln challenges
            atlas_fr_t f[64]; size_t nf = 0;
            rc = atlas_instance_final_claims(i_hb, f, 64, &nf);
            Point p_hb(ln), p_bo(lv + ln);
            for (size_t q = 0; q < ln; q++) p_hb[q] = rs[mr - 1 - q];
            if (!rc) rc = O.append_virtual(Tr, gr::oid(gr::virt(gr::VP_HammingWeight), gr::SC_RamHammingBooleanity), p_hb, *reinterpret_cast<H::Fr*>(&f[0]));
            if (!rc) rc = atlas_instance_final_claims(i_bool, f, 64, &nf);
            for (size_t q = 0; q < lv; q++) p_bo[q] = rs[lv - 1 - q];
            for (size_t q = 0; q < ln; q++) p_bo[lv + q] = rs[mr - 1 - q];
            if (!rc) rc = O.append_sparse(Tr, gr::CP_GatherRa, 0, gr::SC_Booleanity, p_bo, *reinterpret_cast<H::Fr*>(&f[0]));
        }
        if (b) atlas_batched_free(b);
        for (atlas_instance_t i : {i_hb, i_bool}) if (i) atlas_instance_free(i);
        if (rc) return rc;
        // stage 3: HammingWeightSumcheckProver, d = 1, gamma_powers = [1]: sum_k ra(k, r_cycle) = 1
        rc = atlas_hamming_weight_new(Gh.data(), 1, lv, (const atlas_fr_t*)&one, &i_hw);
        std::vector<H::Fr> rs3, fin3;
        if (!rc) rc = run_single(i_hw, one, gr::PT_RaHammingWeight, rs3, fin3);
        if (i_hw) atlas_instance_free(i_hw);
        if (rc) return rc;
        Point p_hw = reversed(rs3);
        p_hw.insert(p_hw.end(), r_index.begin(), r_index.end());
        return O.append_sparse(Tr, gr::CP_GatherRa, 0, gr::SC_HammingWeight, p_hw, fin3[0]);
    }

    // ---- SoftmaxLastAxis (ops/softmax_last_axis/mod.rs:177-262): the auxiliary vectors, then four BatchedSumcheck stages
    // one BatchedSumcheck::prove over the members already added to `b`: proof rows under `proof_type`, the challenges as field elements
    int run_batch(atlas_batched_t b, size_t stride, size_t rounds_cap, uint8_t proof_type, std::vector<H::Fr>& rs) {
        std::vector<atlas_fr_t> rows(rounds_cap * stride); std::vector<uint32_t> nco(rounds_cap); std::vector<atlas_u128_t> ch(rounds_cap);
        size_t mr = 0;
        int rc = atlas_batched_prove(b, &t, rows.data(), stride, nco.data(), ch.data(), &mr);
        if (rc) return rc;
        rs.resize(mr);
        for (size_t i = 0; i < mr; i++) rs[i] = ch_fr(ch[i]);
        Out O = out();
        return O.put_proof(rows, stride, nco, mr, proof_type);
    }
    // cache_openings of a PS-Shout / IdentityRC member of n = log_K + log_T rounds: its ra polynomial at (address challenges as drawn,
    // cycle challenges reversed) (identity_range_check.rs, ps_shout/mod.rs:150-158)
    int ra_opening(const Node& nd, atlas_instance_t inst, uint8_t vp, size_t log_K, size_t log_T, const std::vector<H::Fr>& rs, std::vector<atlas_fr_t>& ra_point, H::Fr& ra_claim) {
        const size_t n = log_K + log_T, o = rs.size() - n;
        atlas_fr_t f[64]; size_t nf = 0;
        int rc = atlas_instance_final_claims(inst, f, 64, &nf);
        if (rc) return rc;
        std::memcpy(&ra_claim, &f[0], 32);
        Point pt(n);
        for (size_t q = 0; q < n; q++) pt[q] = q < log_K ? rs[o + q] : rs[o + log_K + (n - 1 - q)];
        ra_point.resize(n); std::memcpy(ra_point.data(), pt.data(), n * 32);
        return append_advice(nd, vp, pt, ra_claim);
    }
    int op_softmax(const Node& nd) {
        const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T), LS = gr::MODEL_SCALE;
        const gr::Opening& R = red(nd);
        NodeWitness& W = G.wit[nd.idx];
        SoftmaxWitness& Sm = *W.softmax;
        const size_t F = Sm.F, lf = gr::log2u(F), ln = gr::log2u(Sm.N);
        const H::Fr S_fr = H::from_u64((uint64_t)1 << LS);
        const size_t phases = identity_rc_phases(LS);                        // (node_flow.hip.h: the prover's own cut of the address rounds)
        const ExpLut* L = nullptr;
        int rc = atlas_rt_exp_lut(&L);
        if (rc) return rc;
        mark(nullptr);
        // send_auxiliary_vectors (:392-413): exp_sum_q[k], max_k[k], argmax_k[k] as F::from_u32(v as u32), at the empty point
        std::vector<int32_t> aux(3 * F);
        {
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            HIP_TRY(hipMemcpyAsync(aux.data(), Sm.exp_sum.p, F * 4, hipMemcpyDeviceToHost, rt().stream));
            HIP_TRY(hipMemcpyAsync(aux.data() + F, Sm.max_k.p, F * 4, hipMemcpyDeviceToHost, rt().stream));
            HIP_TRY(hipMemcpyAsync(aux.data() + 2 * F, Sm.argmax_k.p, F * 4, hipMemcpyDeviceToHost, rt().stream));
            HIP_TRY(hipStreamSynchronize(rt().stream));
        }
        {
            Out O = out();
            const Point empty;
            for (size_t k = 0; k < F && !rc; k++) {
                const uint8_t vps[3] = {gr::VP_SoftmaxSumOutput, gr::VP_SoftmaxMaxOutput, gr::VP_SoftmaxMaxIndex};
                for (int q = 0; q < 3 && !rc; q++)
                    rc = O.append_virtual(Tr, gr::node_exec(gr::virt(vps[q], nd.idx, k), nd.idx), empty, H::from_u64((uint64_t)(uint32_t)aux[q * F + k]));
            }
            if (rc) return rc;
        }
        auto wrap = [&](const DevBuf& b, size_t len, atlas_poly_t* p) { return atlas_poly_wrap_device_i32(b.as<int32_t>(), len, p); };
        auto eval1 = [&](const DevBuf& b, size_t len, const Point& pt, H::Fr* o) { const int32_t* tp = b.as<int32_t>(); return eval_i32(&tp, 1, len, pt, o); };
        const Point& r0 = R.point;
        const Point r_lead(r0.begin(), r0.begin() + lf);
        H::Fr exp_sum_claim, R_claim;
        rc = eval1(Sm.exp_sum, F, r_lead, &exp_sum_claim);
        if (!rc) rc = append_advice(nd, gr::VP_SoftmaxExpSum, r_lead, exp_sum_claim);                       // cache_exp_sum
        if (!rc) rc = eval1(Sm.R, T, r0, &R_claim);
        if (!rc) rc = append_advice(nd, gr::VP_SoftmaxRecipMultRemainder, r0, R_claim);                     // cache_R
        if (rc) return rc;

        mark("softmax: aux vectors + claims");
        std::vector<H::Fr> rs;
        std::vector<atlas_fr_t> Rra_point, Era_point, Cra_point;
        H::Fr Rra_claim, Era_claim, Cra_claim, exp_q_claim;
        Point r1, r2;
        {   // ---- stage 1: RecipMult, ExpSum, IdentityRC of R (build_stage1_instances :450-495)
            atlas_poly_t p_expq = nullptr, p_inv = nullptr;
            atlas_instance_t i_recip = nullptr, i_sum = nullptr, i_rc = nullptr;
            atlas_batched_t b = nullptr;
            rc = wrap(Sm.exp_q, T, &p_expq);
            if (!rc) rc = wrap(Sm.inv_sum, F, &p_inv);
            if (!rc) rc = atlas_softmax_instance_new(ATLAS_SM_RECIP_MULT, p_expq, p_inv, lf, ln, (const atlas_fr_t*)r0.data(), &i_recip);
            if (!rc) rc = atlas_softmax_instance_new(ATLAS_SM_EXP_SUM, p_expq, nullptr, lf, ln, (const atlas_fr_t*)r_lead.data(), &i_sum);
            if (!rc) rc = atlas_identity_range_check_new(Sm.idx_R.as<uint64_t>(), log_T, LS, phases, (const atlas_fr_t*)r0.data(), &i_rc);
            const H::Fr c_recip = H::add(H::mul(R.claim, S_fr), R_claim);                                    // RecipMultParams::input_claim
            if (!rc) rc = atlas_batched_new(&b);
            if (!rc) rc = atlas_batched_add_instance(b, i_recip, (const atlas_fr_t*)&c_recip);
            if (!rc) rc = atlas_batched_add_instance(b, i_sum, (const atlas_fr_t*)&exp_sum_claim);
            if (!rc) rc = atlas_batched_add_instance(b, i_rc, (const atlas_fr_t*)&R_claim);
            mark("softmax: stage 1 instances");
            if (!rc) rc = run_batch(b, 8, LS + log_T, gr::PT_SoftmaxStage1, rs);
            mark("softmax: stage 1 sumcheck");
            if (!rc) {
                r1.assign(rs.rbegin(), rs.rbegin() + log_T);                                                 // LITTLE_ENDIAN challenges of the last log_T rounds -> BIG_ENDIAN
                atlas_fr_t f[64]; size_t nf = 0;
                rc = atlas_instance_final_claims(i_recip, f, 64, &nf);
                std::memcpy(&exp_q_claim, &f[0], 32);
                if (!rc) rc = append_advice(nd, gr::VP_SoftmaxExpQ, r1, exp_q_claim);                        // RecipMultProver::cache_openings
                if (!rc) rc = atlas_instance_final_claims(i_sum, f, 64, &nf);
                if (!rc) rc = append_advice(nd, gr::VP_SoftmaxExpQ, r1, *reinterpret_cast<H::Fr*>(&f[0]));   // ExpSumProver::cache_openings (the same opening again)
                if (!rc) rc = ra_opening(nd, i_rc, gr::VP_SoftmaxRemainderRa, LS, log_T, rs, Rra_point, Rra_claim);
            }
            if (b) atlas_batched_free(b);
            for (atlas_instance_t i : {i_recip, i_sum, i_rc}) if (i) atlas_instance_free(i);
            for (atlas_poly_t p : {p_expq, p_inv}) if (p) atlas_poly_free(p);
            if (rc) return rc;
        }
        H::Fr exp_hi_claim, exp_lo_claim;
        {   // ---- stage 2: Mult, MaxIndicator, IdentityRC of r_exp, the one-hot checks of R (build_stage2_instances :521-579)
            H::Fr r_exp_claim, max_k_eval;
            rc = eval1(Sm.r_exp, T, r1, &r_exp_claim);
            if (!rc) rc = append_advice(nd, gr::VP_SoftmaxExpRemainder, r1, r_exp_claim);                    // cache_r_exp
            const Point r1_k(r1.begin(), r1.begin() + lf);
            if (!rc) rc = eval1(Sm.max_k, F, r1_k, &max_k_eval);
            atlas_poly_t p_hi = nullptr, p_lo = nullptr, p_x = nullptr, p_e = nullptr;
            atlas_instance_t i_mult = nullptr, i_max = nullptr, i_rc = nullptr;
            atlas_batched_t b = nullptr;
            std::vector<atlas_instance_t> oh;
            std::vector<OneHotFamily> fams(1);
            if (!rc) rc = wrap(Sm.exp_hi, T, &p_hi);
            if (!rc) rc = wrap(Sm.exp_lo, T, &p_lo);
            if (!rc) rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(G.tensor(nd.inputs[0])), T, &p_x);
            if (!rc) rc = wrap(Sm.e, T, &p_e);
            if (!rc) { atlas_poly_t ops[2] = {p_hi, p_lo}; rc = atlas_elementwise_new(ATLAS_EW_MUL, ops, 2, (const atlas_fr_t*)r1.data(), log_T, nullptr, 0, &i_mult); }
            if (!rc) rc = atlas_softmax_instance_new(ATLAS_SM_MAX_INDICATOR, p_x, p_e, lf, ln, (const atlas_fr_t*)r1_k.data(), &i_max);
            if (!rc) rc = atlas_identity_range_check_new(Sm.idx_rexp.as<uint64_t>(), log_T, LS, phases, (const atlas_fr_t*)r1.data(), &i_rc);
            const H::Fr c_mult = H::add(H::mul(exp_q_claim, S_fr), r_exp_claim);                             // MultParams::input_claim
            if (!rc) rc = atlas_batched_new(&b);
            if (!rc) rc = atlas_batched_add_instance(b, i_mult, (const atlas_fr_t*)&c_mult);
            if (!rc) rc = atlas_batched_add_instance(b, i_max, (const atlas_fr_t*)&max_k_eval);
            if (!rc) rc = atlas_batched_add_instance(b, i_rc, (const atlas_fr_t*)&r_exp_claim);
            fams[0].d_lookups = Sm.idx_R.as<uint64_t>(); fams[0].log_K = LS; fams[0].r_cycle = (const atlas_fr_t*)r0.data();       // SoftmaxRaEncoding::remainder
            fams[0].ra_point = Rra_point; fams[0].ra_claim = Rra_claim; fams[0].rad_cp = gr::CP_SoftmaxRemainderRaD;
            if (!rc) rc = onehot_families_build(fams, log_T, &t, b, oh, nullptr);
            mark("softmax: stage 2 instances");
            if (!rc) rc = run_batch(b, 8, LS + log_T, gr::PT_SoftmaxStage2, rs);
            mark("softmax: stage 2 sumcheck");
            if (!rc) {
                r2.assign(rs.rbegin(), rs.rbegin() + log_T);
                atlas_fr_t f[64]; size_t nf = 0;
                rc = atlas_instance_final_claims(i_mult, f, 64, &nf);
                std::memcpy(&exp_hi_claim, &f[0], 32); std::memcpy(&exp_lo_claim, &f[1], 32);
                if (!rc) rc = append_advice(nd, gr::VP_SoftmaxExpHi, r2, exp_hi_claim);                      // MultProver::cache_openings
                if (!rc) rc = append_advice(nd, gr::VP_SoftmaxExpLo, r2, exp_lo_claim);
                if (!rc) rc = atlas_instance_final_claims(i_max, f, 64, &nf);
                if (!rc) rc = append_nodeio(nd, 0, r2, *reinterpret_cast<H::Fr*>(&f[0]));                    // MaxIndicatorProver::cache_openings: X(r2)
                if (!rc) rc = ra_opening(nd, i_rc, gr::VP_SoftmaxExpRemainderRa, LS, log_T, rs, Era_point, Era_claim);
                Out O = out();
                if (!rc) rc = onehot_families_cache(fams, oh.data(), log_T, rs, &t, O);
            }
            if (b) atlas_batched_free(b);
            for (atlas_instance_t i : {i_mult, i_max, i_rc}) if (i) atlas_instance_free(i);
            for (atlas_instance_t i : oh) if (i) atlas_instance_free(i);
            for (atlas_poly_t p : {p_hi, p_lo, p_x, p_e}) if (p) atlas_poly_free(p);
            if (rc) return rc;
        }
        std::vector<atlas_fr_t> hi_point, lo_point;
        H::Fr hi_claim, lo_claim;
        {   // ---- stage 3: the two exp-digit Shout lookups, the significance clamp lookup, the one-hot checks of r_exp (:611-671)
            H::Fr zc[3];                                                                                     // z_hi(r2), z_lo(r2), z(r2)
            const int32_t* zs[3] = {Sm.z_hi.as<int32_t>(), Sm.z_lo.as<int32_t>(), Sm.z.as<int32_t>()};
            rc = eval_i32(zs, 3, T, r2, zc);
            if (!rc) rc = append_advice(nd, gr::VP_SoftmaxZHi, r2, zc[0]);                                   // cache_z_hi_lo
            if (!rc) rc = append_advice(nd, gr::VP_SoftmaxZLo, r2, zc[1]);
            if (rc) return rc;
            atlas_poly_t eq = nullptr, G_hi = nullptr, G_lo = nullptr;
            atlas_dot_prover_t d_hi = nullptr, d_lo = nullptr;
            atlas_instance_t i_hi = nullptr, i_lo = nullptr;               // the same two lookups as host-arithmetic instances (tables of <= 4096 entries)
            static const bool host_raf_off = getenv("ATLAS_SM_DEVICE_RAF") != nullptr;      // A-B
            const bool host_raf = !host_raf_off && Sm.lk_hi <= 12 && Sm.lk_lo <= 12;
            atlas_instance_t i_clamp = nullptr;
            atlas_batched_t b = nullptr;
            std::vector<atlas_instance_t> oh;
            std::vector<OneHotFamily> fams(1);
            rc = atlas_eq_evals((const atlas_fr_t*)r2.data(), log_T, nullptr, &eq);
            const H::Fr g_hi = H::tr_challenge_scalar(Tr);                                                   // ReadRafParams::new (shout.rs:112-130)
            if (!rc) rc = atlas_shout_read_raf_G(Sm.idx_zhi.as<uint64_t>(), T, Sm.lk_hi, eq, &G_hi);
            if (!rc && host_raf) { rc = atlas_rt_shout_read_raf_host_new(G_hi, L->hi.data(), Sm.lk_hi, (const atlas_fr_t*)&g_hi, &i_hi); if (G_hi) atlas_poly_free(G_hi); }
            else if (!rc) { rc = atlas_shout_read_raf_prover_new(G_hi, L->hi.data(), Sm.lk_hi, (const atlas_fr_t*)&g_hi, &d_hi); if (rc && G_hi) atlas_poly_free(G_hi); }
            const H::Fr g_lo = H::tr_challenge_scalar(Tr);
            if (!rc) rc = atlas_shout_read_raf_G(Sm.idx_zlo.as<uint64_t>(), T, Sm.lk_lo, eq, &G_lo);
            if (!rc && host_raf) { rc = atlas_rt_shout_read_raf_host_new(G_lo, L->lo.data(), Sm.lk_lo, (const atlas_fr_t*)&g_lo, &i_lo); if (G_lo) atlas_poly_free(G_lo); }
            else if (!rc) { rc = atlas_shout_read_raf_prover_new(G_lo, L->lo.data(), Sm.lk_lo, (const atlas_fr_t*)&g_lo, &d_lo); if (rc && G_lo) atlas_poly_free(G_lo); }
            if (eq) atlas_poly_free(eq);
            if (!rc) rc = append_advice(nd, gr::VP_SoftmaxClampWitness, r2, zc[2]);                          // append_raf_claims_prover (op_lookups/mod.rs:404-418)
            const H::Fr g_c = H::tr_challenge_scalar(Tr);                                                    // ps_read_raf_prover (unary.rs:112)
            const size_t bound = Sm.lk_hi + Sm.log2_base;                                                    // SOFTMAX_CLAMP_BOUND (common/src/consts/softmax.rs)
            if (!rc) rc = atlas_ps_shout_clamp_new(Sm.idx_z.as<uint64_t>(), log_T, 32, bound, 0, (const atlas_fr_t*)r2.data(), (const atlas_fr_t*)&g_c, &i_clamp);
            const H::Fr c_hi = H::add(exp_hi_claim, H::mul(g_hi, zc[0])), c_lo = H::add(exp_lo_claim, H::mul(g_lo, zc[1]));      // rv_claim + gamma raf_claim
            const H::Fr rv = H::add(H::mul(zc[0], H::from_u64((uint64_t)1 << Sm.log2_base)), zc[1]);         // significance_clamp.rs:61-69
            const H::Fr c_clamp = H::add(rv, H::mul(g_c, zc[2]));
            if (!rc) rc = atlas_batched_new(&b);
            if (!rc) rc = host_raf ? atlas_batched_add_instance(b, i_hi, (const atlas_fr_t*)&c_hi) : atlas_batched_add_dot(b, d_hi, (const atlas_fr_t*)&c_hi);
            if (!rc) rc = host_raf ? atlas_batched_add_instance(b, i_lo, (const atlas_fr_t*)&c_lo) : atlas_batched_add_dot(b, d_lo, (const atlas_fr_t*)&c_lo);
            if (!rc) rc = atlas_batched_add_instance(b, i_clamp, (const atlas_fr_t*)&c_clamp);
            fams[0].d_lookups = Sm.idx_rexp.as<uint64_t>(); fams[0].log_K = LS; fams[0].r_cycle = (const atlas_fr_t*)r1.data();     // SoftmaxRaEncoding::exp_remainder
            fams[0].ra_point = Era_point; fams[0].ra_claim = Era_claim; fams[0].rad_cp = gr::CP_SoftmaxExpRemainderRaD;
            if (!rc) rc = onehot_families_build(fams, log_T, &t, b, oh, nullptr);
            mark("softmax: stage 3 instances");
            if (!rc) rc = run_batch(b, 8, 32 + log_T, gr::PT_SoftmaxStage3, rs);
            mark("softmax: stage 3 sumcheck");
            if (!rc) {
                const size_t mr = rs.size();
                auto shout_open = [&](atlas_dot_prover_t dp, atlas_instance_t ip, size_t lk, uint8_t vp, std::vector<atlas_fr_t>& pt_out, H::Fr& claim) {     // ReadRafProver::cache_openings: [challenges | r]
                    atlas_fr_t f[64]; size_t nf = 0;
                    int rc2 = ip ? atlas_instance_final_claims(ip, f, 64, &nf) : atlas_dot_final_claims(dp, f);
                    if (rc2) return rc2;
                    std::memcpy(&claim, &f[0], 32);
                    Point pt(rs.begin() + (mr - lk), rs.end());
                    pt.insert(pt.end(), r2.begin(), r2.end());
                    pt_out.resize(pt.size()); std::memcpy(pt_out.data(), pt.data(), pt.size() * 32);
                    return append_advice(nd, vp, pt, claim);
                };
                rc = shout_open(d_hi, i_hi, Sm.lk_hi, gr::VP_SoftmaxZHiRa, hi_point, hi_claim);
                if (!rc) rc = shout_open(d_lo, i_lo, Sm.lk_lo, gr::VP_SoftmaxZLoRa, lo_point, lo_claim);
                if (!rc) rc = ra_opening(nd, i_clamp, gr::VP_SoftmaxClampRa, 32, log_T, rs, Cra_point, Cra_claim);
                Out O = out();
                if (!rc) rc = onehot_families_cache(fams, oh.data(), log_T, rs, &t, O);
            }
            if (b) atlas_batched_free(b);
            if (d_hi) atlas_dot_prover_free(d_hi);
            if (d_lo) atlas_dot_prover_free(d_lo);
            for (atlas_instance_t i : {i_hi, i_lo}) if (i) atlas_instance_free(i);
            if (i_clamp) atlas_instance_free(i_clamp);
            for (atlas_instance_t i : oh) if (i) atlas_instance_free(i);
            if (rc) return rc;
        }
        // ---- stage 4: the one-hot checks of z_hi, z_lo and the clamp lookup in one BatchedSumcheck (build_stage4_instances :686-730)
        std::vector<OneHotFamily> fams(3);
        fams[0].d_lookups = Sm.idx_zhi.as<uint64_t>(); fams[0].log_K = Sm.lk_hi; fams[0].ra_point = hi_point; fams[0].ra_claim = hi_claim; fams[0].rad_cp = gr::CP_SoftmaxZHiRaD;
        fams[1].d_lookups = Sm.idx_zlo.as<uint64_t>(); fams[1].log_K = Sm.lk_lo; fams[1].ra_point = lo_point; fams[1].ra_claim = lo_claim; fams[1].rad_cp = gr::CP_SoftmaxZLoRaD;
        fams[2].d_lookups = Sm.idx_z.as<uint64_t>(); fams[2].log_K = 32; fams[2].ra_point = Cra_point; fams[2].ra_claim = Cra_claim; fams[2].rad_cp = gr::CP_SoftmaxClampRaD;
        for (auto& f : fams) f.r_cycle = (const atlas_fr_t*)r2.data();       // the points of SoftmaxExpHi / SoftmaxExpLo / SoftmaxZHi: all r2
        Out O = out();
        mark("softmax: stage 3 openings");
        rc = prove_onehot_checks_multi(fams, log_T, &t, O, gr::PT_SoftmaxStage4);
        mark("softmax: stage 4");
        return rc;
    }

    // Sin / Cos by neural teleportation (ops/sin.rs:56-186, cos.rs; ReductionFlow::Custom): the Euclidean division by the period modulus at a
    // FRESH point (NeuralTeleport), the downscaling right-shift lookup batched with the table read-raf (Execution), the downscale lookup's
    // one-hot checks (TrigDownscaleRaChecks), the committed quotient, the node's eval reduction, then prove_range_and_onehot
    // (neural_teleport/range_and_onehot.rs:68-140): the remainder's range check batched with the table lookup's one-hot checks
    // (RaOneHotChecks), and the range check's own one-hot checks (RaHammingWeight)
    int op_trig(const Node& nd) {
        const size_t T = gr::padded_len(nd.dims), log_T = gr::log2u(T), LK = gr::TRIG_TABLE_VARS, K = (size_t)1 << LK;
        NodeWitness& W = G.wit[nd.idx];
        const uint8_t vp_ra = nd.op == ATLAS_OP_SIN ? gr::VP_SinRa : gr::VP_CosRa, cp_rad = nd.op == ATLAS_OP_SIN ? gr::CP_SinRaD : gr::CP_CosRaD;
        const H::Fr tau = fr_from_i64_host(gr::TRIG_PERIOD_MODULUS);
        Out O = out();
        // ---- 1a: TeleportDivisionProver at a point from the transcript
        const Point r = challenge_point(log_T);
        atlas_poly_t dops[3] = {nullptr, nullptr, nullptr};
        int rc = ATLAS_OK;
        std::vector<H::Fr> rs, fin;
        if (log_T == 0) {                                                     // one element: zero rounds, the three single values
            rc = zero_rounds(H::zero(), gr::PT_NeuralTeleport);
            fin.resize(3);
            if (!rc) rc = scalar_of(G.tensor(nd.inputs[0]), &fin[0]);
            if (!rc) rc = scalar_of(W.rem2.as<int32_t>(), &fin[1]);
            if (!rc) rc = scalar_of(W.rem.as<int32_t>(), &fin[2]);
        } else {
            rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(G.tensor(nd.inputs[0])), T, &dops[0]);
            if (!rc) rc = atlas_poly_wrap_device_i32(W.rem2.as<int32_t>(), T, &dops[1]);
            if (!rc) rc = atlas_poly_wrap_device_i32(W.rem.as<int32_t>(), T, &dops[2]);
            atlas_instance_t i_div = nullptr;
            if (!rc) rc = atlas_elementwise_new(ATLAS_EW_TELEPORT_DIV, dops, 3, (const atlas_fr_t*)r.data(), log_T, (const atlas_fr_t*)&tau, 1, &i_div);
            for (atlas_poly_t p : dops) if (p) atlas_poly_free(p);
            if (!rc) rc = run_single(i_div, H::zero(), gr::PT_NeuralTeleport, rs, fin);
            if (i_div) atlas_instance_free(i_div);
        }
        if (rc) return rc;
        const Point pt = reversed(rs);
        rc = append_nodeio(nd, 0, pt, fin[0]);
        if (!rc) rc = append_advice(nd, gr::VP_TeleportQuotient, pt, fin[1]);
        if (!rc) rc = append_advice(nd, gr::VP_TeleportRemainder, pt, fin[2]);
        if (rc) return rc;
        const H::Fr q_claim = fin[1], rem_claim = fin[2];
        // ---- 1b: cache_downscaled_prove, the right-shift read-raf (witness = the remainder, appended again as its raf claim), the table read-raf
        H::Fr ev[2];
        { const int32_t* tp[2] = {W.clamped.as<int32_t>(), G.tensor(nd.idx)}; rc = eval_i32(tp, 2, T, pt, ev); }
        const H::Fr down_claim = ev[0], out_claim = ev[1];
        if (!rc) rc = append_advice(nd, gr::VP_TrigDownscaled, pt, down_claim);
        if (!rc) rc = append_advice(nd, gr::VP_TeleportRemainder, pt, rem_claim);                              // append_raf_claims_prover
        if (rc) return rc;
        const H::Fr g_d = H::tr_challenge_scalar(Tr);
        atlas_instance_t i_dsc = nullptr, i_tab = nullptr;
        rc = atlas_ps_shout_rshift_new(W.lookups.as<uint64_t>(), log_T, 32, gr::TRIG_DOWNSCALE_BITS, (const atlas_fr_t*)pt.data(), (const atlas_fr_t*)&g_d, &i_dsc);
        const H::Fr g_s = H::tr_challenge_scalar(Tr);                                                           // SinParams::new
        if (!rc) rc = O.append_virtual(Tr, gr::node_exec(gr::virt(gr::VP_NodeOutput, nd.idx), nd.idx), pt, out_claim);     // SinProver::initialize: Target::Current
        {
            atlas_poly_t eq = nullptr, tops[3] = {nullptr, nullptr, nullptr};
            if (!rc) rc = atlas_eq_evals((const atlas_fr_t*)pt.data(), log_T, nullptr, &eq);
            if (!rc) rc = atlas_shout_read_raf_G(W.lookups2.as<uint64_t>(), T, LK, eq, &tops[0]);               // compute_ra_evals_direct
            if (eq) atlas_poly_free(eq);
            const int32_t* d_table = nullptr;
            if (!rc) rc = atlas_rt_trig_table(nd.op, &d_table, nullptr);
            if (!rc) rc = atlas_poly_wrap_device_i32(const_cast<int32_t*>(d_table), K, &tops[1]);
            std::vector<int32_t> ident(K);
            for (size_t i = 0; i < K; i++) ident[i] = (int32_t)i;
            if (!rc) rc = atlas_poly_upload_i32(ident.data(), K, &tops[2]);
            if (!rc) rc = atlas_elementwise_new(ATLAS_EW_GATHER, tops, 3, nullptr, LK, (const atlas_fr_t*)&g_s, 1, &i_tab);
            for (atlas_poly_t p : tops) if (p) atlas_poly_free(p);
        }
        std::vector<atlas_fr_t> Dra_point, tab_point;
        H::Fr Dra_claim, tab_claim;
        {
            atlas_batched_t b = nullptr;
            const H::Fr c_dsc = H::add(down_claim, H::mul(g_d, rem_claim)), c_tab = H::add(out_claim, H::mul(g_s, down_claim));
            if (!rc) rc = atlas_batched_new(&b);
            if (!rc) rc = atlas_batched_add_instance(b, i_dsc, (const atlas_fr_t*)&c_dsc);
            if (!rc) rc = atlas_batched_add_instance(b, i_tab, (const atlas_fr_t*)&c_tab);
            if (!rc) rc = run_batch(b, 8, 32 + log_T, gr::PT_Execution, rs);
            if (!rc) rc = ra_opening(nd, i_dsc, gr::VP_TrigDownscaleRa, 32, log_T, rs, Dra_point, Dra_claim);
            if (!rc) {
                atlas_fr_t f[64]; size_t nf = 0;
                rc = atlas_instance_final_claims(i_tab, f, 64, &nf);
                std::memcpy(&tab_claim, &f[0], 32);
                Point tp(rs.rbegin(), rs.rbegin() + LK);                                                        // the last LK challenges, reversed
                tp.insert(tp.end(), pt.begin(), pt.end());
                tab_point.resize(tp.size()); std::memcpy(tab_point.data(), tp.data(), tp.size() * 32);
                if (!rc) rc = append_advice(nd, vp_ra, tp, tab_claim);
            }
            if (b) atlas_batched_free(b);
        }
        for (atlas_instance_t i : {i_dsc, i_tab}) if (i) atlas_instance_free(i);
        // ---- 1c: the downscale lookup's one-hot checks
        if (!rc) rc = prove_onehot_checks(W.lookups.as<uint64_t>(), log_T, 32, (const atlas_fr_t*)pt.data(), Dra_point, Dra_claim, &t, O, gr::CP_TrigDownscaleRaD, gr::PT_TrigDownscaleRaChecks);
        // ---- the committed quotient at the quotient's point, the eval reduction
        if (!rc) rc = append_dense(nd, gr::CP_TeleportNodeQuotient, pt, q_claim);
        if (!rc) rc = eval_reduction(nd);
        if (rc) return rc;
        // ---- prove_range_and_onehot
        std::vector<atlas_fr_t> Rra_point; H::Fr Rra_claim;
        {
            atlas_instance_t rc_i = nullptr; H::Fr cl;
            atlas_batched_t b = nullptr;
            std::vector<atlas_instance_t> oh;
            std::vector<OneHotFamily> fams(1);
            rc = range_check_new(W.cidx.as<uint64_t>(), log_T, pt, rem_claim, tau, &rc_i, &cl);
            if (!rc) rc = atlas_batched_new(&b);
            if (!rc) rc = atlas_batched_add_instance(b, rc_i, (const atlas_fr_t*)&cl);
            fams[0].d_lookups = W.lookups2.as<uint64_t>(); fams[0].log_K = LK; fams[0].r_cycle = (const atlas_fr_t*)pt.data();
            fams[0].ra_point = tab_point; fams[0].ra_claim = tab_claim; fams[0].rad_cp = cp_rad;
            if (!rc) rc = onehot_families_build(fams, log_T, &t, b, oh, nullptr);
            if (!rc) rc = run_batch(b, 8, 64 + log_T, gr::PT_RaOneHotChecks, rs);
            if (!rc) rc = ra_opening(nd, rc_i, gr::VP_TeleportRangeCheckRa, 64, log_T, rs, Rra_point, Rra_claim);
            if (!rc) rc = onehot_families_cache(fams, oh.data(), log_T, rs, &t, O);
            if (b) atlas_batched_free(b);
            if (rc_i) atlas_instance_free(rc_i);
            for (atlas_instance_t i : oh) if (i) atlas_instance_free(i);
        }
        if (!rc) rc = prove_onehot_checks(W.cidx.as<uint64_t>(), log_T, 64, (const atlas_fr_t*)pt.data(), Rra_point, Rra_claim, &t, O, gr::CP_TeleportRangeCheckRaD, gr::PT_RaHammingWeight);
        return rc;
    }

    int prove_node(const Node& nd) {
        cur = nd.idx;
        mark(nullptr);
        rt().chan.long_mark();                                               // the node is the scope of its long-lived mail (channel.hpp)
        int rc = prove_node_flow(nd);
        if (!rc && !rt().chan.long_check()) rc = fail(ATLAS_ESTATE, "prove_graph: the long-lived mail ring wrapped inside one node");
        mark("(rest of the node)");
        return rc;
    }
    int prove_node_flow(const Node& nd) {
        // (the lookup operators, the gathers over ONE element / ONE index and a softmax of ONE row are composed — DESIGN 11.8b, 12.10; rows of ONE
        // element are refused when the graph is built (graph_exec.hip: 2 <= last axis): the four batched stages of ops/softmax_last_axis bind
        // (position, row) pairs and the instances of softmax.hip were never walked without position variables; the oracle composes [1, 1])
        if (nd.op == ATLAS_OP_SOFTMAX && gr::padded_len(nd.dims) == 1)
            return fail(ATLAS_EINVAL, "prove_graph: SoftmaxLastAxis over ONE element is not composed");
        if (nd.op == ATLAS_OP_DIV) return op_div(nd);                          // ReductionFlow::Custom
        if (nd.op == ATLAS_OP_RSQRT) return op_rsqrt(nd);
        if (nd.op == ATLAS_OP_SIN || nd.op == ATLAS_OP_COS) return op_trig(nd);
        int rc = eval_reduction(nd);                                           // ReductionFlow::Default
        if (rc) return rc;
        mark("eval reduction");
        const gr::Opening& R = red(nd);
        switch (nd.op) {
            case ATLAS_OP_INPUT: case ATLAS_OP_CONSTANT: return ATLAS_OK;    // the verifier evaluates the public tensor itself
            case ATLAS_OP_IDENTITY: return append_nodeio(nd, 0, R.point, R.claim);
            case ATLAS_OP_ADD: case ATLAS_OP_SUB: return op_addsub(nd);
            case ATLAS_OP_EINSUM: case ATLAS_OP_MUL: case ATLAS_OP_SQUARE: case ATLAS_OP_CUBE: return op_fused(nd);
            case ATLAS_OP_AND: return ew_sumcheck(nd, ATLAS_EW_MUL, 2, R.claim, gr::PT_Execution);       // impl_standard_sumcheck_proof_api!(And, MulParams, ..)
            case ATLAS_OP_IFF: return ew_sumcheck(nd, ATLAS_EW_IFF, 3, R.claim, gr::PT_Execution);
            case ATLAS_OP_RELU: case ATLAS_OP_CLAMP: return op_relu(nd);
            case ATLAS_OP_NEG: case ATLAS_OP_IS_NAN: return op_operand_at_point(nd);
            case ATLAS_OP_RESHAPE: return op_reshape(nd);
            case ATLAS_OP_MOVEAXIS: return op_moveaxis(nd);
            case ATLAS_OP_BROADCAST: return op_broadcast(nd);
            case ATLAS_OP_SUM: return op_sum(nd);
            case ATLAS_OP_SCALAR_CONST_DIV: return op_scalar_const_div(nd);
            case ATLAS_OP_SLICE: return op_slice(nd);
            case ATLAS_OP_CONCAT: return op_concat(nd);
            case ATLAS_OP_MEAN_OF_SQUARES: return op_mean_of_squares(nd);
            case ATLAS_OP_TANH: case ATLAS_OP_ERF: case ATLAS_OP_SIGMOID: return op_tanh(nd);      // prove_clamped_activation<Table>
            case ATLAS_OP_GATHER_LARGE: case ATLAS_OP_GATHER_SMALL: return op_gather(nd);
            case ATLAS_OP_SOFTMAX: return op_softmax(nd);
            default: return fail(ATLAS_EINVAL, "prove_graph: operator without a prover composition");
        }
    }

    // ---------------------------------------------------------------- prove_reduced_openings (prover.rs:141-176)
    struct Reduced {
        std::vector<atlas_fr_t> rows; std::vector<uint32_t> nco; std::vector<atlas_u128_t> ch; size_t rounds = 0;
        std::vector<atlas_fr_t> claims; std::vector<atlas_g1_affine_t> com, w; std::vector<atlas_fr_t> v;
        bool present = false;
    } ro;
    int reduced_openings(double* ms_sumcheck_and_open) {
        (void)ms_sumcheck_and_open;
        if (committed.empty()) return ATLAS_OK;
        std::vector<atlas_opening_t> ops;
        for (auto& kv : committed) {                                          // BTreeMap<CommittedPoly> order; the one-hot chunks stay on the device (lookups + shift)
            gr::Committed& c = *kv.second;
            if (!c.opened) return fail(ATLAS_ESTATE, "prove_graph: a committed polynomial was never opened");
            atlas_opening_t O; std::memset(&O, 0, sizeof(O));
            O.kind = c.kind; O.point = (const atlas_fr_t*)c.point.data(); std::memcpy(&O.claim, &c.claim, 32);
            if (c.kind == 1) {
                const size_t d = (c.log_K + c.lkc - 1) / c.lkc;
                O.d_lookups = c.d_lookups; O.chunk_shift = c.lkc * (d - 1 - c.chunk);                                      // OneHotParams::lookup_index_chunk
                O.log_K = c.lkc; O.log_T = c.log_T;
            } else { O.poly = c.dense; O.n = c.log_T; }
            ops.push_back(O);
        }
        size_t maxr = 0;
        for (auto& O : ops) { const size_t n = O.kind ? O.log_K + O.log_T : O.n; maxr = n > maxr ? n : maxr; }
        ro.rows.resize(maxr * 3); ro.nco.resize(maxr); ro.ch.resize(maxr); ro.claims.resize(ops.size());
        ro.com.resize(maxr ? maxr - 1 : 0); ro.w.resize(3); ro.v.resize(3 * maxr);
        int rc = atlas_rt_prove_reduced_openings(ops.data(), ops.size(), srs, &t, ro.rows.data(), ro.nco.data(), ro.ch.data(), &ro.rounds, ro.claims.data(),
                                                 ro.com.data(), ro.w.data(), ro.v.data(), sh && sh->world > 1 ? sh : nullptr);
        ro.present = rc == ATLAS_OK;
        return rc;
    }

    // ---------------------------------------------------------------- the ONNXProof container (proof_serialization.rs:200-224)
    static void put_u64(std::vector<uint8_t>& o, uint64_t v) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i))); }
    static void put_fr(std::vector<uint8_t>& o, const H::Fr& f) { uint8_t b[32]; atlas_fr_to_bytes((const atlas_fr_t*)&f, b); o.insert(o.end(), b, b + 32); }
    static void put_g1(std::vector<uint8_t>& o, const atlas_g1_affine_t& p) { uint8_t b[32]; atlas_g1_to_bytes_compressed(&p, b); o.insert(o.end(), b, b + 32); }
    static void put_opening_id(std::vector<uint8_t>& o, const OpeningId& id) {              // opening_proof.rs:1313-1333 (Committed = tag 0, Virtual = tag 1)
        o.push_back(id.poly.committed ? 0 : 1);
        o.push_back(id.poly.var);                                                            // canonical_serde_enum!: u8 variant index, then the usize fields
        const int ar = id.poly.committed ? gr::cp_arity(id.poly.var) : gr::vp_arity(id.poly.var);
        if (ar >= 1) put_u64(o, id.poly.a);
        if (ar >= 2) put_u64(o, id.poly.b);
        o.push_back(id.sc);
        if (id.sc == gr::SC_NodeExecution || id.sc == gr::SC_RLC) put_u64(o, id.sc_idx);
    }
    int serialize(std::vector<uint8_t>& o) {
        put_u64(o, openings.size());                                          // Claims: (OpeningId, claim) pairs, points dropped
        for (auto& kv : openings) { put_opening_id(o, kv.first); put_fr(o, kv.second.claim); }
        put_u64(o, proofs.size());                                            // BTreeMap<ProofId, SumcheckInstanceProof>
        for (auto& kv : proofs) { put_u64(o, kv.first.node); o.push_back(kv.first.type); o.insert(o.end(), kv.second.begin(), kv.second.end()); }
        put_u64(o, committed.size());                                         // Vec<HyperKZGCommitment>
        for (auto& kv : committed) put_g1(o, kv.second->commitment);
        put_u64(o, evalred.size());                                           // BTreeMap<usize, EvalReductionProof { h: UniPoly { coeffs } }>
        for (auto& kv : evalred) { put_u64(o, kv.first); put_u64(o, kv.second.size()); for (auto& c : kv.second) put_fr(o, c); }
        o.push_back(ro.present ? 1 : 0);                                      // Option<ReducedOpeningProof>
        if (ro.present) {
            size_t l = 0;
            int rc = atlas_sumcheck_proof_serialize(ro.rows.data(), 3, ro.nco.data(), ro.rounds, nullptr, 0, &l);
            if (rc) return rc;
            const size_t at = o.size(); o.resize(at + l);
            rc = atlas_sumcheck_proof_serialize(ro.rows.data(), 3, ro.nco.data(), ro.rounds, o.data() + at, l, &l);
            if (rc) return rc;
            put_u64(o, ro.claims.size());
            for (auto& c : ro.claims) put_fr(o, *reinterpret_cast<const H::Fr*>(&c));
            rc = atlas_hyperkzg_proof_serialize(ro.com.data(), ro.com.size(), ro.w.data(), ro.v.data(), ro.rounds, nullptr, 0, &l);
            if (rc) return rc;
            const size_t at2 = o.size(); o.resize(at2 + l);
            rc = atlas_hyperkzg_proof_serialize(ro.com.data(), ro.com.size(), ro.w.data(), ro.v.data(), ro.rounds, o.data() + at2, l, &l);
            if (rc) return rc;
        }
        return ATLAS_OK;
    }
};

}  // namespace

static int prove_graph_impl(atlas_graph_t G, atlas_srs_t srs, atlas_shard_group_t sh, const int32_t* const* inputs, size_t n_inputs, uint8_t* proof, size_t cap,
                            size_t* proof_len, atlas_transcript_t* final_transcript, atlas_graph_timing_t* timing) {
    NEED_INIT();
    if (!G || !srs || (!inputs && n_inputs) || !proof_len) return fail(ATLAS_EINVAL, "prove_graph: null argument");
    if (G->outputs.empty()) return fail(ATLAS_EINVAL, "prove_graph: no output node marked");
    if (int vrc = atlas_rt_validate_graph(*G)) return vrc;
    atlas_rt::HostSampler::Scope host_samples;                                // ATLAS_HOST_SAMPLE=<file>: backtraces of this thread every 50 us (diagnosis)
    auto now = [] { atlas_sync(); return std::chrono::steady_clock::now(); };
    // ATLAS_MEM_TRACE=1: what this runtime holds on its device at the stage boundaries, and the peak inside each stage (devpool.hpp accounting)
    static const bool mem_trace = getenv("ATLAS_MEM_TRACE") != nullptr;
    auto mem_mark = [&](const char* what) {
        if (!mem_trace) return;
        atlas_rt::DevPool& DP = atlas_rt::dev_pool();
        std::lock_guard<std::mutex> lk(DP.mu);
        fprintf(stderr, "[atlas mem] %-28s in use %7.3f GB, peak since the last mark %7.3f GB, cached %7.3f GB\n", what, DP.in_use / 1073741824.0, DP.peak / 1073741824.0, DP.cached / 1073741824.0);
        DP.peak = DP.in_use;
    };
    mem_mark("before the proof");
    const auto t0 = now();
    if (atlas_rt::Prof::on()) atlas_rt::Prof::get().reset();
    int rc = atlas_graph_trace(G, inputs, n_inputs);                          // pp.model().trace(inputs)
    if (rc) return rc;
    mem_mark("trace"); const auto t1 = now();
    if (atlas_rt::Prof::on()) { atlas_rt::Prof::get().dump(stderr, "Model::trace"); atlas_rt::Prof::get().reset(); }
    Prover P(*G, srs);
    P.sh = sh;
    rc = atlas_transcript_new(&P.t, (const uint8_t*)"ONNXProof", 9);
    if (rc) return rc;
    {   // append_inputs_to_transcript (onnx_proof/mod.rs:90-122)
        H::Transcript& Tr = P.Tr;
        const std::vector<size_t> in_nodes = G->input_nodes();
        H::tr_append_message(Tr, "model_inputs");
        H::tr_append_u64(Tr, in_nodes.size());
        H::tr_append_u64(Tr, in_nodes.size());
        for (size_t i = 0; i < in_nodes.size(); i++) {
            const Node& nd = G->nodes.at(in_nodes[i]);
            H::tr_append_u64(Tr, nd.idx);
            H::tr_append_u64(Tr, nd.dims.size());
            for (size_t d : nd.dims) H::tr_append_u64(Tr, d);
            H::tr_append_bytes(Tr, (const uint8_t*)inputs[i], nd.numel() * 4);                  // i32 little-endian
        }
    }
    rc = P.collect_committed();
    if (!rc) rc = P.commit_all();
    mem_mark("witness commitments"); const auto t2 = now();
    if (!rc) rc = P.output_claim();
    if (atlas_rt::Prof::on()) atlas_rt::Prof::get().reset();
    const bool gtrace = getenv("ATLAS_GRAPH_TRACE") != nullptr;             // per-operator wall clock of the node loop on stderr
    std::map<int, std::pair<double, size_t>> per_op;
    struct LongScope { atlas_rt::Channel& C; bool was; explicit LongScope(atlas_rt::Channel& c) : C(c), was(c.long_scoped) { C.long_scoped = true; } ~LongScope() { C.long_scoped = was; } } long_scope(rt().chan);
    for (auto it = G->nodes.rbegin(); it != G->nodes.rend() && !rc; ++it) {
        const auto tn0 = gtrace ? now() : t2;
        rc = P.prove_node(it->second);
        if (gtrace) { auto& e = per_op[it->second.op]; e.first += ms_between(tn0, now()); e.second++; }
    }
    if (gtrace) fprintf(stderr, "[atlas graph] device pool: %zu requests reached hipMalloc so far in this process (%.3f ms), %zu passed over for their stream\n",
                        atlas_rt::dev_pool().n_real, atlas_rt::dev_pool().real_ms, atlas_rt::dev_pool().cross_stream);
    if (atlas_rt::Prof::on()) { atlas_rt::Prof::get().dump(stderr, "node loop (iop)"); atlas_rt::Prof::get().reset(); }
    if (gtrace) for (auto& kv : per_op) fprintf(stderr, "[atlas graph] op %2d  x%-4zu %9.3f ms  (%.3f ms each)\n", kv.first, kv.second.second, kv.second.first, kv.second.first / kv.second.second);
    mem_mark("node loop (iop)"); const auto t3 = now();
    rt().chan.long_mark();                                                   // the opening reduction is a scope of its own
    if (!rc) rc = P.reduced_openings(nullptr);
    if (!rc && !rt().chan.long_check()) rc = fail(ATLAS_ESTATE, "prove_graph: the long-lived mail ring wrapped inside the opening reduction");
    mem_mark("reduction + HyperKZG::open"); const auto t4 = now();
    if (atlas_rt::Prof::on()) { atlas_rt::Prof::get().dump(stderr, "prove_reduced_openings"); atlas_rt::Prof::get().reset(); }
    if (rc) return rc;
    std::vector<uint8_t> bytes;
    rc = P.serialize(bytes);
    if (rc) return rc;
    *proof_len = bytes.size();
    if (proof) {
        if (cap < bytes.size()) return fail(ATLAS_EINVAL, "prove_graph: proof buffer too small (the needed size is in *proof_len)");
        std::memcpy(proof, bytes.data(), bytes.size());
    }
    if (final_transcript) *final_transcript = P.t;
    if (timing) {
        timing->trace_ms = ms_between(t0, t1); timing->commit_ms = ms_between(t1, t2); timing->iop_ms = ms_between(t2, t3);
        timing->hyperkzg_ms = atlas_rt_last_hyperkzg_ms(); timing->reduction_ms = ms_between(t3, t4) - timing->hyperkzg_ms; timing->total_ms = ms_between(t0, now());
        timing->n_nodes = G->nodes.size(); timing->n_committed = P.committed.size(); timing->n_sumchecks = P.proofs.size() + (P.ro.present ? 1 : 0);
    }
    return ATLAS_OK;
}

extern "C" int atlas_prove_graph(atlas_graph_t G, atlas_srs_t srs, const int32_t* const* inputs, size_t n_inputs, uint8_t* proof, size_t cap, size_t* proof_len,
                                 atlas_transcript_t* final_transcript, atlas_graph_timing_t* timing) {
    return prove_graph_impl(G, srs, nullptr, inputs, n_inputs, proof, cap, proof_len, final_transcript, timing);
}
// ONNXProof::prove with the ranks of a shard group (one process per GPU, BASELINE config 4): every rank holds the model, the inputs and the SRS and
// calls this with the same arguments.  The trace and the IOP (a serial Fiat-Shamir chain of latency-bound sumchecks) run on every rank; the stages
// that are sums over independent terms are split: commit_witness_polynomials by polynomial range (prover.rs:71-87), the four commitment groups of
// HyperKZG::open over the joint polynomial by point range (prover.rs:141-176, hyperkzg/mod.rs:400-447).  Partial results cross the shared-memory
// board, every rank runs the same transcript and returns the SAME proof bytes as atlas_prove_graph on one GPU.
extern "C" int atlas_prove_graph_sharded(atlas_graph_t G, atlas_srs_t srs, atlas_shard_group_t grp, const int32_t* const* inputs, size_t n_inputs, uint8_t* proof,
                                         size_t cap, size_t* proof_len, atlas_transcript_t* final_transcript, atlas_graph_timing_t* timing) {
    if (!grp) return fail(ATLAS_EINVAL, "prove_graph_sharded: null group");
    // the channel's waits scaled with the world for the length of the call: N ranks' processes may share ONE device (the test box), where a
    // launch waits for its challenge through the other ranks' time slices
    const double dev_wait = rt().chan.device_wait_s, host_wait = rt().chan.host_wait_s;
    if (grp->world > 1 && rt().ready) {
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        rt().chan.host_wait_s = host_wait * grp->world;
        (void)hipStreamSynchronize(rt().stream);
        (void)rt().chan.set_device_timeout(dev_wait * grp->world, rt().stream);
    }
    // Do ranks share a device?  (The test box: N processes on ONE GPU.)  Every rank's lane streams, gates and resident launches then compete
    // for the same hardware queues, and from three ranks on a lane's polling launch can sit in front of the launch it waits for — the 12-layer
    // graph at world 4 on one GPU stalled until the channel's timeout with lane streams and takes 15 s without.  The ranks exchange their
    // devices' PCI addresses; a rank whose device serves three or more of them keeps the lanes of its batches on the library stream for the call.
    bool lanes_were = rt().no_lane_streams, shared_checked = false;
    const bool shared_was = rt().device_shared;
    if (grp->world > 1 && rt().ready) {
        int dom = 0, bus = 0, dev = 0;
        (void)hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, rt().device);
        (void)hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, rt().device);
        (void)hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, rt().device);
        char host[64] = {0};
        (void)gethostname(host, sizeof(host) - 1);
        uint64_t hh = 1469598103934665603ull;
        for (const char* c = host; *c; c++) hh = (hh ^ (uint8_t)*c) * 1099511628211ull;
        const uint64_t mine[2] = {((uint64_t)(uint32_t)dom << 32) | ((uint64_t)(uint32_t)bus << 16) | (uint32_t)dev, hh};
        std::vector<uint64_t> all(2 * (size_t)grp->world);
        if (!grp->allgather(mine, sizeof(mine), all.data())) return fail(ATLAS_ENODEV, "prove_graph_sharded: a rank did not answer (device addresses)");
        int sharing = 0;
        for (int r = 0; r < grp->world; r++) sharing += all[2 * r] == mine[0] && all[2 * r + 1] == mine[1];
        shared_checked = true;
        if (sharing >= 3) rt().no_lane_streams = true;
        if (sharing >= 2) rt().device_shared = true;
    }
    int rc = prove_graph_impl(G, srs, grp, inputs, n_inputs, proof, cap, proof_len, final_transcript, timing);
    if (shared_checked) { rt().no_lane_streams = lanes_were; rt().device_shared = shared_was; }
    if (grp->world > 1 && rt().ready) {
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        rt().chan.host_wait_s = host_wait;
        (void)hipStreamSynchronize(rt().stream);
        (void)rt().chan.set_device_timeout(dev_wait, rt().stream);
    }
    if (grp->world > 1) {
        // The failure handshake (shard_group.hpp): a rank whose part failed for a reason of its own — not because another rank said so —
        // marks the exchange the others are about to make, so that nobody waits for the board's timeout and the exchange numbers stay equal;
        // a proof that succeeded here ends with one status exchange, which is where a failure after the last data exchange is learnt.
        if (rc) { if (grp->remote_failed < 0) grp->fail_exchange((uint64_t)(uint32_t)rc); }
        else {
            uint64_t ok = 0;
            std::vector<uint64_t> all((size_t)grp->world);
            if (!grp->allgather(&ok, sizeof(ok), all.data()))
                rc = grp->remote_failed >= 0 ? fail((int)(uint32_t)grp->remote_code ? (int)(uint32_t)grp->remote_code : ATLAS_ENODEV, "prove_graph_sharded: another rank's proof failed")
                                             : fail(ATLAS_ENODEV, "prove_graph_sharded: a rank did not answer (final status)");
        }
    }
    return rc;
}
