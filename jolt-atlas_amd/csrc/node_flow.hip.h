// The operator flows shared by the single-node entry points (node_einsum.hip) and the graph prover (graph_prove.hip):
// witness kernels, Sumcheck::prove of one instance + its ra opening, ra_onehot_provers + BatchedSumcheck::prove, the
// fused-rescale skeleton (fused_rebase.rs:215-285).  What a flow appends to the ProverOpeningAccumulator goes to the
// transcript, to the flat claims list of the single-node API and — when a sink is installed — to the graph prover's
// accumulator under the reference's OpeningId.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/atlas_hip.h"
#include "field.hip.h"
#include "host_field.hpp"
#include "internal.hpp"
#include "graph.hpp"
#include "runtime.hpp"

// where the graph prover receives what a flow appends to the accumulator (nullptr for the single-node entry points)
struct FlowSink {
    virtual ~FlowSink() {}
    virtual void virt(const gr::OpeningId& id, const gr::Point& point, const atlas_host::Fr& claim) = 0;
    // append_sparse of ONE committed one-hot chunk polynomial CommittedPoly::<cp_var>(node, chunk) under SumcheckId `sc`
    virtual void sparse(uint8_t cp_var, size_t chunk, uint8_t sc, const gr::Point& point, const atlas_host::Fr& claim) = 0;
    virtual void proof(uint8_t proof_type, const uint8_t* bytes, size_t len) = 0;
    virtual void mark(const char*) {}                                         // ATLAS_GRAPH_TRACE=2: a stage of the node's flow ends here
};
// the identifiers of one lookup family: its virtual ra polynomial and its committed chunks
struct RaIds { uint8_t ra_vp, rad_cp; };

using namespace atlas;
namespace H = atlas_host;
using atlas_rt::fail;
using atlas_rt::rt;

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
    FlowSink* sink = nullptr; uint64_t node = 0;
    // ProverOpeningAccumulator::append_virtual: transcript, flat list, accumulator
    int append_virtual(H::Transcript& Tr, const gr::OpeningId& id, const gr::Point& point, const H::Fr& c) {
        H::tr_append_scalar(Tr, c);
        if (sink) sink->virt(id, point, c);
        return put_claim(c);
    }
    int append_sparse(H::Transcript& Tr, uint8_t cp_var, size_t chunk, uint8_t sc, const gr::Point& point, const H::Fr& c) {
        H::tr_append_scalar(Tr, c);
        if (sink) sink->sparse(cp_var, chunk, sc, point, c);
        return put_claim(c);
    }
    void mark(const char* what) { if (sink) sink->mark(what); }
    int put_claim(const H::Fr& c) {
        if (!claims) return ATLAS_OK;
        if (n_claims >= claims_cap) return fail(ATLAS_EINVAL, "prove_einsum_node: claims buffer too small");
        std::memcpy(&claims[n_claims++], &c, 32);
        return ATLAS_OK;
    }
    int put_proof(const std::vector<atlas_fr_t>& rows, size_t stride, const std::vector<uint32_t>& nco, size_t n_rounds, uint8_t proof_type = 0) {
        size_t l = 0;
        if (sink) {
            int rc = atlas_sumcheck_proof_serialize(rows.data(), stride, nco.data(), n_rounds, nullptr, 0, &l);
            if (rc) return rc;
            std::vector<uint8_t> buf(l);
            rc = atlas_sumcheck_proof_serialize(rows.data(), stride, nco.data(), n_rounds, buf.data(), buf.size(), &l);
            if (rc) return rc;
            sink->proof(proof_type, buf.data(), l);
            if (!proofs) return ATLAS_OK;
        }
        int rc = atlas_sumcheck_proof_serialize(rows.data(), stride, nco.data(), n_rounds, proofs + len, cap - len, &l);
        if (rc) return rc;
        lens[n_proofs++] = l; len += l;
        return ATLAS_OK;
    }
};

inline gr::Point to_point(const atlas_fr_t* r, size_t n) { gr::Point p(n); if (n) std::memcpy(p.data(), r, n * 32); return p; }
inline H::Fr ch_fr(const atlas_u128_t& c) { return H::challenge_to_fr(c.lo, c.hi, rt().challenge_mode); }

// Sumcheck::prove of one read-raf instance (PS-Shout / IdentityRC: log_K address rounds then log_T cycle rounds) + its
// single cache_openings claim: append_virtual(ra_vp(node), NodeExecution(node)) at normalize_opening_point = the address
// challenges as drawn, the cycle challenges reversed (ps_shout/mod.rs:150-158, identity_range_check.rs).
int prove_single(atlas_instance_t inst, const H::Fr& input_claim, atlas_transcript_t* t, Out& O, std::vector<atlas_u128_t>& ch, H::Fr* final0,
                 size_t log_K = 0, uint8_t ra_vp = 0, uint8_t proof_type = 0, std::vector<atlas_fr_t>* ra_point_out = nullptr) {
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
    gr::Point pt(n);
    for (size_t i = 0; i < n; i++) pt[i] = ch_fr(i < log_K ? ch[i] : ch[log_K + (n - 1 - i)]);
    if (ra_point_out) { ra_point_out->resize(n); std::memcpy(ra_point_out->data(), pt.data(), n * 32); }
    rc = O.append_virtual(*reinterpret_cast<H::Transcript*>(t), gr::node_exec(gr::virt(ra_vp, O.node), O.node), pt, *final0);
    if (!rc) rc = O.put_proof(rows, stride, nco, n, proof_type);
    return rc;
}

// ra_onehot_provers (shout.rs:399-466) + BatchedSumcheck::prove over [RaVirtual, HammingWeight, Booleanity] and their
// cache_openings (ra_virtual.rs:157-184, hamming_weight.rs:149-168, booleanity.rs:350-368): d claims each.  Several lookup
// families of one node (Rsqrt: its two range checks, ops/rsqrt.rs:558-583) share ONE batched sumcheck: the families draw their
// challenges in order, the instances are batched as [ra, hw, bool] per family, the openings cached in that order.
struct OneHotFamily {
    const uint64_t* d_lookups; size_t log_K;
    const atlas_fr_t* r_cycle;                   // the r_cycle_source opening point (log_T)
    std::vector<atlas_fr_t> ra_point; H::Fr ra_claim;
    uint8_t rad_cp;
    const std::vector<H::Fr>* G_pre = nullptr;    // compute_ra_evals(lookups, r_cycle) computed ahead (NodePre); null: computed here
};

// What the flows of ONE node share at its reduced opening point r (log_T variables), launched before the node's first wait so that the wait
// covers them: EqPolynomial::evals(r) on the device — u_evals of the clamp lookup and of the remainder range check, the weights of the
// one-hot G tables — and compute_ra_evals of the node's lookup families (shout.rs:550-598: they depend on r_cycle = r alone, not on
// anything the sumchecks in between draw).  Was: the table rebuilt four times per fused-rescale node and a synchronisation per G table.
struct NodePre {
    atlas_poly_t eq = nullptr;
    struct Fam { const uint64_t* lookups; size_t log_K; atlas_rt_ra_ticket* ticket; std::vector<H::Fr> G; bool have; };
    std::vector<Fam> fams;
    static bool on() { static const bool off = getenv("ATLAS_NO_NODE_PRE") != nullptr; return !off; }      // A-B
    // launches only (library stream)
    int begin(const atlas_fr_t* r, size_t log_T, std::initializer_list<std::pair<const uint64_t*, size_t>> lookups) {
        if (!on() || log_T == 0 || log_T > 16) return ATLAS_OK;
        int rc = atlas_eq_evals(r, log_T, nullptr, &eq);
        if (!rc) rc = atlas_rt_eval_event_record(eq);              // the node's evaluation waits for the table, not for what follows it on the stream
        for (auto& lk : lookups) {
            if (rc) break;
            fams.push_back(Fam{lk.first, lk.second, nullptr, {}, false});
            rc = atlas_rt_shout_ra_evals_launch(lk.first, (size_t)1 << log_T, lk.second, 4, eq, &fams.back().ticket);
        }
        return rc;
    }
    // after the caller has waited for the library stream (or wait = true).  With the node's evaluation on its own stream (atlas_rt_eval_event_record)
    // the caller's wait was NOT for the library stream: collect(false) then leaves the tables to G_for(), which waits when they are first asked for
    // (the one-hot checks, a whole sumcheck later).
    int collect(bool wait) {
        if (!wait && atlas_rt_eval_event_live(eq)) return ATLAS_OK;
        for (auto& F : fams) {
            if (F.have || !F.ticket) continue;
            int rc = atlas_rt_shout_ra_evals_finish(F.ticket, wait, F.G);
            if (rc) return rc;
            wait = false;
            F.have = true;
        }
        return ATLAS_OK;
    }
    const std::vector<H::Fr>* G_for(const uint64_t* lookups, size_t log_K) const {
        for (auto& F : fams) if (!F.have && F.ticket) { if (const_cast<NodePre*>(this)->collect_now()) return nullptr; break; }
        for (auto& F : fams) if (F.have && F.lookups == lookups && F.log_K == log_K) return &F.G;
        return nullptr;
    }
    int collect_now() {
        bool wait = true;
        for (auto& F : fams) {
            if (F.have || !F.ticket) continue;
            int rc = atlas_rt_shout_ra_evals_finish(F.ticket, wait, F.G);
            if (rc) return rc;
            wait = false;
            F.have = true;
        }
        return ATLAS_OK;
    }
    // The node's prefix-suffix read-raf instances built AHEAD, behind the launches above and still before the node's first wait: their
    // constructors depend on r_node_output and the witness alone (u_evals = the shared eq table, the sign scan, phase 0's tables; gamma only
    // enters the host's round arithmetic and is set when it has been drawn), so their launches — ~45 us of this thread per instance — and the
    // device's work for them sit under a wait the node makes anyway, and nothing waits for them when their proofs start.
    // ATLAS_NO_PREBUILD=1 is the A-B.
    atlas_instance_t clamp = nullptr, rc_inst = nullptr, relu = nullptr;
    static bool prebuild_on() { static const bool off = getenv("ATLAS_NO_PREBUILD") != nullptr; return !off && on(); }
    int prebuild_clamp(const uint64_t* d_cidx, size_t log_T, const atlas_fr_t* r) {
        if (!prebuild_on() || !eq) return ATLAS_OK;
        return atlas_rt_ps_shout_clamp_new(d_cidx, log_T, 64, 31, 1, r, nullptr, eq, &clamp, true);
    }
    int prebuild_rc(const uint64_t* d_ridx, size_t log_T, size_t S, size_t phases, const atlas_fr_t* r) {
        if (!prebuild_on() || !eq) return ATLAS_OK;
        return atlas_rt_identity_range_check_new(d_ridx, log_T, S, phases, r, eq, &rc_inst, true);
    }
    int prebuild_relu(const uint64_t* d_lookups, size_t log_T, size_t xlen, const atlas_fr_t* r) {
        if (!prebuild_on() || !eq) return ATLAS_OK;
        return atlas_rt_ps_shout_relu_new(d_lookups, log_T, xlen, r, nullptr, eq, &relu, true);
    }
    atlas_instance_t take(atlas_instance_t& slot) { atlas_instance_t i = slot; slot = nullptr; return i; }
    ~NodePre() {
        for (atlas_instance_t i : {clamp, rc_inst, relu}) if (i) atlas_instance_free(i);      // (before the eq table they borrow)
        for (auto& F : fams) if (F.ticket) atlas_rt_shout_ra_evals_drop(F.ticket);
        if (eq) { atlas_rt_eval_event_drop(eq); atlas_poly_free(eq); }
    }
    NodePre() = default;
    NodePre(const NodePre&) = delete;
    NodePre& operator=(const NodePre&) = delete;
};
// ra_onehot_provers of every family, in order (the challenge draws of each: HammingWeight's gamma powers, Booleanity's gammas and
// r_address): appends [RaVirtual, HammingWeight, Booleanity] per family to the batch and to `insts` (owned by the caller)
int onehot_families_build(std::vector<OneHotFamily>& fams, size_t log_T, atlas_transcript_t* t, atlas_batched_t b, std::vector<atlas_instance_t>& insts, size_t* dmax_out) {
    const size_t lkc = 4;                                            // OneHotParams::new: LOG_K_CHUNK = 4 (common/src/consts/general.rs:2)
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(t);
    const bool trace = getenv("ATLAS_TRACE") != nullptr;
    auto tr0 = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!trace) return;
        atlas_sync();
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[atlas trace] onehot_checks x%zu %-24s %8.3f ms\n", fams.size(), what, std::chrono::duration<double, std::milli>(t1 - tr0).count());
        tr0 = t1;
    };
    int rc = ATLAS_OK;
    size_t dmax = 0;
    const H::Fr zero = H::zero();
    for (auto& F : fams) {
        const size_t d = (F.log_K + lkc - 1) / lkc;
        dmax = d > dmax ? d : dmax;
        std::vector<H::Fr> gamma_powers(d);                         // challenge_scalar_powers(d)
        { const H::Fr q = H::tr_challenge_scalar(T); gamma_powers[0] = H::one(); for (size_t i = 1; i < d; i++) gamma_powers[i] = H::mul(gamma_powers[i - 1], q); }
        std::vector<H::Fr> gammas(d), r_addr(lkc);                   // challenge_vector_optimized
        for (size_t i = 0; i < d; i++) { uint64_t lo, hi; H::tr_challenge_u128(T, lo, hi); gammas[i] = H::challenge_to_fr(lo, hi, rt().challenge_mode); }
        for (size_t i = 0; i < lkc; i++) { uint64_t lo, hi; H::tr_challenge_u128(T, lo, hi); r_addr[i] = H::challenge_to_fr(lo, hi, rt().challenge_mode); }
        // G = compute_ra_evals(lookup_indices, params, r_cycle) (shout.rs:550-598)
        std::vector<H::Fr> Gh;
        if (F.G_pre) Gh = *F.G_pre;
        else {
            atlas_poly_t eq_rc = nullptr;
            if (!rc) rc = atlas_eq_evals(F.r_cycle, log_T, nullptr, &eq_rc);
            if (!rc) rc = atlas_rt_shout_ra_evals_host(F.d_lookups, (size_t)1 << log_T, F.log_K, lkc, eq_rc, Gh);
            if (eq_rc) atlas_poly_free(eq_rc);
        }
        if (rc) break;
        std::vector<atlas_fr_t> G(d << lkc);
        std::memcpy(G.data(), Gh.data(), G.size() * sizeof(atlas_fr_t));
        mark("ra_evals G");
        // RaVirtual: (r_address, r_cycle) = the ra opening point split at log_K
        atlas_instance_t ra = nullptr, hw = nullptr, bo = nullptr;
        rc = atlas_ra_virtual_from_lookups_new(F.d_lookups, log_T, F.log_K, lkc, F.ra_point.data(), F.ra_point.data() + F.log_K, &ra);
        if (!rc) rc = atlas_hamming_weight_new(G.data(), d, lkc, (const atlas_fr_t*)gamma_powers.data(), &hw);
        if (!rc) rc = atlas_booleanity_from_lookups_new(G.data(), F.d_lookups, log_T, F.log_K, lkc, (const atlas_fr_t*)gammas.data(), (const atlas_fr_t*)r_addr.data(), F.r_cycle, &bo);
        mark("booleanity_new");
        for (atlas_instance_t i : {ra, hw, bo}) insts.push_back(i);
        H::Fr hw_claim = H::zero();
        for (auto& x : gamma_powers) hw_claim = H::add(hw_claim, x);    // hamming_weight.rs:49-57
        if (!rc) rc = atlas_batched_add_instance(b, ra, (const atlas_fr_t*)&F.ra_claim);
        if (!rc) rc = atlas_batched_add_instance(b, hw, (const atlas_fr_t*)&hw_claim);
        if (!rc) rc = atlas_batched_add_instance(b, bo, (const atlas_fr_t*)&zero);
        if (rc) break;
    }
    if (dmax_out) *dmax_out = dmax;
    return rc;
}
// cache_openings of the families' instances (insts[3 f + {0, 1, 2}]) after a BatchedSumcheck of `mr` rounds with challenges rs: an
// instance of n rounds sees the LAST n challenges (sumcheck.rs:150-170)
int onehot_families_cache(std::vector<OneHotFamily>& fams, const atlas_instance_t* insts, size_t log_T, const std::vector<H::Fr>& rs, atlas_transcript_t* t, Out& O) {
    const size_t lkc = 4, mr = rs.size(), b0 = mr - lkc - log_T;     // b0: where Booleanity's lkc + log_T challenges start
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(t);
    int rc = ATLAS_OK;
    for (size_t f = 0; f < fams.size() && !rc; f++) {
        OneHotFamily& F = fams[f];
        const size_t d = (F.log_K + lkc - 1) / lkc, pad = d * lkc - F.log_K;
        for (int which = 0; which < 3 && !rc; which++) {
            atlas_fr_t fin[64]; size_t nf = 0;
            rc = atlas_instance_final_claims(insts[3 * f + which], fin, 64, &nf);
            for (size_t i = 0; i < d && !rc; i++) {                  // append_sparse: one transcript append per claim (:335-339)
                gr::Point pt;
                if (O.sink) {
                    pt.resize(lkc + log_T);
                    if (which == 0) {            // RaVirtual (ra_virtual.rs:157-184): chunk i of the zero-padded r_address, reversed cycle challenges
                        for (size_t q = 0; q < lkc; q++) { const size_t pos = i * lkc + q; pt[q] = pos < pad ? H::zero() : *reinterpret_cast<const H::Fr*>(&F.ra_point[pos - pad]); }
                        for (size_t q = 0; q < log_T; q++) pt[lkc + q] = rs[mr - 1 - q];
                    } else if (which == 1) {     // HammingWeight (hamming_weight.rs:149-168): reversed address challenges, r_cycle of the source opening
                        for (size_t q = 0; q < lkc; q++) pt[q] = rs[mr - 1 - q];
                        for (size_t q = 0; q < log_T; q++) pt[lkc + q] = *reinterpret_cast<const H::Fr*>(&F.r_cycle[q]);
                    } else {                     // Booleanity (booleanity.rs:71-76, 350-368): both halves reversed
                        for (size_t q = 0; q < lkc; q++) pt[q] = rs[b0 + lkc - 1 - q];
                        for (size_t q = 0; q < log_T; q++) pt[lkc + q] = rs[mr - 1 - q];
                    }
                }
                rc = O.append_sparse(T, F.rad_cp, i, which == 0 ? gr::SC_RaVirtualization : which == 1 ? gr::SC_HammingWeight : gr::SC_Booleanity, pt,
                                     *reinterpret_cast<H::Fr*>(&fin[i]));
            }
        }
    }
    return rc;
}
int prove_onehot_checks_multi(std::vector<OneHotFamily>& fams, size_t log_T, atlas_transcript_t* t, Out& O, uint8_t proof_type) {
    const size_t lkc = 4;
    std::vector<atlas_instance_t> insts;
    atlas_batched_t b = nullptr;
    int rc = atlas_batched_new(&b);
    size_t dmax = 0;
    if (!rc) rc = onehot_families_build(fams, log_T, t, b, insts, &dmax);
    size_t stride = dmax + 2, max_rounds = lkc + log_T;
    std::vector<atlas_fr_t> rows(max_rounds * stride);
    std::vector<uint32_t> nco(max_rounds);
    std::vector<atlas_u128_t> ch(max_rounds);
    size_t mr = 0;
    if (!rc) rc = atlas_batched_prove(b, t, rows.data(), stride, nco.data(), ch.data(), &mr);
    std::vector<H::Fr> rs(mr);
    for (size_t i = 0; i < mr && !rc; i++) rs[i] = ch_fr(ch[i]);
    if (!rc) rc = onehot_families_cache(fams, insts.data(), log_T, rs, t, O);
    if (!rc) rc = O.put_proof(rows, stride, nco, mr, proof_type);
    if (b) atlas_batched_free(b);
    for (atlas_instance_t inst : insts) if (inst) atlas_instance_free(inst);
    return rc;
}
int prove_onehot_checks(const uint64_t* d_lookups, size_t log_T, size_t log_K, const atlas_fr_t* r_cycle, const std::vector<atlas_fr_t>& ra_point,
                        const H::Fr& ra_claim, atlas_transcript_t* t, Out& O, uint8_t rad_cp = 0, uint8_t proof_type = gr::PT_RaOneHotChecks,
                        const NodePre* pre = nullptr) {
    std::vector<OneHotFamily> f(1);
    f[0].d_lookups = d_lookups; f[0].log_K = log_K; f[0].r_cycle = r_cycle; f[0].ra_point = ra_point; f[0].ra_claim = ra_claim; f[0].rad_cp = rad_cp;
    if (pre) f[0].G_pre = pre->G_for(d_lookups, log_K);
    return prove_onehot_checks_multi(f, log_T, t, O, proof_type);
}

}  // namespace

namespace {

// try_rebase_intermediates on the device (fused_rebase.rs:110-128): rescaled quotient (i64 and as Fr), remainder, the
// clamped i32 output and the two lookup-index vectors.  The graph prover builds it once per node (commit phase) and keeps
// it for the IOP; the single-node entry points build it inside the call.
struct RescaleWitness {
    size_t T = 0, S = 0;
    DevBuf quot, rem, cidx, ridx, qfr, out_own;
    int32_t* d_output = nullptr;
};
// fill_acc launches the kernels that leave the T i64 accumulators in d_acc (library stream)
template <class FillAcc>
int make_rescale_witness(size_t T, size_t S, FillAcc&& fill_acc, int32_t* d_output, RescaleWitness& W) {
    W.T = T; W.S = S;
    hipError_t e = W.quot.alloc(T * 8);
    if (e == hipSuccess) e = W.rem.alloc(T * 4);
    if (e == hipSuccess) e = W.cidx.alloc(T * 8);
    if (e == hipSuccess) e = W.ridx.alloc(T * 8);
    if (e == hipSuccess) e = W.qfr.alloc(T * sizeof(Fr));
    if (e == hipSuccess && !d_output) { e = W.out_own.alloc(T * 4); d_output = W.out_own.as<int32_t>(); }
    if (e != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(fused-rescale witness)", e);
    W.d_output = d_output;
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    size_t gb = (T + 255) / 256; if (gb > 4096) gb = 4096;
    int rc = fill_acc(W.quot.as<int64_t>());
    if (rc) return rc;
    k_einsum_rebase<<<(unsigned)gb, 256, 0, rt().stream>>>(W.quot.as<int64_t>(), T, (uint32_t)S, W.rem.as<int32_t>(), d_output, W.cidx.as<uint64_t>(), W.ridx.as<uint64_t>());
    k_i64_to_fr<<<(unsigned)gb, 256, 0, rt().stream>>>(W.quot.as<int64_t>(), W.qfr.as<Fr>(), T);
    return ATLAS_OK;
}

// How many phases the address rounds of an identity range check over log_K bits are cut into.  The reference's default
// (IdentityRCProvider::phases, identity_range_check.rs:506-518) is log_m = 4, or 2 when 4 does not divide log_K: seven phases for the
// 14-bit remainder of every fused rescale.  The cut is the PROVER's bookkeeping — round j's polynomial is sum over the remaining
// variables of the summand at (r_0 .. r_{j-1}, X, .), whatever tables it is accumulated through — and every phase boundary costs a pass
// over T plus a table hand-over (~45 us at these sizes).  So: the largest chunk of at most 8 bits that divides log_K (14 -> two phases
// of 7 bits: one boundary instead of six); ATLAS_RC_REF_PHASES=1 restores the reference's cut (A-B: same proof bytes).
inline size_t identity_rc_phases(size_t log_K) {
    static const bool ref_cut = getenv("ATLAS_RC_REF_PHASES") != nullptr;
    if (ref_cut) return log_K <= 2 ? 1 : log_K % 4 == 0 ? log_K / 4 : log_K % 2 == 0 ? log_K / 2 : log_K;
    for (size_t log_m = 8; log_m >= 1; log_m--) if (log_K % log_m == 0) return log_K / log_m;
    return log_K;
}

// prove_clamp_lookup (clamp_lookups/mod.rs:264-309) after its raf claim: gamma, PS-Shout read-raf over SaturationTable =
// ClampBoundedTable<64, 31, true> (ProofType::Execution) with its ClampRa opening, then ra_onehot_provers + BatchedSumcheck
// (ProofType::RaOneHotChecks) over the ClampRaD chunks.  acc_claim = the i64 accumulation's opening (already appended),
// out_claim = the node output's reduced opening.  stage_ms[0..1]: lookup, one-hot checks.
inline int prove_clamp_lookup_flow(const uint64_t* d_cidx, size_t log_T, const atlas_fr_t* r_node_output, const H::Fr& acc_claim, const H::Fr& out_claim,
                                   atlas_transcript_t* t, Out& O, double* stage_ms, NodePre* pre = nullptr) {
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point a) { atlas_sync(); return std::chrono::duration<double, std::milli>(now() - a).count(); };
    H::Transcript& Tr = *reinterpret_cast<H::Transcript*>(t);
    auto t0 = now();
    const H::Fr gamma = H::tr_challenge_scalar(Tr);                          // ps_read_raf_prover (unary.rs:112)
    atlas_instance_t exec = pre ? pre->take(pre->clamp) : nullptr;           // built ahead (NodePre::prebuild_clamp)?
    int rc = exec ? atlas_rt_ps_set_gamma(exec, (const atlas_fr_t*)&gamma)
           : pre && pre->eq ? atlas_rt_ps_shout_clamp_new(d_cidx, log_T, 64, 31, 1, r_node_output, (const atlas_fr_t*)&gamma, pre->eq, &exec)
                            : atlas_ps_shout_clamp_new(d_cidx, log_T, 64, 31, 1, r_node_output, (const atlas_fr_t*)&gamma, &exec);
    const H::Fr exec_claim = H::add(out_claim, H::mul(gamma, acc_claim));     // rv_claim + gamma * operand_claim (ps_shout/mod.rs:142-144)
    std::vector<atlas_u128_t> ch;
    H::Fr ra_claim;
    std::vector<atlas_fr_t> ra_point;                                         // ClampRa opening point (normalize_opening_point)
    if (!rc) rc = prove_single(exec, exec_claim, t, O, ch, &ra_claim, 64, gr::VP_ClampRa, gr::PT_Execution, &ra_point);
    if (exec) atlas_instance_free(exec);
    if (stage_ms) stage_ms[0] = ms_since(t0);
    O.mark("clamp lookup: read-raf (64)");
    t0 = now();
    if (!rc) rc = prove_onehot_checks(d_cidx, log_T, 64, r_node_output, ra_point, ra_claim, t, O, gr::CP_ClampRaD, gr::PT_RaOneHotChecks, pre);
    if (stage_ms) stage_ms[1] = ms_since(t0);
    O.mark("clamp lookup: one-hot (d=16)");
    return rc;
}

// The fused-rescale operator flow shared by Einsum, Mul, ... (impl_fused_rescale_proof_api, ops/mod.rs:569-612):
//   fused_rebase::prove_pre (fused_rebase.rs:215-250): remainder advice + rescaled-accumulator claim, saturating clamp lookup
//   (PS-Shout over ClampBoundedTable<64, 31, true>) and its one-hot checks;
//   the operator's own sumcheck over the i64 accumulator, input claim fused_input_claim = rescaled(r) 2^S + R(r);
//   fused_rebase::prove_remainder_rc (:252-285): identity range check of the remainder in [0, 2^S) and its one-hot checks.
// inner(in_claim) runs the operator's sumcheck, appends its openings and stores its proof.  stage_ms[6]: witness openings,
// clamp, one-hot checks, inner, range check, remainder one-hot checks.
template <class Inner>
int prove_fused_rescale(RescaleWitness& W, Inner&& inner, const atlas_fr_t* r_node_output, const atlas_fr_t* output_claim,
                        atlas_transcript_t* t, Out& O, double* stage_ms) {
    const size_t T = W.T, S = W.S, log_T = ilog2(T);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point a) { atlas_sync(); return std::chrono::duration<double, std::milli>(now() - a).count(); };
    H::Transcript& Tr = *reinterpret_cast<H::Transcript*>(t);
    const gr::Point r0 = O.sink ? to_point(r_node_output, log_T) : gr::Point();

    auto t0 = now();
    int rc = ATLAS_OK;
    NodePre pre;                                                             // eq(r_node_output) and the G tables of both lookups: in flight under the wait below
    if (T > 1) rc = pre.begin(r_node_output, log_T, {{W.cidx.as<uint64_t>(), (size_t)64}, {W.ridx.as<uint64_t>(), S}});
    if (!rc && T > 1) rc = pre.prebuild_clamp(W.cidx.as<uint64_t>(), log_T, r_node_output);
    if (!rc && T > 1) rc = pre.prebuild_rc(W.ridx.as<uint64_t>(), log_T, S, identity_rc_phases(S), r_node_output);
    if (rc) return rc;
    atlas_poly_t p_rem = nullptr, p_quot = nullptr, p_out = nullptr;
    {   // borrowed views for evaluate
        rc = atlas_poly_wrap_device_fr(W.qfr.p, T, &p_quot);
        if (!rc) rc = atlas_poly_wrap_device_i32(W.rem.as<int32_t>(), T, &p_rem);
        if (!rc) rc = atlas_poly_wrap_device_i32(W.d_output, T, &p_out);
    }
    H::Fr eval_R, acc_claim, out_claim;
    if (!rc) {   // the three openings at r_node_output share their eq tables and one synchronisation
        const atlas_poly_t ps[3] = {p_rem, p_quot, p_out};
        H::Fr ev[3];
        rc = pre.eq ? atlas_rt_evaluate_with_eq(ps, output_claim ? 2 : 3, pre.eq, (atlas_fr_t*)ev)                  // (one pass against the node's eq table)
                    : atlas_poly_evaluate_many(ps, output_claim ? 2 : 3, r_node_output, log_T, (atlas_fr_t*)ev);
        eval_R = ev[0]; acc_claim = ev[1];
        if (output_claim) std::memcpy(&out_claim, output_claim, 32); else out_claim = ev[2];
    }
    for (atlas_poly_t p : {p_rem, p_quot, p_out}) if (p) atlas_poly_free(p);
    if (!rc) rc = pre.collect(false);                                        // (atlas_poly_evaluate_many waited for the library stream)
    if (stage_ms) stage_ms[0] = ms_since(t0);
    if (rc) return rc;
    O.mark("fused: witness openings");

    // ---- prove_pre: remainder advice (cache_remainder_prove), the rescaled accumulator's raf claim, clamp lookup
    rc = O.append_virtual(Tr, gr::node_exec(gr::virt(gr::VP_RescaleRemainder, O.node), O.node), r0, eval_R);
    if (!rc) rc = O.append_virtual(Tr, gr::node_exec(gr::virt(gr::VP_ClampAcc, O.node), O.node), r0, acc_claim);      // append_raf_claims_prover
    // (a scalar node — is_scalar, clamp_lookups/mod.rs:67 — has no lookups: prove_append_acc only, fused_rebase.rs:215-231; its verifier reads
    // the accumulation and the remainder in the clear)
    const bool scalar = T == 1;
    if (!rc && !scalar) rc = prove_clamp_lookup_flow(W.cidx.as<uint64_t>(), log_T, r_node_output, acc_claim, out_claim, t, O, stage_ms ? stage_ms + 1 : nullptr, &pre);

    // ---- the operator's sumcheck over the accumulator
    t0 = now();
    if (!rc) {
        // fused_input_claim: rescaled(r0) 2^S + R(r0)
        const H::Fr in_claim = H::add(H::mul(acc_claim, H::from_u64((uint64_t)1 << S)), eval_R);
        rc = inner(in_claim);
    }
    if (stage_ms) stage_ms[3] = ms_since(t0);
    O.mark("fused: operator sumcheck");

    if (scalar) return rc;
    // ---- prove_remainder_rc
    t0 = now();
    std::vector<atlas_fr_t> rr_point;
    std::vector<atlas_u128_t> ch;
    H::Fr rr_claim;
    if (!rc) {
        const size_t phases = identity_rc_phases(S);
        atlas_instance_t rcq = pre.take(pre.rc_inst);                         // built ahead (NodePre::prebuild_rc)?
        if (!rcq) rc = pre.eq ? atlas_rt_identity_range_check_new(W.ridx.as<uint64_t>(), log_T, S, phases, r_node_output, pre.eq, &rcq)
                              : atlas_identity_range_check_new(W.ridx.as<uint64_t>(), log_T, S, phases, r_node_output, &rcq);
        if (!rc) rc = prove_single(rcq, eval_R, t, O, ch, &rr_claim, S, gr::VP_RescaleRemainderRa, gr::PT_RangeCheck, &rr_point);
        if (rcq) atlas_instance_free(rcq);
    }
    if (stage_ms) stage_ms[4] = ms_since(t0);
    O.mark("fused: remainder range check");
    t0 = now();
    if (!rc) rc = prove_onehot_checks(W.ridx.as<uint64_t>(), log_T, S, r_node_output, rr_point, rr_claim, t, O, gr::CP_RescaleRemainderRaD, gr::PT_RescaleRemainderRaChecks, &pre);
    if (stage_ms) stage_ms[5] = ms_since(t0);
    O.mark("fused: remainder one-hot");
    return rc;
}

}  // namespace

