// EQ tables, MLE evaluation and the Gruen split-eq product sumcheck: host side + C-ABI.
// Mirrors (paths under the jolt-atlas tree):
//   EqPolynomial::evals                    joltworks/src/poly/eq_poly.rs:77-101
//   MultilinearPolynomial::evaluate        joltworks/src/poly/multilinear_polynomial.rs:766-862
//   MulProver as SumcheckInstanceProver    jolt-atlas-core/src/onnx_proof/ops/mul.rs:125-185
//   Sumcheck::prove                        joltworks/src/subprotocols/sumcheck.rs:565-599
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_field.hpp"
#include "runtime.hpp"
#include "sc_consts.hpp"
#include "spliteq_kernels.hip.h"

using namespace atlas;
namespace H = atlas_host;
using atlas_rt::fail;
using atlas_rt::rt;
using atlas_rt::MAX_ROUNDS;

struct atlas_mul_prover {
    atlas_poly_t left = nullptr, right = nullptr;
    size_t n = 0, m = 0;
    Fr* d_w = nullptr;          // w (n Fr)
    Fr* d_eout = nullptr;       // prefix tables of w_out = w[0..m): 2^(m+1) - 1 Fr
    Fr* d_ein = nullptr;        // prefix tables of w_in = w[m..n-1)
    std::vector<H::Fr> w;
    H::Fr scalar;               // current_scalar (host copy for the round API)
    bool consumed = false;
};

namespace {

inline int grid_for(size_t work) {
    size_t b = (work + SC_THREADS - 1) / SC_THREADS;
    if (b < 1) b = 1;
    if (b > (size_t)SC_MAX_BLOCKS) b = SC_MAX_BLOCKS;
    return (int)b;
}
inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
inline unsigned ilog2(size_t x) { unsigned n = 0; while (x > 1) { x >>= 1; n++; } return n; }
Fr to_dev(const H::Fr& f) { Fr o; std::memcpy(o.v, f.l, 32); return o; }

__global__ __launch_bounds__(SC_THREADS) void k_reduce1(const Fr* partials, int n_partials, Fr* out, int width) {
    __shared__ Fr red[SC_THREADS / 64][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < width; k++) {
        Fr acc = fe_zero();
        for (int b = threadIdx.x; b < n_partials; b += SC_THREADS) acc = fr_add(acc, fe_load(partials + (size_t)b * width + k));
        Fr s = fr_wave_sum(acc);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < width) {
        Fr s = red[0][threadIdx.x];
        for (int w = 1; w < SC_THREADS / 64; w++) s = fr_add(s, red[w][threadIdx.x]);
        fe_store(out + threadIdx.x, s);
    }
}

struct EqPointArgs { Fr v[32]; };
__global__ void k_store_point(EqPointArgs a, uint32_t n, Fr* __restrict__ dst) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) fe_store(dst + i, a.v[i]);
}

// EqPolynomial::evals of at most 16 variables in ONE launch: the point is a kernel argument; every workgroup grows the ten-level head in LDS
// (k_eq_head's passes), owns 4096 entries of the table and multiplies in the factors of the bits above — two direct ones and the product
// over its own index bits, formed once.  Four launches (point, head, two doublings) were one constructor step of every instance over an eq
// table (~1800 per GPT-2-shaped proof, on the critical path).  Same residues: x - x r = x (1 - r), exact arithmetic.
__global__ __launch_bounds__(1024) void k_eq_full(Fr* ev, EqPointArgs pt, uint32_t n, Fr scale) {
    __shared__ Fr tab[1024];
    __shared__ Fr s_wg;
    const uint32_t LL = n < 10 ? n : 10;
    if (threadIdx.x == 0) {
        tab[0] = scale;
        Fr f = fr_one();                                           // the bits of this workgroup's index: table bits 12, 13, ...
        for (uint32_t q = 12; q < n; q++) { const Fr rq = pt.v[n - 1 - q]; f = fr_mul(f, ((blockIdx.x >> (q - 12)) & 1u) ? rq : fr_sub(fr_one(), rq)); }
        s_wg = f;
    }
    __syncthreads();
    for (uint32_t p = 0; p < LL; p++) {
        const uint32_t size = 1u << p;
        const Fr rp = pt.v[n - 1 - p];
        const bool act = threadIdx.x < size;
        Fr x = fe_zero(), y = fe_zero();
        if (act) { x = tab[threadIdx.x]; y = fr_mul(x, rp); }
        if (act) { tab[threadIdx.x + size] = y; tab[threadIdx.x] = fr_sub(x, y); }
        __syncthreads();
    }
    const uint32_t top = n < 12 ? n : 12, per_wg = 1u << top, mask = (1u << LL) - 1;
    Fr hi_r[2], hi_c[2];
    for (uint32_t q = 0; q + LL < top; q++) { hi_r[q] = pt.v[n - 1 - (LL + q)]; hi_c[q] = fr_sub(fr_one(), hi_r[q]); }
    const Fr wgf = s_wg;
    for (uint32_t i = threadIdx.x; i < per_wg; i += 1024) {
        Fr v = tab[i & mask];
        for (uint32_t q = 0; q + LL < top; q++) v = fr_mul(v, ((i >> (LL + q)) & 1) ? hi_r[q] : hi_c[q]);
        if (n > 12) v = fr_mul(v, wgf);
        fe_store(ev + (size_t)blockIdx.x * per_wg + i, v);
    }
}
inline bool eq_full_on() { static const bool off = getenv("ATLAS_NO_EQ_FULL") != nullptr; return !off; }      // A-B
inline void launch_eq_full(const H::Fr* r, size_t n, const H::Fr* scaling, Fr* ev) {
    EqPointArgs a;
    if (n) std::memcpy(a.v, r, n * sizeof(Fr));
    k_eq_full<<<n > 12 ? 1u << (n - 12) : 1u, 1024, 0, rt().stream>>>(ev, a, (uint32_t)n, to_dev(scaling ? *scaling : H::one()));
}

// EqPolynomial::evals into a fresh device buffer (2^n Fr)
int eq_evals_device(const H::Fr* r, size_t n, const H::Fr* scaling, Fr** out) {
    const size_t len = (size_t)1 << n;
    Fr* ev = nullptr;
    hipError_t e = hipMalloc(&ev, len * sizeof(Fr));
    if (e != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(eq)", e);
    if (n <= 16 && eq_full_on()) {
        launch_eq_full(r, n, scaling, ev);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) { hipFree(ev); return fail(ATLAS_ENODEV, "eq_evals", le); }
        *out = ev;
        return ATLAS_OK;
    }
    // the point travels as a kernel argument (<= 30 field elements): no staging copy, no synchronisation — the table is
    // complete in stream order, which is all its users (kernels on the library stream) need
    Fr* d_r = nullptr;
    HIP_TRY(hipMalloc(&d_r, (n ? n : 1) * sizeof(Fr)));
    if (n) {
        EqPointArgs a;
        std::memcpy(a.v, r, n * sizeof(Fr));
        k_store_point<<<1, 64, 0, rt().stream>>>(a, (uint32_t)n, d_r);
    }
    const uint32_t head = n < 12 ? (uint32_t)n : 12u;
    k_eq_head<<<1, 1024, 0, rt().stream>>>(ev, d_r, (uint32_t)n, head, to_dev(scaling ? *scaling : H::one()));
    for (size_t p = head; p < n; p++)
        k_eq_double<<<grid_for((size_t)1 << p), SC_THREADS, 0, rt().stream>>>(ev, (size_t)1 << p, to_dev(r[n - 1 - p]));
    hipError_t le = hipGetLastError();
    hipFree(d_r);                                 // (pool: reused in stream order)
    if (le != hipSuccess) { hipFree(ev); return fail(ATLAS_ENODEV, "eq_evals", le); }
    *out = ev;
    return ATLAS_OK;
}

}  // namespace

// EqPolynomial::evals into a caller-provided device buffer of 2^n Fr (library stream; the caller holds rt().mu): opening.hip's pool
int atlas_rt_eq_evals_into(const H::Fr* r, size_t n, Fr* ev) {
    if (n <= 16 && eq_full_on()) {
        launch_eq_full(r, n, nullptr, ev);
        hipError_t le0 = hipGetLastError();
        return le0 == hipSuccess ? ATLAS_OK : fail(ATLAS_ENODEV, "eq_evals_into", le0);
    }
    Fr* d_r = nullptr;
    HIP_TRY(hipMalloc(&d_r, (n ? n : 1) * sizeof(Fr)));
    if (n) {
        EqPointArgs a;
        std::memcpy(a.v, r, n * sizeof(Fr));
        k_store_point<<<1, 64, 0, rt().stream>>>(a, (uint32_t)n, d_r);
    }
    const uint32_t head = n < 12 ? (uint32_t)n : 12u;
    k_eq_head<<<1, 1024, 0, rt().stream>>>(ev, d_r, (uint32_t)n, head, to_dev(H::one()));
    for (size_t p = head; p < n; p++)
        k_eq_double<<<grid_for((size_t)1 << p), SC_THREADS, 0, rt().stream>>>(ev, (size_t)1 << p, to_dev(r[n - 1 - p]));
    hipError_t le = hipGetLastError();
    hipFree(d_r);                                 // (pool: reused in stream order)
    return le == hipSuccess ? ATLAS_OK : fail(ATLAS_ENODEV, "eq_evals_into", le);
}

extern "C" {

int atlas_eq_evals(const atlas_fr_t* r, size_t n, const atlas_fr_t* scaling, atlas_poly_t* out) {
    PROF("atlas_eq_evals");
    NEED_INIT();
    if ((!r && n) || !out || n > 30) return fail(ATLAS_EINVAL, "eq_evals");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    Fr* ev = nullptr;
    int rc = eq_evals_device(reinterpret_cast<const H::Fr*>(r), n, reinterpret_cast<const H::Fr*>(scaling), &ev);
    if (rc) return rc;
    atlas_poly* p = new atlas_poly();
    p->d = ev; p->len = (size_t)1 << n; p->cap_bytes = p->len * sizeof(Fr); p->is_i32 = false; p->owned = true;
    *out = p;
    return ATLAS_OK;
}

int atlas_poly_evaluate_many(const atlas_poly_t* polys, size_t count, const atlas_fr_t* r, size_t n, atlas_fr_t* out) {
    PROF("atlas_poly_evaluate_many");
    NEED_INIT();
    if (!polys || !count || count > 64 || (!r && n) || !out) return fail(ATLAS_EINVAL, "poly_evaluate_many");
    for (size_t i = 0; i < count; ++i)
        if (!polys[i] || polys[i]->len != ((size_t)1 << n)) return fail(ATLAS_EINVAL, "poly_evaluate: point length != num_vars");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    // DensePolynomial::evaluate: r = (r2 | r1), eq_one = evals(r2) outer, eq_two = evals(r1) inner.
    // The two tables are shared by every polynomial of the call; each result goes straight into
    // the pinned staging area, so the whole call costs one synchronisation.
    const size_t m = n / 2;
    const H::Fr* rr = reinterpret_cast<const H::Fr*>(r);
    Fr *eq1 = nullptr, *eq2 = nullptr;
    int rc = eq_evals_device(rr, m, nullptr, &eq1);
    if (rc) return rc;
    rc = eq_evals_device(rr + m, n - m, nullptr, &eq2);
    if (rc) { hipFree(eq1); return rc; }
    const int grid = grid_for((size_t)1 << n);
    const ScConsts K = make_consts();
    for (size_t i = 0; i < count; ++i) {
        const atlas_poly_t p = polys[i];
        if (p->is_i32)
            k_mle_evaluate<int32_t><<<grid, SC_THREADS, 0, rt().stream>>>((const int32_t*)p->d, p->len, eq1, eq2, (uint32_t)(n - m), rt().d_partials, K);
        else
            k_mle_evaluate<Fr><<<grid, SC_THREADS, 0, rt().stream>>>((const Fr*)p->d, p->len, eq1, eq2, (uint32_t)(n - m), rt().d_partials, K);
        k_reduce1<<<1, SC_THREADS, 0, rt().stream>>>(rt().d_partials, grid, (Fr*)rt().h_pinned + i, 1);
    }
    hipError_t e = hipStreamSynchronize(rt().stream);
    hipFree(eq1); hipFree(eq2);
    if (e != hipSuccess) return fail(ATLAS_ENODEV, "poly_evaluate", e);
    std::memcpy(out, rt().h_pinned, count * sizeof(Fr));
    return ATLAS_OK;
}

}  // extern "C"

// MultilinearPolynomial::evaluate of up to three polynomials at the point whose FULL eq table the caller holds (node_flow.hip.h: NodePre):
// one pass — sum_i eq[i] Z_q[i] for every q — and one reduction, against two tables, a pass and a reduction per polynomial above.
namespace {
struct EvalEqArgs { const void* p[3]; uint32_t is_i32[3]; uint32_t count; };
__global__ __launch_bounds__(SC_THREADS) void k_eval_with_eq(EvalEqArgs A, const Fr* __restrict__ eq, size_t len, Fr* partials, ScConsts K) {
    Fr acc[3];
#pragma unroll
    for (int q = 0; q < 3; q++) acc[q] = fe_zero();
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < len; i += (size_t)gridDim.x * SC_THREADS) {
        const Fr e = fe_load(eq + i);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (q >= (int)A.count) break;
            const Fr z = A.is_i32[q] ? fr_from_i32(reinterpret_cast<const int32_t*>(A.p[q])[i], K.k32) : fe_load(reinterpret_cast<const Fr*>(A.p[q]) + i);
            acc[q] = fr_add(acc[q], fr_mul(e, z));
        }
    }
    block_reduce_store<3>(acc, partials);
}
}  // namespace
// NodePre::begin calls this right after it has enqueued the node's eq table: the evaluation below then runs on a stream of its own behind THIS point of
// the library stream — the operands are tensors / witness columns of the trace, complete long before — and the host waits for the evaluation alone,
// while the lookup tables and prebuilt instances NodePre enqueues next (100-150 us of device work per lookup node, needed only when their proofs start)
// run beside it.  Before, the wait was for all of it: 170 us per node, 343 nodes of the GPT-2-shaped graph (`sync atlas_rt_evaluate_with_eq`, r06m).
// ATLAS_NO_SIDE_EVAL=1: everything on the library stream (A/B).
int atlas_rt_eval_event_record(atlas_poly_t eq_full) {
    static const bool off = getenv("ATLAS_NO_SIDE_EVAL") != nullptr;
    if (off || !eq_full) return ATLAS_OK;
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    // Not while ANOTHER runtime drives this device (ranks of a sharded proof that share a GPU, threads of one process on one GPU: the test boxes): the
    // evaluation's stream waits for an EVENT of the library stream, and with the other runtime's polling launches in the same hardware queues that is a
    // cycle — its eq table behind their polling launch, their next launch behind the evaluation's barrier (seen: 1 run in 6 of the two-thread proof).
    if (rt().stream != rt().lib_stream || rt().no_lane_streams || rt().device_shared) return ATLAS_OK;
    if (rt().device >= 0 && rt().device < 64 && __atomic_load_n(&atlas_rt::g_device_runtimes[rt().device], __ATOMIC_SEQ_CST) > 1) return ATLAS_OK;
    if (!rt().eval_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&rt().eval_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&rt().eval_event, hipEventDisableTiming));
        HIP_TRY(hipMalloc(&rt().d_eval_partials, sizeof(Fr) * SC_MAX_BLOCKS * 3));
    }
    HIP_TRY(hipEventRecord(rt().eval_event, rt().stream));
    rt().eval_event_eq = eq_full->d;
    return ATLAS_OK;
}
void atlas_rt_eval_event_drop(atlas_poly_t eq_full) {
    if (eq_full && rt().eval_event_eq == eq_full->d) rt().eval_event_eq = nullptr;
}
bool atlas_rt_eval_event_live(atlas_poly_t eq_full) { return eq_full && rt().eval_event_eq == eq_full->d; }

int atlas_rt_evaluate_with_eq(const atlas_poly_t* polys, size_t count, atlas_poly_t eq_full, atlas_fr_t* out) {
    PROF("atlas_rt_evaluate_with_eq");
    NEED_INIT();
    if (!polys || !count || count > 3 || !eq_full || eq_full->is_i32 || !out) return fail(ATLAS_EINVAL, "evaluate_with_eq");
    EvalEqArgs A{};
    A.count = (uint32_t)count;
    for (size_t q = 0; q < count; q++) {
        if (!polys[q] || polys[q]->len != eq_full->len) return fail(ATLAS_EINVAL, "evaluate_with_eq: length != the eq table's");
        A.p[q] = polys[q]->d; A.is_i32[q] = polys[q]->is_i32 ? 1u : 0u;
    }
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    const int grid = grid_for(eq_full->len);
    if (rt().eval_event_eq == eq_full->d && rt().eval_stream && rt().stream == rt().lib_stream) {
        Fr* slot = reinterpret_cast<Fr*>(static_cast<char*>(rt().h_pinned) + atlas_rt::PINNED_BYTES - 4 * sizeof(Fr));      // (the synchronous users of h_pinned start at 0)
        HIP_TRY(hipStreamWaitEvent(rt().eval_stream, rt().eval_event, 0));
        k_eval_with_eq<<<grid, SC_THREADS, 0, rt().eval_stream>>>(A, (const Fr*)eq_full->d, eq_full->len, rt().d_eval_partials, make_consts());
        k_reduce1<<<1, SC_THREADS, 0, rt().eval_stream>>>(rt().d_eval_partials, grid, slot, 3);
        HIP_TRY(hipStreamSynchronize(rt().eval_stream));
        std::memcpy(out, slot, count * sizeof(Fr));
        return ATLAS_OK;
    }
    k_eval_with_eq<<<grid, SC_THREADS, 0, rt().stream>>>(A, (const Fr*)eq_full->d, eq_full->len, rt().d_partials, make_consts());
    k_reduce1<<<1, SC_THREADS, 0, rt().stream>>>(rt().d_partials, grid, (Fr*)rt().h_pinned, 3);
    HIP_TRY(hipStreamSynchronize(rt().stream));
    std::memcpy(out, rt().h_pinned, count * sizeof(Fr));
    return ATLAS_OK;
}

extern "C" {

int atlas_poly_evaluate(atlas_poly_t p, const atlas_fr_t* r, size_t n, atlas_fr_t* out) {
    PROF("atlas_poly_evaluate");
    if (!p) return fail(ATLAS_EINVAL, "poly_evaluate");
    return atlas_poly_evaluate_many(&p, 1, r, n, out);
}

// ------------------------------------------------------------------ MulProver
int atlas_mul_prover_new(atlas_poly_t left, atlas_poly_t right, const atlas_fr_t* w, size_t n, atlas_mul_prover_t* out) {
    NEED_INIT();
    if (!left || !right || !w || !out) return fail(ATLAS_EINVAL, "mul_prover_new: null argument");
    if (left->len != right->len || !is_pow2(left->len) || ilog2(left->len) != n || n == 0 || n > MAX_ROUNDS)
        return fail(ATLAS_EINVAL, "mul_prover_new: operand length must be 2^n, n >= 1");
    if (left->is_i32 != right->is_i32) return fail(ATLAS_EINVAL, "mul_prover_new: mixed operand types");
    if (n / 2 > 12 || n - 1 - n / 2 > 12) return fail(ATLAS_EINVAL, "mul_prover_new: n > 25 not supported");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    atlas_mul_prover* P = new atlas_mul_prover();
    P->left = left; P->right = right; P->n = n; P->m = n / 2;
    P->w.assign(reinterpret_cast<const H::Fr*>(w), reinterpret_cast<const H::Fr*>(w) + n);
    P->scalar = H::one();
    const size_t k_out = P->m, k_in = n - 1 - P->m;
    HIP_TRY(hipMalloc(&P->d_w, n * sizeof(Fr)));
    HIP_TRY(hipMalloc(&P->d_eout, (((size_t)2 << k_out)) * sizeof(Fr)));
    HIP_TRY(hipMalloc(&P->d_ein, (((size_t)2 << k_in)) * sizeof(Fr)));
    HIP_TRY(hipMemcpyAsync(P->d_w, w, n * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
    // GruenSplitEqPolynomial::new, LowToHigh: w = [w_out | w_in | w_last] (split_eq_poly.rs:97-121)
    k_eq_cached<<<1, 1024, 0, rt().stream>>>(P->d_eout, P->d_w, (uint32_t)k_out);
    k_eq_cached<<<1, 1024, 0, rt().stream>>>(P->d_ein, P->d_w + P->m, (uint32_t)k_in);
    HIP_TRY(hipStreamSynchronize(rt().stream));
    *out = P;
    return ATLAS_OK;
}

int atlas_mul_prover_free(atlas_mul_prover_t P) {
    if (!P) return ATLAS_OK;
    atlas_poly_free(P->left); atlas_poly_free(P->right);
    if (P->d_w) hipFree(P->d_w);
    if (P->d_eout) hipFree(P->d_eout);
    if (P->d_ein) hipFree(P->d_ein);
    delete P;
    return ATLAS_OK;
}

static SplitEqView view_for_round(const atlas_mul_prover* P, size_t round) {
    const size_t n_free = P->n - round - 1;
    const size_t out_bits = n_free < P->m ? n_free : P->m;
    const size_t in_bits = n_free - out_bits;
    SplitEqView E;
    E.e_out = P->d_eout + (((size_t)1 << out_bits) - 1);
    E.e_in = P->d_ein + (((size_t)1 << in_bits) - 1);
    E.in_bits = (uint32_t)in_bits;
    return E;
}

// input_claim = sum_x eq(w, x) L(x) R(x)  (the node-output opening in the reference, mul.rs:60-66)
int atlas_mul_input_claim(atlas_mul_prover_t P, atlas_fr_t* out) {
    NEED_INIT();
    if (!P || !out) return fail(ATLAS_EINVAL, "mul_input_claim");
    if (P->consumed || P->left->len != ((size_t)1 << P->n)) return fail(ATLAS_ESTATE, "mul_input_claim: instance already bound");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    // s(0) + s(1) of round 0: eq0*q0 + eq1*q1 with the round-0 sums
    const ScConsts K = make_consts();
    const size_t groups = P->left->len / 2;
    const int grid = grid_for(groups);
    SplitEqView E = view_for_round(P, 0);
    if (P->left->is_i32) k_mul_eval<int32_t><<<grid, SC_THREADS, 0, rt().stream>>>((const int32_t*)P->left->d, (const int32_t*)P->right->d, E, groups, rt().d_partials, K);
    else k_mul_eval<Fr><<<grid, SC_THREADS, 0, rt().stream>>>((const Fr*)P->left->d, (const Fr*)P->right->d, E, groups, rt().d_partials, K);
    k_reduce1<<<1, SC_THREADS, 0, rt().stream>>>(rt().d_partials, grid, rt().d_finals + 3, 3);
    HIP_TRY(hipMemcpyAsync(rt().h_pinned, rt().d_finals + 3, 3 * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    H::Fr s[3]; std::memcpy(s, rt().h_pinned, sizeof s);
    const H::Fr wl = P->w[P->n - 1];
    H::Fr claim = H::add(H::mul(H::sub(H::one(), wl), s[0]), H::mul(wl, s[1]));
    std::memcpy(out, &claim, 32);
    return ATLAS_OK;
}

// Sumcheck::prove for the Mul instance, transcript resident on the device.
//   compressed_polys : n * 3 Fr (c0, c2, c3 per round)   challenges : n raw u128
//   final_claims     : left(r), right(r), eq(w, r)
int atlas_sumcheck_prove_mul(atlas_mul_prover_t P, const atlas_fr_t* input_claim, atlas_transcript_t* transcript,
                             atlas_fr_t* compressed_polys, atlas_u128_t* challenges, atlas_fr_t final_claims[3]) {
    NEED_INIT();
    if (!P || !input_claim || !transcript || !compressed_polys || !challenges || !final_claims)
        return fail(ATLAS_EINVAL, "sumcheck_prove_mul: null argument");
    if (P->consumed || P->left->len != ((size_t)1 << P->n)) return fail(ATLAS_ESTATE, "sumcheck_prove_mul: prover already consumed");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    const ScConsts K = make_consts();
    const size_t n = P->n;
    const int mode = rt().challenge_mode;
    const int hi_only = mode == 0;

    MulCtx* d_cx = reinterpret_cast<MulCtx*>(rt().d_ctx);          // same scratch block (sizeof(MulCtx) <= sizeof(ScCtx))
    static_assert(sizeof(MulCtx) <= sizeof(ScCtx), "control block size");
    MulCtx* hcx = reinterpret_cast<MulCtx*>(rt().h_pinned);
    std::memset(hcx, 0, sizeof(MulCtx));
    std::memcpy(&hcx->tr, transcript, sizeof(DevTranscript));
    std::memcpy(&hcx->claim, input_claim, sizeof(Fr));
    H::Fr one = H::one(); std::memcpy(&hcx->scalar, &one, sizeof(Fr));
    HIP_TRY(hipMemcpyAsync(d_cx, hcx, sizeof(MulCtx), hipMemcpyHostToDevice, rt().stream));

    size_t len = P->left->len;
    size_t rounds_done = 0;
    int pending = 0;
    const void *curL = P->left->d, *curR = P->right->d;
    bool cur_i32 = P->left->is_i32;
    Fr* pp[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // ping-pong [which][L/R]
    int which = 0;

    if (n > (size_t)SC_TAIL_LOG) {
        HIP_TRY(hipMalloc(&pp[0][0], (len / 2) * sizeof(Fr))); HIP_TRY(hipMalloc(&pp[0][1], (len / 2) * sizeof(Fr)));
        HIP_TRY(hipMalloc(&pp[1][0], (len / 4) * sizeof(Fr))); HIP_TRY(hipMalloc(&pp[1][1], (len / 4) * sizeof(Fr)));
        {
            const size_t groups = len / 2;
            const int grid = grid_for(groups);
            SplitEqView E = view_for_round(P, 0);
            if (cur_i32) k_mul_eval<int32_t><<<grid, SC_THREADS, 0, rt().stream>>>((const int32_t*)curL, (const int32_t*)curR, E, groups, rt().d_partials, K);
            else k_mul_eval<Fr><<<grid, SC_THREADS, 0, rt().stream>>>((const Fr*)curL, (const Fr*)curR, E, groups, rt().d_partials, K);
            k_mul_fs_round<<<1, SC_THREADS, 0, rt().stream>>>(d_cx, rt().d_partials, grid, P->d_w + (n - 1), rt().d_proof, rt().d_chal, K, 1, mode);
            rounds_done = 1; pending = 1;
        }
        while (len > ((size_t)1 << SC_TAIL_LOG)) {
            const size_t groups_new = len / 4;
            const int grid = grid_for(groups_new);
            SplitEqView E = view_for_round(P, rounds_done);
            Fr* dL = pp[which][0]; Fr* dR = pp[which][1];
            if (cur_i32)
                k_mul_bind_eval<int32_t><<<grid, SC_THREADS, 0, rt().stream>>>((const int32_t*)curL, (const int32_t*)curR, dL, dR, E, groups_new, d_cx, rt().d_partials, K, hi_only);
            else
                k_mul_bind_eval<Fr><<<grid, SC_THREADS, 0, rt().stream>>>((const Fr*)curL, (const Fr*)curR, dL, dR, E, groups_new, d_cx, rt().d_partials, K, hi_only);
            curL = dL; curR = dR; cur_i32 = false; which ^= 1;
            len /= 2;
            k_mul_fs_round<<<1, SC_THREADS, 0, rt().stream>>>(d_cx, rt().d_partials, grid, P->d_w + (n - 1 - rounds_done),
                                                        rt().d_proof + rounds_done * 3, rt().d_chal + 2 * rounds_done, K, 0, mode);
            rounds_done += 1;
        }
    }
    {
        MulTailArgs A;
        A.L = curL; A.R = curR; A.len = (uint32_t)len; A.src_i32 = cur_i32 ? 1 : 0;
        A.e_out_tabs = P->d_eout; A.e_in_tabs = P->d_ein; A.w = P->d_w;
        A.n = (uint32_t)n; A.m = (uint32_t)P->m; A.round0 = (uint32_t)rounds_done;
        A.first = rounds_done == 0 ? 1 : 0; A.pending_bind = pending; A.challenge_mode = mode;
        k_mul_tail<<<1, SC_THREADS, 3 * (sizeof(Fr) << SC_TAIL_LOG), rt().stream>>>(A, d_cx, rt().d_proof, rt().d_chal, rt().d_finals, K);
    }
    hipError_t le = hipGetLastError();
    uint8_t* hp = reinterpret_cast<uint8_t*>(rt().h_pinned);
    const size_t proof_bytes = n * 3 * sizeof(Fr), chal_bytes = n * 2 * sizeof(uint64_t);
    if (le == hipSuccess) le = hipMemcpyAsync(hp, rt().d_proof, proof_bytes, hipMemcpyDeviceToHost, rt().stream);
    if (le == hipSuccess) le = hipMemcpyAsync(hp + 8192, rt().d_chal, chal_bytes, hipMemcpyDeviceToHost, rt().stream);
    if (le == hipSuccess) le = hipMemcpyAsync(hp + 12288, rt().d_finals, 3 * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream);
    if (le == hipSuccess) le = hipMemcpyAsync(hp + 16384, d_cx, sizeof(MulCtx), hipMemcpyDeviceToHost, rt().stream);
    if (le == hipSuccess) le = hipStreamSynchronize(rt().stream);
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) if (pp[a][b]) hipFree(pp[a][b]);
    if (le != hipSuccess) return fail(ATLAS_ENODEV, "sumcheck_prove_mul", le);
    std::memcpy(compressed_polys, hp, proof_bytes);
    std::memcpy(challenges, hp + 8192, chal_bytes);
    std::memcpy(final_claims, hp + 12288, 3 * sizeof(Fr));
    std::memcpy(transcript, hp + 16384, sizeof(DevTranscript));
    P->consumed = true;
    return ATLAS_OK;
}

// ---- trait-shaped round API (the caller owns the loop and the transcript, e.g. inside
//      BatchedSumcheck::prove): MulProver::compute_message / ingest_challenge, mul.rs:160-185
int atlas_mul_compute_message(atlas_mul_prover_t P, size_t round, const atlas_fr_t* previous_claim, atlas_fr_t* coeffs_out,
                              size_t* n_coeffs) {
    NEED_INIT();
    if (!P || !previous_claim || !coeffs_out || !n_coeffs) return fail(ATLAS_EINVAL, "mul_compute_message");
    if (P->consumed || round >= P->n || P->left->len != ((size_t)1 << (P->n - round))) return fail(ATLAS_ESTATE, "mul_compute_message: round out of order");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    const ScConsts K = make_consts();
    const size_t groups = P->left->len / 2;
    const int grid = grid_for(groups);
    SplitEqView E = view_for_round(P, round);
    if (P->left->is_i32) k_mul_eval<int32_t><<<grid, SC_THREADS, 0, rt().stream>>>((const int32_t*)P->left->d, (const int32_t*)P->right->d, E, groups, rt().d_partials, K);
    else k_mul_eval<Fr><<<grid, SC_THREADS, 0, rt().stream>>>((const Fr*)P->left->d, (const Fr*)P->right->d, E, groups, rt().d_partials, K);
    k_reduce1<<<1, SC_THREADS, 0, rt().stream>>>(rt().d_partials, grid, rt().d_finals + 3, 3);
    HIP_TRY(hipMemcpyAsync(rt().h_pinned, rt().d_finals + 3, 3 * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    H::Fr s[3]; std::memcpy(s, rt().h_pinned, sizeof s);
    // gruen_poly_deg_3 (split_eq_poly.rs:379-429) with q(1) taken from the third running sum
    const H::Fr w_cur = P->w[P->n - 1 - round];
    const H::Fr eq1 = H::mul(P->scalar, w_cur), eq0 = H::sub(P->scalar, eq1), eqm = H::sub(eq1, eq0);
    const H::Fr eq2 = H::add(eq1, eqm), eq3 = H::add(eq2, eqm);
    const H::Fr q0 = s[0], q1 = s[1], e2 = H::add(s[2], s[2]);
    const H::Fr q2 = H::add(H::sub(H::add(q1, q1), q0), e2);
    const H::Fr q3 = H::add(H::add(H::sub(H::add(q2, q1), q0), e2), e2);
    const H::Fr ev[3] = {H::mul(eq0, q0), H::mul(eq2, q2), H::mul(eq3, q3)};
    H::Fr c[4];
    const int nc = H::unipoly_from_evals_and_hint(*reinterpret_cast<const H::Fr*>(previous_claim), ev, 3, c);
    std::memcpy(coeffs_out, c, nc * sizeof(H::Fr));
    *n_coeffs = (size_t)nc;
    return ATLAS_OK;
}

int atlas_mul_ingest_challenge(atlas_mul_prover_t P, const atlas_u128_t* r_j, size_t round) {
    NEED_INIT();
    if (!P || !r_j) return fail(ATLAS_EINVAL, "mul_ingest_challenge");
    if (P->consumed || round >= P->n || P->left->len != ((size_t)1 << (P->n - round))) return fail(ATLAS_ESTATE, "mul_ingest_challenge: round out of order");
    int rc = atlas_poly_bind(P->left, r_j, ATLAS_LOW_TO_HIGH);
    if (!rc) rc = atlas_poly_bind(P->right, r_j, ATLAS_LOW_TO_HIGH);
    if (rc) return rc;
    // GruenSplitEqPolynomial::bind: current_scalar *= 1 - w - r + 2 w r (split_eq_poly.rs:336-339)
    const H::Fr r = H::challenge_to_fr(r_j->lo, r_j->hi, rt().challenge_mode);
    const H::Fr w_cur = P->w[P->n - 1 - round];
    const H::Fr wr = H::mul(w_cur, r);
    P->scalar = H::mul(P->scalar, H::add(H::add(H::sub(H::sub(H::one(), w_cur), r), wr), wr));
    return ATLAS_OK;
}

int atlas_mul_final_claims(atlas_mul_prover_t P, atlas_fr_t out[3]) {
    NEED_INIT();
    if (!P || !out) return fail(ATLAS_EINVAL, "mul_final_claims");
    if (P->consumed) return fail(ATLAS_ESTATE, "mul_final_claims: prover consumed by atlas_sumcheck_prove_mul");
    if (P->left->len != 1) return fail(ATLAS_ESTATE, "mul_final_claims: rounds remaining");
    int rc = atlas_poly_final_claim(P->left, &out[0]);
    if (!rc) rc = atlas_poly_final_claim(P->right, &out[1]);
    if (rc) return rc;
    std::memcpy(&out[2], &P->scalar, 32);
    return ATLAS_OK;
}

size_t atlas_mul_num_rounds(atlas_mul_prover_t P) { return P ? P->n : 0; }

}  // extern "C"
