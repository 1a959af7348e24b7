// EQ tables, MLE evaluation and the Gruen split-eq product sumcheck on gfx950.
//
// Device counterpart of (paths under /root/reference):
//   EqPolynomial::{evals, evals_parallel, evals_cached}   joltworks/src/poly/eq_poly.rs:77-101,149-252
//   DensePolynomial::evaluate / split_eq_evaluate          joltworks/src/poly/dense_mlpoly.rs:265-305
//   GruenSplitEqPolynomial::{new, bind, gruen_poly_deg_3, par_fold_out_in_unreduced}
//                                                          joltworks/src/poly/split_eq_poly.rs:86-145,331-372,379-429,526-597
//   MulProver::{compute_message, ingest_challenge}         jolt-atlas-core/src/onnx_proof/ops/mul.rs:160-185
//
// Split-eq layout: w = [w_out (m = n/2 vars) | w_in (n-1-m vars) | w_last], bound
// LowToHigh.  In round j the weight of group g over the n-j-1 still-free variables is
// eq(w[..n-j-1], g) = E_out[g >> in_bits] * E_in[g & in_mask], E_* being PREFIX tables
// (evals_cached): all prefixes of one table family live in one buffer, table k at offset
// 2^k - 1.  Operands are bound low-to-high into a ping-pong buffer: a thread reads four
// consecutive coefficients (128 B), writes the two bound ones (64 B) and accumulates the
// next round's three sums from them.
//
// The reference recovers q(1) by a field division (split_eq_poly.rs:410-413); a modular
// inverse is ~380 dependent multiplications — hopeless on one wavefront — so the device
// accumulates q(1) = sum w_g * lo(1) * ro(1) directly (one more product per group).  For an
// honest prover both give the same value (exact arithmetic), hence the same proof bytes.
#pragma once
#include "sumcheck_kernels.hip.h"

namespace atlas {

// ---- EQ tables -----------------------------------------------------------------------
// One doubling pass of evals_parallel (eq_poly.rs:225-252), in place: y = x*r; x -= y.
static __global__ __launch_bounds__(SC_THREADS) void k_eq_double(Fr* ev, size_t size, Fr r) {
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < size; i += (size_t)gridDim.x * SC_THREADS) {
        Fr x = fe_load(ev + i);
        Fr y = fr_mul(x, r);
        fe_store(ev + i + size, y);
        fe_store(ev + i, fr_sub(x, y));
    }
}

// first `levels` doubling passes in one workgroup (tables up to 2^12 entries), same index
// convention as k_eq_double: pass p uses r[n-1-p]
static __global__ __launch_bounds__(1024) void k_eq_head(Fr* ev, const Fr* r, uint32_t n, uint32_t levels, Fr scale) {
    // The first min(levels, 10) passes grow the table in LDS (in place: pass p reads entry i < 2^p and writes i and i + 2^p), the
    // remaining ones (at most two: levels <= 12) are folded into the write-out as direct factors.  Every constructor of an instance
    // over an eq table starts with this launch (~1400 per nanoGPT-shaped proof, on the critical path); with each pass going through
    // HBM and a fence it took 17.6 us.  Exact field arithmetic: the same residues (x - x r = x (1 - r)).
    __shared__ Fr tab[1024];
    const uint32_t LL = levels < 10 ? levels : 10;
    if (threadIdx.x == 0) tab[0] = scale;
    __syncthreads();
    for (uint32_t p = 0; p < LL; p++) {
        const uint32_t size = 1u << p;
        const Fr rp = fe_load(r + (n - 1 - p));
        const bool act = threadIdx.x < size;
        Fr x = fe_zero(), y = fe_zero();
        if (act) { x = tab[threadIdx.x]; y = fr_mul(x, rp); }
        if (act) { tab[threadIdx.x + size] = y; tab[threadIdx.x] = fr_sub(x, y); }
        __syncthreads();
    }
    Fr hi_r[2], hi_c[2];
    for (uint32_t q = 0; q + LL < levels; q++) { hi_r[q] = fe_load(r + (n - 1 - (LL + q))); hi_c[q] = fr_sub(fr_one(), hi_r[q]); }
    const uint32_t total = 1u << levels, mask = (1u << LL) - 1;
    for (uint32_t i = threadIdx.x; i < total; i += 1024) {
        Fr v = tab[i & mask];
        for (uint32_t q = 0; q + LL < levels; q++) v = fr_mul(v, ((i >> (LL + q)) & 1) ? hi_r[q] : hi_c[q]);
        fe_store(ev + i, v);
    }
}

// evals_cached (eq_poly.rs:174-192): all prefix tables of w[0..k), table j at offset 2^j-1.
// One workgroup; k <= 12.
static __global__ __launch_bounds__(1024) void k_eq_cached(Fr* tabs, const Fr* w, uint32_t k) {
    if (threadIdx.x == 0) fe_store(tabs, fr_one());
    __syncthreads();
    for (uint32_t j = 0; j < k; j++) {
        const Fr wj = fe_load(w + j);
        const Fr* cur = tabs + ((1u << j) - 1);
        Fr* nxt = tabs + ((2u << j) - 1);
        for (uint32_t i = threadIdx.x; i < (1u << j); i += 1024) {
            Fr s = fe_load(cur + i);
            Fr hi = fr_mul(s, wj);
            fe_store(nxt + 2 * i + 1, hi);
            fe_store(nxt + 2 * i, fr_sub(s, hi));
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ---- MLE evaluation: sum_i eq1[i >> n2] * eq2[i & mask] * Z[i] ------------------------
template <class T>
__global__ __launch_bounds__(SC_THREADS) void k_mle_evaluate(const T* __restrict__ Z, size_t len,
                                                             const Fr* __restrict__ eq1, const Fr* __restrict__ eq2,
                                                             uint32_t n2, Fr* partials, ScConsts K) {
    Fr acc[1];
    acc[0] = fe_zero();
    const size_t mask = ((size_t)1 << n2) - 1;
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < len; i += (size_t)gridDim.x * SC_THREADS) {
        Fr z = Src<T>::get(Z, i, K);
        Fr t = fr_mul(fr_mul(fe_load(eq2 + (i & mask)), z), fe_load(eq1 + (i >> n2)));
        acc[0] = fr_add(acc[0], t);
    }
    block_reduce_store<1>(acc, partials);
}

// ---- Gruen split-eq product (Mul) -----------------------------------------------------
struct SplitEqView {
    const Fr* e_out;     // current E_out table (prefix table of w_out)
    const Fr* e_in;      // current E_in table
    uint32_t in_bits;    // log2(len(E_in))
};

struct MulCtx {          // control block of one split-eq instance (HBM)
    DevTranscript tr;
    Fr claim;            // previous_claim
    Fr r;                // challenge just drawn (Montgomery limbs)
    Fr scalar;           // GruenSplitEqPolynomial::current_scalar
};

// three running sums for one group g from the pair values (a0,a1) of L and (b0,b1) of R
__device__ __forceinline__ void mul_accumulate(Fr acc[3], const Fr& a0, const Fr& a1, const Fr& b0, const Fr& b1,
                                               const SplitEqView& E, size_t g) {
    const Fr wgt = fr_mul(fe_load(E.e_out + (g >> E.in_bits)), fe_load(E.e_in + (g & (((size_t)1 << E.in_bits) - 1))));
    const Fr c0 = fr_mul(a0, b0);                         // q(0) term  (mul.rs:176)
    const Fr c1 = fr_mul(a1, b1);                         // q(1) term
    const Fr e = fr_mul(fr_sub(a1, a0), fr_sub(b1, b0));  // q(inf) term (mul.rs:177)
    acc[0] = fr_add(acc[0], fr_mul(wgt, c0));
    acc[1] = fr_add(acc[1], fr_mul(wgt, c1));
    acc[2] = fr_add(acc[2], fr_mul(wgt, e));
}

// round-0 message: no bind. n_groups = len/2.
template <class T>
__global__ __launch_bounds__(SC_THREADS) void k_mul_eval(const T* __restrict__ L, const T* __restrict__ R,
                                                         SplitEqView E, size_t n_groups, Fr* partials, ScConsts K) {
    Fr acc[3];
    acc[0] = fe_zero(); acc[1] = fe_zero(); acc[2] = fe_zero();
    for (size_t g = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; g < n_groups; g += (size_t)gridDim.x * SC_THREADS) {
        Fr a0 = Src<T>::get(L, 2 * g, K), a1 = Src<T>::get(L, 2 * g + 1, K);
        Fr b0 = Src<T>::get(R, 2 * g, K), b1 = Src<T>::get(R, 2 * g + 1, K);
        mul_accumulate(acc, a0, a1, b0, b1, E, g);
    }
    block_reduce_store<3>(acc, partials);
}

// fused: bind L, R low-to-high with r (src -> dst, len -> len/2) and accumulate the next
// round's sums over the len/4 new groups.  T = Fr or int32 (first bind of I32Scalars).
template <class T>
__global__ __launch_bounds__(SC_THREADS) void k_mul_bind_eval(const T* Lsrc, const T* Rsrc, Fr* Ldst, Fr* Rdst,
                                                              SplitEqView E, size_t n_groups_new, const MulCtx* cx,
                                                              Fr* partials, ScConsts K, int r_hi_only) {
    Fr acc[3];
    acc[0] = fe_zero(); acc[1] = fe_zero(); acc[2] = fe_zero();
    const Fr r = fe_load(&cx->r);
    Fr r_s64;
    if constexpr (sizeof(T) == 4) r_s64 = fr_mul(r, K.k64);
    const bool hi = r_hi_only != 0;
    for (size_t g = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; g < n_groups_new; g += (size_t)gridDim.x * SC_THREADS) {
        Fr a0, a1, b0, b1;
        if constexpr (sizeof(T) == 4) {
            a0 = bind_pair_i32(Lsrc[4 * g], Lsrc[4 * g + 1], nullptr, K, r_s64);
            a1 = bind_pair_i32(Lsrc[4 * g + 2], Lsrc[4 * g + 3], nullptr, K, r_s64);
            b0 = bind_pair_i32(Rsrc[4 * g], Rsrc[4 * g + 1], nullptr, K, r_s64);
            b1 = bind_pair_i32(Rsrc[4 * g + 2], Rsrc[4 * g + 3], nullptr, K, r_s64);
        } else {
            const Fr* Lp = reinterpret_cast<const Fr*>(Lsrc) + 4 * g;
            const Fr* Rp = reinterpret_cast<const Fr*>(Rsrc) + 4 * g;
            a0 = bind_pair(fe_load(Lp), fe_load(Lp + 1), r, hi);
            a1 = bind_pair(fe_load(Lp + 2), fe_load(Lp + 3), r, hi);
            b0 = bind_pair(fe_load(Rp), fe_load(Rp + 1), r, hi);
            b1 = bind_pair(fe_load(Rp + 2), fe_load(Rp + 3), r, hi);
        }
        fe_store(Ldst + 2 * g, a0); fe_store(Ldst + 2 * g + 1, a1);
        fe_store(Rdst + 2 * g, b0); fe_store(Rdst + 2 * g + 1, b1);
        mul_accumulate(acc, a0, a1, b0, b1, E, g);
    }
    block_reduce_store<3>(acc, partials);
}

// gruen_poly_deg_3 (split_eq_poly.rs:379-429) from q(0), q(1), e and the eq factor, then the
// common round spine (compress, absorb, squeeze, evaluate) and GruenSplitEq::bind's scalar
// update.  Runs on one wavefront; every lane holds the same field values.
__device__ __forceinline__ void mul_round_wave(WaveTranscript& T, FsScratch* S, const WaveBlakeSched& W,
                                               uint32_t lane, const Fr sums[3], const Fr& w_cur, Fr& claim,
                                               Fr& scalar, Fr& r_out, Fr* proof_row, uint64_t* chal_row,
                                               const ScConsts& K, int challenge_mode) {
    const Fr eq1 = fr_mul(scalar, w_cur);
    const Fr eq0 = fr_sub(scalar, eq1);
    const Fr eqm = fr_sub(eq1, eq0);
    const Fr eq2 = fr_add(eq1, eqm);
    const Fr eq3 = fr_add(eq2, eqm);
    const Fr q0 = sums[0], q1 = sums[1];
    const Fr e2 = fr_dbl(sums[2]);
    const Fr q2 = fr_add(fr_sub(fr_add(q1, q1), q0), e2);
    const Fr q3 = fr_add(fr_add(fr_sub(fr_add(q2, q1), q0), e2), e2);
    const Fr s0 = fr_mul(eq0, q0);
    const Fr s1 = fr_sub(claim, s0);              // = eq1*q1 for a consistent claim (split_eq_poly.rs:408)
    const Fr s2 = fr_mul(eq2, q2);
    const Fr s3 = fr_mul(eq3, q3);
    // UniPoly::from_evals degree 3 (unipoly.rs:78-89)
    Fr c[4];
    c[0] = s0;
    Fr u = fr_sub(s1, s2);
    u = fr_add(fr_dbl(u), u);
    c[3] = fr_mul(fr_add(fr_sub(s3, s0), u), K.six_inv);
    Fr t = fr_half(fr_add(fr_sub(fr_sub(s0, s1), s1), s2));
    c[2] = fr_sub(fr_sub(fr_sub(t, c[3]), c[3]), c[3]);
    c[1] = fr_sub(fr_sub(fr_sub(s1, s0), c[2]), c[3]);
    const Fr k0 = fe_from_mont<FrParams>(c[0]), k2 = fe_from_mont<FrParams>(c[2]), k3 = fe_from_mont<FrParams>(c[3]);
    wt_append_label(T, &S->wt, W, lane, K.lbl_begin);
    wt_append_canonical(T, &S->wt, W, lane, k0);
    wt_append_canonical(T, &S->wt, W, lane, k2);
    wt_append_canonical(T, &S->wt, W, lane, k3);
    wt_append_label(T, &S->wt, W, lane, K.lbl_end);
    uint64_t lo, hi;
    wt_challenge_u128(T, &S->wt, W, lane, lo, hi);
    const Fr r = challenge_to_mont(lo, hi, challenge_mode);
    Fr ev = c[0], pw = r;
#pragma unroll
    for (int k = 1; k <= 3; k++) {
        ev = fr_add(ev, fr_mul(pw, c[k]));
        if (k < 3) pw = fr_mul(pw, r);
    }
    claim = ev;
    // current_scalar *= 1 - w - r + 2 w r   (split_eq_poly.rs:336-339)
    const Fr wr = fr_mul(w_cur, r);
    scalar = fr_mul(scalar, fr_add(fr_add(fr_sub(fr_sub(fr_one(), w_cur), r), wr), wr));
    r_out = r;
    if (lane == 0) {
        fe_store(proof_row + 0, c[0]); fe_store(proof_row + 1, c[2]); fe_store(proof_row + 2, c[3]);
        chal_row[0] = lo; chal_row[1] = hi;
    }
}

static __global__ __launch_bounds__(SC_THREADS) void k_mul_fs_round(MulCtx* cx, const Fr* partials, int n_partials,
                                                             const Fr* w_cur_ptr, Fr* proof_row, uint64_t* chal_row,
                                                             ScConsts K, int first, int challenge_mode) {
    __shared__ Fr red[SC_THREADS / 64][3];
    __shared__ FsScratch fs;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Fr acc[3];
    acc[0] = fe_zero(); acc[1] = fe_zero(); acc[2] = fe_zero();
    for (int b = threadIdx.x; b < n_partials; b += SC_THREADS)
#pragma unroll
        for (int k = 0; k < 3; k++) acc[k] = fr_add(acc[k], fe_load(partials + (size_t)b * 3 + k));
#pragma unroll
    for (int k = 0; k < 3; k++) {
        Fr s = fr_wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    if (wave == 0) wt_init_lds(&fs.wt, lane);
    __syncthreads();
    if (wave != 0) return;
    const uint32_t vz = vgpr_zero();
    const WaveBlakeSched W = wave_blake_sched(lane & 3);
    Fr sums[3];
    const Fr* rp = &red[0][0] + vz;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        Fr s = rp[k];
        for (int w = 1; w < SC_THREADS / 64; w++) s = fr_add(s, rp[w * 3 + k]);
        sums[k] = s;
    }
    WaveTranscript T = wt_load(&cx->tr, lane, vz);
    Fr claim = fe_load(&cx->claim + vz), scalar = fe_load(&cx->scalar + vz);
    const Fr w_cur = fe_load(w_cur_ptr + vz);
    if (first) wt_append_canonical(T, &fs.wt, W, lane, fe_from_mont<FrParams>(claim));
    Fr r;
    mul_round_wave(T, &fs, W, lane, sums, w_cur, claim, scalar, r, proof_row, chal_row, K, challenge_mode);
    wt_store(&cx->tr, T, lane);
    if (lane == 0) {
        fe_store(&cx->claim, claim);
        fe_store(&cx->scalar, scalar);
        fe_store(&cx->r, r);
    }
}

// tail: remaining rounds with len <= 2^SC_TAIL_LOG in one launch (operands ping-pong in LDS)
struct MulTailArgs {
    const void* L; const void* R;
    uint32_t len;                 // current operand length (>= 1)
    int src_i32;
    const Fr* e_out_tabs;         // prefix tables of w_out (table k at 2^k-1), m vars
    const Fr* e_in_tabs;          // prefix tables of w_in, n-1-m vars
    const Fr* w;                  // full challenge vector w (n Fr)
    uint32_t n, m;                // n = total rounds, m = n/2
    uint32_t round0;
    int first, pending_bind, challenge_mode;
};

static __global__ __launch_bounds__(SC_THREADS) void k_mul_tail(MulTailArgs A, MulCtx* cx, Fr* proof, uint64_t* chal,
                                                         Fr* finals, ScConsts K) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Fr* bufL[2]; Fr* bufR[2];
    bufL[0] = reinterpret_cast<Fr*>(smem_raw);
    bufR[0] = bufL[0] + (1u << SC_TAIL_LOG);
    bufL[1] = bufR[0] + (1u << SC_TAIL_LOG);
    bufR[1] = bufL[1] + (1u << (SC_TAIL_LOG - 1));
    __shared__ Fr red[SC_THREADS / 64][3];
    __shared__ Fr s_r;
    __shared__ FsScratch fs;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool hi_only = A.challenge_mode == 0;
    uint32_t len = A.len;
    int cur = 0;
    for (uint32_t i = tid; i < len; i += SC_THREADS) {
        if (A.src_i32) {
            bufL[0][i] = fr_from_i32(reinterpret_cast<const int32_t*>(A.L)[i], K.k32);
            bufR[0][i] = fr_from_i32(reinterpret_cast<const int32_t*>(A.R)[i], K.k32);
        } else {
            bufL[0][i] = fe_load(reinterpret_cast<const Fr*>(A.L) + i);
            bufR[0][i] = fe_load(reinterpret_cast<const Fr*>(A.R) + i);
        }
    }
    WaveTranscript T; WaveBlakeSched W; Fr claim, scalar; uint32_t vz = 0;
    if (wave == 0) {
        vz = vgpr_zero();
        W = wave_blake_sched(lane & 3);
        wt_init_lds(&fs.wt, lane);
        T = wt_load(&cx->tr, lane, vz);
        claim = fe_load(&cx->claim + vz);
        scalar = fe_load(&cx->scalar + vz);
        if (A.first) wt_append_canonical(T, &fs.wt, W, lane, fe_from_mont<FrParams>(claim));
    }
    if (A.pending_bind && tid == 0) s_r = fe_load(&cx->r);
    __syncthreads();
    int pending = A.pending_bind;
    uint32_t round = A.round0;
    for (;;) {
        if (pending) {   // ingest_challenge: bind low-to-high into the other LDS buffer
            const Fr r = s_r;
            const uint32_t half = len / 2;
            Fr *sL = bufL[cur], *sR = bufR[cur], *dL = bufL[cur ^ 1], *dR = bufR[cur ^ 1];
            // dst buffers alternate between the full-size and half-size regions; the half-size
            // region is big enough because len halves before it is written
            for (uint32_t i = tid; i < half; i += SC_THREADS) {
                dL[i] = bind_pair(sL[2 * i], sL[2 * i + 1], r, hi_only);
                dR[i] = bind_pair(sR[2 * i], sR[2 * i + 1], r, hi_only);
            }
            len = half;
            cur ^= 1;
            __syncthreads();
        }
        if (round == A.n) break;
        // compute_message(round): n_free = n - round - 1 variables besides the current one
        const uint32_t n_free = A.n - round - 1;
        const uint32_t out_bits = n_free < A.m ? n_free : A.m;
        const uint32_t in_bits = n_free - out_bits;
        SplitEqView E;
        E.e_out = A.e_out_tabs + ((1u << out_bits) - 1);
        E.e_in = A.e_in_tabs + ((1u << in_bits) - 1);
        E.in_bits = in_bits;
        Fr acc[3];
        acc[0] = fe_zero(); acc[1] = fe_zero(); acc[2] = fe_zero();
        const Fr *sL = bufL[cur], *sR = bufR[cur];
        for (uint32_t g = tid; g < len / 2; g += SC_THREADS)
            mul_accumulate(acc, sL[2 * g], sL[2 * g + 1], sR[2 * g], sR[2 * g + 1], E, g);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            Fr s = fr_wave_sum(acc[k]);
            if (lane == 0) red[wave][k] = s;
        }
        __syncthreads();
        if (wave == 0) {
            Fr sums[3];
            const Fr* rp = &red[0][0] + vz;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                Fr s = rp[k];
                for (int w = 1; w < SC_THREADS / 64; w++) s = fr_add(s, rp[w * 3 + k]);
                sums[k] = s;
            }
            const Fr w_cur = fe_load(A.w + (A.n - 1 - round) + vz);
            Fr r;
            mul_round_wave(T, &fs, W, lane, sums, w_cur, claim, scalar, r, proof + (size_t)round * 3,
                           chal + 2 * (size_t)round, K, A.challenge_mode);
            if (lane == 0) s_r = r;
        }
        __syncthreads();
        pending = 1;
        round += 1;
    }
    if (wave == 0) wt_store(&cx->tr, T, lane);
    if (tid == 0) {
        fe_store(&cx->claim, claim);
        fe_store(&cx->scalar, scalar);
        fe_store(&cx->r, s_r);
        fe_store(finals + 0, bufL[cur][0]);
        fe_store(finals + 1, bufR[cur][0]);
        fe_store(finals + 2, scalar);
    }
}

}  // namespace atlas
