// Quantised model execution on the device: Model::trace (atlas-onnx-tracer/src/model/trace.rs:8-21 ->
// execute.rs) for the operators the proving graph knows — every node's Tensor<i32> output stays in HBM, and the nodes
// that fuse a rescale keep their i64 accumulation / remainder / lookup indices (the witness of f1) from the same pass.
// Operator semantics follow atlas-onnx-tracer/src/ops/*.rs (cited per kernel).  Tensors are row-major; every dimension
// must be a power of two (the reference pads each dimension to the next power of two before building MLEs,
// tensor/mod.rs:474-481 — a graph description pads its shapes up front, e.g. vocab 65 -> 128).
#include <hip/hip_runtime.h>

#include <cmath>
#include <mutex>

#include "graph_state.hip.h"

using gr::Node;

namespace {

constexpr int MAXR = 6;
struct Strides { uint32_t n; uint32_t dim[MAXR]; uint32_t a[MAXR]; uint32_t b[MAXR]; };

// acc[o] = sum_{l<K} L[loff(o) + l lsk] R[roff(o) + l rsk] in i64: einsum_acc_i64 (ops/einsum.rs:34-213) for the layouts with one
// contraction axis; o runs over the output dims (row-major), S.a / S.b = each output axis' stride in the left / right operand
__global__ __launch_bounds__(256) void k_einsum_acc_generic(const int32_t* __restrict__ L, const int32_t* __restrict__ R, Strides S, uint32_t K, uint32_t lsk,
                                                            uint32_t rsk, size_t T, int64_t* __restrict__ acc) {
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < T; o += (size_t)gridDim.x * 256) {
        size_t rem = o, lo = 0, ro = 0;
        for (int d = (int)S.n - 1; d >= 0; d--) { const size_t c = rem % S.dim[d]; rem /= S.dim[d]; lo += c * S.a[d]; ro += c * S.b[d]; }
        int64_t s = 0;
        for (uint32_t l = 0; l < K; l++) s += (int64_t)L[lo + (size_t)l * lsk] * (int64_t)R[ro + (size_t)l * rsk];
        acc[o] = s;
    }
}
// Cube accumulators x^3 (ops/cube.rs:33-46)
__global__ __launch_bounds__(256) void k_cube_acc(const int32_t* __restrict__ x, size_t n, int64_t* __restrict__ acc) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const int64_t v = x[i]; acc[i] = v * v * v; }
}
// And = a * b, Iff = mask * a + (1 - mask) * b on 0/1 masks (tensor/ops.rs:42-71, 206-257)
__global__ __launch_bounds__(256) void k_select(int op, const int32_t* __restrict__ a, const int32_t* __restrict__ b, const int32_t* __restrict__ c, size_t n,
                                                int32_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = op == ATLAS_OP_AND ? a[i] * b[i] : (a[i] ? b[i] : c[i]);
}
// out[o] = in[sum_d coord_d(o) stride_d]: MoveAxis, Broadcast (stride 0 on expanded axes), Slice (with a base offset)
__global__ __launch_bounds__(256) void k_gather_strided(const int32_t* __restrict__ in, Strides S, size_t base, size_t T, int32_t* __restrict__ out) {
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < T; o += (size_t)gridDim.x * 256) {
        size_t rem = o, off = base;
        for (int d = (int)S.n - 1; d >= 0; d--) { off += (rem % S.dim[d]) * S.a[d]; rem /= S.dim[d]; }
        out[o] = in[off];
    }
}

// Neg / Clamp (tensor::ops::nonlinearities::clamp, tensor/ops.rs:3216-3220; the lookup index is `value as u32 as u64`)
__global__ __launch_bounds__(256) void k_neg(const int32_t* __restrict__ x, size_t n, int32_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (int32_t)(0u - (uint32_t)x[i]);
}
__global__ __launch_bounds__(256) void k_clamp_witness(const int32_t* __restrict__ x, size_t n, int32_t bound, int32_t* __restrict__ out, uint64_t* __restrict__ lookups) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int32_t v = x[i];
        out[i] = v < -bound ? -bound : v > bound - 1 ? bound - 1 : v;
        lookups[i] = (uint64_t)(uint32_t)v;
    }
}
// Concat: out[o(i)] = in[i], i over one operand (S.dim = its dims, S.a = the OUTPUT's strides, base = its offset along the axis)
__global__ __launch_bounds__(256) void k_scatter_strided(const int32_t* __restrict__ in, Strides S, size_t base, size_t T, int32_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < T; i += (size_t)gridDim.x * 256) {
        size_t rem = i, off = base;
        for (int d = (int)S.n - 1; d >= 0; d--) { off += (rem % S.dim[d]) * S.a[d]; rem /= S.dim[d]; }
        out[off] = in[i];
    }
}

// Sum over one axis of a [m][n] tensor in i64 (sum_axes_i64, ops/sum.rs:18-58): axis 0 -> n outputs, axis 1 -> m outputs; the clamped
// output and the clamp lookup index; with `squares`: sum of squares (mos_acc_i64, ops/mean_of_squares.rs:19-52)
__global__ __launch_bounds__(256) void k_sum_axis(const int32_t* __restrict__ x, uint32_t m, uint32_t n, int axis, int squares, int64_t* __restrict__ acc) {
    const uint32_t outs = axis == 0 ? n : m, len = axis == 0 ? m : n;
    for (uint32_t o = blockIdx.x * 256 + threadIdx.x; o < outs; o += gridDim.x * 256) {
        int64_t s = 0;
        for (uint32_t l = 0; l < len; l++) { const int64_t v = axis == 0 ? x[(size_t)l * n + o] : x[(size_t)o * n + l]; s += squares ? v * v : v; }
        acc[o] = s;
    }
}
__global__ __launch_bounds__(256) void k_clamp_acc(const int64_t* __restrict__ acc, size_t n, int32_t* __restrict__ out, uint64_t* __restrict__ idx) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int64_t a = acc[i];
        out[i] = (int32_t)(a > 2147483647ll ? 2147483647ll : a < -2147483648ll ? -2147483648ll : a);
        idx[i] = (uint64_t)a;
    }
}
// floor division with the remainder of the divisor's sign (ops/div.rs:9-23, scalar_const_div.rs:9-22; utils/mod.rs:24-30 adjusted_remainder);
// b == nullptr: the scalar constant `c`
__global__ __launch_bounds__(256) void k_floor_div(const int32_t* __restrict__ a, const int32_t* __restrict__ b, int32_t c, size_t n, int32_t* __restrict__ q,
                                                   int32_t* __restrict__ r) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int32_t x = a[i], d = b ? b[i] : c;
        int32_t qq = d ? x / d : 0, rr = d ? x % d : 0;
        if ((rr < 0 && d > 0) || (rr > 0 && d < 0)) { qq -= 1; rr += d; }
        q[i] = qq; r[i] = rr;
    }
}
// mean of squares: quotient / remainder of the accumulation by D = count 2^scale (mos_intermediate_and_remainder), the clamp lookup
// index, the clamped output and the constant bound tensor of the range check
__global__ __launch_bounds__(256) void k_mos_rebase(int64_t* __restrict__ acc /* in: sums, out: quotients */, size_t n, int64_t D, int32_t* __restrict__ rem,
                                                    int32_t* __restrict__ out, uint64_t* __restrict__ cidx, int32_t* __restrict__ bound) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int64_t a = acc[i];
        int64_t q = a / D, r = a % D;
        if (r < 0) { r += D; q -= 1; }
        acc[i] = q; rem[i] = (int32_t)r; cidx[i] = (uint64_t)q; bound[i] = (int32_t)D;
        out[i] = (int32_t)(q > 2147483647ll ? 2147483647ll : q < -2147483648ll ? -2147483648ll : q);
    }
}
// Rsqrt (ops/rsqrt.rs:18-29 + jolt-atlas-core ops/rsqrt.rs:238-250): out = isqrt(2^(3 scale) / x), quotient, both remainders and the
// second range check's bound 2 out + 1
__global__ __launch_bounds__(256) void k_rsqrt(const int32_t* __restrict__ x, size_t n, int64_t s_cubed, int32_t* __restrict__ out, int64_t* __restrict__ quot,
                                               int32_t* __restrict__ div_rem, int32_t* __restrict__ sqrt_rem, int32_t* __restrict__ bound) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int64_t v = x[i];
        int64_t q = 0, dr = 0, o = 0;
        if (v > 0) {
            q = s_cubed / v; dr = s_cubed % v;
            o = (int64_t)sqrt((double)q);
            while (o * o > q) o--;
            while ((o + 1) * (o + 1) <= q) o++;
        }
        out[i] = (int32_t)o; quot[i] = q; div_rem[i] = (int32_t)dr; sqrt_rem[i] = (int32_t)(q - o * o); bound[i] = (int32_t)(2 * o + 1);
    }
}

// Tanh (atlas-onnx-tracer ops/tanh.rs:9-14): clamp to [-2^B, 2^B - 1], then the table; witness: the clamped tensor, its (B+1)-bit
// two's-complement table index (n_bits_to_usize) and the raw input's `as u32 as u64` lookup index of the clamp table
__global__ __launch_bounds__(256) void k_tanh(const int32_t* __restrict__ x, size_t n, uint32_t B, const int32_t* __restrict__ table, int32_t* __restrict__ out,
                                              int32_t* __restrict__ clamped, uint64_t* __restrict__ idx_small, uint64_t* __restrict__ idx_clamp) {
    const int32_t bound = 1 << B;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int32_t v = x[i], c = v < -bound ? -bound : v > bound - 1 ? bound - 1 : v;
        const uint32_t k = c < 0 ? (uint32_t)(c + (1 << (B + 1))) : (uint32_t)c;
        clamped[i] = c; idx_small[i] = k; idx_clamp[i] = (uint64_t)(uint32_t)v; out[i] = table[k];
    }
}
// Sin / Cos (eval_trig, atlas-onnx-tracer/src/ops/mod.rs:317-336; compute_division, neural_teleport/division.rs:48-66): Euclidean quotient and
// remainder by the period modulus, the remainder shifted down, the table (already rescaled by 2^DOWNSCALE_BITS)
__global__ __launch_bounds__(256) void k_trig(const int32_t* __restrict__ x, size_t n, int32_t tau, uint32_t shift, const int32_t* __restrict__ table, int32_t* __restrict__ out,
                                              int32_t* __restrict__ quot, int32_t* __restrict__ rem, int32_t* __restrict__ down, uint64_t* __restrict__ idx_rem,
                                              uint64_t* __restrict__ idx_down) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int32_t v = x[i];
        int32_t r = v % tau, q = v / tau;
        if (r < 0) { r += tau; q -= 1; }
        const int32_t d = r >> shift;
        quot[i] = q; rem[i] = r; down[i] = d; idx_rem[i] = (uint64_t)(uint32_t)r; idx_down[i] = (uint64_t)(uint32_t)d; out[i] = table[d];
    }
}
__global__ __launch_bounds__(256) void k_fill_i32(int32_t* p, size_t n, int32_t v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
// Gather along axis 0 (ops/gather.rs): out[j][w] = dict[idx[j]][w]
// An index outside [0, V) — the reference's tensor indexing panics on it — raises *err (the trace fails after its last launch) and reads row 0
// instead of memory outside the dictionary.
__global__ __launch_bounds__(256) void k_gather_rows(const int32_t* __restrict__ dict, const int32_t* __restrict__ idx, size_t n_idx, size_t word, size_t V, int32_t* __restrict__ out,
                                                     uint64_t* __restrict__ lookups, uint32_t* __restrict__ err) {
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n_idx * word; o += (size_t)gridDim.x * 256) {
        const size_t j = o / word, w = o % word;
        const int32_t k = idx[j];
        const bool ok = k >= 0 && (size_t)k < V;
        if (!ok && w == 0) atomicOr(err, 1u);
        out[o] = dict[(size_t)(ok ? k : 0) * word + w];
        if (w == 0) lookups[j] = ok ? (uint64_t)(uint32_t)k : 0;
    }
}

// SoftmaxLastAxis (atlas-onnx-tracer/src/ops/softmax.rs:74-214), one workgroup per row of the last axis: max and first argmax,
// z = max - x saturated to the sub-table range, the two digit lookups, exp_q = floor(hi lo / S) with its remainder, the row sum,
// inv_sum = floor(S^2 / sum), softmax_q = floor(exp_q inv_sum / S) with its remainder; plus the lookup-index vectors of the prover
struct SoftmaxOut {
    int32_t *out, *R, *exp_q, *exp_hi, *exp_lo, *r_exp, *z, *z_hi, *z_lo, *e, *max_k, *argmax_k, *exp_sum, *inv_sum;
    uint64_t *idx_R, *idx_rexp, *idx_z, *idx_zhi, *idx_zlo;
};
__global__ __launch_bounds__(256) void k_softmax_rows(const int32_t* __restrict__ x, uint32_t N, int32_t S, uint32_t log2_base, int32_t z_bound,
                                                      const int32_t* __restrict__ lut_hi, const int32_t* __restrict__ lut_lo, SoftmaxOut O) {
    __shared__ int32_t s_v[256];
    __shared__ uint32_t s_p[256];
    const size_t k = blockIdx.x, off = k * (size_t)N;
    const uint32_t tid = threadIdx.x;
    int32_t mv = -2147483647 - 1; uint32_t mp = 0xffffffffu;
    for (uint32_t j = tid; j < N; j += 256) { const int32_t v = x[off + j]; if (v > mv || (v == mv && j < mp)) { mv = v; mp = j; } }
    s_v[tid] = mv; s_p[tid] = mp;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) { const int32_t v = s_v[tid + s]; const uint32_t q = s_p[tid + s]; if (v > s_v[tid] || (v == s_v[tid] && q < s_p[tid])) { s_v[tid] = v; s_p[tid] = q; } }
        __syncthreads();
    }
    mv = s_v[0]; mp = s_p[0];
    __syncthreads();
    const int32_t base_mask = (1 << log2_base) - 1;
    int32_t part = 0;
    for (uint32_t j = tid; j < N; j += 256) {
        const size_t i = off + j;
        const int32_t z = mv - x[i], zc = z < z_bound - 1 ? z : z_bound - 1;
        const int32_t zh = zc >> log2_base, zl = zc & base_mask, hi = lut_hi[zh], lo = lut_lo[zl];
        const int64_t prod = (int64_t)hi * lo;
        const int32_t eq = (int32_t)(prod / S), re = (int32_t)(prod - (int64_t)eq * S);
        O.z[i] = z; O.z_hi[i] = zh; O.z_lo[i] = zl; O.exp_hi[i] = hi; O.exp_lo[i] = lo; O.exp_q[i] = eq; O.r_exp[i] = re; O.e[i] = j == mp ? 1 : 0;
        O.idx_z[i] = (uint64_t)(uint32_t)z; O.idx_zhi[i] = (uint64_t)zh; O.idx_zlo[i] = (uint64_t)zl; O.idx_rexp[i] = (uint64_t)re;
        part += eq;
    }
    s_v[tid] = part;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) { if (tid < s) s_v[tid] += s_v[tid + s]; __syncthreads(); }
    const int32_t sum = s_v[0], inv = sum ? (S * S) / sum : 0;
    if (tid == 0) { O.max_k[k] = mv; O.argmax_k[k] = (int32_t)mp; O.exp_sum[k] = sum; O.inv_sum[k] = inv; }
    for (uint32_t j = tid; j < N; j += 256) {
        const size_t i = off + j;
        const int32_t p = O.exp_q[i] * inv, sq = p / S, rem = p - sq * S;
        O.out[i] = sq; O.R[i] = rem; O.idx_R[i] = (uint64_t)rem;
    }
}

unsigned grid_for(size_t n) { size_t b = (n + 255) / 256; return (unsigned)(b > 4096 ? 4096 : b ? b : 1); }

std::vector<size_t> row_major(const std::vector<size_t>& dims) {
    std::vector<size_t> s(dims.size(), 1);
    for (int d = (int)dims.size() - 2; d >= 0; d--) s[d] = s[d + 1] * dims[d + 1];
    return s;
}

}  // namespace

// the output strides of the two einsum operands for the layouts of ops/einsum/mod.rs:137-153 (dims as atlas_einsum_fold takes them)
int atlas_rt_einsum_strides(int layout, const std::vector<size_t>& d, std::vector<size_t>& out_dims, std::vector<size_t>& la, std::vector<size_t>& ra, size_t& K,
                            size_t& lsk, size_t& rsk) {
    switch (layout) {
        case ATLAS_EINSUM_MK_KN_MN: { if (d.size() != 3) return ATLAS_EINVAL; const size_t m = d[0], k = d[1], n = d[2]; out_dims = {m, n}; la = {k, 0}; ra = {0, 1}; K = k; lsk = 1; rsk = n; return 0; }
        case ATLAS_EINSUM_BMK_BKN_MBN: { if (d.size() != 4) return ATLAS_EINVAL; const size_t b = d[0], m = d[1], k = d[2], n = d[3]; out_dims = {m, b, n}; la = {k, m * k, 0}; ra = {0, k * n, 1}; K = k; lsk = 1; rsk = n; return 0; }
        case ATLAS_EINSUM_BMK_KBN_MBN: { if (d.size() != 4) return ATLAS_EINVAL; const size_t b = d[0], m = d[1], k = d[2], n = d[3]; out_dims = {m, b, n}; la = {k, m * k, 0}; ra = {0, n, 1}; K = k; lsk = 1; rsk = b * n; return 0; }
        case ATLAS_EINSUM_MBK_BNK_BMN: { if (d.size() != 4) return ATLAS_EINVAL; const size_t b = d[0], m = d[1], k = d[2], n = d[3]; out_dims = {b, m, n}; la = {k, b * k, 0}; ra = {n * k, 0, k}; K = k; lsk = 1; rsk = 1; return 0; }
        case ATLAS_EINSUM_MBK_NBK_BMN: { if (d.size() != 4) return ATLAS_EINVAL; const size_t b = d[0], m = d[1], k = d[2], n = d[3]; out_dims = {b, m, n}; la = {k, b * k, 0}; ra = {k, 0, b * k}; K = k; lsk = 1; rsk = 1; return 0; }
        case ATLAS_EINSUM_K_NK_N: { if (d.size() != 2) return ATLAS_EINVAL; const size_t k = d[0], n = d[1]; out_dims = {n}; la = {0}; ra = {k}; K = k; lsk = 1; rsk = 1; return 0; }
        default: return ATLAS_EINVAL;
    }
}

// i64 accumulators of an Einsum node into d_acc (library stream; the caller holds rt().mu)
int atlas_rt_einsum_acc(const Node& nd, const int32_t* L, const int32_t* R, int64_t* d_acc) {
    std::vector<size_t> od, la, ra; size_t K, lsk, rsk;
    if (atlas_rt_einsum_strides((int)nd.p[0], nd.shape, od, la, ra, K, lsk, rsk)) return fail(ATLAS_EINVAL, "graph: einsum layout / dims");
    Strides S{}; S.n = (uint32_t)od.size();
    size_t T = 1;
    for (size_t i = 0; i < od.size(); i++) { S.dim[i] = (uint32_t)od[i]; S.a[i] = (uint32_t)la[i]; S.b[i] = (uint32_t)ra[i]; T *= od[i]; }
    static const bool generic_only = getenv("ATLAS_EINSUM_GENERIC") != nullptr;      // A-B
    if ((int)nd.p[0] == ATLAS_EINSUM_MK_KN_MN && !generic_only) {
        // the projections (and the lm head: 16 x 1024 . 1024 x 65536): a thread owns a column of B for 8 rows of A and a slice of k, so B is read
        // once per row TILE — the one-thread-per-output kernel below re-reads it once per row of A: 42 of the 52 ms of a GPT-2-shaped trace
        const size_t m = nd.shape[0], k = nd.shape[1], n = nd.shape[2];
        uint32_t slices = 1;
        while (slices < 64 && (n * ((m + EB_ROWS - 1) / EB_ROWS)) * slices < ((size_t)1 << 16) && k / (slices * 2) >= 32) slices *= 2;
        const uint32_t k_slice = (uint32_t)((k + slices - 1) / slices);
        HIP_TRY(hipMemsetAsync(d_acc, 0, T * 8, rt().stream));
        k_einsum_acc_mk_kn<<<dim3((unsigned)((n + 255) / 256), (unsigned)((m + EB_ROWS - 1) / EB_ROWS), slices), 256, 0, rt().stream>>>(
            L, R, (uint32_t)m, (uint32_t)k, (uint32_t)n, k_slice, (unsigned long long*)d_acc);
        return ATLAS_OK;
    }
    k_einsum_acc_generic<<<grid_for(T), 256, 0, rt().stream>>>(L, R, S, (uint32_t)K, (uint32_t)lsk, (uint32_t)rsk, T, d_acc);
    return ATLAS_OK;
}

// sum_config (jolt-atlas-core/src/utils/dims.rs:545-607): leading dimensions of 1 are dropped until the operand is [m][n]; a 1-d operand is [m][1], axis 0
int atlas_rt_sum_config(const std::vector<size_t>& idims, size_t axis, size_t& m, size_t& n, int& ax) {
    std::vector<size_t> d = idims;
    while (d.size() > 2) { if (d[0] != 1 || axis == 0) return fail(ATLAS_EINVAL, "graph: Sum supports a leading batch of 1 only"); d.erase(d.begin()); axis--; }
    if (d.size() == 1) { m = d[0]; n = 1; ax = 0; return axis == 0 ? ATLAS_OK : fail(ATLAS_EINVAL, "graph: Sum axis"); }
    m = d[0]; n = d[1]; ax = (int)axis;
    return axis < 2 ? ATLAS_OK : fail(ATLAS_EINVAL, "graph: Sum axis");
}

// erffunc of the tracer (tensor/ops.rs:3671-3735): Chebyshev fit of erfc with the coefficients below, the same operations in the same order
static double atlas_erf_cheb(double x) {
    static const double COF[28] = {-1.3026537197817094, 6.419697923564902e-1, 1.9476473204185836e-2, -9.56151478680863e-3, -9.46595344482036e-4,
                                   3.66839497852761e-4, 4.2523324806907e-5, -2.0278578112534e-5, -1.624290004647e-6, 1.303655835580e-6,
                                   1.5626441722e-8, -8.5238095915e-8, 6.529054439e-9, 5.059343495e-9, -9.91364156e-10,
                                   -2.27365122e-10, 9.6467911e-11, 2.394038e-12, -6.886027e-12, 8.94487e-13,
                                   3.13092e-13, -1.12708e-13, 3.81e-16, 7.106e-15, -1.523e-15,
                                   -9.4e-17, 1.21e-16, -2.8e-17};
    auto erfccheb = [&](double z) {
        double d = 0.0, dd = 0.0;
        const double t = 2.0 / (2.0 + z), ty = 4.0 * t - 2.0;
        for (int j = 28 - 2; j >= 1; j--) { const double tmp = d; d = ty * d - dd + COF[j]; dd = tmp; }
        return t * std::exp(-(z * z) + 0.5 * (COF[0] + ty * d) - dd);
    };
    return x >= 0.0 ? 1.0 - erfccheb(x) : erfccheb(-x) - 1.0;
}
// The device copies of the tables below hang off the RUNTIME they were uploaded under (Runtime::graph_tables) and go with it in atlas_shutdown:
// several threads may share the process runtime, a thread may own one on another device, and a runtime may be shut down and brought up again
// on another device — a thread_local pointer freed by whichever thread happened to shut down served stale memory afterwards.  The host tables
// are values only: one copy per process, built once under g_host_tables_mu.
struct GraphDevTables { int32_t* trig[2] = {nullptr, nullptr}; int32_t* act[3] = {nullptr, nullptr, nullptr}; int32_t* exp = nullptr; ExpLut exp_lut; };
static std::mutex g_host_tables_mu;
static GraphDevTables& graph_dev_tables() {                 // callers hold rt().mu
    atlas_rt::Runtime& R = rt();
    if (!R.graph_tables) {
        R.graph_tables = new GraphDevTables();
        R.at_shutdown.push_back([] {
            GraphDevTables* t = static_cast<GraphDevTables*>(rt().graph_tables);
            if (!t) return;
            for (int32_t* p : {t->trig[0], t->trig[1], t->act[0], t->act[1], t->act[2], t->exp}) if (p) (void)hipFree(p);
            delete t;
            rt().graph_tables = nullptr;
        });
    }
    return *static_cast<GraphDevTables*>(R.graph_tables);
}
int atlas_rt_trig_table(int op, const int32_t** d_table, const std::vector<int32_t>** h_table) {
    static std::vector<int32_t> host[2];
    const int k = op == ATLAS_OP_SIN ? 0 : op == ATLAS_OP_COS ? 1 : -1;
    if (k < 0) return fail(ATLAS_EINVAL, "trig_table: not Sin / Cos");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    std::lock_guard<std::mutex> hl(g_host_tables_mu);
    int32_t** dev = graph_dev_tables().trig;
    if (host[k].empty()) {                                                    // SinTable / CosTable::materialize (neural_teleport/sin.rs:26-41)
        const size_t n = (size_t)1 << gr::TRIG_TABLE_VARS;
        const double scale = (double)((uint64_t)1 << (gr::MODEL_SCALE - gr::TRIG_DOWNSCALE_BITS));
        host[k].resize(n);
        for (size_t i = 0; i < n; i++) {
            const double xx = (double)(int32_t)i / scale;
            host[k][i] = (int32_t)std::round(scale * (k == 0 ? std::sin(xx) : std::cos(xx))) * (int32_t)(1 << gr::TRIG_DOWNSCALE_BITS);
        }
    }
    if (d_table && !dev[k]) {
        HIP_TRY(hipMalloc(&dev[k], host[k].size() * 4));
        HIP_TRY(hipMemcpyAsync(dev[k], host[k].data(), host[k].size() * 4, hipMemcpyHostToDevice, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
    }
    if (d_table) *d_table = dev[k];
    if (h_table) *h_table = &host[k];
    return ATLAS_OK;
}
int atlas_rt_activation_table(int op, const int32_t** d_table, const std::vector<int32_t>** h_table) {
    static std::vector<int32_t> host[3];
    const int k = op == ATLAS_OP_TANH ? 0 : op == ATLAS_OP_ERF ? 1 : op == ATLAS_OP_SIGMOID ? 2 : -1;
    if (k < 0) return fail(ATLAS_EINVAL, "activation_table: not a small-table activation");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    std::lock_guard<std::mutex> hl(g_host_tables_mu);
    int32_t** dev = graph_dev_tables().act;
    if (host[k].empty()) {
        const size_t n = (size_t)1 << gr::ACTIVATION_TABLE_VARS;
        const double scale = (double)((uint64_t)1 << gr::MODEL_SCALE);
        host[k].resize(n);
        for (size_t i = 0; i < n; i++) {                                      // usize_to_n_bits, then tensor::ops::nonlinearities::{tanh, erffunc, sigmoid} (tensor/ops.rs:3583-3591, 3671-3735, 3101-3109)
            const int32_t v = i >= n / 2 ? (int32_t)i - (int32_t)n : (int32_t)i;
            const double x = (double)v / scale;
            const double f = k == 0 ? scale * std::tanh(x) : k == 1 ? scale * atlas_erf_cheb(x) : scale / (1.0 + std::exp(-x));
            host[k][i] = (int32_t)std::round(f);
        }
    }
    if (d_table && !dev[k]) {
        HIP_TRY(hipMalloc(&dev[k], host[k].size() * 4));
        HIP_TRY(hipMemcpyAsync(dev[k], host[k].data(), host[k].size() * 4, hipMemcpyHostToDevice, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
    }
    if (d_table) *d_table = dev[k];
    if (h_table) *h_table = &host[k];
    return ATLAS_OK;
}

int atlas_rt_exp_lut(const ExpLut** out) {
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    GraphDevTables& GT = graph_dev_tables();
    ExpLut& L = GT.exp_lut;                                  // (the sub-tables are a few hundred entries: a copy per runtime, with ITS device pointers)
    int32_t*& dev = GT.exp;
    if (L.hi.empty()) {
        const double sf = (double)((uint64_t)1 << gr::MODEL_SCALE);
        const size_t needed = (size_t)std::ceil(sf * std::log(2.0 * sf)) + 2;                  // the flat LUT's cutoff exp(-i/S) S < 0.5
        const unsigned log2_b = (unsigned)std::ceil(std::log2((double)needed) / 2.0);
        const size_t base = (size_t)1 << log2_b, hi_size = needed / base + 2;
        auto entry = [&](double v) { const double r = std::round(v); return (int32_t)(r > 0.0 ? r : 0.0); };
        for (size_t h = 0; h < hi_size; h++) L.hi.push_back(entry(sf * std::exp(-((double)h * (double)base) / sf)));
        for (size_t l = 0; l < base; l++) L.lo.push_back(entry(sf * std::exp(-(double)l / sf)));
        L.hi.resize(gr::next_pow2(L.hi.size()), 0);
        L.log2_base = log2_b;
    }
    if (!dev) {
        HIP_TRY(hipMalloc(&dev, (L.hi.size() + L.lo.size()) * 4));
        HIP_TRY(hipMemcpyAsync(dev, L.hi.data(), L.hi.size() * 4, hipMemcpyHostToDevice, rt().stream));
        HIP_TRY(hipMemcpyAsync(dev + L.hi.size(), L.lo.data(), L.lo.size() * 4, hipMemcpyHostToDevice, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
        L.d_hi = dev; L.d_lo = dev + L.hi.size();
    }
    *out = &L;
    return ATLAS_OK;
}

// Everything exec_node would refuse, decided from the description alone (operand counts and shapes, operator parameters, the
// scales / bounds the prover's tables are compiled for).  atlas_prove_graph and atlas_verify_graph call it for every node before
// they touch one: a verifier's graph is never traced, and its operator formulas index nd.dims / nd.p / nd.shape without checks of
// their own, so a model that would fail to trace is refused here instead of being verified under other semantics.
int atlas_rt_validate_node(const atlas_graph& G, const Node& nd) {
    auto bad = [&](const char* what) { return fail(ATLAS_EINVAL, what); };
    if (nd.dims.empty() || nd.dims.size() > MAXR || !gr::all_pow2(nd.dims)) return bad("graph: a node needs 1..6 dimensions, each a power of two");
    const size_t T = gr::padded_len(nd.dims);
    if (T > ((size_t)1 << 26)) return bad("graph: a node output above 2^26 elements");
    for (size_t i : nd.inputs) { auto it = G.nodes.find(i); if (it == G.nodes.end() || i >= nd.idx) return bad("graph: inputs must name earlier nodes"); }
    auto in_node = [&](size_t i) -> const Node& { return G.nodes.at(nd.inputs[i]); };
    auto need_inputs = [&](size_t n) { return nd.inputs.size() == n; };
    auto same_len = [&]() { for (size_t i = 0; i < nd.inputs.size(); i++) if (gr::padded_len(in_node(i).dims) != T) return false; return true; };
    switch (nd.op) {
        case ATLAS_OP_INPUT: return need_inputs(0) ? ATLAS_OK : bad("graph: Input takes no operands");
        case ATLAS_OP_CONSTANT: return need_inputs(0) && nd.constant.size() == T ? ATLAS_OK : bad("graph: constant length != padded shape");
        case ATLAS_OP_IDENTITY: case ATLAS_OP_RESHAPE: case ATLAS_OP_RELU: case ATLAS_OP_NEG: case ATLAS_OP_IS_NAN:
            return need_inputs(1) && same_len() ? ATLAS_OK : bad("graph: one operand of the output's length expected");
        case ATLAS_OP_ADD: case ATLAS_OP_SUB: case ATLAS_OP_AND:
            return need_inputs(2) && same_len() ? ATLAS_OK : bad("graph: two operands of the output's shape expected");
        case ATLAS_OP_IFF: return need_inputs(3) && same_len() ? ATLAS_OK : bad("graph: Iff takes (mask, a, b) of the output's shape");
        case ATLAS_OP_CLAMP:
            return need_inputs(1) && same_len() && nd.p[0] == (int64_t)gr::CLAMP_BOUND ? ATLAS_OK
                   : bad("graph: Clamp needs one operand and bound_log = CLAMP_BOUND (9): the prover's table is compiled for it");
        case ATLAS_OP_MOVEAXIS: {
            if (!need_inputs(1)) return bad("graph: one operand expected");
            const std::vector<size_t>& idims = in_node(0).dims;
            const size_t r = idims.size();
            if (nd.p[0] < 0 || nd.p[1] < 0 || r != nd.dims.size() || (size_t)nd.p[0] >= r || (size_t)nd.p[1] >= r) return bad("graph: MoveAxis axes");
            std::vector<size_t> perm;
            for (size_t a = 0; a < r; a++) if (a != (size_t)nd.p[0]) perm.push_back(a);
            perm.insert(perm.begin() + nd.p[1], (size_t)nd.p[0]);
            for (size_t a = 0; a < r; a++) if (nd.dims[a] != idims[perm[a]]) return bad("graph: MoveAxis output dims");
            return ATLAS_OK;
        }
        case ATLAS_OP_BROADCAST: {
            if (!need_inputs(1)) return bad("graph: one operand expected");
            const std::vector<size_t>& idims = in_node(0).dims;
            if (idims.size() > nd.dims.size()) return bad("graph: Broadcast rank");
            const size_t off = nd.dims.size() - idims.size();
            for (size_t a = off; a < nd.dims.size(); a++) if (idims[a - off] != nd.dims[a] && idims[a - off] != 1) return bad("graph: Broadcast dims");
            return ATLAS_OK;
        }
        case ATLAS_OP_SLICE: {
            if (!need_inputs(1)) return bad("graph: one operand expected");
            const std::vector<size_t>& idims = in_node(0).dims;
            if (nd.p[0] < 0 || nd.p[1] < 0 || nd.p[2] < 0) return bad("graph: Slice");
            const size_t ax = (size_t)nd.p[0], st = (size_t)nd.p[1], en = (size_t)nd.p[2];
            if (idims.size() != nd.dims.size() || ax >= idims.size() || en <= st || en > idims[ax] || nd.dims[ax] != en - st) return bad("graph: Slice");
            for (size_t a = 0; a < idims.size(); a++) if (a != ax && idims[a] != nd.dims[a]) return bad("graph: Slice non-axis dimensions must match");
            return ATLAS_OK;
        }
        case ATLAS_OP_CONCAT: {
            const size_t r = nd.dims.size();
            if (nd.inputs.empty() || nd.inputs.size() > 8 || nd.p[0] < 0 || (size_t)nd.p[0] >= r) return bad("graph: Concat takes 1..8 operands and an axis of the output");
            size_t off = 0;
            for (size_t k = 0; k < nd.inputs.size(); k++) {
                const std::vector<size_t>& idims = in_node(k).dims;
                if (idims.size() != r) return bad("graph: Concat operand rank");
                for (size_t a = 0; a < r; a++) if (a != (size_t)nd.p[0] && idims[a] != nd.dims[a]) return bad("graph: Concat non-axis dimensions must match");
                off += idims[(size_t)nd.p[0]];
            }
            return off == nd.dims[(size_t)nd.p[0]] ? ATLAS_OK : bad("graph: Concat output axis dimension must equal the sum of the operands'");
        }
        case ATLAS_OP_EINSUM: case ATLAS_OP_MUL: case ATLAS_OP_SQUARE: case ATLAS_OP_CUBE: {
            const int64_t bits = nd.op == ATLAS_OP_EINSUM ? nd.p[1] : nd.op == ATLAS_OP_CUBE ? 2 * nd.p[0] : nd.p[0];
            if (bits < 1 || bits > 30) return bad("graph: fused rescale needs 1 <= bits <= 30 (unfused building-block products are not modelled)");
            if (!need_inputs(nd.op == ATLAS_OP_EINSUM || nd.op == ATLAS_OP_MUL ? 2 : 1)) return bad("graph: operand count");
            if (nd.op != ATLAS_OP_EINSUM) return same_len() ? ATLAS_OK : bad("graph: element-wise operands must have the output's shape");
            std::vector<size_t> od, la, ra; size_t K, lsk, rsk;
            if (!gr::all_pow2(nd.shape) || atlas_rt_einsum_strides((int)nd.p[0], nd.shape, od, la, ra, K, lsk, rsk)) return bad("graph: einsum layout / dims");
            size_t To = 1; for (size_t d : od) To *= d;
            const size_t b = nd.shape.size() == 4 ? nd.shape[0] : 1, m = nd.shape.size() == 2 ? 1 : nd.shape[nd.shape.size() - 3], n = nd.shape.back();
            if (To != T || gr::padded_len(in_node(0).dims) != b * m * K || gr::padded_len(in_node(1).dims) != b * K * n) return bad("graph: einsum operand / output sizes do not match its dims");
            return ATLAS_OK;
        }
        case ATLAS_OP_SUM: {
            if (!need_inputs(1) || nd.shape.size() != 1) return bad("graph: Sum needs one operand and one axis");
            size_t m, n; int axis;
            if (int rc = atlas_rt_sum_config(in_node(0).dims, nd.shape[0], m, n, axis)) return rc;
            return T == (axis == 0 ? n : m) ? ATLAS_OK : bad("graph: Sum output dims");
        }
        case ATLAS_OP_SCALAR_CONST_DIV:
            if (nd.p[0] == 0 || nd.p[0] > 2147483647ll || nd.p[0] < -2147483648ll) return bad("graph: ScalarConstDiv divisor");
            return need_inputs(1) && same_len() ? ATLAS_OK : bad("graph: Div / ScalarConstDiv operands");
        case ATLAS_OP_DIV: return need_inputs(2) && same_len() ? ATLAS_OK : bad("graph: Div / ScalarConstDiv operands");
        case ATLAS_OP_MEAN_OF_SQUARES: {
            if (!need_inputs(1) || nd.p[0] < 0 || nd.p[0] > 30) return bad("graph: MeanOfSquares operand / scale");
            const size_t N = in_node(0).dims.back(), K = gr::padded_len(in_node(0).dims) / N;
            if (nd.p[1] <= 0 || (uint64_t)nd.p[1] > (uint64_t)N || N > ((size_t)1 << 31)) return bad("graph: MeanOfSquares dims / count / divisor");   // (bounded before the product: a hostile count must not overflow it)
            const int64_t D = ((int64_t)1 << nd.p[0]) * nd.p[1];       // < 2^30 * 2^31
            return T == K && D <= 2147483647ll ? ATLAS_OK : bad("graph: MeanOfSquares dims / count / divisor");
        }
        case ATLAS_OP_RSQRT: return need_inputs(1) && same_len() && nd.p[0] > 0 && nd.p[0] <= 14 ? ATLAS_OK : bad("graph: Rsqrt operand / scale (1..14)");
        case ATLAS_OP_TANH: case ATLAS_OP_ERF: case ATLAS_OP_SIGMOID: case ATLAS_OP_SIN: case ATLAS_OP_COS:
            return need_inputs(1) && same_len() && nd.p[0] == (int64_t)gr::MODEL_SCALE ? ATLAS_OK
                   : bad("graph: Tanh / Erf / Sigmoid / Sin / Cos need one operand and scale = MODEL_SCALE (14): the prover's tables and the period modulus are compiled for it");
        case ATLAS_OP_GATHER_LARGE: case ATLAS_OP_GATHER_SMALL: {
            if (!need_inputs(2) || nd.p[0] != 0) return bad("graph: Gather needs (dictionary, indexes) and axis 0");
            const size_t V = in_node(0).dims[0], word = gr::padded_len(in_node(0).dims) / V, N = gr::padded_len(in_node(1).dims);
            if (T != N * word || nd.p[1] < 0 || (size_t)nd.p[1] > V) return bad("graph: Gather dims / dict_len");
            if (nd.op == ATLAS_OP_GATHER_SMALL && (V > 65536 || V < 2)) return bad("graph: GatherSmall is the tracer's choice for dictionaries of at most 2^16 words (handlers/index.rs:33-45)");
            return ATLAS_OK;
        }
        case ATLAS_OP_SOFTMAX: {
            if (!need_inputs(1) || !same_len() || nd.p[0] != (int64_t)gr::MODEL_SCALE)
                return bad("graph: SoftmaxLastAxis needs one operand of the output's shape and scale = MODEL_SCALE (14): its clamp table is compiled for it");
            const size_t N = nd.dims.back(), F = T / N;
            return F >= 1 && N >= 2 && N <= 65536 ? ATLAS_OK : bad("graph: SoftmaxLastAxis needs 2 <= last axis <= 65536");
        }
        default: return bad("graph: operator not supported");
    }
}
int atlas_rt_validate_graph(const atlas_graph& G) {
    if (G.nodes.empty() || G.outputs.empty()) return fail(ATLAS_EINVAL, "graph: no nodes / no outputs");
    for (auto& kv : G.nodes) if (int rc = atlas_rt_validate_node(G, kv.second)) return rc;
    return ATLAS_OK;
}

namespace {

int exec_node(atlas_graph& G, const Node& nd, const int32_t* const* host_inputs, size_t& next_input) {
    static const char* const op_names[] = {"trace Input", "trace Constant", "trace Identity", "trace Add", "trace Sub", "trace Mul", "trace Square", "trace Cube", "trace And", "trace Iff",
                                           "trace ReLU", "trace Einsum", "trace Reshape", "trace MoveAxis", "trace Broadcast", "trace Slice", "trace Concat", "trace Sum",
                                           "trace ScalarConstDiv", "trace Div", "trace MeanOfSquares", "trace Rsqrt", "trace SoftmaxLastAxis", "trace Tanh", "trace GatherLarge",
                                           "trace GatherSmall", "trace Erf", "trace Sigmoid", "trace Neg", "trace IsNan", "trace Clamp", "trace Sin", "trace Cos"};
    PROF(nd.op >= 0 && nd.op < (int)(sizeof(op_names) / sizeof(op_names[0])) ? op_names[nd.op] : "trace ?");
    if (int rc = atlas_rt_validate_node(G, nd)) return rc;
    const size_t T = gr::padded_len(nd.dims);
    DevBuf& out = G.out[nd.idx];
    if (nd.op == ATLAS_OP_CONSTANT && out.p) return ATLAS_OK;                 // uploaded by an earlier trace of this model
    auto in = [&](size_t i) -> const int32_t* { return G.tensor(nd.inputs[i]); };
    auto in_node = [&](size_t i) -> const Node& { return G.nodes.at(nd.inputs[i]); };
    for (size_t i = 0; i < nd.inputs.size(); i++) if (!in(i)) return fail(ATLAS_ESTATE, "graph_trace: operand not executed yet (nodes must be in topological index order)");
    auto need_inputs = [&](size_t n) { return nd.inputs.size() == n; };
    auto same_len = [&]() { for (size_t i = 0; i < nd.inputs.size(); i++) if (gr::padded_len(in_node(i).dims) != T) return false; return true; };
    if (nd.op != ATLAS_OP_EINSUM && nd.op != ATLAS_OP_MUL && nd.op != ATLAS_OP_SQUARE && nd.op != ATLAS_OP_CUBE) HIP_TRY(out.alloc(T * 4));
    if (T > ((size_t)1 << 26)) return fail(ATLAS_EINVAL, "graph: a node output above 2^26 elements");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    switch (nd.op) {
        case ATLAS_OP_INPUT:
            HIP_TRY(hipMemcpyAsync(out.p, host_inputs[next_input++], T * 4, hipMemcpyHostToDevice, rt().stream));
            HIP_TRY(hipStreamSynchronize(rt().stream));
            return ATLAS_OK;
        case ATLAS_OP_CONSTANT:
            if (nd.constant.size() != T) return fail(ATLAS_EINVAL, "graph: constant length != padded shape");
            HIP_TRY(hipMemcpyAsync(out.p, nd.constant.data(), T * 4, hipMemcpyHostToDevice, rt().stream));
            HIP_TRY(hipStreamSynchronize(rt().stream));
            return ATLAS_OK;
        case ATLAS_OP_IDENTITY: case ATLAS_OP_RESHAPE:                       // same flat order when every dimension is a power of two
            if (!need_inputs(1) || !same_len()) return fail(ATLAS_EINVAL, "graph: Identity / Reshape operand length");
            HIP_TRY(hipMemcpyAsync(out.p, in(0), T * 4, hipMemcpyDeviceToDevice, rt().stream));
            return ATLAS_OK;
        case ATLAS_OP_ADD: case ATLAS_OP_SUB: {                               // sat_binop (ops/mod.rs:263-274) + the clamp lookup's witness
            if (!need_inputs(2) || !same_len()) return fail(ATLAS_EINVAL, "graph: Add / Sub need two operands of the output's shape");
            NodeWitness& W = G.wit[nd.idx];
            HIP_TRY(W.acc.alloc(T * 8)); HIP_TRY(W.cidx.alloc(T * 8)); HIP_TRY(W.acc_fr.alloc(T * sizeof(Fr)));
            k_addsub_witness<<<grid_for(T), 256, 0, rt().stream>>>(in(0), in(1), T, nd.op == ATLAS_OP_SUB, W.acc.as<int64_t>(), out.as<int32_t>(), W.cidx.as<uint64_t>());
            k_i64_to_fr<<<grid_for(T), 256, 0, rt().stream>>>(W.acc.as<int64_t>(), W.acc_fr.as<Fr>(), T);
            return ATLAS_OK;
        }
        case ATLAS_OP_AND: case ATLAS_OP_IFF:
            if (!need_inputs(nd.op == ATLAS_OP_AND ? 2 : 3) || !same_len()) return fail(ATLAS_EINVAL, "graph: And / Iff operands");
            k_select<<<grid_for(T), 256, 0, rt().stream>>>(nd.op, in(0), in(1), nd.op == ATLAS_OP_IFF ? in(2) : nullptr, T, out.as<int32_t>());
            return ATLAS_OK;
        case ATLAS_OP_RELU: {
            if (!need_inputs(1) || !same_len()) return fail(ATLAS_EINVAL, "graph: ReLU operand");
            NodeWitness& W = G.wit[nd.idx];
            HIP_TRY(W.lookups.alloc(T * 8));
            k_relu_witness<<<grid_for(T), 256, 0, rt().stream>>>(in(0), T, out.as<int32_t>(), W.lookups.as<uint64_t>());
            return ATLAS_OK;
        }
        case ATLAS_OP_NEG:                                                    // ops/neg.rs of the tracer (wrapping, like i32 negation in release builds)
            if (!need_inputs(1) || !same_len()) return fail(ATLAS_EINVAL, "graph: Neg operand");
            k_neg<<<grid_for(T), 256, 0, rt().stream>>>(in(0), T, out.as<int32_t>());
            return ATLAS_OK;
        case ATLAS_OP_IS_NAN:                                                 // quantised tensors hold no NaN: all zeros
            if (!need_inputs(1) || !same_len()) return fail(ATLAS_EINVAL, "graph: IsNan operand");
            HIP_TRY(hipMemsetAsync(out.p, 0, T * 4, rt().stream));
            return ATLAS_OK;
        case ATLAS_OP_CLAMP: {
            if (!need_inputs(1) || !same_len() || nd.p[0] != (int64_t)gr::CLAMP_BOUND) return fail(ATLAS_EINVAL, "graph: Clamp needs one operand and bound_log = CLAMP_BOUND (9): the prover's table is compiled for it");
            NodeWitness& W = G.wit[nd.idx];
            HIP_TRY(W.lookups.alloc(T * 8));
            k_clamp_witness<<<grid_for(T), 256, 0, rt().stream>>>(in(0), T, (int32_t)1 << gr::CLAMP_BOUND, out.as<int32_t>(), W.lookups.as<uint64_t>());
            return ATLAS_OK;
        }
        case ATLAS_OP_MOVEAXIS: case ATLAS_OP_BROADCAST: case ATLAS_OP_SLICE: {
            if (!need_inputs(1)) return fail(ATLAS_EINVAL, "graph: one operand expected");
            const std::vector<size_t>& idims = in_node(0).dims;
            std::vector<size_t> istr = row_major(idims);
            Strides S{}; S.n = (uint32_t)nd.dims.size();
            if (S.n > MAXR) return fail(ATLAS_EINVAL, "graph: rank > 6");
            size_t base = 0;
            for (size_t i = 0; i < nd.dims.size(); i++) S.dim[i] = (uint32_t)nd.dims[i];
            if (nd.op == ATLAS_OP_MOVEAXIS) {                                 // tensor move_axis(source, destination): out axes = in axes with `source` moved to `destination`
                const size_t src = (size_t)nd.p[0], dst = (size_t)nd.p[1], r = idims.size();
                if (r != nd.dims.size() || src >= r || dst >= r) return fail(ATLAS_EINVAL, "graph: MoveAxis axes");
                std::vector<size_t> perm;                                     // perm[out axis] = in axis
                for (size_t a = 0; a < r; a++) if (a != src) perm.push_back(a);
                perm.insert(perm.begin() + dst, src);
                for (size_t a = 0; a < r; a++) { if (nd.dims[a] != idims[perm[a]]) return fail(ATLAS_EINVAL, "graph: MoveAxis output dims"); S.a[a] = (uint32_t)istr[perm[a]]; }
            } else if (nd.op == ATLAS_OP_BROADCAST) {                         // Tensor::expand (tensor/mod.rs:1042-1100): trailing alignment
                if (idims.size() > nd.dims.size()) return fail(ATLAS_EINVAL, "graph: Broadcast rank");
                const size_t off = nd.dims.size() - idims.size();
                for (size_t a = 0; a < nd.dims.size(); a++) {
                    if (a < off) { S.a[a] = 0; continue; }
                    const size_t id = idims[a - off];
                    if (id != nd.dims[a] && id != 1) return fail(ATLAS_EINVAL, "graph: Broadcast dims");
                    S.a[a] = id == nd.dims[a] ? (uint32_t)istr[a - off] : 0;
                }
            } else {                                                          // Slice { axis, start, end } (ops/slice.rs)
                const size_t ax = (size_t)nd.p[0], st = (size_t)nd.p[1], en = (size_t)nd.p[2];
                if (idims.size() != nd.dims.size() || ax >= idims.size() || en <= st || en > idims[ax] || nd.dims[ax] != en - st) return fail(ATLAS_EINVAL, "graph: Slice");
                for (size_t a = 0; a < idims.size(); a++) S.a[a] = (uint32_t)istr[a];
                base = st * istr[ax];
            }
            k_gather_strided<<<grid_for(T), 256, 0, rt().stream>>>(in(0), S, base, T, out.as<int32_t>());
            return ATLAS_OK;
        }
        case ATLAS_OP_CONCAT: {                                               // tensor::ops::concat along p[0] (atlas-onnx-tracer/src/tensor/ops.rs:2772)
            const size_t ax = (size_t)nd.p[0], r = nd.dims.size();
            if (nd.inputs.empty() || nd.inputs.size() > 8 || ax >= r || r > MAXR) return fail(ATLAS_EINVAL, "graph: Concat takes 1..8 operands and an axis of the output");
            const std::vector<size_t> ostr = row_major(nd.dims);
            size_t off = 0;
            for (size_t k = 0; k < nd.inputs.size(); k++) {
                const std::vector<size_t>& idims = in_node(k).dims;
                if (idims.size() != r) return fail(ATLAS_EINVAL, "graph: Concat operand rank");
                for (size_t a = 0; a < r; a++) if (a != ax && idims[a] != nd.dims[a]) return fail(ATLAS_EINVAL, "graph: Concat non-axis dimensions must match");
                Strides S{}; S.n = (uint32_t)r;
                for (size_t a = 0; a < r; a++) { S.dim[a] = (uint32_t)idims[a]; S.a[a] = (uint32_t)ostr[a]; }
                const size_t Ti = gr::padded_len(idims);
                k_scatter_strided<<<grid_for(Ti), 256, 0, rt().stream>>>(in(k), S, off * ostr[ax], Ti, out.as<int32_t>());
                off += idims[ax];
            }
            if (off != nd.dims[ax]) return fail(ATLAS_EINVAL, "graph: Concat output axis dimension must equal the sum of the operands'");
            return ATLAS_OK;
        }
        case ATLAS_OP_EINSUM: case ATLAS_OP_MUL: case ATLAS_OP_SQUARE: case ATLAS_OP_CUBE: {
            const size_t S = nd.op == ATLAS_OP_EINSUM ? (size_t)nd.p[1] : nd.op == ATLAS_OP_CUBE ? 2 * (size_t)nd.p[0] : (size_t)nd.p[0];      // rebase_bits (fused_rebase.rs:71-79)
            if (S == 0 || S > 30) return fail(ATLAS_EINVAL, "graph: fused rescale needs 1 <= bits <= 30 (unfused building-block products are not modelled)");
            if (!need_inputs(nd.op == ATLAS_OP_EINSUM || nd.op == ATLAS_OP_MUL ? 2 : 1)) return fail(ATLAS_EINVAL, "graph: operand count");
            if (nd.op != ATLAS_OP_EINSUM && !same_len()) return fail(ATLAS_EINVAL, "graph: element-wise operands must have the output's shape");
            NodeWitness& W = G.wit[nd.idx];
            W.rescale.reset(new RescaleWitness());
            auto fill = [&](int64_t* d_acc) -> int {
                if (nd.op == ATLAS_OP_EINSUM) return atlas_rt_einsum_acc(nd, in(0), in(1), d_acc);
                if (nd.op == ATLAS_OP_CUBE) k_cube_acc<<<grid_for(T), 256, 0, rt().stream>>>(in(0), T, d_acc);
                else k_mul_acc<<<grid_for(T), 256, 0, rt().stream>>>(in(0), nd.op == ATLAS_OP_SQUARE ? in(0) : in(1), T, d_acc);
                return ATLAS_OK;
            };
            int rc = make_rescale_witness(T, S, fill, nullptr, *W.rescale);
            if (rc) return rc;
            out.p = W.rescale->out_own.release();                             // the node output = SatClamp_i32(acc >> S)
            W.rescale->d_output = out.as<int32_t>();
            return ATLAS_OK;
        }
        case ATLAS_OP_SUM: {                                                  // Sum { axes: [a] } over a 2-d operand (utils/dims.rs:545-607 normalises to 2-d)
            if (!need_inputs(1) || nd.shape.size() != 1) return fail(ATLAS_EINVAL, "graph: Sum needs one operand and one axis");
            size_t m, n; int axis;
            if (int rc = atlas_rt_sum_config(in_node(0).dims, nd.shape[0], m, n, axis)) return rc;
            if (T != (axis == 0 ? n : m)) return fail(ATLAS_EINVAL, "graph: Sum output dims");
            NodeWitness& W = G.wit[nd.idx];
            HIP_TRY(W.acc.alloc(T * 8)); HIP_TRY(W.cidx.alloc(T * 8)); HIP_TRY(W.acc_fr.alloc(T * sizeof(Fr)));
            k_sum_axis<<<grid_for(T), 256, 0, rt().stream>>>(in(0), (uint32_t)m, (uint32_t)n, axis, 0, W.acc.as<int64_t>());
            k_clamp_acc<<<grid_for(T), 256, 0, rt().stream>>>(W.acc.as<int64_t>(), T, out.as<int32_t>(), W.cidx.as<uint64_t>());
            k_i64_to_fr<<<grid_for(T), 256, 0, rt().stream>>>(W.acc.as<int64_t>(), W.acc_fr.as<Fr>(), T);
            return ATLAS_OK;
        }
        case ATLAS_OP_SCALAR_CONST_DIV: case ATLAS_OP_DIV: {
            const bool sc = nd.op == ATLAS_OP_SCALAR_CONST_DIV;
            if (!need_inputs(sc ? 1 : 2) || !same_len()) return fail(ATLAS_EINVAL, "graph: Div / ScalarConstDiv operands");
            if (sc && nd.p[0] == 0) return fail(ATLAS_EINVAL, "graph: ScalarConstDiv by zero");
            NodeWitness& W = G.wit[nd.idx];
            HIP_TRY(W.rem.alloc(T * 4));
            k_floor_div<<<grid_for(T), 256, 0, rt().stream>>>(in(0), sc ? nullptr : in(1), (int32_t)nd.p[0], T, out.as<int32_t>(), W.rem.as<int32_t>());
            if (!sc) { uint64_t* lk2 = nullptr; int rc = atlas_lookup_indices_from_operands(W.rem.as<int32_t>(), in(1), T, &lk2); if (rc) return rc; W.lookups.p = lk2; }   // interleave(R, divisor)
            return ATLAS_OK;
        }
        case ATLAS_OP_MEAN_OF_SQUARES: {                                      // MeanOfSquares { axes: [last], scale, count, padded_count }: operand [K][N]
            if (!need_inputs(1)) return fail(ATLAS_EINVAL, "graph: MeanOfSquares operand");
            const size_t N = in_node(0).dims.back(), K = gr::padded_len(in_node(0).dims) / N;
            const int64_t D = ((int64_t)1 << nd.p[0]) * (int64_t)nd.p[1];
            if (T != K || nd.p[1] <= 0 || (size_t)nd.p[1] > N || D > 2147483647ll) return fail(ATLAS_EINVAL, "graph: MeanOfSquares dims / count / divisor");
            NodeWitness& W = G.wit[nd.idx];
            W.rescale.reset(new RescaleWitness());
            RescaleWitness& R = *W.rescale;
            R.T = T; R.S = 0;
            HIP_TRY(R.quot.alloc(T * 8)); HIP_TRY(R.rem.alloc(T * 4)); HIP_TRY(R.cidx.alloc(T * 8)); HIP_TRY(R.qfr.alloc(T * sizeof(Fr))); HIP_TRY(W.bound.alloc(T * 4));
            k_sum_axis<<<grid_for(T), 256, 0, rt().stream>>>(in(0), (uint32_t)K, (uint32_t)N, 1, 1, R.quot.as<int64_t>());
            k_mos_rebase<<<grid_for(T), 256, 0, rt().stream>>>(R.quot.as<int64_t>(), T, D, R.rem.as<int32_t>(), out.as<int32_t>(), R.cidx.as<uint64_t>(), W.bound.as<int32_t>());
            k_i64_to_fr<<<grid_for(T), 256, 0, rt().stream>>>(R.quot.as<int64_t>(), R.qfr.as<Fr>(), T);
            R.d_output = out.as<int32_t>();
            { uint64_t* lk2 = nullptr; int rc = atlas_lookup_indices_from_operands(R.rem.as<int32_t>(), W.bound.as<int32_t>(), T, &lk2); if (rc) return rc; W.lookups.p = lk2; }
            return ATLAS_OK;
        }
        case ATLAS_OP_RSQRT: {
            if (!need_inputs(1) || !same_len() || nd.p[0] <= 0 || nd.p[0] > 14) return fail(ATLAS_EINVAL, "graph: Rsqrt operand / scale (1..14)");
            NodeWitness& W = G.wit[nd.idx];
            DevBuf q64;
            HIP_TRY(q64.alloc(T * 8)); HIP_TRY(W.quot_fr.alloc(T * sizeof(Fr))); HIP_TRY(W.rem.alloc(T * 4)); HIP_TRY(W.rem2.alloc(T * 4)); HIP_TRY(W.bound.alloc(T * 4));
            k_rsqrt<<<grid_for(T), 256, 0, rt().stream>>>(in(0), T, (int64_t)1 << (3 * nd.p[0]), out.as<int32_t>(), q64.as<int64_t>(), W.rem.as<int32_t>(), W.rem2.as<int32_t>(), W.bound.as<int32_t>());
            k_i64_to_fr<<<grid_for(T), 256, 0, rt().stream>>>(q64.as<int64_t>(), W.quot_fr.as<Fr>(), T);
            HIP_TRY(hipStreamSynchronize(rt().stream));                          // q64 leaves scope
            { uint64_t* lk2 = nullptr; int rc = atlas_lookup_indices_from_operands(W.rem.as<int32_t>(), in(0), T, &lk2); if (rc) return rc; W.lookups.p = lk2; }
            { uint64_t* lk2 = nullptr; int rc = atlas_lookup_indices_from_operands(W.rem2.as<int32_t>(), W.bound.as<int32_t>(), T, &lk2); if (rc) return rc; W.lookups2.p = lk2; }
            return ATLAS_OK;
        }
        case ATLAS_OP_TANH: case ATLAS_OP_ERF: case ATLAS_OP_SIGMOID: {       // ops/{tanh,erf,sigmoid}.rs of the tracer: clamp, then the function = the table
            if (!need_inputs(1) || !same_len() || nd.p[0] != (int64_t)gr::MODEL_SCALE) return fail(ATLAS_EINVAL, "graph: Tanh / Erf / Sigmoid need one operand and scale = MODEL_SCALE (14): the prover's table is compiled for it");
            const int32_t* d_table = nullptr;
            if (int rc = atlas_rt_activation_table(nd.op, &d_table, nullptr)) return rc;
            NodeWitness& W = G.wit[nd.idx];
            HIP_TRY(W.clamped.alloc(T * 4)); HIP_TRY(W.lookups.alloc(T * 8)); HIP_TRY(W.lookups2.alloc(T * 8));
            k_tanh<<<grid_for(T), 256, 0, rt().stream>>>(in(0), T, (uint32_t)gr::ACTIVATION_BOUND, d_table, out.as<int32_t>(), W.clamped.as<int32_t>(), W.lookups2.as<uint64_t>(), W.lookups.as<uint64_t>());
            return ATLAS_OK;
        }
        case ATLAS_OP_SIN: case ATLAS_OP_COS: {
            if (!need_inputs(1) || !same_len() || nd.p[0] != (int64_t)gr::MODEL_SCALE) return fail(ATLAS_EINVAL, "graph: Sin / Cos need one operand and scale = MODEL_SCALE (14): the period modulus and the table are compiled for it");
            const int32_t* d_table = nullptr;
            if (int rc = atlas_rt_trig_table(nd.op, &d_table, nullptr)) return rc;
            NodeWitness& W = G.wit[nd.idx];
            // rem = the teleportation remainder, rem2 = the quotient, clamped = the downscaled remainder; lookups = the remainder (the right-shift
            // lookup), lookups2 = the table index, cidx = interleave(remainder, tau) (the range check)
            HIP_TRY(W.rem.alloc(T * 4)); HIP_TRY(W.rem2.alloc(T * 4)); HIP_TRY(W.clamped.alloc(T * 4)); HIP_TRY(W.bound.alloc(T * 4));
            HIP_TRY(W.lookups.alloc(T * 8)); HIP_TRY(W.lookups2.alloc(T * 8));
            k_trig<<<grid_for(T), 256, 0, rt().stream>>>(in(0), T, (int32_t)gr::TRIG_PERIOD_MODULUS, (uint32_t)gr::TRIG_DOWNSCALE_BITS, d_table, out.as<int32_t>(), W.rem2.as<int32_t>(),
                                                      W.rem.as<int32_t>(), W.clamped.as<int32_t>(), W.lookups.as<uint64_t>(), W.lookups2.as<uint64_t>());
            k_fill_i32<<<grid_for(T), 256, 0, rt().stream>>>(W.bound.as<int32_t>(), T, (int32_t)gr::TRIG_PERIOD_MODULUS);
            { uint64_t* lk2 = nullptr; int rc = atlas_lookup_indices_from_operands(W.rem.as<int32_t>(), W.bound.as<int32_t>(), T, &lk2); if (rc) return rc; W.cidx.p = lk2; }
            return ATLAS_OK;
        }
        case ATLAS_OP_GATHER_LARGE: case ATLAS_OP_GATHER_SMALL: {             // inputs (dictionary [V][D...], indexes [N]); output [N][D...]
            if (!need_inputs(2) || nd.p[0] != 0) return fail(ATLAS_EINVAL, "graph: Gather needs (dictionary, indexes) and axis 0");
            const size_t V = in_node(0).dims[0], word = gr::padded_len(in_node(0).dims) / V, N = gr::padded_len(in_node(1).dims);
            if (T != N * word || (size_t)nd.p[1] > V) return fail(ATLAS_EINVAL, "graph: Gather dims / dict_len");
            if (nd.op == ATLAS_OP_GATHER_SMALL && (V > 65536 || V < 2)) return fail(ATLAS_EINVAL, "graph: GatherSmall is the tracer's choice for dictionaries of at most 2^16 words (handlers/index.rs:33-45)");
            NodeWitness& W = G.wit[nd.idx];
            HIP_TRY(W.lookups.alloc(N * 8));
            if (!G.trace_err.p) { HIP_TRY(G.trace_err.alloc(4)); HIP_TRY(hipMemsetAsync(G.trace_err.p, 0, 4, rt().stream)); }
            k_gather_rows<<<grid_for(T), 256, 0, rt().stream>>>(in(0), in(1), N, word, V, out.as<int32_t>(), W.lookups.as<uint64_t>(), G.trace_err.as<uint32_t>());
            return ATLAS_OK;
        }
        case ATLAS_OP_SOFTMAX: {                                              // SoftmaxLastAxis { scale }: rows = the leading dimensions, N = the last one
            if (!need_inputs(1) || !same_len() || nd.p[0] != (int64_t)gr::MODEL_SCALE)
                return fail(ATLAS_EINVAL, "graph: SoftmaxLastAxis needs one operand of the output's shape and scale = MODEL_SCALE (14): its clamp table is compiled for it");
            const size_t N = nd.dims.back(), F = T / N;
            if (F < 1 || N < 2 || N > 65536) return fail(ATLAS_EINVAL, "graph: SoftmaxLastAxis needs 2 <= last axis <= 65536");
            const ExpLut* L = nullptr;
            if (int rc = atlas_rt_exp_lut(&L)) return rc;
            NodeWitness& W = G.wit[nd.idx];
            W.softmax.reset(new SoftmaxWitness());
            SoftmaxWitness& Sm = *W.softmax;
            Sm.F = F; Sm.N = N; Sm.log2_base = L->log2_base; Sm.lk_hi = gr::log2u(L->hi.size()); Sm.lk_lo = gr::log2u(L->lo.size());
            for (DevBuf* b : {&Sm.R, &Sm.exp_q, &Sm.exp_hi, &Sm.exp_lo, &Sm.r_exp, &Sm.z, &Sm.z_hi, &Sm.z_lo, &Sm.e}) HIP_TRY(b->alloc(T * 4));
            for (DevBuf* b : {&Sm.max_k, &Sm.argmax_k, &Sm.exp_sum, &Sm.inv_sum}) HIP_TRY(b->alloc(F * 4));
            for (DevBuf* b : {&Sm.idx_R, &Sm.idx_rexp, &Sm.idx_z, &Sm.idx_zhi, &Sm.idx_zlo}) HIP_TRY(b->alloc(T * 8));
            SoftmaxOut O{out.as<int32_t>(), Sm.R.as<int32_t>(), Sm.exp_q.as<int32_t>(), Sm.exp_hi.as<int32_t>(), Sm.exp_lo.as<int32_t>(), Sm.r_exp.as<int32_t>(),
                         Sm.z.as<int32_t>(), Sm.z_hi.as<int32_t>(), Sm.z_lo.as<int32_t>(), Sm.e.as<int32_t>(), Sm.max_k.as<int32_t>(), Sm.argmax_k.as<int32_t>(),
                         Sm.exp_sum.as<int32_t>(), Sm.inv_sum.as<int32_t>(), Sm.idx_R.as<uint64_t>(), Sm.idx_rexp.as<uint64_t>(), Sm.idx_z.as<uint64_t>(),
                         Sm.idx_zhi.as<uint64_t>(), Sm.idx_zlo.as<uint64_t>()};
            k_softmax_rows<<<(unsigned)F, 256, 0, rt().stream>>>(in(0), (uint32_t)N, (int32_t)1 << gr::MODEL_SCALE, (uint32_t)L->log2_base,
                                                             (int32_t)(L->hi.size() << L->log2_base), L->d_hi, L->d_lo, O);
            return ATLAS_OK;
        }
        default: return fail(ATLAS_EINVAL, "graph_trace: operator not supported by the device executor");
    }
}

}  // namespace

extern "C" {

int atlas_graph_new(atlas_graph_t* out) {
    if (!out) return fail(ATLAS_EINVAL, "graph_new");
    *out = new atlas_graph();
    return ATLAS_OK;
}
int atlas_graph_free(atlas_graph_t G) { delete G; return ATLAS_OK; }

int atlas_graph_add_node(atlas_graph_t G, size_t idx, int op, const size_t* inputs, size_t n_inputs, const size_t* dims, size_t n_dims, const int64_t* params,
                         size_t n_params, const size_t* shape, size_t n_shape, const int32_t* constant) {
    if (!G || (!inputs && n_inputs) || !dims || n_dims == 0 || n_dims > MAXR || (!params && n_params) || n_params > 6 || (!shape && n_shape))
        return fail(ATLAS_EINVAL, "graph_add_node: null / out-of-range argument");
    if (G->nodes.count(idx)) return fail(ATLAS_EINVAL, "graph_add_node: node index already present");
    Node nd; nd.idx = idx; nd.op = op;
    nd.inputs.assign(inputs, inputs + n_inputs);
    nd.dims.assign(dims, dims + n_dims);
    if (!gr::all_pow2(nd.dims)) return fail(ATLAS_EINVAL, "graph_add_node: every dimension must be a power of two (pad the model's shapes)");
    for (size_t i = 0; i < n_inputs; i++) if (inputs[i] >= idx || !G->nodes.count(inputs[i])) return fail(ATLAS_EINVAL, "graph_add_node: inputs must name earlier nodes");
    for (size_t i = 0; i < n_params; i++) nd.p[i] = params[i];
    if (n_shape) nd.shape.assign(shape, shape + n_shape);
    if (op == ATLAS_OP_CONSTANT) {
        if (!constant) return fail(ATLAS_EINVAL, "graph_add_node: Constant without data");
        nd.constant.assign(constant, constant + nd.numel());
    }
    G->nodes[idx] = std::move(nd);
    G->clear_trace();
    return ATLAS_OK;
}

int atlas_graph_set_outputs(atlas_graph_t G, const size_t* idx, size_t n) {
    if (!G || !idx || n == 0) return fail(ATLAS_EINVAL, "graph_set_outputs");
    for (size_t i = 0; i < n; i++) if (!G->nodes.count(idx[i])) return fail(ATLAS_EINVAL, "graph_set_outputs: unknown node");
    G->outputs.assign(idx, idx + n);
    return ATLAS_OK;
}

size_t atlas_graph_num_nodes(atlas_graph_t G) { return G ? G->nodes.size() : 0; }

// Model::trace: inputs in the order of the graph's Input nodes (ascending index), each padded_len(dims) i32 on the host
int atlas_graph_trace(atlas_graph_t G, const int32_t* const* inputs, size_t n_inputs) {
    PROF("atlas_graph_trace");
    NEED_INIT();
    if (!G || (!inputs && n_inputs)) return fail(ATLAS_EINVAL, "graph_trace: null argument");
    if (n_inputs != G->input_nodes().size()) return fail(ATLAS_EINVAL, "graph_trace: one tensor per Input node expected");
    for (size_t i = 0; i < n_inputs; i++) if (!inputs[i]) return fail(ATLAS_EINVAL, "graph_trace: null input tensor");
    G->clear_trace_keep_constants();
    G->trace_err.free();
    size_t next = 0;
    for (auto& kv : G->nodes) {
        int rc = exec_node(*G, kv.second, inputs, next);
        if (rc) { G->clear_trace(); return rc; }
    }
    if (G->trace_err.p) {                          // a Gather ran: did every index name a row of its dictionary?  (one word back, one wait)
        uint32_t bad = 0;
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        hipError_t e = hipMemcpyAsync(&bad, G->trace_err.p, 4, hipMemcpyDeviceToHost, rt().stream);
        if (e == hipSuccess) e = hipStreamSynchronize(rt().stream);
        if (e != hipSuccess) { G->clear_trace(); return fail(ATLAS_ENODEV, "graph_trace: reading the gather range flag", e); }
        if (bad) { G->clear_trace(); return fail(ATLAS_EINVAL, "graph_trace: a Gather index lies outside its dictionary (the reference's tensor indexing panics)"); }
    }
    G->traced = true;
    return ATLAS_OK;
}

int atlas_graph_node_output(atlas_graph_t G, size_t idx, int32_t* host_out, size_t cap, size_t* len) {
    if (!G || !len) return fail(ATLAS_EINVAL, "graph_node_output: null argument");
    auto it = G->nodes.find(idx);
    if (it == G->nodes.end() || !G->tensor(idx)) return fail(ATLAS_ESTATE, "graph_node_output: node not traced");
    *len = gr::padded_len(it->second.dims);
    if (!host_out) return ATLAS_OK;
    if (cap < *len) return fail(ATLAS_EINVAL, "graph_node_output: buffer too small");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    HIP_TRY(hipMemcpyAsync(host_out, G->tensor(idx), *len * 4, hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    return ATLAS_OK;
}

}  // extern "C"
