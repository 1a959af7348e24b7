// SRS handle, multi-scalar multiplication and one-hot commitment: host side + C-ABI.
// Mirrors (paths under the jolt-atlas tree):
//   SRS / KZGProverKey                 joltworks/src/poly/commitment/hyperkzg/kzg.rs:18-143
//   VariableBaseMSM::{msm, msm_field_elements}   joltworks/src/msm/mod.rs:27-190
//   UnivariateKZG::commit_as_univariate          kzg.rs:285-298
//   HyperKZG::commit_one_hot                     hyperkzg/mod.rs:520-554
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_curve.hpp"
#include "shard_group.hpp"
#include "host_field.hpp"
#include "hyperkzg_kernels.hip.h"
#include "msm_kernels.hip.h"
#include "msm_tab_kernels.hip.h"
#include "runtime.hpp"
#include "srs.hpp"

using namespace atlas;
namespace H = atlas_host;
using atlas_rt::fail;
using atlas_rt::rt;


namespace {

// grow-only device workspace shared by MSM calls (serialised by rt().mu)
void release_arenas();
struct Workspace {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        bool hooked = false;              // callers hold rt().mu
        for (auto f : rt().at_shutdown) hooked |= f == &release_arenas;
        if (!hooked) rt().at_shutdown.push_back(&release_arenas);
        if (bytes <= cap) return ATLAS_OK;
        if (p) hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(msm workspace)", e);
        cap = bytes;
        return ATLAS_OK;
    }
};
// the arenas and side streams of a runtime (one per process by default, one per thread that owns a device: runtime.hpp)
struct MsmState {
    Workspace ws;
    Workspace hk_arena;            // HyperKZG::open: Pi_0.., B, h_k and the scan scratch
    Workspace ws_side, ws_tab;     // second lane: the narrow pipeline of msm_device_multi; third: its fixed-base pipeline
    hipStream_t side_stream = nullptr, side_stream2 = nullptr;
    hipEvent_t side_event = nullptr;
};
inline MsmState& msm_state() {       // callers hold rt().mu
    atlas_rt::Runtime& R = rt();
    if (!R.msm_ws) R.msm_ws = new MsmState();
    return *static_cast<MsmState*>(R.msm_ws);
}
#define ws (msm_state().ws)
#define hk_arena (msm_state().hk_arena)
#define ws_side (msm_state().ws_side)
#define ws_tab (msm_state().ws_tab)
#define side_stream (msm_state().side_stream)
#define side_stream2 (msm_state().side_stream2)
#define side_event (msm_state().side_event)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

MsmShape pick_shape(size_t n) {
    MsmShape S;
    // window width by problem size: bucket folding costs ~2 * n_windows * 2^(c-1) additions; c = 13 keeps a
    // window's 4096 counters in LDS for the counting sort (measured at 2^22: 11.9 ms vs 19.1 ms for c = 16)
    S.c = n >= (1u << 24) ? 14 : n >= (1u << 18) ? 13 : n >= (1u << 13) ? 12 : n >= (1u << 8) ? 8 : 6;   // 2^24: 38.7 ms (c = 14) vs 40.1 ms
    if (const char* e = getenv("ATLAS_MSM_C")) { int v = atoi(e); if (v >= 2 && v <= 16) S.c = (uint32_t)v; }   // experiments
    S.n_windows = (255 + S.c - 1) / S.c;
    S.bpw = 1u << (S.c - 1);
    return S;
}

inline int grid_for(size_t work, int cap = 4096) {
    size_t b = (work + MSM_THREADS - 1) / MSM_THREADS;
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = cap;
    return (int)b;
}

void to_out(const H::G1Aff& a, atlas_g1_affine_t* out) {
    std::memcpy(out->x.l, a.x.l, 32);
    std::memcpy(out->y.l, a.y.l, 32);
    out->infinity = (H::q_is_zero(a.x) && H::q_is_zero(a.y)) ? 1 : 0;
}

// core: `launch_digits(digits)` fills the window-major signed digits for shape S
// Several scalar vectors over the same bases (prefixes of the SRS) share one pipeline: n = total number of scalars
// in the concatenation, vector k = [offs[k], offs[k] + lens[k]); out[k] = its MSM.  K = 1 is the plain case.
struct MsmMulti { size_t K; const size_t* lens; const size_t* offs; };

// A lane = the stream and workspace one pipeline runs in.  The default lane is the library stream; HyperKZG's short
// vectors run in a second lane next to the long ones.  With `pend` the call only enqueues; msm_finish() reads the
// window sums back and runs the host Horner.
struct MsmLane { hipStream_t st; Workspace* wk; };
struct MsmPending {
    hipStream_t st = nullptr; const G1Xyzz* wsum = nullptr; uint32_t V = 0; size_t K = 0, n = 0; MsmShape S{};
    atlas_g1_affine_t* out = nullptr;
    std::vector<MsmTile> tiles;      // host source of an enqueued H2D copy: alive until the finish
    std::vector<TabTile> tab_tiles;
    const G1Xyzz* fin = nullptr;     // wide-set fold: 1 + fin_bits points per set, the Horner over them is the host's (msm_finish)
    uint32_t fin_bits = 0, chunk_log = 0;
};

// bucket reduction (msm_kernels.hip.h): lists and run sums live in one carved block
struct ReduceBufs {
    uint32_t* n_lists; uint32_t* medium_list; uint32_t* big_list; uint32_t* big_run_off; G1Xyzz* run_sums;
    static size_t bytes(size_t TB, size_t s_max) {
        const size_t nb = s_max / MSM_MEDIUM_SEGS + 2, nr = s_max / MSM_BIG_RUN + nb + 2;
        return align_up(16 + TB * 4 + nb * 4 * 2, 256) + nr * sizeof(G1Xyzz);
    }
    ReduceBufs(unsigned char* p, size_t TB, size_t s_max) {
        const size_t nb = s_max / MSM_MEDIUM_SEGS + 2;
        n_lists = (uint32_t*)p; medium_list = n_lists + 4; big_list = medium_list + TB; big_run_off = big_list + nb;
        run_sums = (G1Xyzz*)(p + align_up(16 + TB * 4 + nb * 4 * 2, 256));
    }
};
void launch_bucket_reduce(hipStream_t st, const G1Xyzz* partial, const SegMap map, uint32_t TB, G1Xyzz* buckets, const ReduceBufs& B) {
    hipMemsetAsync(B.n_lists, 0, 16, st);
    k_msm_bucket_reduce_small<<<(TB + MSM_THREADS - 1) / MSM_THREADS, MSM_THREADS, 0, st>>>(partial, map, TB, buckets, B.medium_list, B.big_list, B.n_lists);
    k_msm_bucket_reduce_medium<<<TB / 16 + 1 < 1024u ? TB / 16 + 1 : 1024u, MSM_THREADS, 0, st>>>(partial, map, B.medium_list, B.n_lists, buckets);
    k_msm_big_prefix<<<1, 1024, 0, st>>>(map, B.big_list, B.n_lists, B.big_run_off);
    k_msm_bucket_reduce_big1<<<2048, MSM_THREADS, 0, st>>>(partial, map, B.big_list, B.n_lists, B.big_run_off, B.run_sums);
    k_msm_bucket_reduce_big2<<<256, MSM_THREADS, 0, st>>>(B.run_sums, B.big_list, B.n_lists, B.big_run_off, buckets);
}

int msm_finish(const MsmPending& P) {
    std::vector<H::G1X> hw(P.V);
    if (P.fin) {       // set sum = fin[0] + 2^chunk_log * sum_b 2^b fin[1 + b]  (k_msm_fold_sets)
        const uint32_t nf = 1 + P.fin_bits;
        std::vector<H::G1X> hf((size_t)P.V * nf);
        hipError_t ce = hipMemcpyAsync(hf.data(), P.fin, hf.size() * sizeof(G1Xyzz), hipMemcpyDeviceToHost, P.st);
        if (ce == hipSuccess) ce = hipStreamSynchronize(P.st);
        if (ce != hipSuccess) return fail(ATLAS_ENODEV, "msm result copy", ce);
        for (uint32_t v = 0; v < P.V; v++) {
            const H::G1X* f = hf.data() + (size_t)v * nf;
            H::G1X acc = f[P.fin_bits];
            for (int b = (int)P.fin_bits - 1; b >= 1; b--) acc = H::gx_add(H::gx_dbl(acc), f[b]);
            for (uint32_t d = 0; d < P.chunk_log; d++) acc = H::gx_dbl(acc);
            hw[v] = H::gx_add(acc, f[0]);
        }
    } else {
        hipError_t ce = hipMemcpyAsync(hw.data(), P.wsum, P.V * sizeof(G1Xyzz), hipMemcpyDeviceToHost, P.st);
        if (ce == hipSuccess) ce = hipStreamSynchronize(P.st);
        if (ce != hipSuccess) return fail(ATLAS_ENODEV, "msm result copy", ce);
    }
    const bool trace = getenv("ATLAS_TRACE") != nullptr;
    const auto th0 = std::chrono::steady_clock::now();
    // Horner over the windows of each vector: acc = 2^c * acc + W_w
    for (size_t k = 0; k < P.K; k++) {
        const H::G1X* hk = hw.data() + k * P.S.n_windows;
        H::G1X acc = hk[P.S.n_windows - 1];
        for (int w = (int)P.S.n_windows - 2; w >= 0; w--) {
            for (uint32_t d = 0; d < P.S.c; d++) acc = H::gx_dbl(acc);
            acc = H::gx_add(acc, hk[w]);
        }
        to_out(H::gx_to_aff(acc), P.out + k);
    }
    if (trace)
        fprintf(stderr, "[atlas trace] msm n=%zu K=%zu c=%u host Horner %8.3f ms\n", P.n, P.K, P.S.c,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count());
    return ATLAS_OK;
}

template <class DigitsFn>
int msm_core(const G1Affine* bases, size_t n, const MsmShape S, DigitsFn&& launch_digits, atlas_g1_affine_t* out,
             const MsmMulti* multi = nullptr, const MsmLane* lane = nullptr, MsmPending* pend = nullptr) {
    const size_t K = multi ? multi->K : 1;
    const hipStream_t st = lane ? lane->st : rt().stream;
    Workspace& wk = lane ? *lane->wk : ws;
    if (n == 0) { H::G1Aff z{H::q_zero(), H::q_zero()}; for (size_t k = 0; k < K; k++) to_out(z, out + k); return ATLAS_OK; }
    // 32-bit positions: sorted entries, bucket offsets and tile bounds
    if (n * (size_t)S.n_windows >= ((size_t)1 << 32) || K * (size_t)S.n_windows * S.bpw >= ((size_t)1 << 31))
        return fail(ATLAS_EINVAL, "msm: more than 2^32 (scalar, window) pairs in one call; split the input");
    const uint32_t V = (uint32_t)K * S.n_windows;          // virtual windows
    const uint32_t TB = V * S.bpw;
    uint32_t chunk_max = (uint32_t)MSM_CHUNK;
    if (const char* e = getenv("ATLAS_MSM_CHUNK")) { int v = atoi(e); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) chunk_max = (uint32_t)v; }   // experiments
    const uint32_t chunk = S.bpw < chunk_max ? S.bpw : chunk_max;
    const uint32_t n_chunks = TB / chunk;
    const uint32_t chunks_per_window = S.bpw / chunk;

    // tiles of the LDS counting sort: <= `tile` consecutive scalars of one vector
    size_t tile = MSM_TILE;
    if (const char* e = getenv("ATLAS_MSM_TILE")) { long v = atol(e); if (v >= 1024 && v <= (1 << 22)) tile = (size_t)v; }   // experiments
    std::vector<MsmTile> h_tiles;
    for (size_t k = 0; k < K; k++) {
        const size_t o = multi ? multi->offs[k] : 0, len = multi ? multi->lens[k] : n;
        for (size_t t0 = 0; t0 < len; t0 += tile)
            h_tiles.push_back(MsmTile{(uint32_t)(o + t0), (uint32_t)(o + (t0 + tile < len ? t0 + tile : len)),
                                      (uint32_t)(k * S.n_windows * S.bpw), (uint32_t)o});
    }

    // workspace carve-up
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_digits = carve(n * (size_t)S.n_windows * sizeof(int16_t));
    const size_t n_scan_blocks = ((size_t)TB + SCAN_BLOCK - 1) / SCAN_BLOCK;
    const size_t o_bsum = carve((n_scan_blocks + 1) * 4 * 3);
    const size_t o_counts = carve((size_t)(TB + 1) * 4);
    const size_t o_offsets = carve((size_t)(TB + 1) * 4);
    const size_t o_cursor = carve((size_t)(TB + 1) * 4);
    const size_t o_sorted = carve(n * (size_t)S.n_windows * 4);
    // load-balanced accumulation: segments of <= MSM_SEG_LEN sorted entries (see msm_kernels.hip.h)
    // segment length: about eight segments per bucket whatever the size (a bucket of a window holds ~ n / bpw entries;
    // 128 at 2^22 scalars with 13-bit windows).  More segments per bucket than MSM_SMALL_SEGS would send every bucket
    // through the workgroup-per-bucket reduction (measured: 23 ms instead of 6 ms at 2^21 with 32-entry segments, 16.7 ms
    // of a 55 ms MSM at 2^24 with 128-entry ones); fewer leave the accumulation short of threads.
    size_t n_vec = n;
    if (multi) { n_vec = 0; for (size_t k = 0; k < K; k++) n_vec = multi->lens[k] > n_vec ? multi->lens[k] : n_vec; }
    uint32_t seg_len = 8;
    while (seg_len < 2048 && (size_t)seg_len * 8 * S.bpw < n_vec) seg_len <<= 1;
    // ... but never fewer than ~2^17 segments in all (narrow scalars have one or two windows: few, very long buckets —
    // there the workgroup-per-bucket reduction is the right tool and the accumulation needs the threads)
    while (seg_len > 8 && (n * (size_t)S.n_windows) / seg_len < ((size_t)1 << 17)) seg_len >>= 1;
    if (const char* e = getenv("ATLAS_MSM_SEG")) { int v = atoi(e); if (v >= 8 && v <= 4096 && (v & (v - 1)) == 0) seg_len = (uint32_t)v; }   // experiments
    const size_t s_max = (n * (size_t)S.n_windows) / seg_len + TB + 1;
    const size_t o_segc = carve((size_t)(TB + 1) * 4);
    const size_t o_segoff = carve((size_t)(TB + 1) * 4);
    const size_t o_segcur = carve((size_t)(TB + 1) * 4);
    const size_t o_partial = carve(s_max * sizeof(G1Xyzz));
    const size_t o_buckets = carve((size_t)TB * sizeof(G1Xyzz));
    const size_t o_chunks = carve((size_t)n_chunks * sizeof(G1Xyzz));
    const size_t o_wsum = carve((size_t)V * sizeof(G1Xyzz));
    const size_t o_biglist = carve(ReduceBufs::bytes(TB, s_max));
    const size_t o_tiles = carve(h_tiles.size() * sizeof(MsmTile));
    int rc = wk.ensure(off);
    if (rc) return rc;
    unsigned char* W = (unsigned char*)wk.p;
    MsmTile* d_tiles = (MsmTile*)(W + o_tiles);
    int16_t* digits = (int16_t*)(W + o_digits);
    uint32_t* bsum = (uint32_t*)(W + o_bsum);
    uint32_t* boff = bsum + (n_scan_blocks + 1);
    uint32_t* bcur = boff + (n_scan_blocks + 1);
    uint32_t* counts = (uint32_t*)(W + o_counts);
    uint32_t* offsets = (uint32_t*)(W + o_offsets);
    uint32_t* cursor = (uint32_t*)(W + o_cursor);
    uint32_t* sorted = (uint32_t*)(W + o_sorted);
    G1Xyzz* buckets = (G1Xyzz*)(W + o_buckets);
    G1Xyzz* partial = (G1Xyzz*)(W + o_partial);
    uint32_t* segc = (uint32_t*)(W + o_segc);
    uint32_t* seg_off = (uint32_t*)(W + o_segoff);
    uint32_t* seg_cur = (uint32_t*)(W + o_segcur);
    G1Xyzz* chunks = (G1Xyzz*)(W + o_chunks);
    G1Xyzz* wsum = (G1Xyzz*)(W + o_wsum);
    const ReduceBufs rbufs(W + o_biglist, TB, s_max);

    const bool timing = rt().timing && !pend;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    if (timing) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3); hipEventRecord(e0, st); }

    HIP_TRY(hipMemsetAsync(counts, 0, (size_t)(TB + 1) * 4, st));
    // signed digits once, window-major; histogram of all windows; scan; scatter window by window
    launch_digits(digits, st);
    const bool lds_sort = S.bpw <= MSM_LDS_BPW;
    if (multi && !lds_sort) return fail(ATLAS_EINVAL, "msm: batched vectors need an LDS-sortable window width");
    const unsigned n_tiles = (unsigned)h_tiles.size();
    if (lds_sort) HIP_TRY(hipMemcpyAsync(d_tiles, h_tiles.data(), h_tiles.size() * sizeof(MsmTile), hipMemcpyHostToDevice, st));
    if (lds_sort) k_msm_hist_lds<<<dim3(n_tiles, S.n_windows), MSM_THREADS, 0, st>>>(digits, n, d_tiles, S, counts);
    else k_msm_hist_w<<<dim3((unsigned)grid_for(n, 256), S.n_windows), MSM_THREADS, 0, st>>>(digits, n, S, counts);
    k_scan_block_sums<<<(unsigned)n_scan_blocks, 256, 0, st>>>(counts, TB, bsum);
    k_exclusive_scan<<<1, 1024, 0, st>>>(bsum, (uint32_t)n_scan_blocks, boff, bcur);
    k_scan_apply<<<(unsigned)n_scan_blocks, 256, 0, st>>>(counts, TB, boff, offsets, cursor, (uint32_t)n_scan_blocks);
    if (lds_sort) k_msm_scatter_lds<<<dim3(n_tiles, S.n_windows), MSM_THREADS, 0, st>>>(digits, n, d_tiles, S.bpw, cursor, sorted);
    else
        for (uint32_t w = 0; w < S.n_windows; w++)
            k_msm_scatter_w<<<grid_for(n, 1024), MSM_THREADS, 0, st>>>(digits + (size_t)w * n, n, cursor + (size_t)w * S.bpw, sorted);
    if (timing) hipEventRecord(e1, st);
    k_msm_seg_counts<<<(TB + MSM_THREADS - 1) / MSM_THREADS, MSM_THREADS, 0, st>>>(offsets, TB, seg_len, segc);
    k_scan_block_sums<<<(unsigned)n_scan_blocks, 256, 0, st>>>(segc, TB, bsum);
    k_exclusive_scan<<<1, 1024, 0, st>>>(bsum, (uint32_t)n_scan_blocks, boff, bcur);
    k_scan_apply<<<(unsigned)n_scan_blocks, 256, 0, st>>>(segc, TB, boff, seg_off, seg_cur, (uint32_t)n_scan_blocks);
    k_msm_accumulate_seg<<<(unsigned)((s_max + MSM_THREADS - 1) / MSM_THREADS), MSM_THREADS, 0, st>>>(bases, sorted, offsets, seg_off, TB, seg_len, partial);
    launch_bucket_reduce(st, partial, SegMap{seg_off, nullptr, 0}, TB, buckets, rbufs);
    if (timing) hipEventRecord(e2, st);
    k_msm_fold_chunks<<<(n_chunks + MSM_THREADS - 1) / MSM_THREADS, MSM_THREADS, 0, st>>>(buckets, S, chunk, n_chunks, chunks);
    k_g1_group_sum<<<V, MSM_THREADS, 0, st>>>(chunks, chunks_per_window, wsum);
    if (timing) hipEventRecord(e3, st);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(ATLAS_ENODEV, "msm launch", le);

    MsmPending P;
    P.st = st; P.wsum = wsum; P.V = V; P.K = K; P.n = n; P.S = S; P.out = out;
    P.tiles = std::move(h_tiles);
    if (pend) { *pend = std::move(P); return ATLAS_OK; }      // (no timing events on deferred calls)
    rc = msm_finish(P);
    if (rc) return rc;

    if (timing) {
        float a = 0, b = 0, c = 0;
        hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2); hipEventElapsedTime(&c, e2, e3);
        atlas_timing_t t{};
        t.total_ms = a + b + c; t.pass_ms = b; t.fs_ms = a + c;
        t.pass_bytes = (uint64_t)n * (sizeof(G1Affine) + sizeof(Fr));
        t.n_pass = 1; t.n_fs = S.c;   // n_fs carries the window width for the caller
        rt().last_timing = t;
        hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2); hipEventDestroy(e3);
    }
    return ATLAS_OK;
}


// ---- fixed-base path (msm_tab_kernels.hip.h) ----------------------------------------------------------------------
// Digit width for `n` scalars against a table of stride tab_c: the divisor c' = tab_c / q that minimises
// n * ceil(255 / c') mixed additions + ~3 * q * 2^(c'-1) additions of bucket folding.  Returns q = 0 when the
// variable-base plan (20 windows of 13 bits) is at least as cheap.
// bucket bits resolved by the second (LDS) pass of the sort; the first pass partitions by the rest
uint32_t tab_lo_bits() {
    static const uint32_t v = [] {
        if (const char* e = getenv("ATLAS_TAB_LO")) { const int x = atoi(e); if (x >= 6 && x <= (int)TAB_LO_BITS) return (uint32_t)x; }   // experiments
        return TAB_LO_BITS;
    }();
    return v;
}
bool tab_q_ok(uint32_t tab_c, uint32_t levels, uint32_t q) {
    if (q == 0 || tab_c % q) return false;
    const uint32_t c = tab_c / q;
    if (c < 4 || c > 24) return false;
    const uint32_t nd = (255 + c - 1) / c;
    if ((nd + q - 1) / q > levels) return false;
    const uint32_t hi = c - 1 > tab_lo_bits() ? c - 1 - tab_lo_bits() : 0;
    return (q << hi) <= TAB_MAX_BINS;
}
uint32_t pick_tab_q(size_t n, uint32_t tab_c, uint32_t levels) {
    if (const char* e = getenv("ATLAS_MSM_TAB")) { if (atoi(e) == 0) return 0; }
    if (const char* e = getenv("ATLAS_MSM_TAB_Q")) { int v = atoi(e); if (v >= 1 && tab_q_ok(tab_c, levels, (uint32_t)v)) return (uint32_t)v; }   // experiments, tests
    const MsmShape V = pick_shape(n);
    double best = (double)n * V.n_windows + 3.0 * V.n_windows * V.bpw;
    uint32_t best_q = 0;
    for (uint32_t q = 1; q <= 4; q++) {
        if (!tab_q_ok(tab_c, levels, q) || tab_c / q < 8) continue;
        const uint32_t c = tab_c / q;
        const double cost = (double)n * ((255 + c - 1) / c) + 3.0 * q * (double)(1u << (c - 1));
        if (cost < best * 0.95) { best = cost; best_q = q; }
    }
    return best_q;
}

// K scalar vectors (vector k = d_scalars[offs[k] .. offs[k] + lens[k])) against table points [pt_off, pt_off + lens[k]).
int msm_tab_core(const atlas_srs* srs, size_t pt_off, const Fr* d_scalars, size_t n, uint32_t q, atlas_g1_affine_t* out,
                 const MsmMulti* multi = nullptr, const MsmLane* lane = nullptr, MsmPending* pend = nullptr) {
    const size_t K = multi ? multi->K : 1;
    const hipStream_t st = lane ? lane->st : rt().stream;
    Workspace& wk = lane ? *lane->wk : ws;
    TabShape S;
    S.c = srs->tab_c / q; S.q = q;
    S.n_digits = (255 + S.c - 1) / S.c;
    S.lo_bits = S.c - 1 < tab_lo_bits() ? S.c - 1 : tab_lo_bits();
    S.hi_bits = S.c - 1 - S.lo_bits;
    S.level_stride = (uint32_t)srs->tab_len;
    const uint32_t bpw = 1u << (S.c - 1);
    const uint32_t bins_per_vec = q << S.hi_bits;
    const size_t n_ent = n * (size_t)S.n_digits;
    if (n_ent >= ((size_t)1 << 32) || K * (size_t)q * bpw >= ((size_t)1 << 31) ||
        (size_t)srs->tab_levels * srs->tab_len >= ((size_t)1 << 31))
        return fail(ATLAS_EINVAL, "msm: more than 2^32 (scalar, digit) pairs in one call; split the input");
    const uint32_t V = (uint32_t)(K * q);                  // bucket sets
    const uint32_t TB = V * bpw;
    const uint32_t NB = (uint32_t)K * bins_per_vec;
    uint32_t chunk_max = (uint32_t)MSM_CHUNK * 2;          // 2^19 buckets: 8 per thread keeps one wave per SIMD busy
    if (const char* e = getenv("ATLAS_MSM_CHUNK")) { int v = atoi(e); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) chunk_max = (uint32_t)v; }   // experiments
    const uint32_t chunk = bpw < chunk_max ? bpw : chunk_max;
    const uint32_t n_chunks = TB / chunk;
    const uint32_t chunks_per_set = bpw / chunk;

    std::vector<TabTile> h_tiles;
    for (size_t k = 0; k < K; k++) {
        const size_t o = multi ? multi->offs[k] : 0, len = multi ? multi->lens[k] : n;
        for (size_t t0 = 0; t0 < len; t0 += TAB_TILE)
            h_tiles.push_back(TabTile{(uint32_t)(o + t0), (uint32_t)(o + (t0 + TAB_TILE < len ? t0 + TAB_TILE : len)),
                                      (uint32_t)(k * bins_per_vec), (uint32_t)(pt_off - srs->tab_off - o)});      // (table index = point index - tab_off)
    }
    const unsigned n_tiles = (unsigned)h_tiles.size();

    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_ent = carve(n_ent * sizeof(uint2));
    const size_t o_bins = carve((size_t)(NB + 1) * 4 * 3);
    const size_t n_scan_blocks = ((size_t)TB + SCAN_BLOCK - 1) / SCAN_BLOCK;
    const size_t o_bsum = carve((n_scan_blocks + 1) * 4 * 3);
    const size_t o_counts = carve((size_t)(TB + 1) * 4);
    const size_t o_offsets = carve((size_t)(TB + 1) * 4);
    const size_t o_cursor = carve((size_t)(TB + 1) * 4);
    const size_t o_sorted = carve(n_ent * 4);
    // even segments (k_msm_accumulate_even): 32 entries per thread, shorter while that leaves fewer than 2^19 threads
    uint32_t seg_len = 64;
    while (seg_len > 8 && n_ent / seg_len < ((size_t)1 << 19)) seg_len >>= 1;
    while (seg_len < 1024 && (size_t)seg_len * 8 * TB < n_ent) seg_len <<= 1;      // ... but no more than ~8 partials per bucket
    if (const char* e = getenv("ATLAS_MSM_SEG")) { int v = atoi(e); if (v >= 8 && v <= 4096 && (v & (v - 1)) == 0) seg_len = (uint32_t)v; }   // experiments
    const size_t s_max = n_ent / seg_len + TB + 1;           // slots t + b of k_msm_accumulate_even
    const size_t o_partial = carve(s_max * sizeof(G1Xyzz));
    const size_t o_buckets = carve((size_t)TB * sizeof(G1Xyzz));
    const size_t o_chunks = carve((size_t)n_chunks * sizeof(G1Xyzz));
    const size_t o_wsum = carve((size_t)V * sizeof(G1Xyzz));
    uint32_t gs_split = 1;
    while (chunks_per_set / gs_split > 1024 && gs_split < 256) gs_split <<= 1;
    const size_t o_gsum = carve((size_t)V * gs_split * sizeof(G1Xyzz));
    // wide sets (>= 256 chunks each): the fold without scalar multiples (msm_kernels.hip.h, k_msm_fold_pairs)
    const bool wide_fold = chunks_per_set >= MSM_THREADS && chunks_per_set % MSM_THREADS == 0 && !getenv("ATLAS_MSM_FOLD_MUL");
    const uint32_t f_groups = wide_fold ? chunks_per_set / MSM_THREADS : 0;
    uint32_t f_hi = 0, chunk_log = 0;
    while ((1u << f_hi) < f_groups) f_hi++;
    while ((1u << chunk_log) < chunk) chunk_log++;
    const uint32_t f_bits = 8 + f_hi;
    const size_t o_fruns = carve(wide_fold ? (size_t)n_chunks * sizeof(G1Xyzz) : 0);
    const size_t o_fpart = carve(wide_fold ? (size_t)V * 10 * f_groups * sizeof(G1Xyzz) : 0);
    const size_t o_ffin = carve(wide_fold ? (size_t)V * (1 + f_bits) * sizeof(G1Xyzz) : 0);
    const size_t o_biglist = carve(ReduceBufs::bytes(TB, s_max));
    const size_t o_tiles = carve(h_tiles.size() * sizeof(TabTile));
    int rc = wk.ensure(off);
    if (rc) return rc;
    unsigned char* W = (unsigned char*)wk.p;
    uint2* ent = (uint2*)(W + o_ent);
    uint32_t* bin_counts = (uint32_t*)(W + o_bins);
    uint32_t* bin_off = bin_counts + (NB + 1);
    uint32_t* bin_cur = bin_off + (NB + 1);
    uint32_t* bsum = (uint32_t*)(W + o_bsum);
    uint32_t* boff = bsum + (n_scan_blocks + 1);
    uint32_t* bcur = boff + (n_scan_blocks + 1);
    uint32_t* counts = (uint32_t*)(W + o_counts);
    uint32_t* offsets = (uint32_t*)(W + o_offsets);
    uint32_t* cursor = (uint32_t*)(W + o_cursor);
    uint32_t* sorted = (uint32_t*)(W + o_sorted);
    G1Xyzz* partial = (G1Xyzz*)(W + o_partial);
    G1Xyzz* buckets = (G1Xyzz*)(W + o_buckets);
    G1Xyzz* chunks = (G1Xyzz*)(W + o_chunks);
    G1Xyzz* wsum = (G1Xyzz*)(W + o_wsum);
    G1Xyzz* gsum = (G1Xyzz*)(W + o_gsum);
    const ReduceBufs rbufs(W + o_biglist, TB, s_max);
    TabTile* d_tiles = (TabTile*)(W + o_tiles);

    const bool timing = rt().timing && !pend;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    if (timing) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3); hipEventRecord(e0, st); }

    HIP_TRY(hipMemsetAsync(bin_counts, 0, (size_t)(NB + 1) * 4, st));
    HIP_TRY(hipMemsetAsync(counts, 0, (size_t)(TB + 1) * 4, st));
    HIP_TRY(hipMemcpyAsync(d_tiles, h_tiles.data(), h_tiles.size() * sizeof(TabTile), hipMemcpyHostToDevice, st));
    k_tab_part_hist<<<n_tiles, MSM_THREADS, 0, st>>>(d_scalars, d_tiles, S, bin_counts);
    k_exclusive_scan<<<1, 1024, 0, st>>>(bin_counts, NB, bin_off, bin_cur);
    k_tab_part_scatter<<<n_tiles, MSM_THREADS, 0, st>>>(d_scalars, d_tiles, S, bin_cur, ent);
    const unsigned n_chunk_wgs = (unsigned)(((n_ent + TAB_CHUNK - 1) / TAB_CHUNK + 7) / 8 * 8);      // a multiple of the XCD count (tab_chunk_of_block)
    k_tab_bin_hist<<<n_chunk_wgs, MSM_THREADS, 0, st>>>(ent, bin_off, NB, S.lo_bits, counts);
    k_scan_block_sums<<<(unsigned)n_scan_blocks, 256, 0, st>>>(counts, TB, bsum);
    k_exclusive_scan<<<1, 1024, 0, st>>>(bsum, (uint32_t)n_scan_blocks, boff, bcur);
    k_scan_apply<<<(unsigned)n_scan_blocks, 256, 0, st>>>(counts, TB, boff, offsets, cursor, (uint32_t)n_scan_blocks);
    k_tab_bin_scatter<<<n_chunk_wgs, MSM_THREADS, 0, st>>>(ent, bin_off, NB, S.lo_bits, cursor, sorted);
    if (timing) hipEventRecord(e1, st);
    uint32_t seg_log = 0;
    while ((1u << seg_log) < seg_len) seg_log++;
    const SegMap smap{nullptr, offsets, seg_log};
    k_msm_accumulate_even<<<(unsigned)(((n_ent >> seg_log) + 1 + MSM_THREADS - 1) / MSM_THREADS), MSM_THREADS, 0, st>>>(srs->tab, sorted, offsets, TB, seg_log, partial);
    launch_bucket_reduce(st, partial, smap, TB, buckets, rbufs);
    if (timing) hipEventRecord(e2, st);
    MsmShape SS; SS.c = S.c; SS.n_windows = q; SS.bpw = bpw;
    const bool host_horner = wide_fold && V <= 4;
    if (wide_fold) {
        G1Xyzz* fruns = (G1Xyzz*)(W + o_fruns);
        G1Xyzz* fpart = (G1Xyzz*)(W + o_fpart);
        G1Xyzz* ffin = (G1Xyzz*)(W + o_ffin);
        k_msm_fold_pairs<<<(n_chunks + MSM_THREADS - 1) / MSM_THREADS, MSM_THREADS, 0, st>>>(buckets, chunk, n_chunks, chunks, fruns);
        k_msm_fold_groups<<<dim3(f_groups, 2, V), MSM_THREADS, 2 * MSM_THREADS * sizeof(G1Xyzz), st>>>(chunks, fruns, chunks_per_set, fpart);
        k_msm_fold_sets<<<dim3(10 + (f_hi > 8 ? f_hi - 8 : 0), V), MSM_THREADS, 2 * MSM_THREADS * sizeof(G1Xyzz), st>>>(fpart, f_groups, f_hi, ffin);
        if (!host_horner) k_msm_fold_horner<<<(V + 63) / 64, 64, 0, st>>>(ffin, V, f_bits, chunk_log, wsum);
    } else {
    k_msm_fold_chunks<<<(n_chunks + MSM_THREADS - 1) / MSM_THREADS, MSM_THREADS, 0, st>>>(buckets, SS, chunk, n_chunks, chunks);
    if (gs_split > 1) {      // a set has up to 2^20 chunk sums: one workgroup per set would add them 4096 deep
        k_g1_group_sum<<<V * gs_split, MSM_THREADS, 0, st>>>(chunks, chunks_per_set / gs_split, gsum);
        k_g1_group_sum<<<V, MSM_THREADS, 0, st>>>(gsum, gs_split, wsum);
    } else k_g1_group_sum<<<V, MSM_THREADS, 0, st>>>(chunks, chunks_per_set, wsum);
    }
    if (timing) hipEventRecord(e3, st);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(ATLAS_ENODEV, "msm launch", le);

    MsmPending P;
    P.st = st; P.wsum = wsum; P.V = V; P.K = K; P.n = n; P.S = SS; P.out = out;
    if (host_horner) { P.fin = (const G1Xyzz*)(W + o_ffin); P.fin_bits = f_bits; P.chunk_log = chunk_log; }
    P.tab_tiles = std::move(h_tiles);
    if (pend) { *pend = std::move(P); return ATLAS_OK; }
    rc = msm_finish(P);
    if (rc) return rc;
    if (timing) {
        float a = 0, b = 0, c = 0;
        hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2); hipEventElapsedTime(&c, e2, e3);
        atlas_timing_t t{};
        t.total_ms = a + b + c; t.pass_ms = b; t.fs_ms = a + c;
        t.pass_bytes = (uint64_t)n * (sizeof(G1Affine) + sizeof(Fr));
        t.n_pass = 1; t.n_fs = S.c;
        rt().last_timing = t;
        hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2); hipEventDestroy(e3);
    }
    return ATLAS_OK;
}

// does the table cover points [pt_off, pt_off + n)?
inline bool tab_covers(const atlas_srs* srs, size_t pt_off, size_t n) { return srs && srs->tab && pt_off >= srs->tab_off && pt_off + n <= srs->tab_off + srs->tab_len; }

// scalars are Montgomery Fr already on the device
int msm_device(const G1Affine* bases, const Fr* d_scalars, size_t n, atlas_g1_affine_t* out, const atlas_srs* srs = nullptr,
               size_t pt_off = 0) {
    if (n && tab_covers(srs, pt_off, n))
        if (const uint32_t q = pick_tab_q(n, srs->tab_c, srs->tab_levels)) return msm_tab_core(srs, pt_off, d_scalars, n, q, out);
    const MsmShape S = pick_shape(n);
    return msm_core(bases, n, S, [&](int16_t* digits, hipStream_t st) {
        k_msm_digits<<<grid_for(n), MSM_THREADS, 0, st>>>(d_scalars, n, S, digits);
    }, out);
}

// K vectors stored in d_scalars (vector k at offs[k], lens[k] scalars), each against bases[0..lens[k]).  One pipeline
// has one window width; vectors much shorter than the longest would pay for its 4096 buckets per window with a handful
// of points each (HyperKZG's Pi_j halve from 2^(ell-1) down to 2), so they go through a second, narrower pipeline.
constexpr size_t MSM_MULTI_SPLIT = (size_t)1 << 15;

struct MsmGroup {
    std::vector<size_t> idx, gl, go;
    std::vector<atlas_g1_affine_t> res;
    MsmPending pend;
};

static int msm_multi_group(const G1Affine* bases, const Fr* d_scalars, MsmGroup& G, const size_t* lens, const size_t* offs,
                           const MsmShape S, const MsmLane* lane) {
    size_t lo = SIZE_MAX, hi = 0;
    for (size_t k : G.idx) { lo = offs[k] < lo ? offs[k] : lo; hi = offs[k] + lens[k] > hi ? offs[k] + lens[k] : hi; }
    G.gl.resize(G.idx.size()); G.go.resize(G.idx.size()); G.res.resize(G.idx.size());
    for (size_t i = 0; i < G.idx.size(); i++) { G.gl[i] = lens[G.idx[i]]; G.go[i] = offs[G.idx[i]] - lo; }
    const MsmMulti M{G.idx.size(), G.gl.data(), G.go.data()};
    const Fr* sc = d_scalars + lo;
    const size_t n = hi - lo;
    return msm_core(bases, n, S, [&](int16_t* digits, hipStream_t st) {
        k_msm_digits<<<grid_for(n), MSM_THREADS, 0, st>>>(sc, n, S, digits);
    }, G.res.data(), &M, lane, &G.pend);
}


void release_arenas() {            // atlas_shutdown
    for (Workspace* w : {&ws, &hk_arena, &ws_side, &ws_tab}) { if (w->p) hipFree(w->p); w->p = nullptr; w->cap = 0; }
    if (side_stream) { hipStreamDestroy(side_stream); side_stream = nullptr; }
    if (side_stream2) { hipStreamDestroy(side_stream2); side_stream2 = nullptr; }
    if (side_event) { hipEventDestroy(side_event); side_event = nullptr; }
    delete static_cast<MsmState*>(rt().msm_ws);      // the state itself goes with the runtime it belongs to (atlas_shutdown / atlas_shutdown_thread)
    rt().msm_ws = nullptr;
}

// Up to three pipelines side by side: the vectors long enough for the fixed-base table (when the SRS has one), the other
// long ones (13-bit windows), and the short ones (narrow windows).
int msm_device_multi(const G1Affine* bases, const Fr* d_scalars, size_t n_tot, size_t K, const size_t* lens, const size_t* offs,
                     atlas_g1_affine_t* out, const atlas_srs* srs = nullptr) {
    (void)n_tot;
    size_t mx = 0;
    for (size_t k = 0; k < K; k++) mx = lens[k] > mx ? lens[k] : mx;
    const uint32_t tab_q = tab_covers(srs, 0, mx) ? pick_tab_q(mx, srs->tab_c, srs->tab_levels) : 0;
    MsmGroup tab, big, small;
    size_t mx_small = 0, mx_big = 0;
    for (size_t k = 0; k < K; k++) {
        if (tab_q && pick_tab_q(lens[k], srs->tab_c, srs->tab_levels) == tab_q) tab.idx.push_back(k);
        else if (mx >= MSM_MULTI_SPLIT && lens[k] < MSM_MULTI_SPLIT) { small.idx.push_back(k); mx_small = lens[k] > mx_small ? lens[k] : mx_small; }
        else { big.idx.push_back(k); mx_big = lens[k] > mx_big ? lens[k] : mx_big; }
    }
    const int n_side = (small.idx.empty() ? 0 : 1) + ((!tab.idx.empty() && !big.idx.empty()) ? 1 : 0);
    if (n_side) {
        // side pipelines run behind whatever produced the scalars on the library stream, NOT behind the pipeline enqueued
        // there: the event is recorded before that one is enqueued
        if (!side_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&side_stream2, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&side_event, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(side_event, rt().stream));
        // ... and the host waits for the library stream as well: the event alone did not order a side stream behind the
        // library stream reliably when the two share a hardware queue (see Pipeline::begin in batched.hip)
        HIP_TRY(hipStreamSynchronize(rt().stream));
    }
    auto drain = [&]() { if (side_stream) { hipStreamSynchronize(side_stream); hipStreamSynchronize(side_stream2); } hipStreamSynchronize(rt().stream); };
    int rc = ATLAS_OK;
    if (!tab.idx.empty()) {
        size_t lo = SIZE_MAX, hi = 0;
        for (size_t k : tab.idx) { lo = offs[k] < lo ? offs[k] : lo; hi = offs[k] + lens[k] > hi ? offs[k] + lens[k] : hi; }
        tab.gl.resize(tab.idx.size()); tab.go.resize(tab.idx.size()); tab.res.resize(tab.idx.size());
        for (size_t i = 0; i < tab.idx.size(); i++) { tab.gl[i] = lens[tab.idx[i]]; tab.go[i] = offs[tab.idx[i]] - lo; }
        const MsmMulti M{tab.idx.size(), tab.gl.data(), tab.go.data()};
        const MsmLane lane{rt().stream, &ws_tab};
        rc = msm_tab_core(srs, 0, d_scalars + lo, hi - lo, tab_q, tab.res.data(), &M, &lane, &tab.pend);
        if (rc) { drain(); return rc; }
    }
    if (!big.idx.empty()) {
        const bool aside = !tab.idx.empty();
        if (aside) HIP_TRY(hipStreamWaitEvent(side_stream2, side_event, 0));
        const MsmLane lane{aside ? side_stream2 : rt().stream, &ws};
        rc = msm_multi_group(bases, d_scalars, big, lens, offs, pick_shape(mx_big), &lane);
        if (rc) { drain(); return rc; }
    }
    if (!small.idx.empty()) {
        // narrow pipeline: about 16 points per bucket for its longest vector; its latency-bound tail and host Horner
        // overlap the wide pipelines' accumulation
        uint32_t lg = 0;
        while (((size_t)2 << lg) <= mx_small) lg++;
        MsmShape S;
        S.c = lg < 10 ? 6 : lg - 4 > 12 ? 12 : lg - 4;
        S.n_windows = (255 + S.c - 1) / S.c;
        S.bpw = 1u << (S.c - 1);
        HIP_TRY(hipStreamWaitEvent(side_stream, side_event, 0));
        const MsmLane side{side_stream, &ws_side};
        rc = msm_multi_group(bases, d_scalars, small, lens, offs, S, &side);
        if (rc) { drain(); return rc; }
        rc = msm_finish(small.pend);
        if (rc) { drain(); return rc; }
        for (size_t i = 0; i < small.idx.size(); i++) out[small.idx[i]] = small.res[i];
    }
    if (!big.idx.empty()) {
        rc = msm_finish(big.pend);
        if (rc) { drain(); return rc; }
        for (size_t i = 0; i < big.idx.size(); i++) out[big.idx[i]] = big.res[i];
    }
    if (!tab.idx.empty()) {
        rc = msm_finish(tab.pend);
        if (rc) { drain(); return rc; }
        for (size_t i = 0; i < tab.idx.size(); i++) out[tab.idx[i]] = tab.res[i];
    }
    return ATLAS_OK;
}

// narrow integer scalars (msm_u8 .. msm_u64 and the signed split of I32/I64Scalars,
// joltworks/src/msm/mod.rs:38-181): the window plan follows the data's actual bit length so
// the buckets stay balanced, and a negative scalar flips the sign of its digits instead of
// going through two MSMs.
template <typename T>
int msm_small_device(const G1Affine* bases, const T* d_scalars, size_t n, atlas_g1_affine_t* out) {
    if (n == 0) { H::G1Aff z{H::q_zero(), H::q_zero()}; to_out(z, out); return ATLAS_OK; }
    int rc = ws.ensure(256);
    if (rc) return rc;
    unsigned long long* d_max = (unsigned long long*)ws.p;
    HIP_TRY(hipMemsetAsync(d_max, 0, 8, rt().stream));
    k_abs_max<T><<<grid_for(n, 1024), MSM_THREADS, 0, rt().stream>>>(d_scalars, n, d_max);
    unsigned long long mx = 0;
    HIP_TRY(hipMemcpyAsync(&mx, d_max, 8, hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    if (mx == 0) { H::G1Aff z{H::q_zero(), H::q_zero()}; to_out(z, out); return ATLAS_OK; }
    const uint32_t bits = 64 - (uint32_t)__builtin_clzll(mx);
    // the top window must stay below half after the incoming carry (digits >= half go negative)
    const uint32_t total = bits + 2;
    const uint32_t c0 = pick_shape(n).c;
    MsmShape S;
    S.n_windows = (total + c0 - 1) / c0;
    S.c = (total + S.n_windows - 1) / S.n_windows;
    if (S.c < 2) S.c = 2;
    S.bpw = 1u << (S.c - 1);
    return msm_core(bases, n, S, [&](int16_t* digits, hipStream_t st) {
        k_msm_digits_small<T><<<grid_for(n), MSM_THREADS, 0, st>>>(d_scalars, n, S, digits);
    }, out);
}

// K I32Scalars vectors (each against the prefix of the same SRS) through ONE bucket pipeline: the narrow-scalar plan of msm_small_device over the
// concatenated scalars, one window shape from the largest magnitude of them all.  A graph commits ~50 dense advice polynomials (quotients,
// remainders: i32) — one by one they were an abs-max pass, a synchronisation and a latency-bound pipeline each, 0.45 ms a polynomial (r06z).
int msm_small_multi_i32(const G1Affine* bases, const int32_t* d_cat, size_t n_tot, size_t K, const size_t* lens, const size_t* offs, atlas_g1_affine_t* out) {
    const H::G1Aff z{H::q_zero(), H::q_zero()};
    if (n_tot == 0) { for (size_t k = 0; k < K; k++) to_out(z, out + k); return ATLAS_OK; }
    int rc = ws.ensure(256);
    if (rc) return rc;
    unsigned long long* d_max = (unsigned long long*)ws.p;
    HIP_TRY(hipMemsetAsync(d_max, 0, 8, rt().stream));
    k_abs_max<int32_t><<<grid_for(n_tot, 1024), MSM_THREADS, 0, rt().stream>>>(d_cat, n_tot, d_max);
    unsigned long long mx = 0;
    HIP_TRY(hipMemcpyAsync(&mx, d_max, 8, hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    if (mx == 0) { for (size_t k = 0; k < K; k++) to_out(z, out + k); return ATLAS_OK; }
    size_t mx_len = 0;
    for (size_t k = 0; k < K; k++) mx_len = lens[k] > mx_len ? lens[k] : mx_len;
    const uint32_t bits = 64 - (uint32_t)__builtin_clzll(mx), total = bits + 2;       // (the top window stays below half after the incoming carry)
    const uint32_t c0 = pick_shape(mx_len).c;
    MsmShape S;
    S.n_windows = (total + c0 - 1) / c0;
    S.c = (total + S.n_windows - 1) / S.n_windows;
    if (S.c < 2) S.c = 2;
    S.bpw = 1u << (S.c - 1);
    const MsmMulti M{K, lens, offs};
    return msm_core(bases, n_tot, S, [&](int16_t* digits, hipStream_t st) {
        k_msm_digits_small<int32_t><<<grid_for(n_tot), MSM_THREADS, 0, st>>>(d_cat, n_tot, S, digits);
    }, out, &M);
}

}  // namespace

// Transcript::append_point (blake2b.rs:166-187) on the host transcript
static void host_append_point(H::Transcript& T, const atlas_g1_affine_t* p) {
    uint8_t buf[64];
    if (p->infinity) { std::memset(buf, 0, 64); H::tr_append_bytes(T, buf, 64); return; }
    H::Fq x, y, one{{1, 0, 0, 0}};
    std::memcpy(x.l, p->x.l, 32); std::memcpy(y.l, p->y.l, 32);
    H::Fq cx = H::q_mul(x, one), cy = H::q_mul(y, one);       // canonical integers
    for (int i = 0; i < 32; i++) {
        buf[31 - i] = (uint8_t)(cx.l[i >> 3] >> (8 * (i & 7)));
        buf[63 - i] = (uint8_t)(cy.l[i >> 3] >> (8 * (i & 7)));
    }
    H::tr_append_bytes(T, buf, 64);
}

extern "C" {

static_assert(sizeof(atlas_g1_affine_t) == 72, "arkworks G1Affine image");

int atlas_transcript_append_point(atlas_transcript_t* t, const atlas_g1_affine_t* p) {
    if (!t || !p) return fail(ATLAS_EINVAL, "append_point");
    host_append_point(*reinterpret_cast<H::Transcript*>(t), p);
    return ATLAS_OK;
}

int atlas_transcript_append_points(atlas_transcript_t* t, const atlas_g1_affine_t* p, size_t n) {
    if (!t || (!p && n)) return fail(ATLAS_EINVAL, "append_points");
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(t);
    H::tr_append_message(T, "begin_append_vector");
    for (size_t i = 0; i < n; i++) host_append_point(T, &p[i]);
    H::tr_append_message(T, "end_append_vector");
    return ATLAS_OK;
}

int atlas_srs_upload(const void* bases, size_t n, size_t stride_bytes, atlas_srs_t* out) {
    NEED_INIT();
    if (!bases || !out || n == 0 || stride_bytes < 64) return fail(ATLAS_EINVAL, "srs_upload");
    std::vector<G1Affine> tmp(n);
    const unsigned char* p = (const unsigned char*)bases;
    for (size_t i = 0; i < n; i++) {
        std::memcpy(&tmp[i], p + i * stride_bytes, 64);
        if (stride_bytes >= 65 && p[i * stride_bytes + 64]) std::memset(&tmp[i], 0, 64);   // infinity flag
    }
    atlas_srs* s = new atlas_srs();
    hipError_t e = hipMalloc(&s->d, n * sizeof(G1Affine));
    if (e != hipSuccess) { delete s; return fail(ATLAS_ENOMEM, "hipMalloc(srs)", e); }
    s->len = n;
    e = hipMemcpyAsync(s->d, tmp.data(), n * sizeof(G1Affine), hipMemcpyHostToDevice, rt().stream);
    if (e == hipSuccess) e = hipStreamSynchronize(rt().stream);
    if (e != hipSuccess) { (void)hipFree(s->d); delete s; return fail(ATLAS_ENODEV, "srs_upload: copy", e); }
    *out = s;
    return ATLAS_OK;
}

int atlas_srs_generate(const atlas_fr_t* tau, size_t n, atlas_srs_t* out) {
    NEED_INIT();
    if (!tau || !out || n == 0) return fail(ATLAS_EINVAL, "srs_generate");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    // tau^(2^j), j < 64 ; 2^j * G, j < 254 (host, O(1) work)
    std::vector<H::Fr> tp(64);
    std::memcpy(&tp[0], tau, 32);
    for (int j = 1; j < 64; j++) tp[j] = H::mul(tp[j - 1], tp[j - 1]);
    std::vector<H::G1Aff> dt(254);
    H::G1X cur = H::gx_from_aff(H::G1Aff{H::q_from_u64(1), H::q_from_u64(2)});
    for (int j = 0; j < 254; j++) { dt[j] = H::gx_to_aff(cur); cur = H::gx_dbl(cur); }
    DevBuf tpb, dtb;
    HIP_TRY(tpb.alloc(64 * sizeof(Fr)));
    HIP_TRY(dtb.alloc(254 * sizeof(G1Affine)));
    HIP_TRY(hipMemcpyAsync(tpb.p, tp.data(), 64 * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(dtb.p, dt.data(), 254 * sizeof(G1Affine), hipMemcpyHostToDevice, rt().stream));
    atlas_srs* s = new atlas_srs();
    hipError_t e = hipMalloc(&s->d, n * sizeof(G1Affine));
    if (e != hipSuccess) { delete s; return fail(ATLAS_ENOMEM, "hipMalloc(srs)", e); }
    s->len = n;
    k_srs_generate<<<grid_for(n, 8192), MSM_THREADS, 0, rt().stream>>>(tpb.as<Fr>(), dtb.as<G1Affine>(), n, s->d);
    hipError_t se = hipStreamSynchronize(rt().stream);
    if (se != hipSuccess) { hipFree(s->d); delete s; return fail(ATLAS_ENODEV, "srs_generate", se); }
    *out = s;
    return ATLAS_OK;
}

int atlas_srs_len(atlas_srs_t s, size_t* len) {
    if (!s || !len) return fail(ATLAS_EINVAL, "srs_len");
    *len = s->len;
    return ATLAS_OK;
}

int atlas_srs_download(atlas_srs_t s, size_t offset, size_t n, atlas_g1_affine_t* out) {
    NEED_INIT();
    if (!s || !out || offset + n > s->len) return fail(ATLAS_EINVAL, "srs_download: range");
    std::vector<G1Affine> tmp(n);
    HIP_TRY(hipMemcpyAsync(tmp.data(), s->d + offset, n * sizeof(G1Affine), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    for (size_t i = 0; i < n; i++) {
        H::G1Aff a; std::memcpy(&a, &tmp[i], 64);
        to_out(a, &out[i]);
    }
    return ATLAS_OK;
}

int atlas_srs_free(atlas_srs_t s) {
    if (!s) return ATLAS_OK;
    if (s->d) hipFree(s->d);
    if (s->tab) hipFree(s->tab);
    delete s;
    return ATLAS_OK;
}

// Fixed-base table over the first n_points points of the SRS (0 = all): levels 2^(c j) G_i, j < ceil(255 / c).
// window_bits = 0 picks c from n_points (20 at 2^22).  One-time cost per prover key: ~c doublings per point and level.
// MSMs over Fr scalars whose points lie inside the table use it (msm_tab_kernels.hip.h); results are the same group
// elements, so every output stays bit-identical.  Calling it again replaces the table.
int atlas_srs_precompute(atlas_srs_t srs, size_t n_points, uint32_t window_bits) { return atlas_srs_precompute_range(srs, 0, n_points, window_bits); }
// the table over the powers [first_point, first_point + n_points) only: what rank r of a sharded proof needs of it (its range of every long vector)
int atlas_srs_precompute_range(atlas_srs_t srs, size_t first_point, size_t n_points, uint32_t window_bits) {
    NEED_INIT();
    if (!srs) return fail(ATLAS_EINVAL, "srs_precompute: null SRS");
    if (first_point >= srs->len) return fail(ATLAS_EINVAL, "srs_precompute_range: first point beyond the SRS");
    if (n_points == 0 || first_point + n_points > srs->len) n_points = srs->len - first_point;
    uint32_t c = window_bits;
    if (c == 0) {
        uint32_t lg = 0;
        while (((size_t)2 << lg) <= n_points) lg++;
        // widths whose top digit is not a stub: 16 -> 16 digits, 17 -> 15, 20 -> 13, 22 -> 12 (measured: 2^20 3.5 ms at 20
        // against 5.1 at 18; 2^22 8.7 ms at 20; 2^24 31 ms at 22 against 41 at 21)
        c = lg >= 24 ? 22 : lg >= 19 ? 20 : lg >= 16 ? 17 : 16;
    }
    if (c < 8 || c > 24) return fail(ATLAS_EINVAL, "srs_precompute: window_bits must be 0 or 8..24");
    const uint32_t levels = (255 + c - 1) / c;
    if ((size_t)levels * n_points >= ((size_t)1 << 31)) return fail(ATLAS_EINVAL, "srs_precompute: table beyond 2^31 points");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    if (srs->tab) { hipStreamSynchronize(rt().stream); hipFree(srs->tab); srs->tab = nullptr; srs->tab_len = srs->tab_off = 0; srs->tab_c = srs->tab_levels = 0; }
    G1Affine* tab = nullptr;
    hipError_t e = hipMalloc(&tab, (size_t)levels * n_points * sizeof(G1Affine));
    if (e != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(srs table)", e);
    e = hipMemcpyAsync(tab, srs->d + first_point, n_points * sizeof(G1Affine), hipMemcpyDeviceToDevice, rt().stream);
    for (uint32_t l = 1; l < levels && e == hipSuccess; l++) {
        k_tab_next_level<<<grid_for((n_points + TAB_INV_BATCH - 1) / TAB_INV_BATCH, 1 << 16), MSM_THREADS, 0, rt().stream>>>(
            tab + (size_t)(l - 1) * n_points, tab + (size_t)l * n_points, n_points, c);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(rt().stream);
    if (e != hipSuccess) { hipFree(tab); return fail(ATLAS_ENODEV, "srs_precompute", e); }
    srs->tab = tab; srs->tab_len = n_points; srs->tab_off = first_point; srs->tab_c = c; srs->tab_levels = levels;
    return ATLAS_OK;
}

// table parameters: window_bits = 0 when the SRS has no table
int atlas_srs_table_info(atlas_srs_t srs, size_t* n_points, uint32_t* window_bits, uint32_t* levels) {
    if (!srs) return fail(ATLAS_EINVAL, "srs_table_info: null SRS");
    if (n_points) *n_points = srs->tab_len;
    if (window_bits) *window_bits = srs->tab_c;
    if (levels) *levels = srs->tab_levels;
    return ATLAS_OK;
}

int atlas_msm_fr(atlas_srs_t srs, size_t offset, const atlas_fr_t* scalars, size_t n, atlas_g1_affine_t* out) {
    NEED_INIT();
    if (!srs || !out || (!scalars && n)) return fail(ATLAS_EINVAL, "msm_fr: null argument");
    if (offset + n > srs->len)   // ProofVerifyError::KeyLengthError (msm/mod.rs:35-37)
        return fail(ATLAS_EINVAL, "msm_fr: KeyLengthError (bases shorter than scalars)");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    Fr* d_s = nullptr;
    if (n) {
        HIP_TRY(hipMalloc(&d_s, n * sizeof(Fr)));
        HIP_TRY(hipMemcpyAsync(d_s, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
    }
    int rc = msm_device(srs->d + offset, d_s, n, out, srs, offset);
    if (d_s) hipFree(d_s);
    return rc;
}

int atlas_msm_poly(atlas_srs_t srs, size_t offset, atlas_poly_t poly, atlas_g1_affine_t* out) {
    PROF("atlas_msm_poly");
    NEED_INIT();
    if (!srs || !poly || !out) return fail(ATLAS_EINVAL, "msm_poly: null argument");
    if (offset + poly->len > srs->len) return fail(ATLAS_EINVAL, "msm_poly: KeyLengthError (bases shorter than scalars)");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    if (poly->is_i32) return msm_small_device<int32_t>(srs->d + offset, (const int32_t*)poly->d, poly->len, out);   // I32Scalars, msm/mod.rs:88-130
    return msm_device(srs->d + offset, (const Fr*)poly->d, poly->len, out, srs, offset);
}

// msm over device-resident narrow scalars; kind = ATLAS_SCALAR_* (element type of the
// MultilinearPolynomial variant, msm/mod.rs:38-181)
int atlas_msm_small(atlas_srs_t srs, size_t offset, const void* scalars, size_t n, int kind, atlas_g1_affine_t* out) {
    NEED_INIT();
    if (!srs || !out || (!scalars && n) || offset + n > srs->len) return fail(ATLAS_EINVAL, "msm_small: KeyLengthError");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    static const size_t width[] = {1, 2, 4, 8, 4, 8};
    if (kind < 0 || kind > ATLAS_SCALAR_I64) return fail(ATLAS_EINVAL, "msm_small: kind");
    void* d_s = nullptr;
    if (n) {
        HIP_TRY(hipMalloc(&d_s, n * width[kind]));
        hipError_t e = hipMemcpyAsync(d_s, scalars, n * width[kind], hipMemcpyHostToDevice, rt().stream);
        if (e != hipSuccess) { hipFree(d_s); return fail(ATLAS_ENODEV, "msm_small copy", e); }
    }
    int rc = ATLAS_EINVAL;
    const G1Affine* b = srs->d + offset;
    switch (kind) {
        case ATLAS_SCALAR_U8: rc = msm_small_device<uint8_t>(b, (const uint8_t*)d_s, n, out); break;
        case ATLAS_SCALAR_U16: rc = msm_small_device<uint16_t>(b, (const uint16_t*)d_s, n, out); break;
        case ATLAS_SCALAR_U32: rc = msm_small_device<uint32_t>(b, (const uint32_t*)d_s, n, out); break;
        case ATLAS_SCALAR_U64: rc = msm_small_device<uint64_t>(b, (const uint64_t*)d_s, n, out); break;
        case ATLAS_SCALAR_I32: rc = msm_small_device<int32_t>(b, (const int32_t*)d_s, n, out); break;
        case ATLAS_SCALAR_I64: rc = msm_small_device<int64_t>(b, (const int64_t*)d_s, n, out); break;
    }
    if (d_s) hipFree(d_s);
    return rc;
}

int atlas_g1_sum_indexed(atlas_srs_t srs, const uint32_t* indices, size_t n, atlas_g1_affine_t* out) {
    NEED_INIT();
    if (!srs || !out || (!indices && n)) return fail(ATLAS_EINVAL, "g1_sum_indexed: null argument");
    for (size_t i = 0; i < n; i++)
        if (indices[i] >= srs->len) return fail(ATLAS_EINVAL, "g1_sum_indexed: KeyLengthError (index beyond the SRS)");
    if (n == 0) { H::G1Aff z{H::q_zero(), H::q_zero()}; to_out(z, out); return ATLAS_OK; }
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    const int grid = grid_for(n, 1024);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_idx = carve(n * 4), o_part = carve((size_t)grid * sizeof(G1Xyzz)), o_one = carve(sizeof(G1Xyzz));
    int rc = ws.ensure(off);
    if (rc) return rc;
    unsigned char* W = (unsigned char*)ws.p;
    HIP_TRY(hipMemcpyAsync(W + o_idx, indices, n * 4, hipMemcpyHostToDevice, rt().stream));
    k_g1_sum_indexed<<<grid, MSM_THREADS, 0, rt().stream>>>(srs->d, (const uint32_t*)(W + o_idx), n, (G1Xyzz*)(W + o_part));
    k_g1_group_sum<<<1, MSM_THREADS, 0, rt().stream>>>((const G1Xyzz*)(W + o_part), (uint32_t)grid, (G1Xyzz*)(W + o_one));
    H::G1X r;
    HIP_TRY(hipMemcpyAsync(&r, W + o_one, sizeof(G1Xyzz), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    to_out(H::gx_to_aff(r), out);
    return ATLAS_OK;
}


// HyperKZG::batch_commit_one_hot (hyperkzg/mod.rs:558-596; called per lookup op with d = 8 / 16 polynomials,
// prover.rs:236-249): one upload of all index vectors, one launch for all sums, one copy back.
int atlas_commit_one_hot_batch(atlas_srs_t srs, const int32_t* const* nonzero_indices, const size_t* K, const size_t* T, size_t R,
                               atlas_g1_affine_t* out) {
    NEED_INIT();
    if (!srs || (R && (!nonzero_indices || !K || !T || !out))) return fail(ATLAS_EINVAL, "commit_one_hot_batch: null argument");
    if (R == 0) return ATLAS_OK;
    size_t total = 0, t_max = 0;
    for (size_t r = 0; r < R; r++) {
        if (!nonzero_indices[r] && T[r]) return fail(ATLAS_EINVAL, "commit_one_hot_batch: null index vector");
        if (K[r] * T[r] > srs->len) return fail(ATLAS_EINVAL, "commit_one_hot_batch: KeyLengthError (K*T beyond the SRS)");
        total += T[r]; t_max = T[r] > t_max ? T[r] : t_max;
    }
    if (total >= ((size_t)1 << 32)) return fail(ATLAS_EINVAL, "commit_one_hot_batch: more than 2^32 cycles in one call");
    std::vector<int32_t> flat(total ? total : 1);
    std::vector<OneHotRowDesc> rows(R);
    size_t o = 0;
    for (size_t r = 0; r < R; r++) {
        rows[r] = OneHotRowDesc{(uint32_t)o, (uint32_t)T[r]};
        for (size_t t = 0; t < T[r]; t++) {
            const int32_t k = nonzero_indices[r][t];
            if (k >= 0 && (size_t)k >= K[r]) return fail(ATLAS_EINVAL, "commit_one_hot_batch: index >= K");
            flat[o + t] = k;
        }
        o += T[r];
    }
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    // about 2048 workgroups in flight over the batch, at least one per polynomial
    unsigned gx = (unsigned)(2048 / R); if (gx < 1) gx = 1;
    const unsigned need = (unsigned)((t_max + MSM_THREADS - 1) / MSM_THREADS); if (gx > need) gx = need ? need : 1;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t q = off; off = align_up(off + bytes, 256); return q; };
    const size_t o_idx = carve(flat.size() * 4), o_rows = carve(R * sizeof(OneHotRowDesc)), o_part = carve((size_t)R * gx * sizeof(G1Xyzz)),
                 o_sum = carve(R * sizeof(G1Xyzz));
    int rc = ws.ensure(off);
    if (rc) return rc;
    unsigned char* W = (unsigned char*)ws.p;
    HIP_TRY(hipMemcpyAsync(W + o_idx, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(W + o_rows, rows.data(), R * sizeof(OneHotRowDesc), hipMemcpyHostToDevice, rt().stream));
    k_g1_sum_onehot_rows<<<dim3(gx, (unsigned)R), MSM_THREADS, 0, rt().stream>>>(srs->d, (const int32_t*)(W + o_idx), (const OneHotRowDesc*)(W + o_rows), (G1Xyzz*)(W + o_part));
    k_g1_group_sum<<<(unsigned)R, MSM_THREADS, 0, rt().stream>>>((const G1Xyzz*)(W + o_part), gx, (G1Xyzz*)(W + o_sum));
    std::vector<H::G1X> res(R);
    HIP_TRY(hipMemcpyAsync(res.data(), W + o_sum, R * sizeof(G1Xyzz), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    for (size_t r = 0; r < R; r++) to_out(H::gx_to_aff(res[r]), out + r);
    return ATLAS_OK;
}

// The d RaD commitments of one lookup from its T device-resident lookup indices (generate_node_witnesses, witness.rs:136-200
// + HyperKZG::batch_commit_one_hot, hyperkzg/mod.rs:558-596): d = ceil(log_K / log_k_chunk) one-hot polynomials with
// K_chunk = 2^log_k_chunk addresses each, chunk 0 most significant (OneHotParams::lookup_index_chunk, config.rs:73-75).
int atlas_commit_lookup_chunks(atlas_srs_t srs, const uint64_t* d_lookups, size_t log_T, size_t log_K, size_t log_k_chunk, atlas_g1_affine_t* out) {
    PROF("atlas_commit_lookup_chunks");
    NEED_INIT();
    if (!srs || !d_lookups || !out || log_k_chunk == 0 || log_k_chunk > 16 || log_K == 0 || log_K > 64 || log_T > 26)
        return fail(ATLAS_EINVAL, "commit_lookup_chunks: bad argument");
    const size_t T = (size_t)1 << log_T, d = (log_K + log_k_chunk - 1) / log_k_chunk;
    if ((T << log_k_chunk) > srs->len) return fail(ATLAS_EINVAL, "commit_lookup_chunks: KeyLengthError (K*T beyond the SRS)");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    unsigned gx = (unsigned)(2048 / d); if (gx < 1) gx = 1;
    const unsigned need = (unsigned)((T + MSM_THREADS - 1) / MSM_THREADS); if (gx > need) gx = need ? need : 1;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t q = off; off = align_up(off + bytes, 256); return q; };
    const size_t o_part = carve(d * gx * sizeof(G1Xyzz)), o_sum = carve(d * sizeof(G1Xyzz));
    int rc = ws.ensure(off);
    if (rc) return rc;
    unsigned char* W = (unsigned char*)ws.p;
    k_g1_sum_lookup_chunks<<<dim3(gx, (unsigned)d), MSM_THREADS, 0, rt().stream>>>(srs->d, d_lookups, (uint32_t)T, (uint32_t)d, (uint32_t)log_k_chunk, (G1Xyzz*)(W + o_part));
    k_g1_group_sum<<<(unsigned)d, MSM_THREADS, 0, rt().stream>>>((const G1Xyzz*)(W + o_part), gx, (G1Xyzz*)(W + o_sum));
    std::vector<H::G1X> res(d);
    HIP_TRY(hipMemcpyAsync(res.data(), W + o_sum, d * sizeof(G1Xyzz), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    for (size_t r = 0; r < d; r++) to_out(H::gx_to_aff(res[r]), out + r);
    return ATLAS_OK;
}

// commit_witness_polynomials over the lookups of a whole graph (prover.rs:71-87, witness.rs:136-200): the chunk polynomials of n
// lookup families in ONE launch, one copy and one synchronisation, affine through one shared inversion.  out: the d_f commitments
// of family f after those of family f - 1.
int atlas_commit_lookup_chunks_multi(atlas_srs_t srs, const atlas_lookup_family_t* fams, size_t n, size_t log_k_chunk, atlas_g1_affine_t* out) {
    PROF("atlas_commit_lookup_chunks_multi");
    NEED_INIT();
    if (!srs || !fams || !out || n == 0 || log_k_chunk == 0 || log_k_chunk > 16) return fail(ATLAS_EINVAL, "commit_lookup_chunks_multi: bad argument");
    std::vector<LookupChunkRow> rows;
    size_t maxT = 0;
    for (size_t f = 0; f < n; f++) {
        if (!fams[f].d_lookups || fams[f].log_K == 0 || fams[f].log_K > 64 || fams[f].log_T > 26) return fail(ATLAS_EINVAL, "commit_lookup_chunks_multi: family");
        const size_t T = (size_t)1 << fams[f].log_T, d = (fams[f].log_K + log_k_chunk - 1) / log_k_chunk;
        if ((T << log_k_chunk) > srs->len) return fail(ATLAS_EINVAL, "commit_lookup_chunks_multi: KeyLengthError (K*T beyond the SRS)");
        for (size_t i = 0; i < d; i++) rows.push_back(LookupChunkRow{fams[f].d_lookups, (uint32_t)T, (uint32_t)(log_k_chunk * (d - 1 - i))});
        maxT = T > maxT ? T : maxT;
    }
    const size_t R_all = rows.size();
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    static const bool old_cut = getenv("ATLAS_COMMIT_OLD_CUT") != nullptr;      // A-B: one slice count for the whole launch
    // chunk polynomials of a family that coincide with their left neighbour are not computed again (k_lookup_chunk_diff): alias[r] = the row whose
    // commitment row r shares.  ATLAS_COMMIT_NO_ALIAS=1: every row on its own (A/B).
    std::vector<uint32_t> alias(R_all);
    for (size_t r = 0; r < R_all; r++) alias[r] = (uint32_t)r;
    static const bool no_alias = getenv("ATLAS_COMMIT_NO_ALIAS") != nullptr;
    if (!no_alias && !old_cut) {
        std::vector<LookupFamilyDev> hf(n);
        for (size_t f = 0; f < n; f++) hf[f] = LookupFamilyDev{fams[f].d_lookups, (uint32_t)((size_t)1 << fams[f].log_T), (uint32_t)((fams[f].log_K + log_k_chunk - 1) / log_k_chunk)};
        DevBuf d_f, d_diff;
        HIP_TRY(d_f.alloc(n * sizeof(LookupFamilyDev)));
        HIP_TRY(d_diff.alloc(n * sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(d_f.p, hf.data(), n * sizeof(LookupFamilyDev), hipMemcpyHostToDevice, rt().stream));
        k_lookup_chunk_diff<<<(unsigned)n, MSM_THREADS, 0, rt().stream>>>(d_f.as<LookupFamilyDev>(), (uint32_t)log_k_chunk, d_diff.as<uint32_t>());
        std::vector<uint32_t> diff(n);
        HIP_TRY(hipMemcpyAsync(diff.data(), d_diff.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
        size_t r0 = 0;
        for (size_t f = 0; f < n; f++) {
            const size_t d = hf[f].d;
            for (size_t c = 1; c < d && d <= 32; c++) if (!((diff[f] >> (c - 1)) & 1u)) alias[r0 + c] = alias[r0 + c - 1];
            r0 += d;
        }
    }
    std::vector<uint32_t> slot(R_all, 0);                                       // row -> its place among the rows that are computed
    {
        std::vector<LookupChunkRow> uniq;
        for (size_t r = 0; r < R_all; r++) if (alias[r] == r) { slot[r] = (uint32_t)uniq.size(); uniq.push_back(rows[r]); }
        rows.swap(uniq);
    }
    const size_t R = rows.size();
    DevBuf d_rows, d_part, d_sum, d_slices, d_off;
    HIP_TRY(d_rows.alloc(R * sizeof(LookupChunkRow)));
    HIP_TRY(d_sum.alloc(R * sizeof(G1Xyzz)));
    HIP_TRY(hipMemcpyAsync(d_rows.p, rows.data(), R * sizeof(LookupChunkRow), hipMemcpyHostToDevice, rt().stream));
    std::vector<LookupSlice> slices;                // (kept alive until the synchronisation below: pageable sources of asynchronous copies)
    std::vector<uint32_t> offs;
    if (old_cut) {
        unsigned gx = (unsigned)((maxT + 4 * MSM_THREADS - 1) / (4 * MSM_THREADS)); if (gx < 1) gx = 1; if (gx > 64) gx = 64;
        while ((size_t)gx * R < 2048 && gx < 64 && (size_t)gx * MSM_THREADS < maxT) gx *= 2;
        HIP_TRY(d_part.alloc(R * gx * sizeof(G1Xyzz)));
        k_g1_sum_lookup_rows<<<dim3(gx, (unsigned)R), MSM_THREADS, 0, rt().stream>>>(srs->d, d_rows.as<LookupChunkRow>(), (uint32_t)(((uint64_t)1 << log_k_chunk) - 1), d_part.as<G1Xyzz>());
        k_g1_group_sum<<<(unsigned)R, MSM_THREADS, 0, rt().stream>>>(d_part.as<G1Xyzz>(), gx, d_sum.as<G1Xyzz>());
    } else {
        // slices per row: ~32 points per thread, at most 64 slices; with few rows more slices, so that the launch still fills the chip
        size_t pts = 32;
        { size_t tot = 0; for (auto& r : rows) tot += (r.T + MSM_THREADS * pts - 1) / (MSM_THREADS * pts); while (pts > 1 && tot < 1024) { pts /= 2; tot *= 2; } }
        offs.assign(R + 1, 0);
        for (size_t r = 0; r < R; r++) {
            size_t ns = (rows[r].T + MSM_THREADS * pts - 1) / (MSM_THREADS * pts);
            ns = ns < 1 ? 1 : ns > 64 ? 64 : ns;
            offs[r + 1] = offs[r] + (uint32_t)ns;
            for (size_t q = 0; q < ns; q++) slices.push_back(LookupSlice{(uint32_t)r, (uint32_t)q, (uint32_t)ns});
        }
        HIP_TRY(d_slices.alloc(slices.size() * sizeof(LookupSlice)));
        HIP_TRY(d_off.alloc(offs.size() * sizeof(uint32_t)));
        HIP_TRY(d_part.alloc(slices.size() * sizeof(G1Xyzz)));
        HIP_TRY(hipMemcpyAsync(d_slices.p, slices.data(), slices.size() * sizeof(LookupSlice), hipMemcpyHostToDevice, rt().stream));
        HIP_TRY(hipMemcpyAsync(d_off.p, offs.data(), offs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, rt().stream));
        k_g1_sum_lookup_slices<<<(unsigned)slices.size(), MSM_THREADS, 0, rt().stream>>>(srs->d, d_rows.as<LookupChunkRow>(), d_slices.as<LookupSlice>(), (uint32_t)(((uint64_t)1 << log_k_chunk) - 1),
                                                                                    d_part.as<G1Xyzz>());
        k_g1_group_sum_var<<<(unsigned)R, 64, 0, rt().stream>>>(d_part.as<G1Xyzz>(), d_off.as<uint32_t>(), d_sum.as<G1Xyzz>());
    }
    std::vector<H::G1X> res(R);
    HIP_TRY(hipMemcpyAsync(res.data(), d_sum.p, R * sizeof(G1Xyzz), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    std::vector<H::G1Aff> aff(R);
    H::gx_batch_to_aff(res.data(), R, aff.data());
    for (size_t r = 0; r < R_all; r++) to_out(aff[slot[alias[r]]], out + r);
    return ATLAS_OK;
}

// CommitmentScheme::batch_commit (commitment_scheme.rs:76-90 -> UnivariateKZG::commit_batch, kzg.rs:195-243): n
// polynomials against prefixes of the same SRS.  LargeScalars polynomials share one bucket pipeline (their scalars are
// gathered into one buffer: 32 B per coefficient against ~20 point additions); I32Scalars ones take the narrow-scalar plan.
int atlas_commit_batch(atlas_srs_t srs, const atlas_poly_t* polys, size_t n, atlas_g1_affine_t* out) {
    NEED_INIT();
    if (!srs || (n && (!polys || !out))) return fail(ATLAS_EINVAL, "commit_batch: null argument");
    std::vector<size_t> fr_idx;
    size_t tot = 0;
    for (size_t i = 0; i < n; i++) {
        if (!polys[i]) return fail(ATLAS_EINVAL, "commit_batch: null polynomial");
        if (polys[i]->len > srs->len) return fail(ATLAS_EINVAL, "commit_batch: KeyLengthError (bases shorter than scalars)");
        if (!polys[i]->is_i32) { fr_idx.push_back(i); tot += polys[i]->len; }
    }
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    {
        std::vector<size_t> ii;
        size_t itot = 0;
        for (size_t i = 0; i < n; i++) if (polys[i]->is_i32) { ii.push_back(i); itot += polys[i]->len; }
        static const bool one_by_one = getenv("ATLAS_COMMIT_I32_ONE_BY_ONE") != nullptr;       // A/B
        if (ii.size() >= 2 && !one_by_one && itot < ((size_t)1 << 28)) {
            DevBuf cat;
            HIP_TRY(cat.alloc(itot * sizeof(int32_t)));
            std::vector<size_t> il(ii.size()), io(ii.size());
            size_t o = 0;
            for (size_t j = 0; j < ii.size(); j++) {
                const atlas_poly* P = polys[ii[j]];
                il[j] = P->len; io[j] = o;
                HIP_TRY(hipMemcpyAsync(cat.as<int32_t>() + o, P->d, P->len * sizeof(int32_t), hipMemcpyDeviceToDevice, rt().stream));
                o += P->len;
            }
            std::vector<atlas_g1_affine_t> res(ii.size());
            const int rc = msm_small_multi_i32(srs->d, cat.as<int32_t>(), itot, ii.size(), il.data(), io.data(), res.data());
            (void)hipStreamSynchronize(rt().stream);
            if (rc) return rc;
            for (size_t j = 0; j < ii.size(); j++) out[ii[j]] = res[j];
        } else {
            for (size_t i : ii) { int rc = msm_small_device<int32_t>(srs->d, (const int32_t*)polys[i]->d, polys[i]->len, out + i); if (rc) return rc; }
        }
    }
    if (fr_idx.size() == 1) return msm_device(srs->d, (const Fr*)polys[fr_idx[0]]->d, polys[fr_idx[0]]->len, out + fr_idx[0], srs, 0);
    if (fr_idx.empty()) return ATLAS_OK;
    Fr* cat = nullptr;
    hipError_t e = hipMalloc(&cat, tot * sizeof(Fr));
    if (e != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(commit_batch scalars)", e);
    std::vector<size_t> lens(fr_idx.size()), offs(fr_idx.size());
    size_t o = 0;
    for (size_t j = 0; j < fr_idx.size() && e == hipSuccess; j++) {
        const atlas_poly* P = polys[fr_idx[j]];
        lens[j] = P->len; offs[j] = o;
        e = hipMemcpyAsync(cat + o, P->d, P->len * sizeof(Fr), hipMemcpyDeviceToDevice, rt().stream);
        o += P->len;
    }
    std::vector<atlas_g1_affine_t> res(fr_idx.size());
    int rc = e == hipSuccess ? msm_device_multi(srs->d, cat, tot, fr_idx.size(), lens.data(), offs.data(), res.data(), srs)
                             : fail(ATLAS_ENODEV, "commit_batch: gather", e);
    (void)hipStreamSynchronize(rt().stream);
    (void)hipFree(cat);
    if (rc) return rc;
    for (size_t j = 0; j < fr_idx.size(); j++) out[fr_idx[j]] = res[j];
    return ATLAS_OK;
}


// The ceiling the bucket accumulation runs against: chip-wide v_mad_u64_u32 rate, measured now (tools/microbench.hip k_mad;
// the figure is not in MI355X_MICROARCH.md).  bench.py reports the accumulation's multiply-adds against it.
__global__ void k_mad_peak(uint64_t* out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 1, b = blockIdx.x * 40503u + 7;
    uint64_t x0 = a, x1 = b, x2 = a ^ b, x3 = a + b, x4 = 5, x5 = 6, x6 = 7, x7 = 8;
    for (int i = 0; i < iters; i++) {
        x0 = (uint64_t)(uint32_t)x0 * a + x0; x1 = (uint64_t)(uint32_t)x1 * b + x1;
        x2 = (uint64_t)(uint32_t)x2 * a + x2; x3 = (uint64_t)(uint32_t)x3 * b + x3;
        x4 = (uint64_t)(uint32_t)x4 * a + x4; x5 = (uint64_t)(uint32_t)x5 * b + x5;
        x6 = (uint64_t)(uint32_t)x6 * a + x6; x7 = (uint64_t)(uint32_t)x7 * b + x7;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

extern "C" int atlas_measure_mad_peak(double* mads_per_s) {
    NEED_INIT();
    if (!mads_per_s) return fail(ATLAS_EINVAL, "measure_mad_peak");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    const int blocks = 256 * 8, threads = 256, iters = 4096;
    int rc = ws.ensure((size_t)blocks * threads * 8);
    if (rc) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    k_mad_peak<<<blocks, threads, 0, rt().stream>>>((uint64_t*)ws.p, 16);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, rt().stream);
        k_mad_peak<<<blocks, threads, 0, rt().stream>>>((uint64_t*)ws.p, iters);
        hipEventRecord(e1, rt().stream);
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    *mads_per_s = (double)blocks * threads * iters * 8 / (best * 1e-3);
    return ATLAS_OK;
}

// ------------------------------------------------------------------ HyperKZG::open
// ATLAS_TRACE=1: wall-clock of the phases of one open on stderr (each mark drains the stream)
struct HkTrace {
    bool on = getenv("ATLAS_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char* what) {
        if (!on) return;
        hipStreamSynchronize(rt().stream);
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[atlas trace] hyperkzg_open %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
// hyperkzg/mod.rs:400-447 + kzg_open_batch :231-280.  poly is not consumed.
//   com : ell-1 commitments to the folded polynomials Pi_1..Pi_{ell-1}
//   w   : 3 witness commitments,  v : 3*ell evaluations, v[i*ell + j] = Pi_j(u_i)
// K commitments of one group (the Pi_j of the open, or its three witness polynomials) with the points split by contiguous range over the
// ranks of a shard group (SURVEY §8e "MSM: by point range"): rank r commits coefficients [r len / world, (r + 1) len / world) of every
// vector against the same range of the SRS, the partial points cross the board and every rank adds them up (the sum of the ranks'
// points is the same group element whatever the split, so the affine coordinates — and the proof bytes — are those of one GPU).
// Vectors shorter than 1024 per rank stay whole on rank 0 (the others contribute the identity).
static int commit_group_sharded(const atlas_srs* srs, const Fr* d_scalars, size_t K, const size_t* lens, const size_t* offs, atlas_shard_group* sh,
                                atlas_g1_affine_t* out) {
    const size_t world = (size_t)sh->world, rank = (size_t)sh->rank;
    std::vector<atlas_g1_affine_t> mine(K);
    for (auto& p : mine) { std::memset(&p, 0, sizeof(p)); p.infinity = 1; }
    std::vector<size_t> small_idx;
    for (size_t k = 0; k < K; k++) {
        const size_t part = lens[k] / world;
        if (part < 1024) { small_idx.push_back(k); continue; }
        int rc = msm_device(srs->d + rank * part, d_scalars + offs[k] + rank * part, part, &mine[k], srs, rank * part);
        if (rc) return rc;
    }
    if (rank == 0 && !small_idx.empty()) {
        std::vector<size_t> sl(small_idx.size()), so(small_idx.size());
        for (size_t i = 0; i < small_idx.size(); i++) { sl[i] = lens[small_idx[i]]; so[i] = offs[small_idx[i]]; }
        std::vector<atlas_g1_affine_t> res(small_idx.size());
        size_t hi = 0;
        for (size_t i = 0; i < sl.size(); i++) hi = so[i] + sl[i] > hi ? so[i] + sl[i] : hi;
        int rc = msm_device_multi(srs->d, d_scalars, hi, small_idx.size(), sl.data(), so.data(), res.data(), srs);
        if (rc) return rc;
        for (size_t i = 0; i < small_idx.size(); i++) mine[small_idx[i]] = res[i];
    }
    // exchange: 6 points (432 B) per record
    std::vector<atlas_g1_affine_t> all(6 * world);
    for (size_t k0 = 0; k0 < K; k0 += 6) {
        const size_t cnt = K - k0 < 6 ? K - k0 : 6;
        int rc = atlas_shard_allgather(sh, &mine[k0], cnt * sizeof(atlas_g1_affine_t), all.data());
        if (rc) return rc;
        for (size_t q = 0; q < cnt; q++) {
            H::G1X acc = H::gx_inf();
            for (size_t r = 0; r < world; r++) {
                const atlas_g1_affine_t& pt = all[r * cnt + q];
                if (pt.infinity) continue;
                H::G1Aff a; std::memcpy(a.x.l, pt.x.l, 32); std::memcpy(a.y.l, pt.y.l, 32);
                acc = H::gx_add(acc, H::gx_from_aff(a));
            }
            to_out(H::gx_to_aff(acc), &out[k0 + q]);
        }
    }
    return ATLAS_OK;
}

static int hyperkzg_open_impl(atlas_srs_t srs, atlas_poly_t poly, const atlas_u128_t* point, size_t ell,
                              atlas_transcript_t* transcript, atlas_g1_affine_t* com, atlas_g1_affine_t* w, atlas_fr_t* v, atlas_shard_group* sh) {
    NEED_INIT();
    if (!srs || !poly || !point || !transcript || !w || !v || ell == 0 || ell > 30 || (!com && ell > 1))
        return fail(ATLAS_EINVAL, "hyperkzg_open: bad argument");
    const size_t n = (size_t)1 << ell;
    if (poly->is_i32 || poly->len != n) return fail(ATLAS_EINVAL, "hyperkzg_open: poly must be LargeScalars of length 2^ell");
    if (srs->len < n) return fail(ATLAS_EINVAL, "hyperkzg_open: KeyLengthError (SRS shorter than the polynomial)");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(transcript);
    const int mode = rt().challenge_mode;

    // one buffer for Pi_0..Pi_{ell-1} (2n - 2 coefficients), B, and the three h_k
    // (kept between calls: the arena only grows — 6n Fr = 192 MB at 2^20 out of 288 GB)
    const size_t n_blocks = (n + HK_BLOCK - 1) / HK_BLOCK;
    // scratch: xincl 3*(n_blocks*256) | blocktot 3*n_blocks | G 3*n_blocks | total 3 | pw16 3*257 | q ell | evpart 3*eval_blocks | evout 3*ell
    const size_t xs = n_blocks * HK_THREADS;
    size_t eval_blocks = 0;
    for (size_t j = 0, len = n; j < ell; j++, len >>= 1) eval_blocks += (len + HK_BLOCK - 1) / HK_BLOCK;
    const size_t scr_elems = 3 * xs + 6 * n_blocks + 3 + 3 * 257 + ell + 3 * eval_blocks + 3 * ell;
    { int rc = hk_arena.ensure((6 * n + scr_elems) * sizeof(Fr)); if (rc) return rc; }
    Fr* polys = (Fr*)hk_arena.p; Fr* B = polys + 2 * n; Fr* h = B + n; Fr* scr = h + 3 * n;
    Fr* xincl = scr; Fr* blocktot = xincl + 3 * xs; Fr* G = blocktot + 3 * n_blocks; Fr* total = G + 3 * n_blocks;
    Fr* pw16 = total + 3; Fr* dq = pw16 + 3 * 257; Fr* evpart = dq + ell; Fr* evout = evpart + 3 * eval_blocks;
    // The arenas of an opening of 2^22 points and more (this one: 6 n Fr — 3.2 GB at n = 2^24 —, the bucket pipelines' workspaces: 10 GB for the three
    // witness vectors on one GPU) go back when it ends: kept, they sat under the NEXT proof's reduction, whose pools then came on top (per-rank peak of the
    // sharded 12-layer proof 12-14 GB at world 4).  Smaller ones stay cached.  ATLAS_KEEP_ARENAS=1: as before (A/B).
    auto cleanup = [&]() {
        static const bool keep = getenv("ATLAS_KEEP_ARENAS") != nullptr;
        if (keep) return;
        bool any = false;
        for (Workspace* wk : {&ws, &hk_arena, &ws_side, &ws_tab}) any = any || (wk->p && wk->cap >= ((size_t)1 << 30));
        if (!any) return;
        if (side_stream) { (void)hipStreamSynchronize(side_stream); (void)hipStreamSynchronize(side_stream2); }
        (void)hipStreamSynchronize(rt().stream);
        for (Workspace* wk : {&ws, &hk_arena, &ws_side, &ws_tab})
            if (wk->p && wk->cap >= ((size_t)1 << 30)) { (void)hipFree(wk->p); wk->p = nullptr; wk->cap = 0; }
    };

    HkTrace tr;
    // Phase 1: folds (LowToHigh, variable point[ell-i-1])
    HIP_TRY(hipMemcpyAsync(polys, poly->d, n * sizeof(Fr), hipMemcpyDeviceToDevice, rt().stream));
    {
        size_t off = 0, len = n;
        for (size_t i = 0; i + 1 < ell; i++) {
            H::Fr x = H::challenge_to_fr(point[ell - i - 1].lo, point[ell - i - 1].hi, mode);
            Fr xd; std::memcpy(xd.v, x.l, 32);
            k_hk_fold<<<grid_for(len / 2, 2048), HK_THREADS, 0, rt().stream>>>(polys + off, polys + off + len, len / 2, xd, mode == 0);
            off += len; len >>= 1;
        }
    }
    tr.mark("alloc + folds");
    // commitments to Pi_1.. (commit_variable_batch, kzg.rs:227-243)
    // Pi_1 .. Pi_{ell-1} lie back to back after Pi_0: one batched pipeline over all of them (n - 2 scalars)
    if (ell > 1) {
        std::vector<size_t> lens(ell - 1), offs(ell - 1);
        size_t off = 0, len = n >> 1;
        for (size_t i = 1; i < ell; i++) { lens[i - 1] = len; offs[i - 1] = off; off += len; len >>= 1; }
        int rc = sh ? commit_group_sharded(srs, polys + n, ell - 1, lens.data(), offs.data(), sh, com)
                    : msm_device_multi(srs->d, polys + n, off, ell - 1, lens.data(), offs.data(), com, srs);
        if (rc) { cleanup(); return rc; }
    }
    tr.mark("commit Pi_1..");
    // Phase 2: transcript, r, u = [r, -r, r^2]
    H::tr_append_message(T, "begin_append_vector");
    for (size_t i = 0; i + 1 < ell; i++) host_append_point(T, &com[i]);
    H::tr_append_message(T, "end_append_vector");
    const H::Fr r = H::tr_challenge_scalar(T);
    const H::Fr u[3] = {r, H::neg(r), H::mul(r, r)};
    HkPowers P;
    std::vector<H::Fr> hpw16(3 * 257);
    for (int k = 0; k < 3; k++) {
        H::Fr c = u[k];
        for (int b = 0; b < 32; b++) { std::memcpy(P.p2[k][b].v, c.l, 32); c = H::mul(c, c); }
        H::Fr u16; std::memcpy(u16.l, P.p2[k][4].v, 32);
        H::Fr acc = H::one();
        for (int j = 0; j <= 256; j++) { hpw16[k * 257 + j] = acc; acc = H::mul(acc, u16); }
    }
    HIP_TRY(hipMemcpyAsync(pw16, hpw16.data(), 3 * 257 * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
    tr.mark("transcript + powers");
    // Phase 3a: v[i][j] = Pi_j(u_i), all ell polynomials in two launches and one readback
    std::vector<H::Fr> hv(3 * ell);
    {
        k_hk_eval_blocks<<<(unsigned)eval_blocks, HK_THREADS, 0, rt().stream>>>(polys, n, P, evpart, eval_blocks);
        k_hk_eval_sum<<<(unsigned)ell, HK_THREADS, 0, rt().stream>>>(evpart, eval_blocks, n, (uint32_t)ell, evout);
        HIP_TRY(hipMemcpyAsync(hv.data(), evout, 3 * ell * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
        HIP_TRY(hipStreamSynchronize(rt().stream));
    }
    tr.mark("evaluations v");
    std::memcpy(v, hv.data(), 3 * ell * sizeof(Fr));
    H::tr_append_scalars(T, hv.data(), 3 * ell);
    // q powers (challenge_scalar_powers, blake2b.rs:224-231), B = sum q^j Pi_j
    std::vector<H::Fr> q(ell);
    { H::Fr q1 = H::tr_challenge_scalar(T); q[0] = H::one(); for (size_t j = 1; j < ell; j++) q[j] = H::mul(q[j - 1], q1); }
    HIP_TRY(hipMemcpyAsync(dq, q.data(), ell * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
    k_hk_lincomb<<<grid_for(n, 2048), HK_THREADS, 0, rt().stream>>>(polys, n, (uint32_t)ell, dq, B);
    // Phase 3b: witness polynomials h_k = B / (x - u_k) and their commitments
    k_hk_scan_blocks<<<(unsigned)n_blocks, HK_THREADS, 0, rt().stream>>>(B, n, P, xincl, xs, blocktot, n_blocks);
    k_hk_scan_grid<<<1, HK_THREADS, 0, rt().stream>>>(blocktot, n_blocks, P, G, total);
    k_hk_witness<<<(unsigned)n_blocks, HK_THREADS, 0, rt().stream>>>(B, n, P, xincl, xs, G, n_blocks, pw16, h, n);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { cleanup(); return fail(ATLAS_ENODEV, "hyperkzg launch", le); }
    tr.mark("lincomb + witness polys");
    {   // the three witness commitments share the bases: one batched pipeline
        const size_t lens[3] = {n, n, n}, offs[3] = {0, n, 2 * n};
        int rc = sh ? commit_group_sharded(srs, h, 3, lens, offs, sh, w) : msm_device_multi(srs->d, h, 3 * n, 3, lens, offs, w, srs);
        if (rc) { cleanup(); return rc; }
    }
    tr.mark("commit witnesses");
    H::tr_append_message(T, "begin_append_vector");
    for (int k = 0; k < 3; k++) host_append_point(T, &w[k]);
    H::tr_append_message(T, "end_append_vector");
    (void)H::tr_challenge_scalar(T);      // d_0: keeps the transcript in step with the verifier (:276-277)
    cleanup();
    tr.mark("transcript + free");
    return ATLAS_OK;
}

int atlas_hyperkzg_open(atlas_srs_t srs, atlas_poly_t poly, const atlas_u128_t* point, size_t ell,
                        atlas_transcript_t* transcript, atlas_g1_affine_t* com, atlas_g1_affine_t* w, atlas_fr_t* v) {
    PROF("atlas_hyperkzg_open");
    return hyperkzg_open_impl(srs, poly, point, ell, transcript, com, w, v, nullptr);
}
// HyperKZG::open with its four commitment groups (Pi_1.., the three witness polynomials: ~95 % of the open) split by point range over the
// ranks of a shard group.  Every rank passes the WHOLE polynomial (the replicated folds, evaluations and quotients are HBM-bound passes:
// ~1 ms of a 39 ms open at 2^22) and its copy of the SRS, runs the same transcript and returns the same proof; two board exchanges.
int atlas_hyperkzg_open_sharded(atlas_srs_t srs, atlas_shard_group_t group, atlas_poly_t poly, const atlas_u128_t* point, size_t ell,
                                atlas_transcript_t* transcript, atlas_g1_affine_t* com, atlas_g1_affine_t* w, atlas_fr_t* v) {
    if (!group) return fail(ATLAS_EINVAL, "hyperkzg_open_sharded: null group");
    return hyperkzg_open_impl(srs, poly, point, ell, transcript, com, w, v, group->world > 1 ? group : nullptr);
}

// sum of n affine points on the host: combines the per-rank partial MSMs of a point-range
// sharded commitment (one 72-byte point per GPU)
int atlas_g1_sum_affine(const atlas_g1_affine_t* pts, size_t n, atlas_g1_affine_t* out) {
    if ((!pts && n) || !out) return fail(ATLAS_EINVAL, "g1_sum_affine");
    H::G1X acc = H::gx_inf();
    for (size_t i = 0; i < n; i++) {
        if (pts[i].infinity) continue;
        H::G1Aff a; std::memcpy(a.x.l, pts[i].x.l, 32); std::memcpy(a.y.l, pts[i].y.l, 32);
        acc = H::gx_add(acc, H::gx_from_aff(a));
    }
    to_out(H::gx_to_aff(acc), out);
    return ATLAS_OK;
}

}  // extern "C"
