// libatlas_hip.so — runtime, device polynomials, sumcheck driver and the C-ABI
// (include/atlas_hip.h).  Host logic mirrors the reference's prover-side interfaces:
//   SumcheckInstanceProver for EinsumDotProver  jolt-atlas-core/src/onnx_proof/ops/einsum/dot.rs:255-375
//   Sumcheck::prove                              joltworks/src/subprotocols/sumcheck.rs:565-599
//   PolynomialBinding                            joltworks/src/poly/multilinear_polynomial.rs:657-667
// The product path never touches oracle/ and has no CPU fallback: every entry point
// fails with ATLAS_ENODEV when the HIP device is not usable.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_field.hpp"
#include "runtime.hpp"
#include "sumcheck_kernels.hip.h"
#include "sc_consts.hpp"
#include "sumcheck_f9_kernels.hip.h"
#include "shard_group.hpp"

using namespace atlas;
namespace H = atlas_host;

// ------------------------------------------------------------------ runtime state
namespace atlas_rt {
Runtime g_default;
thread_local Runtime* g_cur = nullptr;
int g_thread_runtimes = 0;
int g_device_runtimes[64] = {0};
thread_local hipStream_t tl_lane_stream = nullptr;
thread_local std::string t_err;       // atlas_last_error is per calling thread (commit is called from Rayon workers)
int fail(int code, const char* what, hipError_t e) {
    t_err = what;
    if (e != hipSuccess) { t_err += ": "; t_err += hipGetErrorString(e); }
    return code;
}
}  // namespace atlas_rt
using atlas_rt::fail;
using atlas_rt::rt;
using atlas_rt::MAX_ROUNDS;
using atlas_rt::PINNED_BYTES;

namespace {

inline int grid_for(size_t work) {
    size_t b = (work + SC_THREADS - 1) / SC_THREADS;
    if (b < 1) b = 1;
    static const size_t cap = [] { const char* e = getenv("ATLAS_SC_BLOCKS"); int v = e ? atoi(e) : 0; return (size_t)(v >= 1 && v <= SC_MAX_BLOCKS ? v : SC_MAX_BLOCKS); }();
    if (b > cap) b = cap;
    return (int)b;
}

inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
inline unsigned ilog2(size_t x) { unsigned n = 0; while (x > 1) { x >>= 1; n++; } return n; }

// ---- timing: hip events on the library stream around each launch --------------------
struct Timer {
    std::vector<hipEvent_t> ev;     // pairs (start, stop)
    std::vector<int> kind;          // 0 = data pass, 1 = fs/tail
    std::vector<uint64_t> bytes;
    void begin(int k, uint64_t b) {
        if (!rt().timing) return;
        hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
        hipEventRecord(a, rt().stream); ev.push_back(a); ev.push_back(c); kind.push_back(k); bytes.push_back(b);
    }
    void end() { if (rt().timing) hipEventRecord(ev.back(), rt().stream); }
    void collect() {
        atlas_timing_t t{};
        if (rt().timing && !ev.empty()) {
            hipEventSynchronize(ev.back());
            float ms = 0;
            for (size_t i = 0; i < kind.size(); i++) {
                hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
                if (kind[i] == 0) { t.pass_ms += ms; t.pass_bytes += bytes[i]; t.n_pass++; }
                else { t.fs_ms += ms; t.n_fs++; }
            }
            hipEventElapsedTime(&ms, ev.front(), ev.back());
            t.total_ms = ms;
            for (auto e : ev) hipEventDestroy(e);
        }
        rt().last_timing = t;
        ev.clear(); kind.clear(); bytes.clear();
    }
};

}  // namespace

// ------------------------------------------------------------------ handles
struct atlas_dot_prover {
    atlas_poly_t left = nullptr, right = nullptr, eq = nullptr;
    int schedule = 0;
    size_t a = 0, b = 0;
    size_t n_rounds = 0;
    bool consumed = false;
    // sharded stepping (atlas_dot_shard_*): messages emitted so far / unapplied challenge
    size_t shard_rounds_done = 0;
    bool shard_pending = false;
    bool shard_active = false;
};

// ------------------------------------------------------------------ runtime API
extern "C" {

int atlas_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int atlas_init(int device_ordinal) {
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    if (rt().ready && rt().device == device_ordinal) return ATLAS_OK;
    if (rt().ready) return fail(ATLAS_ESTATE, "atlas_init: already initialised on another device (atlas_shutdown first)");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(ATLAS_ENODEV, "no HIP device available", e);
    if (device_ordinal < 0 || device_ordinal >= n) return fail(ATLAS_EINVAL, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device_ordinal));
    if (!rt().stream) HIP_TRY(hipStreamCreateWithFlags(&rt().stream, hipStreamNonBlocking));
    rt().lib_stream = rt().stream;
    HIP_TRY(hipMalloc(&rt().d_partials, sizeof(Fr) * SC_MAX_BLOCKS * 3));
    HIP_TRY(hipMalloc(&rt().d_ctx, sizeof(ScCtx)));
    HIP_TRY(hipMalloc(&rt().d_proof, sizeof(Fr) * MAX_ROUNDS * 3));
    HIP_TRY(hipMalloc(&rt().d_chal, sizeof(uint64_t) * MAX_ROUNDS * 2));
    HIP_TRY(hipMalloc(&rt().d_finals, sizeof(Fr) * 8));
    HIP_TRY(hipHostMalloc(&rt().h_pinned, PINNED_BYTES, hipHostMallocDefault));
    HIP_TRY(rt().chan.init());
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dot_tail_ch<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (sizeof(Fr) << SC_TAIL_CH_LOG)));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dot_tail2_f9),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (sizeof(Fr) << SC_TAIL_CH_LOG)));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dot_tail_ch<3>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (sizeof(Fr) << SC_TAIL_LOG)));
    if (const char* e = getenv("ATLAS_FS")) rt().fs_mode = std::strcmp(e, "device") == 0 ? ATLAS_FS_DEVICE : ATLAS_FS_HOST;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dot_tail<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (sizeof(Fr) << SC_TAIL_LOG)));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dot_tail<3>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (sizeof(Fr) << SC_TAIL_LOG)));
    rt().device = device_ordinal;
    rt().ready = true;
    if (device_ordinal >= 0 && device_ordinal < 64) __atomic_fetch_add(&atlas_rt::g_device_runtimes[device_ordinal], 1, __ATOMIC_SEQ_CST);
    return ATLAS_OK;
}

int atlas_shutdown(void) {
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    if (!rt().ready) return ATLAS_OK;
    hipStreamSynchronize(rt().stream);
    for (auto f : rt().at_shutdown) f();
    rt().at_shutdown.clear();
    if (rt().eval_stream) { hipStreamDestroy(rt().eval_stream); hipEventDestroy(rt().eval_event); hipFree(rt().d_eval_partials); rt().eval_stream = nullptr; rt().eval_event = nullptr; rt().d_eval_partials = nullptr; rt().eval_event_eq = nullptr; }
    hipFree(rt().d_partials); hipFree(rt().d_ctx); hipFree(rt().d_proof); hipFree(rt().d_chal); hipFree(rt().d_finals);
    hipHostFree(rt().h_pinned);
    rt().chan.release();
    atlas_rt::dev_pool().release();
    hipStreamDestroy(rt().stream);
    if (rt().device >= 0 && rt().device < 64) __atomic_fetch_sub(&atlas_rt::g_device_runtimes[rt().device], 1, __ATOMIC_SEQ_CST);
    rt().ready = false; rt().stream = nullptr; rt().lib_stream = nullptr; rt().d_partials = nullptr; rt().d_ctx = nullptr; rt().d_proof = nullptr;
    rt().d_chal = nullptr; rt().d_finals = nullptr; rt().h_pinned = nullptr;
    return ATLAS_OK;
}

// One process, N devices (include/atlas_hip.h): the calling thread gets a runtime of its own on `device_ordinal` — stream, round channel,
// allocator, MSM workspace — and every later call of THIS thread uses it; other threads keep the process runtime (atlas_init) or their own.
// A thread that already has one on another device must atlas_shutdown_thread first.
int atlas_init_thread(int device_ordinal) {
    if (atlas_rt::g_cur && atlas_rt::g_cur->ready) return atlas_init(device_ordinal);      // same device: no-op; another: ATLAS_ESTATE
    if (!atlas_rt::g_cur) { atlas_rt::g_cur = new atlas_rt::Runtime(); __atomic_fetch_add(&atlas_rt::g_thread_runtimes, 1, __ATOMIC_SEQ_CST); }
    const int rc = atlas_init(device_ordinal);
    if (rc) { delete atlas_rt::g_cur; atlas_rt::g_cur = nullptr; __atomic_fetch_sub(&atlas_rt::g_thread_runtimes, 1, __ATOMIC_SEQ_CST); }
    return rc;
}
int atlas_shutdown_thread(void) {
    if (!atlas_rt::g_cur) return ATLAS_OK;
    const int rc = atlas_shutdown();
    delete atlas_rt::g_cur->pool;
    delete atlas_rt::g_cur;
    atlas_rt::g_cur = nullptr;
    __atomic_fetch_sub(&atlas_rt::g_thread_runtimes, 1, __ATOMIC_SEQ_CST);
    return rc;
}

const char* atlas_last_error(void) { return atlas_rt::t_err.c_str(); }

int atlas_sync(void) {
    NEED_INIT();
    HIP_TRY(hipStreamSynchronize(rt().stream));
    return ATLAS_OK;
}

// How long the waits of the round channel and of the shard board may last (include/atlas_hip.h).  A value <= 0 keeps the setting.
static double g_board_wait_s = 30.0;
int atlas_set_timeouts(double device_wait_s, double host_wait_s, double board_wait_s) {
    NEED_INIT();
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    if (host_wait_s > 0) rt().chan.host_wait_s = host_wait_s;
    if (board_wait_s > 0) g_board_wait_s = board_wait_s;
    if (device_wait_s > 0) { HIP_TRY(hipStreamSynchronize(rt().stream)); HIP_TRY(rt().chan.set_device_timeout(device_wait_s, rt().stream)); }
    return ATLAS_OK;
}
// what the calling thread's runtime holds on its device through the library's allocator (devpool.hpp): bytes in use now and the peak since the
// last reset — the figure a rank of a sharded proof reports (tests/test_gpu_sharded.py, tools/time_sharded.py)
int atlas_device_memory_stats(size_t* in_use_bytes, size_t* peak_bytes, int reset_peak) {
    atlas_rt::DevPool& P = atlas_rt::dev_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    if (in_use_bytes) *in_use_bytes = P.in_use;
    if (peak_bytes) *peak_bytes = P.peak;
    if (reset_peak) P.peak = P.in_use;
    return ATLAS_OK;
}
// diagnosis (ATLAS_DEV_STAMPS=1): write the device's and the host's stamps of the rounds since the last dump to `path` and reset them
int atlas_rt_stamps_dump(const char* path) {
    NEED_INIT();
    if (!path) return fail(ATLAS_EINVAL, "stamps_dump");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    HIP_TRY(rt().chan.stamps_dump(path));
    return ATLAS_OK;
}
int atlas_get_timeouts(double* device_wait_s, double* host_wait_s, double* board_wait_s) {
    if (device_wait_s) *device_wait_s = rt().chan.device_wait_s;
    if (host_wait_s) *host_wait_s = rt().chan.host_wait_s;
    if (board_wait_s) *board_wait_s = g_board_wait_s;
    return ATLAS_OK;
}

int atlas_set_challenge_mode(int mode) {
    if (mode != 0 && mode != 1) return fail(ATLAS_EINVAL, "challenge mode must be 0 or 1");
    rt().challenge_mode = mode;
    return ATLAS_OK;
}
int atlas_get_challenge_mode(void) { return rt().challenge_mode; }

int atlas_set_fs_mode(int mode) {
    if (mode != ATLAS_FS_HOST && mode != ATLAS_FS_DEVICE) return fail(ATLAS_EINVAL, "fs mode must be ATLAS_FS_HOST or ATLAS_FS_DEVICE");
    rt().fs_mode = mode;
    return ATLAS_OK;
}
int atlas_get_fs_mode(void) { return rt().fs_mode; }

int atlas_set_timing(int enabled) { rt().timing = enabled != 0; return ATLAS_OK; }
int atlas_last_timing(atlas_timing_t* out) {
    if (!out) return fail(ATLAS_EINVAL, "null out");
    *out = rt().last_timing;
    return ATLAS_OK;
}

// ------------------------------------------------------------------ transcript API
static_assert(sizeof(atlas_transcript_t) == sizeof(H::Transcript), "transcript image");
static_assert(sizeof(atlas_transcript_t) == sizeof(DevTranscript), "transcript image (device)");
static_assert(sizeof(atlas_fr_t) == sizeof(Fr), "Fr image");

int atlas_transcript_new(atlas_transcript_t* t, const uint8_t* label, size_t n) {
    if (!t || !label || n > 32) return fail(ATLAS_EINVAL, "transcript_new: label must be <= 32 bytes");
    H::tr_new(*reinterpret_cast<H::Transcript*>(t), label, n);
    return ATLAS_OK;
}
int atlas_transcript_append_message(atlas_transcript_t* t, const uint8_t* msg, size_t n) {
    if (!t || !msg || n > 32) return fail(ATLAS_EINVAL, "append_message: msg must be <= 32 bytes");
    H::tr_append_message(*reinterpret_cast<H::Transcript*>(t), msg, n);
    return ATLAS_OK;
}
int atlas_transcript_append_bytes(atlas_transcript_t* t, const uint8_t* b, size_t n) {
    if (!t || (!b && n)) return fail(ATLAS_EINVAL, "append_bytes");
    H::tr_append_bytes(*reinterpret_cast<H::Transcript*>(t), b, n);
    return ATLAS_OK;
}
int atlas_transcript_append_u64(atlas_transcript_t* t, uint64_t x) {
    if (!t) return fail(ATLAS_EINVAL, "append_u64");
    H::tr_append_u64(*reinterpret_cast<H::Transcript*>(t), x);
    return ATLAS_OK;
}
int atlas_transcript_append_scalar(atlas_transcript_t* t, const atlas_fr_t* s) {
    if (!t || !s) return fail(ATLAS_EINVAL, "append_scalar");
    H::tr_append_scalar(*reinterpret_cast<H::Transcript*>(t), *reinterpret_cast<const H::Fr*>(s));
    return ATLAS_OK;
}
int atlas_transcript_append_scalars(atlas_transcript_t* t, const atlas_fr_t* s, size_t n) {
    if (!t || (!s && n)) return fail(ATLAS_EINVAL, "append_scalars");
    H::tr_append_scalars(*reinterpret_cast<H::Transcript*>(t), reinterpret_cast<const H::Fr*>(s), n);
    return ATLAS_OK;
}
int atlas_transcript_challenge_u128(atlas_transcript_t* t, atlas_u128_t* out) {
    if (!t || !out) return fail(ATLAS_EINVAL, "challenge_u128");
    H::tr_challenge_u128(*reinterpret_cast<H::Transcript*>(t), out->lo, out->hi);
    return ATLAS_OK;
}
int atlas_transcript_challenge_scalar(atlas_transcript_t* t, atlas_fr_t* out) {
    if (!t || !out) return fail(ATLAS_EINVAL, "challenge_scalar");
    H::Fr f = H::tr_challenge_scalar(*reinterpret_cast<H::Transcript*>(t));
    std::memcpy(out, &f, 32);
    return ATLAS_OK;
}
int atlas_challenge_to_fr(const atlas_u128_t* c, atlas_fr_t* out) {
    if (!c || !out) return fail(ATLAS_EINVAL, "challenge_to_fr");
    H::Fr f = H::challenge_to_fr(c->lo, c->hi, rt().challenge_mode);
    std::memcpy(out, &f, 32);
    return ATLAS_OK;
}

// ------------------------------------------------------------------ polynomial API
static int poly_alloc(size_t bytes, bool is_i32, size_t len, atlas_poly_t* out) {
    atlas_poly* p = new atlas_poly();
    hipError_t e = hipMalloc(&p->d, bytes ? bytes : 32);
    if (e != hipSuccess) { delete p; return fail(ATLAS_ENOMEM, "hipMalloc(poly)", e); }
    p->len = len; p->cap_bytes = bytes; p->is_i32 = is_i32; p->owned = true;
    *out = p;
    return ATLAS_OK;
}

int atlas_poly_upload_fr(const atlas_fr_t* host, size_t len, atlas_poly_t* out) {
    NEED_INIT();
    if (!host || !out || !is_pow2(len)) return fail(ATLAS_EINVAL, "poly_upload_fr: len must be a power of two");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);                    // rt().stream is read: not while another thread's pipeline has it on a lane
    int rc = poly_alloc(len * sizeof(Fr), false, len, out);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync((*out)->d, host, len * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    return ATLAS_OK;
}

int atlas_poly_upload_i32(const int32_t* host, size_t len, atlas_poly_t* out) {
    NEED_INIT();
    if (!host || !out || !is_pow2(len)) return fail(ATLAS_EINVAL, "poly_upload_i32: len must be a power of two");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    int rc = poly_alloc(len * sizeof(int32_t), true, len, out);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync((*out)->d, host, len * sizeof(int32_t), hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    return ATLAS_OK;
}

int atlas_poly_wrap_device_fr(void* dptr, size_t len, atlas_poly_t* out) {
    PROF("atlas_poly_wrap_device_fr");
    NEED_INIT();
    if (!dptr || !out || !is_pow2(len)) return fail(ATLAS_EINVAL, "poly_wrap_device_fr");
    atlas_poly* p = new atlas_poly();
    p->d = dptr; p->len = len; p->cap_bytes = len * sizeof(Fr); p->is_i32 = false; p->owned = false;
    *out = p;
    return ATLAS_OK;
}

int atlas_poly_wrap_device_i32(void* dptr, size_t len, atlas_poly_t* out) {
    PROF("atlas_poly_wrap_device_i32");
    NEED_INIT();
    if (!dptr || !out || !is_pow2(len)) return fail(ATLAS_EINVAL, "poly_wrap_device_i32");
    atlas_poly* p = new atlas_poly();
    p->d = dptr; p->len = len; p->cap_bytes = len * sizeof(int32_t); p->is_i32 = true; p->owned = false;
    *out = p;
    return ATLAS_OK;
}

int atlas_poly_len(atlas_poly_t p, size_t* len) {
    if (!p || !len) return fail(ATLAS_EINVAL, "poly_len");
    *len = p->len;
    return ATLAS_OK;
}

// i32 -> Fr promotion without binding (to_field over the whole vector)
__global__ __launch_bounds__(SC_THREADS) void k_i32_to_fr(const int32_t* in, Fr* out, size_t n, ScConsts K) {
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * SC_THREADS)
        fe_store(out + i, fr_from_i32(in[i], K.k32));
}

int atlas_poly_download(atlas_poly_t p, atlas_fr_t* host, size_t cap) {
    NEED_INIT();
    if (!p || !host || cap < p->len) return fail(ATLAS_EINVAL, "poly_download: buffer too small");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    if (p->is_i32) {
        Fr* tmp = nullptr;
        HIP_TRY(hipMalloc(&tmp, p->len * sizeof(Fr)));
        k_i32_to_fr<<<grid_for(p->len), SC_THREADS, 0, rt().stream>>>((const int32_t*)p->d, tmp, p->len, make_consts());
        hipError_t e = hipMemcpyAsync(host, tmp, p->len * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream);
        hipStreamSynchronize(rt().stream);
        hipFree(tmp);
        if (e != hipSuccess) return fail(ATLAS_ENODEV, "poly_download", e);
        return ATLAS_OK;
    }
    HIP_TRY(hipMemcpyAsync(host, p->d, p->len * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    return ATLAS_OK;
}

int atlas_poly_clone(atlas_poly_t p, atlas_poly_t* out) {
    PROF("atlas_poly_clone");
    NEED_INIT();
    if (!p || !out) return fail(ATLAS_EINVAL, "poly_clone");
    size_t bytes = p->len * (p->is_i32 ? sizeof(int32_t) : sizeof(Fr));
    int rc = poly_alloc(bytes, p->is_i32, p->len, out);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync((*out)->d, p->d, bytes, hipMemcpyDeviceToDevice, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    return ATLAS_OK;
}

int atlas_poly_free(atlas_poly_t p) {
    PROF("atlas_poly_free");
    if (!p) return ATLAS_OK;
    if (p->owned && p->d) hipFree(p->d);
    delete p;
    return ATLAS_OK;
}

// first bind of an I32Scalars polynomial into a fresh Fr buffer (either order)
__global__ __launch_bounds__(SC_THREADS) void k_bind_i32(const int32_t* z, Fr* out, size_t half, int order,
                                                         Fr r_s64, ScConsts K) {
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < half; i += (size_t)gridDim.x * SC_THREADS) {
        int32_t a = order == ATLAS_HIGH_TO_LOW ? z[i] : z[2 * i];
        int32_t b = order == ATLAS_HIGH_TO_LOW ? z[i + half] : z[2 * i + 1];
        fe_store(out + i, bind_pair_i32(a, b, nullptr, K, r_s64));
    }
}

__global__ __launch_bounds__(SC_THREADS) void k_bind_hi_val(Fr* z, size_t half, Fr r, int r_hi_only) {
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < half; i += (size_t)gridDim.x * SC_THREADS) {
        Fr a = fe_load(z + i), b = fe_load(z + i + half);
        fe_store(z + i, bind_pair(a, b, r, r_hi_only != 0));
    }
}

__global__ __launch_bounds__(SC_THREADS) void k_bind_lo_val(const Fr* z, Fr* out, size_t half, Fr r, int r_hi_only) {
    for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < half; i += (size_t)gridDim.x * SC_THREADS) {
        Fr a = fe_load(z + 2 * i), b = fe_load(z + 2 * i + 1);
        fe_store(out + i, bind_pair(a, b, r, r_hi_only != 0));
    }
}

static Fr host_to_dev(const H::Fr& f) { Fr o; std::memcpy(o.v, f.l, 32); return o; }

// bind on the library stream, no sync
static int poly_bind_async(atlas_poly* p, const atlas_u128_t* rc128, int order) {
    if (p->len < 2) return fail(ATLAS_ESTATE, "bind: polynomial is fully bound");
    const size_t half = p->len / 2;
    H::Fr rh = H::challenge_to_fr(rc128->lo, rc128->hi, rt().challenge_mode);
    const int hi_only = rt().challenge_mode == 0;
    if (p->is_i32) {
        static const uint64_t two64[4] = {0, 1, 0, 0};
        static const H::Fr k64 = H::from_canonical(two64);
        H::Fr rs = H::mul(rh, k64);
        Fr* out = nullptr;
        hipError_t e = hipMalloc(&out, half * sizeof(Fr));
        if (e != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(bind_i32)", e);
        k_bind_i32<<<grid_for(half), SC_THREADS, 0, rt().stream>>>((const int32_t*)p->d, out, half, order,
                                                               host_to_dev(rs), make_consts());
        if (p->owned) { hipStreamSynchronize(rt().stream); hipFree(p->d); }
        p->d = out; p->is_i32 = false; p->owned = true; p->cap_bytes = half * sizeof(Fr);
    } else if (order == ATLAS_HIGH_TO_LOW) {
        k_bind_hi_val<<<grid_for(half), SC_THREADS, 0, rt().stream>>>((Fr*)p->d, half, host_to_dev(rh), hi_only);
    } else {
        Fr* out = nullptr;
        hipError_t e = hipMalloc(&out, half * sizeof(Fr));
        if (e != hipSuccess) return fail(ATLAS_ENOMEM, "hipMalloc(bind_lo)", e);
        k_bind_lo_val<<<grid_for(half), SC_THREADS, 0, rt().stream>>>((const Fr*)p->d, out, half, host_to_dev(rh), hi_only);
        // keep the caller's buffer: copy back in place (stream-ordered), drop the scratch
        hipMemcpyAsync(p->d, out, half * sizeof(Fr), hipMemcpyDeviceToDevice, rt().stream);
        hipStreamSynchronize(rt().stream);
        hipFree(out);
    }
    p->len = half;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ATLAS_ENODEV, "bind launch", e);
    return ATLAS_OK;
}

int atlas_poly_bind(atlas_poly_t p, const atlas_u128_t* r, int order) {
    NEED_INIT();
    if (!p || !r || (order != ATLAS_HIGH_TO_LOW && order != ATLAS_LOW_TO_HIGH)) return fail(ATLAS_EINVAL, "poly_bind");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    int rc = poly_bind_async(p, r, order);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(rt().stream));
    return ATLAS_OK;
}

int atlas_poly_final_claim(atlas_poly_t p, atlas_fr_t* out) {
    NEED_INIT();
    if (!p || !out) return fail(ATLAS_EINVAL, "final_claim");
    if (p->len != 1) return fail(ATLAS_ESTATE, "final_claim: polynomial not fully bound");
    return atlas_poly_download(p, out, 1);
}

// ------------------------------------------------------------------ dot prover
static EqView eq_view_for_round(const atlas_dot_prover* P, size_t round, const Fr* eq_ptr, size_t eq_len) {
    EqView v; v.p = eq_ptr; v.mode = EQ_NONE; v.shift = 0; v.mask = 0; v.half = 0;
    if (P->schedule == ATLAS_EQ_HIGH) {
        if (round < P->a) { v.mode = EQ_PAIR; v.shift = (uint32_t)P->b; v.half = (uint32_t)(eq_len / 2); }
        else { v.mode = EQ_IDX; v.mask = 0; }
    } else if (P->schedule == ATLAS_EQ_LOW) {
        if (round < P->a) { v.mode = EQ_IDX; v.mask = (uint32_t)((1ull << P->b) - 1); }
        else { v.mode = EQ_PAIR; v.shift = 0; v.half = (uint32_t)(eq_len / 2); }
    }
    return v;
}

int atlas_dot_prover_new(atlas_poly_t left, atlas_poly_t right, atlas_poly_t eq, int schedule, size_t a, size_t b,
                         atlas_dot_prover_t* out) {
    PROF("atlas_dot_prover_new");
    NEED_INIT();
    if (!left || !right || !out) return fail(ATLAS_EINVAL, "dot_prover_new: null operand");
    if (left->len != right->len || !is_pow2(left->len)) return fail(ATLAS_EINVAL, "dot_prover_new: operand lengths");
    if (left->is_i32 != right->is_i32) return fail(ATLAS_EINVAL, "dot_prover_new: mixed operand types");
    const size_t n = ilog2(left->len);
    if (n > MAX_ROUNDS) return fail(ATLAS_EINVAL, "dot_prover_new: too many rounds");
    if (schedule == ATLAS_EQ_NONE) {
        if (eq) return fail(ATLAS_EINVAL, "dot_prover_new: eq given with EqSchedule::None");
    } else if (schedule == ATLAS_EQ_HIGH || schedule == ATLAS_EQ_LOW) {
        if (!eq || eq->is_i32) return fail(ATLAS_EINVAL, "dot_prover_new: eq table required (Fr)");
        if (a + b != n) return fail(ATLAS_EINVAL, "dot_prover_new: schedule bits != num_rounds");
        size_t want = schedule == ATLAS_EQ_HIGH ? ((size_t)1 << a) : ((size_t)1 << b);
        if (eq->len != want) return fail(ATLAS_EINVAL, "dot_prover_new: eq table length");
        if (left->is_i32) return fail(ATLAS_EINVAL, "dot_prover_new: i32 operands only with EqSchedule::None");
    } else {
        return fail(ATLAS_EINVAL, "dot_prover_new: unknown schedule");
    }
    atlas_dot_prover* P = new atlas_dot_prover();
    P->left = left; P->right = right; P->eq = eq; P->schedule = schedule; P->a = a; P->b = b; P->n_rounds = n;
    *out = P;
    return ATLAS_OK;
}

int atlas_dot_prover_free(atlas_dot_prover_t P) {
    if (!P) return ATLAS_OK;
    atlas_poly_free(P->left); atlas_poly_free(P->right); atlas_poly_free(P->eq);
    delete P;
    return ATLAS_OK;
}

}  // extern "C"

// sum of per-workgroup partials -> DEG Fr at out[0..DEG)
template <int DEG>
__global__ __launch_bounds__(SC_THREADS) void k_reduce_partials(const Fr* partials, int n_partials, Fr* out) {
    __shared__ Fr red[SC_THREADS / 64][DEG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Fr acc[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) acc[k] = fe_zero();
    for (int b = threadIdx.x; b < n_partials; b += SC_THREADS)
#pragma unroll
        for (int k = 0; k < DEG; k++) acc[k] = fr_add(acc[k], fe_load(partials + (size_t)b * DEG + k));
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        Fr s = fr_wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < DEG) {
        Fr s = red[0][threadIdx.x];
        for (int w = 1; w < SC_THREADS / 64; w++) s = fr_add(s, red[w][threadIdx.x]);
        fe_store(out + threadIdx.x, s);
    }
}

template <int DEG>
static void launch_eval(const atlas_dot_prover* P, const EqView& eq, size_t half, int grid) {
    const ScConsts K = make_consts();
    if (P->left->is_i32)
        k_dot_eval<DEG, int32_t, DevIo><<<grid, SC_THREADS, 0, rt().stream>>>((const int32_t*)P->left->d, (const int32_t*)P->right->d,
                                                                        eq, half, DevIo{rt().d_ctx, rt().d_partials}, K);
    else
        k_dot_eval<DEG, Fr, DevIo><<<grid, SC_THREADS, 0, rt().stream>>>((const Fr*)P->left->d, (const Fr*)P->right->d, eq, half,
                                                                   DevIo{rt().d_ctx, rt().d_partials}, K);
}

extern "C" {

int atlas_dot_compute_message(atlas_dot_prover_t P, size_t round, const atlas_fr_t* previous_claim,
                              atlas_fr_t* coeffs_out, size_t* n_coeffs) {
    NEED_INIT();
    if (!P || !previous_claim || !coeffs_out || !n_coeffs) return fail(ATLAS_EINVAL, "compute_message");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    if (P->consumed) return fail(ATLAS_ESTATE, "compute_message: prover already consumed");
    if (round >= P->n_rounds || P->left->len != ((size_t)1 << (P->n_rounds - round)))
        return fail(ATLAS_ESTATE, "compute_message: round out of order");
    const int deg = P->schedule == ATLAS_EQ_NONE ? 2 : 3;
    const size_t half = P->left->len / 2;
    const int grid = grid_for(half);
    EqView eq = eq_view_for_round(P, round, P->eq ? (const Fr*)P->eq->d : nullptr, P->eq ? P->eq->len : 0);
    if (deg == 2) { launch_eval<2>(P, eq, half, grid); k_reduce_partials<2><<<1, SC_THREADS, 0, rt().stream>>>(rt().d_partials, grid, rt().d_finals + 3); }
    else { launch_eval<3>(P, eq, half, grid); k_reduce_partials<3><<<1, SC_THREADS, 0, rt().stream>>>(rt().d_partials, grid, rt().d_finals + 3); }
    H::Fr ev[3];
    HIP_TRY(hipMemcpyAsync(rt().h_pinned, rt().d_finals + 3, deg * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    std::memcpy(ev, rt().h_pinned, deg * sizeof(Fr));
    H::Fr c[4];
    int nc = H::unipoly_from_evals_and_hint(*reinterpret_cast<const H::Fr*>(previous_claim), ev, deg, c);
    std::memcpy(coeffs_out, c, nc * sizeof(H::Fr));
    *n_coeffs = (size_t)nc;
    return ATLAS_OK;
}

int atlas_dot_input_claim(atlas_dot_prover_t P, atlas_fr_t* out) {
    NEED_INIT();
    if (!P || !out) return fail(ATLAS_EINVAL, "input_claim");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    if (P->consumed || P->left->len != ((size_t)1 << P->n_rounds)) return fail(ATLAS_ESTATE, "input_claim: instance already bound");
    const size_t len = P->left->len;
    const int grid = grid_for(len);
    EqView eq; eq.p = P->eq ? (const Fr*)P->eq->d : nullptr; eq.mode = EQ_NONE; eq.shift = 0; eq.mask = 0; eq.half = 0;
    if (P->schedule == ATLAS_EQ_HIGH) { eq.mode = EQ_PAIR; eq.shift = (uint32_t)P->b; }
    if (P->schedule == ATLAS_EQ_LOW) { eq.mode = EQ_IDX; eq.mask = (uint32_t)((1ull << P->b) - 1); }
    const ScConsts K = make_consts();
    if (P->left->is_i32)
        k_dot_claim<int32_t><<<grid, SC_THREADS, 0, rt().stream>>>((const int32_t*)P->left->d, (const int32_t*)P->right->d, eq, len, rt().d_partials, K);
    else
        k_dot_claim<Fr><<<grid, SC_THREADS, 0, rt().stream>>>((const Fr*)P->left->d, (const Fr*)P->right->d, eq, len, rt().d_partials, K);
    k_reduce_partials<1><<<1, SC_THREADS, 0, rt().stream>>>(rt().d_partials, grid, rt().d_finals + 3);
    HIP_TRY(hipMemcpyAsync(rt().h_pinned, rt().d_finals + 3, sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    std::memcpy(out, rt().h_pinned, sizeof(Fr));
    return ATLAS_OK;
}

int atlas_dot_ingest_challenge(atlas_dot_prover_t P, const atlas_u128_t* r_j, size_t round) {
    NEED_INIT();
    if (!P || !r_j) return fail(ATLAS_EINVAL, "ingest_challenge");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    if (P->consumed) return fail(ATLAS_ESTATE, "ingest_challenge: prover already consumed");
    if (round >= P->n_rounds || P->left->len != ((size_t)1 << (P->n_rounds - round)))
        return fail(ATLAS_ESTATE, "ingest_challenge: round out of order");
    int rc = poly_bind_async(P->left, r_j, ATLAS_HIGH_TO_LOW);
    if (!rc) rc = poly_bind_async(P->right, r_j, ATLAS_HIGH_TO_LOW);
    if (!rc && P->schedule == ATLAS_EQ_HIGH && round < P->a) rc = poly_bind_async(P->eq, r_j, ATLAS_HIGH_TO_LOW);
    if (!rc && P->schedule == ATLAS_EQ_LOW && round >= P->a) rc = poly_bind_async(P->eq, r_j, ATLAS_HIGH_TO_LOW);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(rt().stream));
    return ATLAS_OK;
}

int atlas_dot_final_claims(atlas_dot_prover_t P, atlas_fr_t out[3]) {
    NEED_INIT();
    if (!P || !out) return fail(ATLAS_EINVAL, "final_claims");
    if (P->consumed) return fail(ATLAS_ESTATE, "final_claims: prover consumed by atlas_sumcheck_prove_dot");
    if (P->left->len != 1) return fail(ATLAS_ESTATE, "final_claims: rounds remaining");
    int rc = atlas_poly_download(P->left, &out[0], 1);
    if (!rc) rc = atlas_poly_download(P->right, &out[1], 1);
    if (rc) return rc;
    if (P->eq) {
        if (P->eq->len != 1) return fail(ATLAS_ESTATE, "final_claims: eq not fully bound");
        return atlas_poly_download(P->eq, &out[2], 1);
    }
    H::Fr one = H::one();
    std::memcpy(&out[2], &one, 32);
    return ATLAS_OK;
}

}  // extern "C"

// ---- whole-instance prove: the round loop as a launch chain, transcript on device ----
template <int DEG>
static int prove_dot_impl(atlas_dot_prover* P, const atlas_fr_t* input_claim, atlas_transcript_t* transcript,
                          atlas_fr_t* compressed_polys, atlas_u128_t* challenges, atlas_fr_t final_claims[3]) {
    const ScConsts K = make_consts();
    const size_t n = P->n_rounds;
    const int mode = rt().challenge_mode;
    const int hi_only = mode == 0;
    Timer tm;

    // control block: transcript + running claim
    ScCtx* hctx = reinterpret_cast<ScCtx*>(rt().h_pinned);
    std::memset(hctx, 0, sizeof(ScCtx));
    std::memcpy(&hctx->tr, transcript, sizeof(DevTranscript));
    std::memcpy(&hctx->claim, input_claim, sizeof(Fr));
    HIP_TRY(hipMemcpyAsync(rt().d_ctx, hctx, sizeof(ScCtx), hipMemcpyHostToDevice, rt().stream));

    size_t len = P->left->len;
    size_t eq_len = P->eq ? P->eq->len : 0;
    Fr* eqp = P->eq ? (Fr*)P->eq->d : nullptr;
    size_t rounds_done = 0;       // messages emitted so far
    int pending = 0;              // challenge of round rounds_done-1 not yet bound
    const size_t esz = P->left->is_i32 ? sizeof(int32_t) : sizeof(Fr);
    // degree-2 / LargeScalars / challenge mode 0: data passes on the 29-bit lazy-limb kernels
    const bool use_f9 = DEG == 2 && !P->left->is_i32 && mode == 0;
    auto grid_f9 = [](size_t work) { size_t b = (work + SC_THREADS - 1) / SC_THREADS; return (int)(b < 1 ? 1 : b > 256 ? 256 : b); };

    if (n > (size_t)SC_TAIL_LOG) {
        // round 0 message over the untouched operands
        {
            const size_t half = len / 2;
            const int grid = use_f9 ? grid_f9(half) : grid_for(half);
            EqView eq = eq_view_for_round(P, 0, eqp, eq_len);
            tm.begin(0, 2 * len * esz);
            if (use_f9) k_dot_eval2_f9<DevIoF9><<<grid, SC_THREADS, 0, rt().stream>>>((const Fr*)P->left->d, (const Fr*)P->right->d, half, DevIoF9{rt().d_ctx, rt().d_partials});
            else launch_eval<DEG>(P, eq, half, grid);
            tm.end();
            tm.begin(1, 0);
            k_fs_round<DEG><<<1, SC_THREADS, 0, rt().stream>>>(rt().d_ctx, rt().d_partials, grid, rt().d_proof, rt().d_chal, K, 1, mode);
            tm.end();
            rounds_done = 1; pending = 1;
        }
        // fused passes: bind r_j, emit message j+1, while the operands exceed the tail size
        while (len > ((size_t)1 << SC_TAIL_LOG)) {
            const size_t j = rounds_done - 1;          // challenge index being bound
            const size_t q = len / 4;
            const int grid = use_f9 ? grid_f9(q) : grid_for(q);
            bool fuse_eq = false;
            if (P->schedule == ATLAS_EQ_HIGH && j < P->a) {
                tm.begin(0, (eq_len + eq_len / 2) * sizeof(Fr));
                k_bind_hi<<<grid_for(eq_len / 2), SC_THREADS, 0, rt().stream>>>(eqp, eq_len / 2, &rt().d_ctx->r, hi_only);
                tm.end();
                eq_len /= 2;
            } else if (P->schedule == ATLAS_EQ_LOW && j >= P->a) {
                fuse_eq = true;
            }
            EqView eq = eq_view_for_round(P, j + 1, eqp, fuse_eq ? eq_len / 2 : eq_len);
            uint64_t bytes = 2 * len * esz + 2 * (len / 2) * sizeof(Fr);
            if (fuse_eq) bytes += (eq_len + eq_len / 2) * sizeof(Fr);
            tm.begin(0, bytes);
            if (P->left->is_i32) {
                DevBuf Lb, Rb;                              // handed to the operands below; freed here on an early return
                HIP_TRY(Lb.alloc((len / 2) * sizeof(Fr)));
                HIP_TRY(Rb.alloc((len / 2) * sizeof(Fr)));
                Fr *Ld = Lb.as<Fr>(), *Rd = Rb.as<Fr>();
                k_dot_bind_eval<DEG, int32_t, false, DevIo><<<grid, SC_THREADS, 0, rt().stream>>>(
                    (const int32_t*)P->left->d, (const int32_t*)P->right->d, Ld, Rd, nullptr, eq, q,
                    DevIo{rt().d_ctx, rt().d_partials}, K, hi_only);
                tm.end();
                HIP_TRY(hipStreamSynchronize(rt().stream));
                if (P->left->owned) (void)hipFree(P->left->d);
                if (P->right->owned) (void)hipFree(P->right->d);
                P->left->d = Lb.release(); P->left->is_i32 = false; P->left->owned = true;
                P->right->d = Rb.release(); P->right->is_i32 = false; P->right->owned = true;
            } else if (fuse_eq) {
                k_dot_bind_eval<DEG, Fr, true, DevIo><<<grid, SC_THREADS, 0, rt().stream>>>(
                    (const Fr*)P->left->d, (const Fr*)P->right->d, (Fr*)P->left->d, (Fr*)P->right->d, eqp, eq, q,
                    DevIo{rt().d_ctx, rt().d_partials}, K, hi_only);
                tm.end();
                eq_len /= 2;
            } else if (use_f9) {
                // last fused pass hands canonical residues to the LDS tail kernel
                if (len / 2 <= ((size_t)1 << SC_TAIL_LOG))
                    k_dot_bind_eval2_f9<true, DevIoF9><<<grid, SC_THREADS, 0, rt().stream>>>((Fr*)P->left->d, (Fr*)P->right->d, q, DevIoF9{rt().d_ctx, rt().d_partials});
                else
                    k_dot_bind_eval2_f9<false, DevIoF9><<<grid, SC_THREADS, 0, rt().stream>>>((Fr*)P->left->d, (Fr*)P->right->d, q, DevIoF9{rt().d_ctx, rt().d_partials});
                tm.end();
            } else {
                k_dot_bind_eval<DEG, Fr, false, DevIo><<<grid, SC_THREADS, 0, rt().stream>>>(
                    (const Fr*)P->left->d, (const Fr*)P->right->d, (Fr*)P->left->d, (Fr*)P->right->d, nullptr, eq, q,
                    DevIo{rt().d_ctx, rt().d_partials}, K, hi_only);
                tm.end();
            }
            len /= 2;
            tm.begin(1, 0);
            k_fs_round<DEG><<<1, SC_THREADS, 0, rt().stream>>>(rt().d_ctx, rt().d_partials, grid, rt().d_proof + rounds_done * DEG,
                                                          rt().d_chal + 2 * rounds_done, K, 0, mode);
            tm.end();
            rounds_done += 1;
        }
    }

    // tail: all remaining rounds in one launch
    {
        TailArgs A;
        A.L = P->left->d; A.R = P->right->d; A.eq = eqp;
        A.len = (uint32_t)len; A.eq_len = (uint32_t)eq_len;
        A.src_i32 = P->left->is_i32 ? 1 : 0;
        A.sched = P->schedule; A.a = (uint32_t)P->a; A.b = (uint32_t)P->b;
        A.round0 = (uint32_t)rounds_done; A.n_rounds = (uint32_t)n;
        A.first = rounds_done == 0 ? 1 : 0;
        A.pending_bind = pending;
        A.challenge_mode = mode;
        tm.begin(1, 0);
        k_dot_tail<DEG><<<1, SC_THREADS, 3 * (sizeof(Fr) << SC_TAIL_LOG), rt().stream>>>(A, rt().d_ctx, rt().d_proof, rt().d_chal,
                                                                                     rt().d_finals, K);
        tm.end();
    }
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(ATLAS_ENODEV, "sumcheck launch", le);

    // results: one D2H each, stream ordered
    uint8_t* hp = reinterpret_cast<uint8_t*>(rt().h_pinned);
    const size_t proof_bytes = n * DEG * sizeof(Fr), chal_bytes = n * 2 * sizeof(uint64_t);
    HIP_TRY(hipMemcpyAsync(hp, rt().d_proof, proof_bytes ? proof_bytes : 32, hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipMemcpyAsync(hp + 8192, rt().d_chal, chal_bytes ? chal_bytes : 16, hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipMemcpyAsync(hp + 12288, rt().d_finals, 3 * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipMemcpyAsync(hp + 16384, rt().d_ctx, sizeof(ScCtx), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    std::memcpy(compressed_polys, hp, proof_bytes);
    std::memcpy(challenges, hp + 8192, chal_bytes);
    std::memcpy(final_claims, hp + 12288, 3 * sizeof(Fr));
    std::memcpy(transcript, hp + 16384, sizeof(DevTranscript));
    tm.collect();
    P->consumed = true;
    P->left->len = 1; P->right->len = 1;
    if (P->eq) P->eq->len = 1;
    return ATLAS_OK;
}

// ---- whole-instance prove over the round channel: transcript on the host, every launch enqueued up front ----
// One round of Sumcheck::prove on the host (sumcheck.rs:578-588) for a degree-2/3 message given as the
// evaluations at 0, 2[, 3]: from_evals_and_hint, compress, append_to_transcript, challenge_scalar_optimized,
// evaluate.  row <- compressed coefficients (DEG of them).
static void host_fs_round(H::Transcript& T, int deg, const H::Fr* ev, H::Fr& claim, H::Fr* row, atlas_u128_t* chal, int mode) {
    H::Fr c[4];
    H::unipoly_from_evals_and_hint(claim, ev, deg, c);
    H::tr_append_message(T, "UniPoly_begin");
    H::tr_append_scalar(T, c[0]);
    for (int k = 2; k <= deg; k++) H::tr_append_scalar(T, c[k]);
    H::tr_append_message(T, "UniPoly_end");
    H::tr_challenge_u128(T, chal->lo, chal->hi);
    const H::Fr r = H::challenge_to_fr(chal->lo, chal->hi, mode);
    H::Fr v = c[deg];
    for (int k = deg - 1; k >= 0; k--) v = H::add(H::mul(v, r), c[k]);       // same value as UniPoly::evaluate (unipoly.rs:229-245)
    claim = v;
    row[0] = c[0];
    for (int k = 2; k <= deg; k++) row[k - 1] = c[k];
}

// grp != null: P holds this rank's strided shard of ONE instance spread over grp->world ranks (coefficient k * world +
// rank; HighToLow pairs stay on a rank).  Each round the ranks exchange their partial sums through the group's shared-
// memory board and every rank runs the same transcript step; after the local rounds the `world` remaining coefficients
// of each operand are exchanged and the last log2(world) rounds are host arithmetic on every rank.  input_claim = the
// claim of the whole instance.  Degree-2 instances.
template <int DEG>
static int prove_dot_channel(atlas_dot_prover* P, const atlas_fr_t* input_claim, atlas_transcript_t* transcript,
                             atlas_fr_t* compressed_polys, atlas_u128_t* challenges, atlas_fr_t final_claims[3],
                             atlas_shard_group* grp = nullptr) {
    using atlas_rt::Channel;
    Channel& C = rt().chan;
    const ScConsts K = make_consts();
    const size_t n = P->n_rounds;
    const int mode = rt().challenge_mode;
    const int hi_only = mode == 0;
    Timer tm;
    if (C.abort_dirty) { HIP_TRY(hipMemsetAsync(C.d_abort, 0, 4, rt().stream)); C.abort_dirty = false; }

    // tags: records of round k = tag0 + k (k = n: the final claims), challenge of round k = tag0 + n + 1 + k
    const uint32_t tag0 = C.take_tags(2 * n + 2);
    const size_t slot0 = C.take_slots(n);
    auto mtag = [&](size_t k) { return tag0 + (uint32_t)k; };
    auto rtag = [&](size_t k) { return tag0 + (uint32_t)(n + 1 + k); };
    struct Mail { atlas::Chunk* base; size_t blocks; int radix, shl; };
    std::vector<Mail> mails(n);
    std::vector<size_t> waiters(n, 1);          // workgroups of the launches that wait for challenge k

    size_t len = P->left->len;
    size_t eq_len = P->eq ? P->eq->len : 0;
    Fr* eqp = P->eq ? (Fr*)P->eq->d : nullptr;
    size_t rounds_done = 0;
    int pending = 0;
    const bool was_i32 = P->left->is_i32;
    const size_t esz = was_i32 ? sizeof(int32_t) : sizeof(Fr);
    const bool use_f9 = DEG == 2 && !was_i32 && mode == 0;
    static const size_t f9_cap = [] { const char* e = getenv("ATLAS_F9_BLOCKS"); int v = e ? atoi(e) : 0; return (size_t)(v >= 1 && v <= 2048 ? v : 256); }();   // experiments
    // experiments: a second cap for the passes of at least 2^ATLAS_F9_BIG_MIN_LOG quads (two wavefronts per SIMD issue the non-multiply instructions at twice the rate, tools/exp_mad.hip)
    static const size_t f9_big_cap = [] { const char* e = getenv("ATLAS_F9_BIG_BLOCKS"); int v = e ? atoi(e) : 0; return (size_t)(v >= 1 && v <= 2048 ? v : 0); }();
    static const size_t f9_big_min = [] { const char* e = getenv("ATLAS_F9_BIG_MIN_LOG"); int v = e ? atoi(e) : 19; return (size_t)1 << (v >= 10 && v <= 30 ? v : 19); }();
    auto grid_f9 = [](size_t work) { const size_t cap = f9_big_cap && work >= f9_big_min ? f9_big_cap : f9_cap; size_t b = (work + SC_THREADS - 1) / SC_THREADS; return (int)(b < 1 ? 1 : b > cap ? cap : b); };
    const size_t tail_log = P->schedule == ATLAS_EQ_NONE ? SC_TAIL_CH_LOG : SC_TAIL_LOG;
    void *old_l = nullptr, *old_r = nullptr;       // i32 sources replaced by the first fused pass

    if (n > tail_log) {
        {
            const size_t half = len / 2;
            static const size_t f9_r0_cap = [] { const char* e = getenv("ATLAS_F9_R0_BLOCKS"); int v = e ? atoi(e) : 0; return (size_t)(v >= 1 && v <= 2048 ? v : 0); }();   // experiments: round 0 alone
            int grid = use_f9 ? grid_f9(half) : grid_for(half);
            if (use_f9 && f9_r0_cap) { const size_t b = (half + SC_THREADS - 1) / SC_THREADS; grid = (int)(b > f9_r0_cap ? f9_r0_cap : b); }
            atlas::Chunk* reg = C.alloc((size_t)grid * ch_stride(DEG));
            const RoundIo io = C.io(reg, mtag(0), (size_t)-1, 0);
            EqView eq = eq_view_for_round(P, 0, eqp, eq_len);
            tm.begin(0, 2 * len * esz);
            if (use_f9) k_dot_eval2_f9<ChanIoF9><<<grid, SC_THREADS, 0, rt().stream>>>((const Fr*)P->left->d, (const Fr*)P->right->d, half, ChanIoF9{io});
            else if (was_i32) k_dot_eval<DEG, int32_t, ChanIo><<<grid, SC_THREADS, 0, rt().stream>>>((const int32_t*)P->left->d, (const int32_t*)P->right->d, eq, half, ChanIo{io, mode}, K);
            else k_dot_eval<DEG, Fr, ChanIo><<<grid, SC_THREADS, 0, rt().stream>>>((const Fr*)P->left->d, (const Fr*)P->right->d, eq, half, ChanIo{io, mode}, K);
            tm.end();
            mails[0] = Mail{reg, (size_t)grid, use_f9 ? 29 : 32, use_f9 ? 5 : 0};
            rounds_done = 1; pending = 1;
        }
    }
    // Review item 3 (round 6): the first bind pass of a 2^22 instance had grown 77 -> 99 us under the tracer while round 0 got faster.  The
    // device's own clock (ATLAS_DEV_STAMPS=1, tools/bind_wait_split.py -> profiles/r06g_bind_wait_split*.txt) says the pass WAITS 3.6 us — one poll
    // of a challenge that is already there — and works 59 us; entry to entry it is 76 us, as in round 4.  The growth is the tracer's: every
    // launch costs this thread more under it, the dozen launches in front of the transcript's first step then end after round 0's sums have
    // arrived, and the resident bind pass shows the host's lateness as its own duration.  Two orders of enqueueing are kept:
    //   * all launches up front (default: 0.426-0.431 ms per step untraced);
    //   * ATLAS_SC_ENQUEUE_LAZY=1: two launches ahead of the transcript, more only while the current round's first record has not arrived
    //     (the host is never late at round 0; 0.433-0.438 ms untraced — the launches that fall into the short rounds cost more than the
    //     lateness did; strictly two ahead: 0.461-0.471 ms).
    bool tail_enqueued = false;
    size_t tail_round0 = 0;
    atlas::Chunk* tail_reg = nullptr;
    int enq_rc = ATLAS_OK;
    auto enqueue_pass = [&]() -> int {
        const size_t j = rounds_done - 1;          // challenge index being bound
        const size_t q = len / 4;
        // at most 2^15 quads: two lanes per quad (k_dot_bind_eval2_f9_pair); ATLAS_F9_NO_PAIR=1 is the A/B
        static const bool no_pair = getenv("ATLAS_F9_NO_PAIR") != nullptr;
        const bool pair_pass = use_f9 && !no_pair && 2 * q <= (size_t)256 * SC_THREADS && !(P->schedule == ATLAS_EQ_LOW && j >= P->a) && !P->left->is_i32;
        const int grid = pair_pass ? (int)((2 * q + SC_THREADS - 1) / SC_THREADS) : use_f9 ? grid_f9(q) : grid_for(q);
        bool fuse_eq = false;
        if (P->schedule == ATLAS_EQ_HIGH && j < P->a) {
            tm.begin(0, (eq_len + eq_len / 2) * sizeof(Fr));
            waiters[j] = (size_t)(grid_for(eq_len / 2) > grid ? grid_for(eq_len / 2) : grid);
            k_bind_hi_io<ChanIo><<<grid_for(eq_len / 2), SC_THREADS, 0, rt().stream>>>(eqp, eq_len / 2, ChanIo{C.io(nullptr, 0, slot0 + j, rtag(j), waiters[j]), mode}, hi_only);
            tm.end();
            eq_len /= 2;
        } else if (P->schedule == ATLAS_EQ_LOW && j >= P->a) {
            fuse_eq = true;
        }
        EqView eq = eq_view_for_round(P, j + 1, eqp, fuse_eq ? eq_len / 2 : eq_len);
        uint64_t bytes = 2 * len * (P->left->is_i32 ? sizeof(int32_t) : sizeof(Fr)) + 2 * (len / 2) * sizeof(Fr);
        if (fuse_eq) bytes += (eq_len + eq_len / 2) * sizeof(Fr);
        atlas::Chunk* reg = C.alloc((size_t)grid * ch_stride(DEG));
        if (waiters[j] < (size_t)grid) waiters[j] = (size_t)grid;
        const RoundIo io = C.io(reg, mtag(rounds_done), slot0 + j, rtag(j), waiters[j]);
        tm.begin(0, bytes);
        if (P->left->is_i32) {
            Fr *Ld = nullptr, *Rd = nullptr;
            hipError_t e = hipMalloc(&Ld, (len / 2) * sizeof(Fr));
            if (e == hipSuccess) e = hipMalloc(&Rd, (len / 2) * sizeof(Fr));
            if (e != hipSuccess) {
                if (Ld) (void)hipFree(Ld);
                for (size_t k = 0; k < n; k++) C.publish(slot0 + k, rtag(k), 0, 0, true);
                (void)hipStreamSynchronize(rt().stream);
                P->consumed = true;
                return fail(ATLAS_ENOMEM, "hipMalloc(bound operands)", e);
            }
            k_dot_bind_eval<DEG, int32_t, false, ChanIo><<<grid, SC_THREADS, 0, rt().stream>>>(
                (const int32_t*)P->left->d, (const int32_t*)P->right->d, Ld, Rd, nullptr, eq, q, ChanIo{io, mode}, K, hi_only);
            old_l = P->left->owned ? P->left->d : nullptr; old_r = P->right->owned ? P->right->d : nullptr;
            P->left->d = Ld; P->left->is_i32 = false; P->left->owned = true;
            P->right->d = Rd; P->right->is_i32 = false; P->right->owned = true;
        } else if (fuse_eq) {
            k_dot_bind_eval<DEG, Fr, true, ChanIo><<<grid, SC_THREADS, 0, rt().stream>>>(
                (const Fr*)P->left->d, (const Fr*)P->right->d, (Fr*)P->left->d, (Fr*)P->right->d, eqp, eq, q, ChanIo{io, mode}, K, hi_only);
            eq_len /= 2;
        } else if (pair_pass) {
            k_dot_bind_eval2_f9_pair<ChanIoF9><<<grid, SC_THREADS, 0, rt().stream>>>((Fr*)P->left->d, (Fr*)P->right->d, q, ChanIoF9{io});
        } else if (use_f9) {
            if (false)                                     // (the lazy-limb tail takes residues < 2.1p as they are)
                k_dot_bind_eval2_f9<true, ChanIoF9><<<grid, SC_THREADS, 0, rt().stream>>>((Fr*)P->left->d, (Fr*)P->right->d, q, ChanIoF9{io});
            else
                k_dot_bind_eval2_f9<false, ChanIoF9><<<grid, SC_THREADS, 0, rt().stream>>>((Fr*)P->left->d, (Fr*)P->right->d, q, ChanIoF9{io});
        } else {
            k_dot_bind_eval<DEG, Fr, false, ChanIo><<<grid, SC_THREADS, 0, rt().stream>>>(
                (const Fr*)P->left->d, (const Fr*)P->right->d, (Fr*)P->left->d, (Fr*)P->right->d, nullptr, eq, q, ChanIo{io, mode}, K, hi_only);
        }
        tm.end();
        mails[rounds_done] = Mail{reg, (size_t)grid, use_f9 ? 29 : 32, use_f9 ? 5 : 0};
        len /= 2;
        rounds_done += 1;
        return ATLAS_OK;
    };
    // Measured and NOT adopted (round 6; ATLAS_SC_RESIDENT=1 enables it): every bind pass from 2^ATLAS_SC_RESIDENT_LOG coefficients (default: all)
    // down to the tail's size in ONE resident launch (k_dot_resident2_f9) instead of a launch per round.  Bit-exact (tools/dbg_resident.py: 420 proofs
    // of 2^12 .. 2^22 against the per-round passes) and no faster: 0.450-0.454 ms per 2^22 step against 0.439-0.456 for the launches on the same box,
    // any starting size (profiles/r06i_resident_sweep.txt) — a dependent kernel boundary is ~1.5 us and the launches are enqueued ahead, so what a
    // round costs beyond its bytes is the challenge's way down (poll, fan-out to the replicas) and the 256 records' way up, which a resident grid
    // pays as well.  What the experiment did find: an inline-asm store needs `s_nop 1` behind it (channel.hip.h).
    static const bool no_resident = getenv("ATLAS_SC_RESIDENT") == nullptr;
    static const size_t res_log = [] { const char* e = getenv("ATLAS_SC_RESIDENT_LOG"); const int v = e ? atoi(e) : 0; return (size_t)(v >= 12 && v <= 30 ? v : 30); }();
    const bool resident_ok = use_f9 && !no_resident && P->schedule == ATLAS_EQ_NONE && (f9_cap & (f9_cap - 1)) == 0;
    auto enqueue_resident = [&]() {
        const size_t W = f9_cap;                                       // workgroups: one per CU
        size_t n_mid = 0;
        for (size_t l = len; l > ((size_t)1 << tail_log); l >>= 1) n_mid++;
        const size_t region = W * ch_stride(2);
        atlas::Chunk* reg = C.alloc(n_mid * region);
        const size_t k0 = rounds_done;                                 // first message round served; it binds challenge k0 - 1
        DotResidentArgs A;
        A.L = (Fr*)P->left->d; A.R = (Fr*)P->right->d; A.len = len;
        A.round0 = (uint32_t)k0; A.n_mid = (uint32_t)n_mid;
        A.mail = reg; A.region_chunks = (uint32_t)region;
        A.r_host0 = C.rslots + Channel::SLOT_CHUNKS * (slot0 + k0 - 1); A.r_dev0 = C.d_rslots + Channel::DEV_SLOT_CHUNKS * (slot0 + k0 - 1);
        A.dev_slot_chunks = (uint32_t)Channel::DEV_SLOT_CHUNKS; A.r_replicas = Channel::replicas_for(W);
        A.abort_flag = C.d_abort; A.tag_mail0 = mtag(k0); A.tag_r0 = rtag(k0 - 1);
        uint64_t bytes = 0;
        size_t l = len;
        for (size_t t = 0; t < n_mid; t++, l >>= 1) {
            const size_t q = l / 4, blocks = q >= W * SC_THREADS ? W : (2 * q + SC_THREADS - 1) / SC_THREADS;
            mails[k0 + t] = Mail{reg + t * region, blocks, 29, 5};
            bytes += 3 * l * sizeof(Fr);                               // two operands read, half of each written back
        }
        tm.begin(0, bytes);
        k_dot_resident2_f9<<<(unsigned)W, SC_THREADS, 0, rt().stream>>>(A);
        tm.end();
        len = l; rounds_done = k0 + n_mid;
    };
    auto enqueue_tail = [&]() {
        tail_reg = C.alloc((n + 2) * ch_stride(3));
        tail_round0 = rounds_done;
        TailChArgs A;
        A.L = P->left->d; A.R = P->right->d; A.eq = eqp;
        A.len = (uint32_t)len; A.eq_len = (uint32_t)eq_len;
        A.src_i32 = P->left->is_i32 ? 1 : 0;
        A.sched = P->schedule; A.a = (uint32_t)P->a; A.b = (uint32_t)P->b;
        A.round0 = (uint32_t)rounds_done; A.n_rounds = (uint32_t)n;
        A.pending_bind = pending; A.challenge_mode = mode; A.cap_log = (uint32_t)tail_log;
        A.mail = tail_reg; A.r_host = C.rslots + Channel::SLOT_CHUNKS * slot0; A.r_slot_chunks = (uint32_t)Channel::SLOT_CHUNKS; A.abort_flag = C.d_abort;
        A.tag_mail0 = mtag(0); A.tag_r0 = rtag(0);
        const bool tail_f9 = DEG == 2 && mode == 0;
        for (size_t k = rounds_done; k < n; k++) mails[k] = Mail{tail_reg + (k - rounds_done) * ch_stride(DEG), 1, tail_f9 ? 29 : 32, tail_f9 ? 5 : 0};
        tm.begin(1, 0);
        if (DEG == 2 && mode == 0) k_dot_tail2_f9<<<1, SC_TAIL_THREADS, 2 * (sizeof(Fr) << tail_log), rt().stream>>>(A, K);
        else k_dot_tail_ch<DEG><<<1, SC_TAIL_X_THREADS, (P->schedule == ATLAS_EQ_NONE ? 2 : 3) * (sizeof(Fr) << tail_log), rt().stream>>>(A, K);
        tm.end();
        tail_enqueued = true;
    };
    // launches for every round up to `upto` (and the tail once the passes are out)
    auto enqueue_more = [&](size_t upto) {
        while (!tail_enqueued && !enq_rc && rounds_done <= upto) {
            if (n > tail_log && len > ((size_t)1 << tail_log)) {
                if (resident_ok && len <= ((size_t)1 << res_log)) enqueue_resident();
                else enq_rc = enqueue_pass();
            } else enqueue_tail();
        }
    };
    static const bool enqueue_all = getenv("ATLAS_SC_ENQUEUE_LAZY") == nullptr;
    constexpr size_t SC_LOOKAHEAD = 2;
    enqueue_more(enqueue_all ? n : SC_LOOKAHEAD);
    if (enq_rc) return enq_rc;
    hipError_t le = hipGetLastError();

    // the transcript, on this thread
    H::Transcript T;
    std::memcpy(&T, transcript, sizeof(T));
    H::Fr claim;
    std::memcpy(&claim, input_claim, 32);
    H::tr_append_scalar(T, claim);                                   // sumcheck.rs:573-574
    bool ok = le == hipSuccess;
    static const bool trace = getenv("ATLAS_TRACE_CH") != nullptr;
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::vector<double> tr_t;
    double t_prev = trace ? now_us() : 0;
    for (size_t round = 0; round < n; round++) {
        uint64_t acc[DEG][9];
        H::Fr ev[DEG];
        double t_first = 0;
        if (ok && !tail_enqueued) {
            enqueue_more(round + SC_LOOKAHEAD);
            // ... and whatever else fits into this round's wait: one launch at a time until its first record is there
            while (!tail_enqueued && !enq_rc && reinterpret_cast<const volatile atlas::Chunk*>(mails[round].base)->tag != mtag(round)) enqueue_more(rounds_done);
            if (enq_rc) { P->consumed = true; return enq_rc; }
            if (hipGetLastError() != hipSuccess) ok = false;
        }
        if (trace && ok) { uint64_t a1[1][9]; C.collect(mails[round].base, mtag(round), 1, 1, a1); t_first = now_us(); }
        if (ok) ok = C.collect(mails[round].base, mtag(round), mails[round].blocks, DEG, acc);
        if (!ok) { C.publish(slot0 + round, rtag(round), 0, 0, true); continue; }
        const double t_coll = trace ? now_us() : 0;
        for (int k = 0; k < DEG; k++) ev[k] = atlas_rt::sum_to_fr(acc[k], mails[round].radix, mails[round].shl);
        if (grp) {                                  // the message of the whole instance = sum over the ranks' shards, in rank order
            H::Fr all[atlas_shard_group::MAX_WORLD * DEG];
            if (!grp->allgather(ev, sizeof(H::Fr) * DEG, all)) { ok = false; C.publish(slot0 + round, rtag(round), 0, 0, true); continue; }
            for (int k = 0; k < DEG; k++) {
                ev[k] = all[k];
                for (int r = 1; r < grp->world; r++) ev[k] = H::add(ev[k], all[r * DEG + k]);
            }
        }
        host_fs_round(T, DEG, ev, claim, reinterpret_cast<H::Fr*>(compressed_polys) + round * DEG, &challenges[round], mode);
        C.publish(slot0 + round, rtag(round), challenges[round].lo, challenges[round].hi);
        // while the device works on the next pass: let the runtime retire the launches that have completed (otherwise it
        // reaps all of them inside the final synchronisation: ~170 us for the 14 launches of a 2^22 instance)
        if (!tail_enqueued || round < tail_round0) (void)hipStreamQuery(rt().stream);
        if (trace) { const double t_pub = now_us(); tr_t.push_back(t_first - t_prev); tr_t.push_back(t_coll - t_first); tr_t.push_back(t_pub - t_coll); t_prev = t_pub; }
    }
    if (trace) {
        fprintf(stderr, "[atlas trace] channel dot n=%zu: per round (us) wait-for-first-record / collect-rest / fs+publish\n", n);
        for (size_t r = 0; r * 3 + 2 < tr_t.size(); r++) fprintf(stderr, "  round %2zu blocks %4zu: %7.2f %7.2f %7.2f\n", r, mails[r].blocks, tr_t[3 * r], tr_t[3 * r + 1], tr_t[3 * r + 2]);
    }
    uint32_t fin[3][9];
    if (ok) ok = C.collect_raw(tail_reg + (n - tail_round0) * ch_stride(DEG), mtag(n), 3, fin);
    P->consumed = true;
    P->left->len = 1; P->right->len = 1;
    if (P->eq) P->eq->len = 1;
    if (old_l || old_r || !ok) (void)hipStreamSynchronize(rt().stream);
    if (old_l) (void)hipFree(old_l);
    if (old_r) (void)hipFree(old_r);
    if (!ok) {
        tm.collect();
        return le != hipSuccess ? fail(ATLAS_ENODEV, "sumcheck launch", le) : fail(ATLAS_ENODEV, "round channel: no answer from the device");
    }
    for (int k = 0; k < 3; k++) std::memcpy(&final_claims[k], fin[k], 32);
    if (grp) {
        // the `world` coefficients left of each operand: rank g holds index g.  Rounds n .. n + log2(world) - 1 on the host.
        const int W = grp->world;
        H::Fr mine[2], all[atlas_shard_group::MAX_WORLD * 2];
        std::memcpy(&mine[0], fin[0], 32); std::memcpy(&mine[1], fin[1], 32);
        if (!grp->allgather(mine, sizeof(mine), all)) return fail(ATLAS_ENODEV, "sharded sumcheck: a rank did not answer");
        std::vector<H::Fr> Lr(W), Rr(W);
        for (int r = 0; r < W; r++) { Lr[r] = all[2 * r]; Rr[r] = all[2 * r + 1]; }
        size_t round = n;
        for (int len = W; len > 1; len /= 2, round++) {
            const int half = len / 2;
            H::Fr ev[2] = {H::zero(), H::zero()};
            for (int i = 0; i < half; i++) {
                ev[0] = H::add(ev[0], H::mul(Lr[i], Rr[i]));
                const H::Fr l2 = H::sub(H::add(Lr[i + half], Lr[i + half]), Lr[i]), r2 = H::sub(H::add(Rr[i + half], Rr[i + half]), Rr[i]);
                ev[1] = H::add(ev[1], H::mul(l2, r2));
            }
            host_fs_round(T, 2, ev, claim, reinterpret_cast<H::Fr*>(compressed_polys) + round * 2, &challenges[round], mode);
            const H::Fr rf = H::challenge_to_fr(challenges[round].lo, challenges[round].hi, mode);
            for (int i = 0; i < half; i++) {
                Lr[i] = H::add(Lr[i], H::mul(rf, H::sub(Lr[i + half], Lr[i])));
                Rr[i] = H::add(Rr[i], H::mul(rf, H::sub(Rr[i + half], Rr[i])));
            }
        }
        std::memcpy(&final_claims[0], &Lr[0], 32); std::memcpy(&final_claims[1], &Rr[0], 32);
    }
    std::memcpy(transcript, &T, sizeof(T));
    // the last launch has mailed its records and is retiring: observing its completion signal here costs microseconds
    // and keeps the runtime's next implicit device synchronisation (a hipFree of the operands) off its slow wait path
    {
        const double t0 = trace ? now_us() : 0;
        static const bool spin_query = getenv("ATLAS_SYNC_QUERY") != nullptr;
        if (spin_query) { while (hipStreamQuery(rt().stream) == hipErrorNotReady) {} }
        else (void)hipStreamSynchronize(rt().stream);
        if (trace) fprintf(stderr, "[atlas trace] final stream sync %.2f us\n", now_us() - t0);
    }
    tm.collect();
    return ATLAS_OK;
}

extern "C" {

int atlas_sumcheck_prove_dot(atlas_dot_prover_t P, const atlas_fr_t* input_claim, atlas_transcript_t* transcript,
                             atlas_fr_t* compressed_polys, atlas_u128_t* challenges, atlas_fr_t final_claims[3]) {
    PROF("atlas_sumcheck_prove_dot");
    NEED_INIT();
    if (!P || !input_claim || !transcript || !compressed_polys || !challenges || !final_claims)
        return fail(ATLAS_EINVAL, "sumcheck_prove_dot: null argument");
    if (P->consumed) return fail(ATLAS_ESTATE, "sumcheck_prove_dot: prover already consumed");
    if (P->left->len != ((size_t)1 << P->n_rounds)) return fail(ATLAS_ESTATE, "sumcheck_prove_dot: rounds already run");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    if (rt().fs_mode == ATLAS_FS_HOST) {
        if (P->schedule == ATLAS_EQ_NONE) return prove_dot_channel<2>(P, input_claim, transcript, compressed_polys, challenges, final_claims);
        return prove_dot_channel<3>(P, input_claim, transcript, compressed_polys, challenges, final_claims);
    }
    if (P->schedule == ATLAS_EQ_NONE) return prove_dot_impl<2>(P, input_claim, transcript, compressed_polys, challenges, final_claims);
    return prove_dot_impl<3>(P, input_claim, transcript, compressed_polys, challenges, final_claims);
}

// ------------------------------------------------------------------ sharded instance (multi-GPU)
// One sumcheck instance split over `world` ranks, one GPU each.  Rank g holds the
// strided shard L_g[k] = L[k*world + g] (HighToLow binding pairs i and i + len/2, which have
// the same residue mod world, so the first n - log2(world) rounds need no data exchange).
// Per round every rank computes its local partial message, the ranks all-gather world*DEG
// field elements (RCCL / torch.distributed, done by the caller) and every rank runs the same
// transcript step on the gathered partials — the challenge is identical everywhere, no
// broadcast.  After the local rounds each rank holds one coefficient per operand; these are
// gathered and the last log2(world) rounds run redundantly on every rank.
// Degree-2 (EqSchedule::None) LargeScalars instances.

int atlas_dot_shard_begin(atlas_dot_prover_t P, const atlas_fr_t* input_claim, const atlas_transcript_t* transcript) {
    NEED_INIT();
    if (!P || !input_claim || !transcript) return fail(ATLAS_EINVAL, "dot_shard_begin: null argument");
    if (P->schedule != ATLAS_EQ_NONE || P->left->is_i32) return fail(ATLAS_EINVAL, "dot_shard_begin: degree-2 LargeScalars instances only");
    if (P->consumed || P->left->len != ((size_t)1 << P->n_rounds)) return fail(ATLAS_ESTATE, "dot_shard_begin: prover already used");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    ScCtx* hctx = reinterpret_cast<ScCtx*>(rt().h_pinned);
    std::memset(hctx, 0, sizeof(ScCtx));
    std::memcpy(&hctx->tr, transcript, sizeof(DevTranscript));
    std::memcpy(&hctx->claim, input_claim, sizeof(Fr));
    HIP_TRY(hipMemcpyAsync(rt().d_ctx, hctx, sizeof(ScCtx), hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    P->shard_rounds_done = 0; P->shard_pending = false; P->shard_active = true;
    return ATLAS_OK;
}

// local partial of the next message: 2 Fr (evals at 0 and 2 over this rank's shard) -> host
int atlas_dot_shard_local_message(atlas_dot_prover_t P, atlas_fr_t* out2) {
    NEED_INIT();
    if (!P || !out2) return fail(ATLAS_EINVAL, "dot_shard_local_message");
    if (!P->shard_active || P->left->len < 2) return fail(ATLAS_ESTATE, "dot_shard_local_message: no local round left");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    const ScConsts K = make_consts();
    const int hi_only = rt().challenge_mode == 0;
    const bool f9 = rt().challenge_mode == 0;
    const size_t len = P->left->len;
    int grid;
    EqView eq; eq.p = nullptr; eq.mode = EQ_NONE; eq.shift = 0; eq.mask = 0; eq.half = 0;
    if (!P->shard_pending) {          // first message: no bind
        const size_t half = len / 2;
        grid = (int)((half + SC_THREADS - 1) / SC_THREADS); if (grid > 256) grid = 256; if (grid < 1) grid = 1;
        if (f9) k_dot_eval2_f9<DevIoF9><<<grid, SC_THREADS, 0, rt().stream>>>((const Fr*)P->left->d, (const Fr*)P->right->d, half, DevIoF9{rt().d_ctx, rt().d_partials});
        else k_dot_eval<2, Fr, DevIo><<<grid, SC_THREADS, 0, rt().stream>>>((const Fr*)P->left->d, (const Fr*)P->right->d, eq, half, DevIo{rt().d_ctx, rt().d_partials}, K);
    } else {                          // bind the last challenge, evaluate the next message
        if (len < 4) return fail(ATLAS_ESTATE, "dot_shard_local_message: use atlas_dot_shard_local_final");
        const size_t q = len / 4;
        grid = (int)((q + SC_THREADS - 1) / SC_THREADS); if (grid > 256) grid = 256; if (grid < 1) grid = 1;
        // canonical residues in HBM: the exact kernels may read them later (local_final)
        if (f9) k_dot_bind_eval2_f9<true, DevIoF9><<<grid, SC_THREADS, 0, rt().stream>>>((Fr*)P->left->d, (Fr*)P->right->d, q, DevIoF9{rt().d_ctx, rt().d_partials});
        else k_dot_bind_eval<2, Fr, false, DevIo><<<grid, SC_THREADS, 0, rt().stream>>>((const Fr*)P->left->d, (const Fr*)P->right->d, (Fr*)P->left->d, (Fr*)P->right->d, nullptr, eq, q, DevIo{rt().d_ctx, rt().d_partials}, K, hi_only);
        P->left->len = len / 2; P->right->len = len / 2;
        P->shard_pending = false;
    }
    k_reduce_partials<2><<<1, SC_THREADS, 0, rt().stream>>>(rt().d_partials, grid, rt().d_finals + 3);
    HIP_TRY(hipMemcpyAsync(rt().h_pinned, rt().d_finals + 3, 2 * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    std::memcpy(out2, rt().h_pinned, 2 * sizeof(Fr));
    return ATLAS_OK;
}

// transcript step on the gathered partials (world * 2 Fr, rank-major); identical on every rank
int atlas_dot_shard_round(atlas_dot_prover_t P, const atlas_fr_t* gathered, size_t world) {
    NEED_INIT();
    if (!P || !gathered || world == 0 || world > SC_MAX_BLOCKS) return fail(ATLAS_EINVAL, "dot_shard_round");
    if (!P->shard_active || P->shard_pending) return fail(ATLAS_ESTATE, "dot_shard_round: out of order");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    const ScConsts K = make_consts();
    HIP_TRY(hipMemcpyAsync(rt().d_partials, gathered, world * 2 * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
    const size_t rd = P->shard_rounds_done;
    if (rd >= MAX_ROUNDS) return fail(ATLAS_ESTATE, "dot_shard_round: too many rounds");
    k_fs_round<2><<<1, SC_THREADS, 0, rt().stream>>>(rt().d_ctx, rt().d_partials, (int)world, rt().d_proof + rd * 2, rt().d_chal + 2 * rd, K,
                                                 rd == 0 ? 1 : 0, rt().challenge_mode);
    HIP_TRY(hipStreamSynchronize(rt().stream));
    P->shard_rounds_done = rd + 1;
    P->shard_pending = true;
    return ATLAS_OK;
}

// after the last local round: apply the pending challenge to the two remaining coefficients
// of each operand and return this rank's (L_g, R_g)
int atlas_dot_shard_local_final(atlas_dot_prover_t P, atlas_fr_t* out2) {
    NEED_INIT();
    if (!P || !out2) return fail(ATLAS_EINVAL, "dot_shard_local_final");
    if (!P->shard_active || !P->shard_pending || P->left->len != 2) return fail(ATLAS_ESTATE, "dot_shard_local_final: local rounds not finished");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    const int hi_only = rt().challenge_mode == 0;
    k_bind_hi<<<1, SC_THREADS, 0, rt().stream>>>((Fr*)P->left->d, 1, &rt().d_ctx->r, hi_only);
    k_bind_hi<<<1, SC_THREADS, 0, rt().stream>>>((Fr*)P->right->d, 1, &rt().d_ctx->r, hi_only);
    HIP_TRY(hipMemcpyAsync(rt().h_pinned, P->left->d, sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipMemcpyAsync((char*)rt().h_pinned + sizeof(Fr), P->right->d, sizeof(Fr), hipMemcpyDeviceToHost, rt().stream));
    HIP_TRY(hipStreamSynchronize(rt().stream));
    std::memcpy(out2, rt().h_pinned, 2 * sizeof(Fr));
    P->left->len = 1; P->right->len = 1; P->shard_pending = false;
    return ATLAS_OK;
}

// gathered_lr: world * 2 Fr, rank-major (L_0, R_0, L_1, R_1, ...).  Runs the last log2(world)
// rounds (same on every rank) and returns the whole proof.
int atlas_dot_shard_finish(atlas_dot_prover_t P, const atlas_fr_t* gathered_lr, size_t world, atlas_transcript_t* transcript,
                           atlas_fr_t* compressed_polys, atlas_u128_t* challenges, atlas_fr_t final_claims[3]) {
    NEED_INIT();
    if (!P || !gathered_lr || !transcript || !compressed_polys || !challenges || !final_claims || !is_pow2(world) ||
        world > ((size_t)1 << SC_TAIL_LOG))
        return fail(ATLAS_EINVAL, "dot_shard_finish");
    if (!P->shard_active || P->left->len != 1 || P->shard_pending) return fail(ATLAS_ESTATE, "dot_shard_finish: out of order");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    const ScConsts K = make_consts();
    const size_t n_total = P->shard_rounds_done + ilog2(world);
    if (n_total > MAX_ROUNDS) return fail(ATLAS_EINVAL, "dot_shard_finish: too many rounds");
    // de-interleave into two device arrays of `world` coefficients
    std::vector<Fr> hl(world), hr(world);
    for (size_t i = 0; i < world; i++) { std::memcpy(&hl[i], &gathered_lr[2 * i], sizeof(Fr)); std::memcpy(&hr[i], &gathered_lr[2 * i + 1], sizeof(Fr)); }
    Fr *dl = nullptr, *dr = nullptr;
    HIP_TRY(hipMalloc(&dl, world * sizeof(Fr))); HIP_TRY(hipMalloc(&dr, world * sizeof(Fr)));
    HIP_TRY(hipMemcpyAsync(dl, hl.data(), world * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
    HIP_TRY(hipMemcpyAsync(dr, hr.data(), world * sizeof(Fr), hipMemcpyHostToDevice, rt().stream));
    TailArgs A;
    A.L = dl; A.R = dr; A.eq = nullptr; A.len = (uint32_t)world; A.eq_len = 0; A.src_i32 = 0;
    A.sched = 0; A.a = 0; A.b = 0; A.round0 = (uint32_t)P->shard_rounds_done; A.n_rounds = (uint32_t)n_total;
    A.first = P->shard_rounds_done == 0 ? 1 : 0;      // one coefficient per rank: no local round absorbed the input claim yet
    A.pending_bind = 0; A.challenge_mode = rt().challenge_mode;
    k_dot_tail<2><<<1, SC_THREADS, 3 * (sizeof(Fr) << SC_TAIL_LOG), rt().stream>>>(A, rt().d_ctx, rt().d_proof, rt().d_chal, rt().d_finals, K);
    uint8_t* hp = reinterpret_cast<uint8_t*>(rt().h_pinned);
    const size_t proof_bytes = n_total * 2 * sizeof(Fr), chal_bytes = n_total * 2 * sizeof(uint64_t);
    hipError_t e = hipMemcpyAsync(hp, rt().d_proof, proof_bytes, hipMemcpyDeviceToHost, rt().stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hp + 8192, rt().d_chal, chal_bytes, hipMemcpyDeviceToHost, rt().stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hp + 12288, rt().d_finals, 3 * sizeof(Fr), hipMemcpyDeviceToHost, rt().stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hp + 16384, rt().d_ctx, sizeof(ScCtx), hipMemcpyDeviceToHost, rt().stream);
    if (e == hipSuccess) e = hipStreamSynchronize(rt().stream);
    hipFree(dl); hipFree(dr);
    if (e != hipSuccess) return fail(ATLAS_ENODEV, "dot_shard_finish", e);
    std::memcpy(compressed_polys, hp, proof_bytes);
    std::memcpy(challenges, hp + 8192, chal_bytes);
    std::memcpy(final_claims, hp + 12288, 3 * sizeof(Fr));
    std::memcpy(transcript, hp + 16384, sizeof(DevTranscript));
    P->consumed = true; P->shard_active = false;
    return ATLAS_OK;
}

// ---- the same over the round channel and a shared-memory board (shard_group.hpp): one call per rank proves the whole
// instance, no per-round call into the library, no collective
int atlas_shard_group_open(const char* name, int world, int rank, atlas_shard_group_t* out) {
    if (!name || !out || world < 1 || world > (int)atlas_shard_group::MAX_WORLD || (world & (world - 1)) || rank < 0 || rank >= world)
        return fail(ATLAS_EINVAL, "shard_group_open: world must be a power of two <= 64, 0 <= rank < world");
    atlas_shard_group* g_ = new atlas_shard_group();
    if (!g_->open(name, world, rank)) { g_->close(); delete g_; return fail(ATLAS_ENODEV, "shard_group_open: shared-memory board not available (shm_open / peers missing)"); }
    if (const char* v = getenv("ATLAS_BOARD_WAIT_S")) { const double x = atof(v); if (x > 0) g_board_wait_s = x; }
    g_->timeout_s_default = g_board_wait_s;
    *out = g_;
    return ATLAS_OK;
}
// the failure handshake of the board (shard_group.hpp): a rank that gives up on a sharded call marks the exchange the others are about to make
int atlas_shard_fail_exchange(atlas_shard_group_t grp, int code) {
    if (!grp) return fail(ATLAS_EINVAL, "shard_fail_exchange: null group");
    grp->fail_exchange((uint64_t)(uint32_t)code);
    return ATLAS_OK;
}
// after an exchange returned an error: *rank = the rank that had given up (-1: none — a timeout), *code = what it passed
int atlas_shard_remote_failed(atlas_shard_group_t grp, int* rank, int* code) {
    if (!grp || !rank || !code) return fail(ATLAS_EINVAL, "shard_remote_failed: null argument");
    *rank = grp->remote_failed; *code = (int)(uint32_t)grp->remote_code;
    return ATLAS_OK;
}
int atlas_shard_group_close(atlas_shard_group_t grp) {
    if (grp) { grp->close(); delete grp; }
    return ATLAS_OK;
}
int atlas_shard_allgather(atlas_shard_group_t grp, const void* mine, size_t n_bytes, void* all) {
    if (!grp || !mine || !all || n_bytes == 0 || n_bytes > atlas_shard_group::PAYLOAD) return fail(ATLAS_EINVAL, "shard_allgather: 1 .. 496 bytes per rank");
    if (!grp->allgather(mine, n_bytes, all)) return fail(ATLAS_ENODEV, grp->remote_failed >= 0 ? "shard_allgather: another rank gave up on the call (atlas_shard_remote_failed)" : "shard_allgather: a rank did not answer");
    return ATLAS_OK;
}
int atlas_sumcheck_prove_dot_sharded(atlas_dot_prover_t P, atlas_shard_group_t grp, const atlas_fr_t* input_claim, atlas_transcript_t* transcript,
                                     atlas_fr_t* compressed_polys, atlas_u128_t* challenges, atlas_fr_t final_claims[3]) {
    NEED_INIT();
    if (!P || !grp || !input_claim || !transcript || !compressed_polys || !challenges || !final_claims)
        return fail(ATLAS_EINVAL, "sumcheck_prove_dot_sharded: null argument");
    if (P->schedule != ATLAS_EQ_NONE || P->left->is_i32) return fail(ATLAS_EINVAL, "sumcheck_prove_dot_sharded: degree-2 LargeScalars instances");
    if (P->consumed || P->left->len != ((size_t)1 << P->n_rounds)) return fail(ATLAS_ESTATE, "sumcheck_prove_dot_sharded: prover already used");
    if (P->n_rounds + ilog2((size_t)grp->world) > MAX_ROUNDS) return fail(ATLAS_EINVAL, "sumcheck_prove_dot_sharded: too many rounds");
    std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
    return prove_dot_channel<2>(P, input_claim, transcript, compressed_polys, challenges, final_claims, grp);
}

size_t atlas_dot_num_rounds(atlas_dot_prover_t P) { return P ? P->n_rounds : 0; }
int atlas_dot_degree(atlas_dot_prover_t P) { return P ? (P->schedule == ATLAS_EQ_NONE ? 2 : 3) : 0; }

// sum of n field elements on the host (combining per-rank input claims)
int atlas_fr_sum(const atlas_fr_t* v, size_t n, atlas_fr_t* out) {
    if ((!v && n) || !out) return fail(ATLAS_EINVAL, "fr_sum");
    H::Fr acc = H::zero();
    for (size_t i = 0; i < n; i++) acc = H::add(acc, *reinterpret_cast<const H::Fr*>(&v[i]));
    std::memcpy(out, &acc, 32);
    return ATLAS_OK;
}

}  // extern "C"
