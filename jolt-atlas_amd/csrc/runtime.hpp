// Shared runtime state of libatlas_hip.so (one HIP stream, scratch buffers, last error).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/atlas_hip.h"
#include "channel.hpp"
#include "prof.hpp"

namespace atlas {
struct Fe;
struct ScCtx;
}  // namespace atlas

// device polynomial handle (MultilinearPolynomial: LargeScalars or I32Scalars)
struct atlas_poly {
    void* d = nullptr;       // current coefficients: Fr if !is_i32, int32 if is_i32
    size_t len = 0;          // current length
    size_t cap_bytes = 0;
    bool is_i32 = false;
    bool owned = true;
};

namespace atlas_rt {

// one lock serialises the entry points that touch the shared stream scratch; recursive because entry points call each
// other (final_claims -> poly_download)
using Mutex = std::recursive_mutex;

struct Runtime {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;      // where the entry points launch; Pipeline::advance (batched.hip) points it at a lane stream
                                       // for the duration of one enqueue() call, under rt().mu
    hipStream_t lib_stream = nullptr;  // the library stream itself: never swapped (the device pool's tag for threads outside a lane)
    int challenge_mode = 0;
    int fs_mode = ATLAS_FS_HOST;       // where the Fiat-Shamir transcript of the whole-instance provers runs
    Channel chan;                      // round channel (pinned mailboxes + challenge slots), ATLAS_FS_HOST
    bool timing = false;
    atlas_timing_t last_timing{};
    atlas::Fe* d_partials = nullptr;   // SC_MAX_BLOCKS * 3 Fr
    atlas::ScCtx* d_ctx = nullptr;
    atlas::Fe* d_proof = nullptr;      // up to 64 rounds * 3
    uint64_t* d_chal = nullptr;        // up to 64 rounds * 2
    atlas::Fe* d_finals = nullptr;     // 3 (+ scratch for reduced evals)
    void* h_pinned = nullptr;          // pinned staging for small D2H/H2D
    bool device_shared = false;        // other ranks' processes run on THIS device for the length of a sharded call (graph_prove.hip): no cross-stream event waits (spliteq.hip)
    bool no_lane_streams = false;      // the lanes of a pipelined batch stay on the library stream (set while several ranks' processes share THIS device: graph_prove.hip)
    // atlas_rt_evaluate_with_eq on a stream of its own (spliteq.hip): the node's first wait is for ONE evaluation, not for everything NodePre put on the
    // library stream behind the eq table.  eval_event: recorded right after the eq table whose device pointer is eval_event_eq.
    hipStream_t eval_stream = nullptr;
    hipEvent_t eval_event = nullptr;
    const void* eval_event_eq = nullptr;
    atlas::Fe* d_eval_partials = nullptr;
    int pending_async = 0;             // launches of shared_message_step calls whose results the driver has not waited for yet (batched.hip)
    std::vector<void (*)()> at_shutdown;   // release hooks of the translation units that keep device arenas
    struct DevPool* pool = nullptr;        // the caching allocator of this runtime (devpool.hpp): blocks are reused in the order of ITS streams
    void* msm_ws = nullptr;                // msm.hip's workspace (one per runtime: it lives on the runtime's device)
    void* graph_tables = nullptr;          // graph_exec.hip's device copies of the trig / activation / exp tables (per runtime: they live on ITS device)
    Mutex mu;
};
// One Runtime per PROCESS by default (g_default: what every thread sees that never asked for its own), and one per THREAD for the threads
// that call atlas_init_thread(device): the shape a single Rust `prove` call over the GPUs of a node needs — N threads of one process, a
// device, stream set, round channel, allocator and MSM workspace each (the reference is one process: onnx_proof/mod.rs:153-156).  A thread's
// runtime is also what the host-thread pool's workers of THAT thread see (host_threads.hpp hands the pointer over with every job).
extern Runtime g_default;
extern thread_local Runtime* g_cur;
extern int g_device_runtimes[64];      // runtimes of THIS process that are up on device d: more than one (threads that share a GPU: tests) rules out the same
extern int g_thread_runtimes;          // how many threads own a runtime right now: while none does, rt() is one load of a global and no TLS access
                                       // (a thread-local read in a shared library is a call: ~1 M of them per GPT-2-shaped proof were +0.8 %)
inline Runtime& rt() {
    if (__builtin_expect(__atomic_load_n(&g_thread_runtimes, __ATOMIC_RELAXED) == 0, 1)) return g_default;
    Runtime* p = g_cur;
    return p ? *p : g_default;
}
// a worker thread that runs a range of the owner's job sees the owner's runtime for the length of the range (batched.hip: the host-thread pool)
struct RtScope {
    Runtime* saved;
    explicit RtScope(Runtime* owner) : saved(g_cur) { g_cur = owner; }
    ~RtScope() { g_cur = saved; }
};
// the lane stream the calling thread is enqueueing on (set by Pipeline::advance around enqueue()), nullptr otherwise.  The device
// pool tags a returned block with THIS thread's stream, so a thread that frees a buffer while another thread's pipeline has
// rt().stream pointed at a lane cannot mislabel it.
extern thread_local hipStream_t tl_lane_stream;
inline hipStream_t pool_tag_stream() { return tl_lane_stream ? tl_lane_stream : rt().lib_stream; }

constexpr size_t MAX_ROUNDS = 64;
constexpr size_t PINNED_BYTES = 1 << 16;

int fail(int code, const char* what, hipError_t e = hipSuccess);

}  // namespace atlas_rt

#include "devpool.hpp"

// device buffer that is returned (to the pool) when the scope is left, whatever the exit path
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    template <class T> T* as() const { return static_cast<T*>(p); }
    void* release() { void* q = p; p = nullptr; return q; }
    void free() { if (p) (void)hipFree(p); p = nullptr; }
};

#define HIP_TRY(x)                                                             \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) return atlas_rt::fail(ATLAS_ENODEV, #x, e_);     \
    } while (0)
#define NEED_INIT()                                                            \
    do {                                                                       \
        if (!atlas_rt::rt().ready) {                                              \
            int rc_ = atlas_init(atlas_rt::rt().device < 0 ? 0 : atlas_rt::rt().device); \
            if (rc_) return rc_;                                               \
        }                                                                      \
    } while (0)
