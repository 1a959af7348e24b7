// Caching device allocator behind hipMalloc / hipFree inside the library.
//
// Why: an operator node is dozens of short-lived device buffers (witness columns, one-hot rows, eq tables, sort scratch);
// hipMalloc costs 50-100 us each and hipFree waits for the device to go idle before it returns (0.4 ms when launches are in
// flight) — about 2.4 of the 21 ms of the Einsum node (tools/time_node.py with ATLAS_TRACE=1).  Buffers are returned to
// per-size free lists instead and handed out again without a runtime call.  Everything the library launches runs on
// the library stream or on side streams that are drained before the entry point returns, so a block that is reused is
// reused in stream order; nothing here relies on hipFree's implicit device synchronisation.
//   * size classes: four per octave (<= 25 % slack), 256-byte minimum; blocks above POOL_MAX_BLOCK (1 GiB) bypass the pool;
//   * at most POOL_MAX_CACHED bytes are kept; beyond that hipFree is real;
//   * atlas_shutdown releases everything (runtime at_shutdown hook).  ATLAS_NO_POOL=1 disables the cache.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace atlas_rt {

struct DevPool {
    static constexpr size_t POOL_MAX_BLOCK = (size_t)1 << 30;
    static constexpr size_t POOL_MAX_CACHED = (size_t)24 << 30;
    std::mutex mu;
    std::unordered_map<void*, uint32_t> live;             // block -> size class (blocks handed out by the pool)
    struct FreeBlock { void* p; hipStream_t freed_on; };  // the library's current stream when the block came back
    std::vector<std::vector<FreeBlock>> free_lists;       // by size class
    size_t double_free = 0;
    size_t cross_stream = 0;                              // blocks passed over because they were freed under another stream
    size_t cached = 0;
    // accounting (atlas_device_memory_stats): bytes handed out and not yet returned — pool blocks by their class size, blocks above POOL_MAX_BLOCK
    // (`big`) by their own — and the highest value seen since the last reset: what a rank of a sharded proof holds on its device
    size_t in_use = 0, peak = 0;
    std::unordered_map<void*, size_t> big;
    void account(ptrdiff_t d) { in_use = (size_t)((ptrdiff_t)in_use + d); if (in_use > peak) peak = in_use; }
    size_t n_real = 0;                                    // hipMalloc calls that reached the runtime (misses), and what they cost
    double real_ms = 0;
    // ATLAS_POOL_POISON=<byte>: no caching, and every block is filled with that byte before it is handed out (fresh
    // blocks have no pending users, so the fill cannot disturb anything): a kernel that relies on zero-initialised memory
    // shows up as a changed result.  ATLAS_POOL_POISON_MIN / _MAX (bytes) narrow the fill to a size range.
    const char* poison_env = getenv("ATLAS_POOL_POISON");
    bool off = getenv("ATLAS_NO_POOL") != nullptr || poison_env != nullptr;
    hipStream_t poison_stream = nullptr;
    hipError_t poisoned(void** out, size_t bytes) {
        hipError_t e = hipMalloc(out, bytes);
        if (e != hipSuccess || !poison_env || !bytes) return e;
        static const size_t lo = getenv("ATLAS_POOL_POISON_MIN") ? strtoull(getenv("ATLAS_POOL_POISON_MIN"), nullptr, 0) : 0;
        static const size_t hi = getenv("ATLAS_POOL_POISON_MAX") ? strtoull(getenv("ATLAS_POOL_POISON_MAX"), nullptr, 0) : ~(size_t)0;
        if (bytes < lo || bytes > hi) return e;
        std::lock_guard<std::mutex> lk(mu);
        if (!poison_stream) (void)hipStreamCreateWithFlags(&poison_stream, hipStreamNonBlocking);
        (void)hipMemsetAsync(*out, atoi(poison_env), bytes, poison_stream);
        (void)hipStreamSynchronize(poison_stream);
        return e;
    }

    static size_t class_bytes(uint32_t c) { return ((size_t)4 + (c & 3)) << (6 + (c >> 2)); }   // (4..7) * 2^(6 + c/4): 256, 320, 384, 448, 512, ...
    static uint32_t class_of(size_t bytes) {
        uint32_t c = 0;
        while (class_bytes(c) < bytes) c++;
        return c;
    }
    hipError_t alloc(void** out, size_t bytes) {
        if (off || bytes == 0 || bytes > POOL_MAX_BLOCK) {
            const hipError_t e = poisoned(out, bytes);
            if (e == hipSuccess && bytes) { std::lock_guard<std::mutex> lk(mu); big[*out] = bytes; account((ptrdiff_t)bytes); }
            return e;
        }
        const uint32_t c = class_of(bytes);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (c < free_lists.size() && !free_lists[c].empty()) {
                // Reuse is in stream order only on the stream the block was freed under: the launches of a pipelined
                // batch go to lane streams (batched.hip), so a block another stream gave back is passed over.
                auto& fl = free_lists[c];
                const hipStream_t cur = pool_tag_stream();
                for (size_t k = fl.size(); k-- > 0;) {
                    static const bool any_stream = getenv("ATLAS_POOL_ANYSTREAM") != nullptr;     // diagnosis only: the pre-4a34137 reuse rule
                    if (fl[k].freed_on != cur && fl[k].freed_on != ANY_STREAM && !any_stream) continue;
                    void* p = fl[k].p;
                    fl.erase(fl.begin() + (ptrdiff_t)k);
                    cached -= class_bytes(c);
                    live[p] = c;
                    account((ptrdiff_t)class_bytes(c));
                    *out = p;
                    return hipSuccess;
                }
                cross_stream++;
                static const bool dbg = getenv("ATLAS_POOL_DEBUG") != nullptr;
                if (dbg) fprintf(stderr, "[atlas pool] class %u (%zu B): %zu cached block(s), none freed under the current stream\n", c, class_bytes(c), fl.size());
            }
        }
        void* p = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&p, class_bytes(c));
        { std::lock_guard<std::mutex> lk(mu); n_real++; real_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
        if (e != hipSuccess) {                    // out of memory: give the cache back and try once more
            (void)hipGetLastError();
            release();
            e = hipMalloc(&p, class_bytes(c));
            if (e != hipSuccess) return e;
        }
        std::lock_guard<std::mutex> lk(mu);
        live[p] = c;
        account((ptrdiff_t)class_bytes(c));
        *out = p;
        return hipSuccess;
    }
    hipError_t free(void* p) {
        if (!p) return hipSuccess;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = live.find(p);
            if (it == live.end()) {               // not handed out by the pool: a block of the pool freed twice must not reach hipFree
                for (auto& l : free_lists)
                    for (auto& b : l)
                        if (b.p == p) {
                            double_free++;
                            fprintf(stderr, "[atlas pool] block %p freed twice (ignored)\n", p);
                            return hipSuccess;
                        }
            }
            auto bg = big.find(p);
            if (bg != big.end()) { account(-(ptrdiff_t)bg->second); big.erase(bg); }
            if (it != live.end()) {
                const uint32_t c = it->second;
                live.erase(it);
                account(-(ptrdiff_t)class_bytes(c));
                if (cached + class_bytes(c) <= POOL_MAX_CACHED) {
                    if (free_lists.size() <= c) free_lists.resize(c + 1);
                    free_lists[c].push_back(FreeBlock{p, pool_tag_stream()});
                    cached += class_bytes(c);
                    return hipSuccess;
                }
            }
        }
        return hipFree(p);
    }
    // Every cached block becomes reusable under ANY stream.  Called where the host has just waited for the streams that could still be using
    // freed blocks: Pipeline::begin (after the library stream drained, before the lanes start) and Pipeline::join (after every lane drained).
    // Without it a block a lane returned is passed over by the library stream's next request and the request goes to the runtime
    // (hipMalloc: 50-500 us on the host thread that runs the transcript).
    static inline const hipStream_t ANY_STREAM = (hipStream_t)(uintptr_t)1;
    void retag_all() {
        std::lock_guard<std::mutex> lk(mu);
        for (auto& l : free_lists) for (auto& b : l) b.freed_on = ANY_STREAM;
    }
    void release() {                              // real hipFree of everything cached
        std::vector<void*> all;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (auto& l : free_lists) { for (auto& b : l) all.push_back(b.p); l.clear(); }
            cached = 0;
        }
        for (void* p : all) (void)hipFree(p);
    }
};

inline DevPool& dev_pool() {                      // the calling thread's runtime's pool; never destroyed: the HIP runtime may be gone at static-destruction time
    Runtime& R = rt();
    if (!R.pool) {
        static std::mutex create_mu;
        std::lock_guard<std::mutex> lk(create_mu);
        if (!R.pool) R.pool = new DevPool();
    }
    return *R.pool;
}

template <class T>
inline hipError_t pool_malloc(T** out, size_t bytes) { return dev_pool().alloc((void**)out, bytes); }
inline hipError_t pool_free(void* p) { return dev_pool().free(p); }

}  // namespace atlas_rt

// every hipMalloc / hipFree below this line in a translation unit of the library goes through the pool
#define hipMalloc(p, n) atlas_rt::pool_malloc((p), (n))
#define hipFree(p) atlas_rt::pool_free((void*)(p))
