// A few host threads for the per-instance arithmetic of very large batches (the opening-reduction sumcheck of a whole graph steps
// thousands of one-hot rows per round: ~15 field multiplications each, 3-9 ms per round on the one thread that runs the transcript).
// The reference does the same loop with Rayon (BatchedSumcheck::prove over Vec<Box<dyn SumcheckInstanceProver>>,
// joltworks/src/subprotocols/sumcheck.rs:91-104).  parallel_for splits [0, n) into contiguous ranges, runs range 0 on the caller and
// returns when every range is done; the workers spin briefly for the next job, then sleep.  All sums they produce are exact field
// elements: the split cannot change a result.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

#if defined(__x86_64__) || defined(__i386__)
#define ATLAS_CPU_RELAX() __builtin_ia32_pause()
#else
#define ATLAS_CPU_RELAX() std::this_thread::yield()
#endif

namespace atlas_host {

class HostThreads {
public:
    static HostThreads& get() { static HostThreads t; return t; }
    size_t threads() const { return n_; }
    // f(lo, hi, part): part in [0, parts)
    void parallel_for(size_t n, const std::function<void(size_t, size_t, size_t)>& f) {
        const size_t parts = n_ < n ? n_ : (n ? n : 1);
        if (parts <= 1 || getpid() != pid_) { f(0, n, 0); return; }     // (a forked child inherits the object without its threads: serial there)
        std::unique_lock<std::mutex> job_lock(job_mu_);                  // one job at a time
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &f; total_ = n; parts_ = parts;
            pending_.store(parts - 1, std::memory_order_relaxed);
            epoch_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        f(0, n / parts, 0);
        const auto t0 = std::chrono::steady_clock::now();
        for (int spin = 0; pending_.load(std::memory_order_acquire) != 0; spin++) {      // the ranges are equal: the others finish about now
            if ((spin & 255) != 255 || std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(2)) { ATLAS_CPU_RELAX(); continue; }
            std::unique_lock<std::mutex> lk(mu_);
            done_cv_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
        }
        { std::lock_guard<std::mutex> lk(mu_); fn_ = nullptr; }          // (a late worker reads fn_ under mu_)
    }

private:
    HostThreads() {
        size_t hw = std::thread::hardware_concurrency();
        size_t want = 8;
        if (const char* e = getenv("ATLAS_HOST_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) want = (size_t)v; }
        if (hw && want > hw / 2) want = hw / 2;                          // leave cores for the transcript thread's neighbours: spinning workers on every core stall each other
        n_ = want < 1 ? 1 : want;
        pid_ = getpid();
        for (size_t w = 1; w < n_; w++) workers_.emplace_back([this, w] { run(w); });
    }
    ~HostThreads() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; epoch_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void run(size_t w) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t, size_t, size_t)>* f = nullptr;
            size_t n = 0, parts = 0;
            // jobs come in bursts (four per round of a batched sumcheck): spin for the next one for a while before sleeping
            {
                const auto t0 = std::chrono::steady_clock::now();
                for (int spin = 0; epoch_.load(std::memory_order_acquire) == seen; spin++) {
                    ATLAS_CPU_RELAX();
                    if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
                }
            }
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return epoch_.load(std::memory_order_acquire) != seen; });
                seen = epoch_.load(std::memory_order_acquire);
                if (stop_) return;
                f = fn_; n = total_; parts = parts_;
            }
            if (f && w < parts) {
                (*f)(n * w / parts, n * (w + 1) / parts, w);
                if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> lk(mu_); done_cv_.notify_one(); }
            }
        }
    }
    size_t n_ = 1;
    pid_t pid_ = 0;
    std::vector<std::thread> workers_;
    std::mutex mu_, job_mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(size_t, size_t, size_t)>* fn_ = nullptr;
    size_t total_ = 0, parts_ = 0;
    std::atomic<size_t> pending_{0};
    std::atomic<uint64_t> epoch_{0};
    bool stop_ = false;
};

}  // namespace atlas_host
