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
    using Fn = std::function<void(size_t, size_t, size_t)>;
    static HostThreads& get() { static HostThreads t; return t; }
    size_t threads() const { return n_; }
    // f(lo, hi, part): part in [0, parts)
    void parallel_for(size_t n, const Fn& f) {
        const size_t parts = n_ < n ? n_ : (n ? n : 1);
        if (parts <= 1 || getpid() != pid_) { f(0, n, 0); return; }     // (a forked child inherits the object without its threads: serial there)
        std::unique_lock<std::mutex> job_lock(job_mu_);                  // one job at a time
        pending_.store(parts - 1, std::memory_order_relaxed);
        {
            // The job is published like a sequence lock: epoch odd while its fields are written, even once they stand.  A spinning worker reads them
            // without mu_ (every worker taking the mutex to read three words was a convoy: 16 or 32 threads were no faster than 8, r06r) and checks
            // the epoch again; only the workers that went to sleep need the mutex and the condition variable.
            std::lock_guard<std::mutex> lk(mu_);
            epoch_.fetch_add(1, std::memory_order_acq_rel);
            fn_.store(&f, std::memory_order_relaxed); total_.store(n, std::memory_order_relaxed); parts_.store(parts, std::memory_order_relaxed);
            epoch_.fetch_add(1, std::memory_order_release);
        }
        if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
        f(0, n / parts, 0);
        const auto t0 = std::chrono::steady_clock::now();
        for (int spin = 0; pending_.load(std::memory_order_acquire) != 0; spin++) {      // the ranges are equal: the others finish about now
            if ((spin & 255) != 255 || std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(2)) { ATLAS_CPU_RELAX(); continue; }
            std::unique_lock<std::mutex> lk(mu_);
            done_cv_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
        }
    }

private:
    HostThreads() {
        size_t hw = std::thread::hardware_concurrency();
        size_t want = 16;                                                // (8 -> 16: 71 -> 61 ms for the GPT-2-shaped reduction's batched sumcheck, r06v; 32 is no faster)
        if (const char* e = getenv("ATLAS_HOST_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) want = (size_t)v; }
        if (hw && want > hw / 2) want = hw / 2;                          // leave cores for the transcript thread's neighbours: spinning workers on every core stall each other
        n_ = want < 1 ? 1 : want;
        // how long a worker spins for the next job before it sleeps: the jobs of a batched sumcheck come four to a round with the round's device work
        // (~1 ms for the GPT-2-shaped reduction) between the bursts — a worker that slept through it paid a wake-up per round
        if (const char* e = getenv("ATLAS_HOST_SPIN_US")) { const long v = atol(e); if (v >= 0 && v <= 1000000) spin_us_ = v; }
        pid_ = getpid();
        for (size_t w = 1; w < n_; w++) workers_.emplace_back([this, w] { run(w); });
    }
    ~HostThreads() {
        { std::lock_guard<std::mutex> lk(mu_); stop_.store(true, std::memory_order_release); epoch_.fetch_add(2, std::memory_order_release); }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    bool ready(uint64_t seen, uint64_t& e) const { e = epoch_.load(std::memory_order_acquire); return e != seen && !(e & 1u); }
    void run(size_t w) {
        uint64_t seen = 0;
        for (;;) {
            uint64_t e = 0;
            bool got = false;
            {
                const auto t0 = std::chrono::steady_clock::now();
                for (int spin = 0; !(got = ready(seen, e)); spin++) {
                    if (stop_.load(std::memory_order_acquire)) return;
                    ATLAS_CPU_RELAX();
                    if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_)) break;
                }
            }
            if (!got) {
                std::unique_lock<std::mutex> lk(mu_);
                sleepers_.fetch_add(1, std::memory_order_acq_rel);
                cv_.wait(lk, [&] { return stop_.load(std::memory_order_acquire) || ready(seen, e); });
                sleepers_.fetch_sub(1, std::memory_order_acq_rel);
            }
            if (stop_.load(std::memory_order_acquire)) return;
            const Fn* f = fn_.load(std::memory_order_relaxed);
            const size_t n = total_.load(std::memory_order_relaxed), parts = parts_.load(std::memory_order_relaxed);
            std::atomic_thread_fence(std::memory_order_acquire);
            if (epoch_.load(std::memory_order_relaxed) != e) continue;   // the next job was being written meanwhile: this worker had no part in job e (its parts all finish before the next is posted)
            seen = e;
            if (f && w < parts) {
                (*f)(n * w / parts, n * (w + 1) / parts, w);
                if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> lk(mu_); done_cv_.notify_one(); }
            }
        }
    }
    size_t n_ = 1;
    long spin_us_ = 2000;
    pid_t pid_ = 0;
    std::vector<std::thread> workers_;
    std::mutex mu_, job_mu_;
    std::condition_variable cv_, done_cv_;
    std::atomic<const Fn*> fn_{nullptr};
    std::atomic<size_t> total_{0}, parts_{0};
    std::atomic<size_t> pending_{0}, sleepers_{0};
    std::atomic<uint64_t> epoch_{0};
    std::atomic<bool> stop_{false};
};

}  // namespace atlas_host
