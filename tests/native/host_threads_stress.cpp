// Stress of the host-thread pool's job hand-off (jolt-atlas_amd/csrc/host_threads.hpp): bursts of jobs with sleeps between them (workers spinning,
// workers asleep), jobs with fewer parts than threads (workers that take no part), every index of every job written exactly once.
// Built and run by tests/test_host_threads.py (g++ -pthread; with -fsanitize=thread when ATLAS_TEST_TSAN=1).
#include "../../jolt-atlas_amd/csrc/host_threads.hpp"

#include <cstdio>
#include <thread>

int main() {
    auto& HT = atlas_host::HostThreads::get();
    std::vector<uint64_t> v(100000);
    uint64_t bad = 0;
    const int iters = 6000;
    for (int it = 0; it < iters; it++) {
        HT.parallel_for(v.size(), [&](size_t lo, size_t hi, size_t) { for (size_t i = lo; i < hi; i++) v[i] += i + (uint64_t)it; });
        if (it % 500 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(it % 1000 == 0 ? 5 : 1));     // past the spin: the workers go to sleep
    }
    for (size_t i = 0; i < v.size(); i++) { const uint64_t want = (uint64_t)iters * i + (uint64_t)(iters - 1) * iters / 2; if (v[i] != want) bad++; }
    std::vector<int> c(3, 0);                                                // three parts on a pool of more threads
    for (int it = 0; it < 3000; it++) HT.parallel_for(3, [&](size_t lo, size_t hi, size_t) { for (size_t i = lo; i < hi; i++) c[i]++; });
    for (int x : c) if (x != 3000) bad++;
    int zero_calls = 0;                                                      // n = 0: one call with an empty range
    HT.parallel_for(0, [&](size_t lo, size_t hi, size_t) { zero_calls += (lo == 0 && hi == 0) ? 1 : 100; });
    if (zero_calls != 1) bad++;
    std::printf("threads %zu bad %llu\n", HT.threads(), (unsigned long long)bad);
    return bad != 0;
}
