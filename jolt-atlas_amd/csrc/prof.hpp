// Host-side wall-clock accounting of the proving thread (diagnosis; ATLAS_PROF=1).  PROF("label") times the enclosing scope
// and adds it to the label's inclusive total; every blocking runtime call of the library (hipStreamSynchronize, hipMemcpy,
// hipDeviceSynchronize) is accounted under "sync <calling function>".  atlas_prof_dump() prints the table (graph_prove.hip
// does at the end of a proof).  Labels are string literals: the table is keyed by their address, then merged by text.
// Without ATLAS_PROF the cost is one predictable branch per scope.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace atlas_rt {

struct Prof {
    struct Entry { const char* label; double us; unsigned long n; };
    static constexpr int CAP = 512;
    Entry e[CAP];
    int n = 0;
    static bool on() { static const bool v = getenv("ATLAS_PROF") != nullptr; return v; }
    static Prof& get() { static thread_local Prof p; return p; }
    static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void add(const char* label, double us) {
        for (int i = 0; i < n; i++) if (e[i].label == label) { e[i].us += us; e[i].n++; return; }
        for (int i = 0; i < n; i++) if (!std::strcmp(e[i].label, label)) { e[i].us += us; e[i].n++; return; }
        if (n < CAP) e[n++] = Entry{label, us, 1};
    }
    void reset() { n = 0; }
    void dump(FILE* f, const char* title) {
        // sorted by time, largest first
        for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) if (e[j].us > e[i].us) { Entry t = e[i]; e[i] = e[j]; e[j] = t; }
        fprintf(f, "[atlas prof] %s: host wall clock of the proving thread per label (inclusive)\n", title);
        for (int i = 0; i < n; i++) fprintf(f, "[atlas prof] %10.3f ms  x%-7lu %8.2f us each  %s\n", e[i].us * 1e-3, e[i].n, e[i].us / (double)e[i].n, e[i].label);
    }
    struct Scope {
        const char* label; double t0;
        explicit Scope(const char* l) : label(on() ? l : nullptr), t0(label ? now_us() : 0) {}
        ~Scope() { if (label) get().add(label, now_us() - t0); }
    };
};

inline hipError_t prof_stream_sync(hipStream_t s, const char* where) {
    if (!Prof::on()) return ::hipStreamSynchronize(s);
    const double t0 = Prof::now_us();
    const hipError_t e = ::hipStreamSynchronize(s);
    char* lbl = nullptr;                                   // "sync <function>": interned by text (add() falls back to strcmp)
    static thread_local char names[256][64];
    static thread_local int n_names = 0;
    for (int i = 0; i < n_names && !lbl; i++) if (!std::strcmp(names[i] + 5, where)) lbl = names[i];
    if (!lbl && n_names < 256) { snprintf(names[n_names], 64, "sync %s", where); lbl = names[n_names++]; }
    Prof::get().add(lbl ? lbl : "sync (other)", Prof::now_us() - t0);
    return e;
}
inline hipError_t prof_memcpy(void* d, const void* s, size_t n, hipMemcpyKind k, const char* where) {
    if (!Prof::on()) return ::hipMemcpy(d, s, n, k);
    const double t0 = Prof::now_us();
    const hipError_t e = ::hipMemcpy(d, s, n, k);
    char* lbl = nullptr;
    static thread_local char names[128][64];
    static thread_local int n_names = 0;
    for (int i = 0; i < n_names && !lbl; i++) if (!std::strcmp(names[i] + 7, where)) lbl = names[i];
    if (!lbl && n_names < 128) { snprintf(names[n_names], 64, "memcpy %s", where); lbl = names[n_names++]; }
    Prof::get().add(lbl ? lbl : "memcpy (other)", Prof::now_us() - t0);
    return e;
}

}  // namespace atlas_rt

#define ATLAS_PROF_CAT2(a, b) a##b
#define ATLAS_PROF_CAT(a, b) ATLAS_PROF_CAT2(a, b)
#define PROF(label) atlas_rt::Prof::Scope ATLAS_PROF_CAT(prof_scope_, __LINE__)(label)
// the library's own blocking calls, accounted per calling function
#define hipStreamSynchronize(s) atlas_rt::prof_stream_sync((s), __func__)
#define hipMemcpy(d, s, n, k) atlas_rt::prof_memcpy((d), (s), (n), (k), __func__)
