// Diagnosis: where the HOST thread of a proof spends its time.  ATLAS_HOST_SAMPLE=<file> arms a 50 us POSIX timer on the thread that
// runs atlas_prove_graph; every tick stores a backtrace; at the end of the call the samples are written as "library+0xoffset" frames, one
// sample per line, leaf first.  tools/host_samples.py resolves the offsets of libatlas_hip.so against its symbol table and prints the
// inclusive / exclusive profile.  Nothing is armed without the variable; the product path never reads the samples.
#pragma once
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace atlas_rt {

struct HostSampler {
    static constexpr int FRAMES = 20;
    struct Sample { int n; void* pc[FRAMES]; };
    static inline std::vector<Sample>* store = nullptr;
    static inline volatile size_t count = 0;
    static inline timer_t timer;
    static inline bool armed = false;

    static void on_tick(int, siginfo_t*, void*) {
        if (!store || count >= store->size()) return;
        Sample& s = (*store)[count];
        s.n = backtrace(s.pc, FRAMES);
        count = count + 1;
    }
    static const char* path() { static const char* p = getenv("ATLAS_HOST_SAMPLE"); return p; }
    static void start() {
        if (!path() || armed) return;
        if (!store) { store = new std::vector<Sample>(400000); void* warm[4]; (void)backtrace(warm, 4); }      // the first backtrace loads the unwinder: not in a handler
        count = 0;
        struct sigaction sa; std::memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = on_tick; sa.sa_flags = SA_SIGINFO | SA_RESTART;
        sigaction(SIGRTMIN + 3, &sa, nullptr);
        struct sigevent ev; std::memset(&ev, 0, sizeof(ev));
        ev.sigev_notify = SIGEV_THREAD_ID; ev.sigev_signo = SIGRTMIN + 3;
        ev._sigev_un._tid = (pid_t)syscall(SYS_gettid);
        if (timer_create(CLOCK_MONOTONIC, &ev, &timer) != 0) return;
        struct itimerspec its; its.it_value.tv_sec = 0; its.it_value.tv_nsec = 50000; its.it_interval = its.it_value;
        timer_settime(timer, 0, &its, nullptr);
        armed = true;
    }
    static void stop_and_dump() {
        if (!armed) return;
        timer_delete(timer); armed = false;
        FILE* f = fopen(path(), "a");
        if (!f) return;
        fprintf(f, "# %zu samples, 50 us apart\n", (size_t)count);
        for (size_t i = 0; i < count; i++) {
            const Sample& s = (*store)[i];
            for (int k = 2; k < s.n; k++) {                          // frames 0-1: the handler and the signal trampoline
                Dl_info di;
                if (dladdr(s.pc[k], &di) && di.dli_fname) {
                    const char* b = strrchr(di.dli_fname, '/');
                    fprintf(f, "%s+0x%zx ", b ? b + 1 : di.dli_fname, (size_t)((char*)s.pc[k] - (char*)di.dli_fbase));
                } else fprintf(f, "?+%p ", s.pc[k]);
            }
            fputc('\n', f);
        }
        fclose(f);
    }
    struct Scope { Scope() { start(); } ~Scope() { stop_and_dump(); } };
};

}  // namespace atlas_rt
