// Experiment: what does it cost the host to READ records a kernel wrote into pinned memory?
// (round channel, channel.hpp: collect()).  Variants: allocation flags, read width, CPU the reader runs on.
// hipcc --offload-arch=gfx950 -O3 -mavx2 -o tools/bin/exp_hostread tools/exp_hostread.hip
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <sched.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Chunk { uint32_t d[3]; uint32_t tag; };

__global__ void k_write(Chunk* m, int per_block, uint32_t tag) {
    if (threadIdx.x < per_block) {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        v4 v = {blockIdx.x, threadIdx.x, 7u, tag};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(m + blockIdx.x * per_block + threadIdx.x), "v"(v) : "memory");
    }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static uint64_t read_scalar(const volatile Chunk* c, int n, uint32_t tag) {
    uint64_t s = 0;
    for (int i = 0; i < n; i++) { while (c[i].tag != tag) {} s += c[i].d[0] + c[i].d[1] + c[i].d[2]; }
    return s;
}
static uint64_t read_sse(const Chunk* c, int n, uint32_t tag) {
    __m128i acc = _mm_setzero_si128();
    for (int i = 0; i < n; i++) {
        __m128i v;
        do { v = _mm_load_si128((const __m128i*)(c + i)); asm volatile("" ::: "memory"); } while ((uint32_t)_mm_extract_epi32(v, 3) != tag);
        acc = _mm_add_epi32(acc, v);
    }
    return (uint64_t)_mm_extract_epi32(acc, 0) + _mm_extract_epi32(acc, 1) + _mm_extract_epi32(acc, 2);
}
static uint64_t read_prefetch_sse(const Chunk* c, int n, uint32_t tag) {
    for (int i = 0; i < n; i += 4) _mm_prefetch((const char*)(c + i), _MM_HINT_T0);
    return read_sse(c, n, tag);
}

static void pin(int cpu) { cpu_set_t s; CPU_ZERO(&s); CPU_SET(cpu, &s); sched_setaffinity(0, sizeof(s), &s); }

int main() {
    CK(hipSetDevice(0));
    system("cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\\n' ' '; echo ' <- GPU numa nodes'; for n in /sys/devices/system/node/node*; do echo $n $(cat $n/cpulist); done; taskset -p $$ ");
    const int per_block = 6, blocks = 256, n = per_block * blocks;
    struct Var { const char* name; unsigned flags; int mode; } vars[] = {
        {"hipHostMalloc default", hipHostMallocDefault, 0}, {"hipHostMalloc NonCoherent", hipHostMallocNonCoherent, 0},
        {"hipHostMalloc Coherent", hipHostMallocCoherent, 0}, {"hipHostMalloc NumaUser", hipHostMallocNumaUser, 0},
        {"malloc + hipHostRegister", 0, 1}};
    const int cpus[] = {-1, 0, 32, 64, 96, 128, 192};
    uint32_t tag = 100;
    for (auto& v : vars) {
        for (int cpu : cpus) {
            if (cpu >= 0) pin(cpu);
            Chunk* m = nullptr;
            if (v.mode == 0) { if (hipHostMalloc(&m, 1 << 20, v.flags) != hipSuccess) { printf("%s: alloc failed\n", v.name); (void)hipGetLastError(); break; } }
            else { m = (Chunk*)aligned_alloc(4096, 1 << 20); memset(m, 0, 1 << 20); CK(hipHostRegister(m, 1 << 20, hipHostRegisterDefault)); }
            memset(m, 0, 1 << 20);
            double best[3] = {1e9, 1e9, 1e9}, poll_best = 1e9;
            for (int rep = 0; rep < 20; rep++) {
                for (int k = 0; k < 3; k++) {
                    tag++;
                    k_write<<<blocks, 64>>>(m, per_block, tag);
                    CK(hipDeviceSynchronize());
                    double t0 = now_us();
                    uint64_t s = k == 0 ? read_scalar(m, n, tag) : k == 1 ? read_sse(m, n, tag) : read_prefetch_sse(m, n, tag);
                    double t1 = now_us();
                    if (s == 12345) printf("x");
                    if (t1 - t0 < best[k]) best[k] = t1 - t0;
                }
                // polling while the kernel runs: launch-to-all-read
                tag++;
                double t0 = now_us();
                k_write<<<blocks, 64>>>(m, per_block, tag);
                uint64_t s = read_sse(m, n, tag);
                double t1 = now_us();
                if (s == 12345) printf("x");
                if (t1 - t0 < poll_best) poll_best = t1 - t0;
                CK(hipDeviceSynchronize());
            }
            printf("%-28s cpu %3d: read %d chunks after sync: scalar %.2f us, sse %.2f us, prefetch+sse %.2f us; launch->all read (polling) %.2f us\n", v.name, cpu, n,
                   best[0], best[1], best[2], poll_best);
            if (v.mode == 0) CK(hipHostFree(m)); else { CK(hipHostUnregister(m)); free(m); }
        }
    }
    return 0;
}
