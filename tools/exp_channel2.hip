// Experiment 2: where does the time go when 256 workgroups wait for one host-published challenge?
// Each workgroup records (100 MHz ticks relative to workgroup 0's entry): when it saw r, when it had mailed.
// Variants of the fan-out: A = every workgroup polls its own HBM replica written by workgroup 0 (sc1 stores / sc1 loads),
// B = same with sc0 sc1 on both, C = one shared HBM word polled with __hip_atomic_load (agent),  D = replicas polled
// with __hip_atomic_load agent scope on a 32-bit tag word + plain loads of the payload after it.
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/exp_channel2 tools/exp_channel2.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Chunk { uint32_t d[3]; uint32_t tag; };
typedef uint32_t v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sys(void* p, v4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_dev(void* p, v4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ v4 ld_sys(const void* p) { v4 v; asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ v4 ld_dev(const void* p) { v4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }

template <int VAR, int MAILVAR>
__global__ __launch_bounds__(256) void k_round(Chunk* mail, const Chunk* r_host, Chunk* r_dev, uint32_t* r_word, uint32_t tag_r, uint32_t tag_mail,
                                               uint64_t* dbg, int work) {
    __shared__ uint32_t s_r;
    const uint64_t t_in = wall_clock64();
    if (threadIdx.x == 0) {
        uint32_t r = 0;
        if (blockIdx.x == 0) {
            for (;;) { v4 a = ld_sys(r_host); if (a.w == tag_r) { r = a.x; break; } __builtin_amdgcn_s_sleep(2); }
        } else if (VAR == 0 || VAR == 1) {
            const Chunk* p = r_dev + 4 * blockIdx.x;
            for (;;) { v4 a = VAR == 0 ? ld_dev(p) : ld_sys(p); if (a.w == tag_r) { r = a.x; break; } __builtin_amdgcn_s_sleep(2); }
        } else if (VAR == 2) {
            for (;;) { uint32_t t = __hip_atomic_load(r_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (t == tag_r) { r = t; break; } __builtin_amdgcn_s_sleep(2); }
        } else {
            uint32_t* p = (uint32_t*)(r_dev + 4 * blockIdx.x);
            for (;;) { uint32_t t = __hip_atomic_load(p + 3, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); if (t == tag_r) { r = p[0]; break; } __builtin_amdgcn_s_sleep(2); }
        }
        s_r = r;
    }
    __syncthreads();
    const uint32_t r = s_r;
    if (blockIdx.x == 0) {
        if (VAR == 0 || VAR == 1) { if (threadIdx.x < gridDim.x) { v4 v = {r, 0, 0, tag_r}; if (VAR == 0) st_dev(r_dev + 4 * threadIdx.x, v); else st_sys(r_dev + 4 * threadIdx.x, v); } }
        else if (VAR == 2) { if (threadIdx.x == 0) __hip_atomic_store(r_word, tag_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else { if (threadIdx.x < gridDim.x) { uint32_t* p = (uint32_t*)(r_dev + 4 * threadIdx.x); p[0] = r; __hip_atomic_store(p + 3, tag_r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); } }
    }
    const uint64_t t_r = wall_clock64();
    uint32_t x = r + threadIdx.x;
    for (int k = 0; k < work; k++) x = x * 1664525u + 1013904223u;
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    if (MAILVAR == 0) {      // two lanes, three single-lane stores each (the product's shape)
        if (threadIdx.x < 2) { Chunk* p = mail + (blockIdx.x * 2 + threadIdx.x) * 3; v4 v = {x, 1, 2, tag_mail}; st_sys(p, v); st_sys(p + 1, v); st_sys(p + 2, v); }
    } else {                 // six lanes, one store instruction
        if (threadIdx.x < 6) { v4 v = {x, 1, 2, tag_mail}; st_sys(mail + blockIdx.x * 6 + threadIdx.x, v); }
    }
    if (threadIdx.x == 0) { dbg[blockIdx.x * 4 + 0] = t_in; dbg[blockIdx.x * 4 + 1] = t_r; dbg[blockIdx.x * 4 + 2] = wall_clock64(); }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int VAR, int MAILVAR>
void run(const char* name, int grid, Chunk* mail, Chunk* rh, Chunk* rd, uint32_t* rw, uint64_t* dbg, uint64_t* hdbg, uint32_t& epoch) {
    const int rounds = 12;
    std::vector<double> host_first(rounds), host_all(rounds);
    std::vector<std::vector<uint64_t>> dbgs;
    uint32_t base = epoch; epoch += 100;
    // chain: kernel i waits for r tag base+i (published by host after it collected mail base+i-1... for i=0 publish immediately)
    double t_pub_prev = 0;
    for (int i = 0; i < rounds; i++)
        k_round<VAR, MAILVAR><<<grid, 256>>>(mail + (i & 3) * 2048, rh + 4 * i, rd + 1024 * i, rw, base + i, base + 50 + i, dbg + (size_t)i * 4 * 256, 2000);
    for (int i = 0; i < rounds; i++) {
        // publish r_i
        volatile Chunk* s = rh + 4 * i; s->d[0] = 7; __atomic_thread_fence(__ATOMIC_RELEASE); s->tag = base + i; __atomic_thread_fence(__ATOMIC_SEQ_CST);
        double t_pub = now_us();
        volatile Chunk* m = mail + (i & 3) * 2048;
        while (m[0].tag != base + 50 + i) {}
        host_first[i] = now_us() - t_pub;
        for (int c = 0; c < grid * 6; c++) while (m[c].tag != base + 50 + i) {}
        host_all[i] = now_us() - t_pub;
        (void)t_pub_prev;
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hdbg, dbg, (size_t)rounds * 4 * 256 * 8, hipMemcpyDeviceToHost));
    // stats over rounds >= 2
    double f = 0, a = 0; std::vector<double> seen_med, seen_max, done_max;
    for (int i = 2; i < rounds; i++) {
        f += host_first[i]; a += host_all[i];
        uint64_t t0 = hdbg[(size_t)i * 1024 + 1];     // block 0 saw r
        std::vector<double> seen;
        double dmax = 0;
        for (int b = 0; b < grid; b++) { seen.push_back((double)(int64_t)(hdbg[(size_t)i * 1024 + b * 4 + 1] - t0) / 100.0); double d = (double)(int64_t)(hdbg[(size_t)i * 1024 + b * 4 + 2] - t0) / 100.0; if (d > dmax) dmax = d; }
        std::sort(seen.begin(), seen.end());
        seen_med.push_back(seen[seen.size() / 2]); seen_max.push_back(seen.back()); done_max.push_back(dmax);
    }
    auto avg = [](std::vector<double>& v) { double s = 0; for (double x : v) s += x; return s / v.size(); };
    printf("%-44s grid %3d: host publish -> first record %.2f us, -> all records %.2f us | on device, after workgroup 0 saw r: median workgroup saw r +%.2f us, last +%.2f us, last mailed +%.2f us\n",
           name, grid, f / (rounds - 2), a / (rounds - 2), avg(seen_med), avg(seen_max), avg(done_max));
}

int main() {
    CK(hipSetDevice(0));
    Chunk *mail, *rh, *rd; uint32_t* rw; uint64_t *dbg, *hdbg;
    CK(hipHostMalloc(&mail, 4 * 2048 * sizeof(Chunk), hipHostMallocDefault)); memset(mail, 0, 4 * 2048 * sizeof(Chunk));
    CK(hipHostMalloc(&rh, 4096 * sizeof(Chunk), hipHostMallocDefault)); memset(rh, 0, 4096 * sizeof(Chunk));
    CK(hipMalloc(&rd, 64 * 1024 * sizeof(Chunk))); CK(hipMemset(rd, 0, 64 * 1024 * sizeof(Chunk)));
    CK(hipMalloc(&rw, 256)); CK(hipMemset(rw, 0, 256));
    CK(hipMalloc(&dbg, 64 * 1024 * 8)); hdbg = (uint64_t*)malloc(64 * 1024 * 8);
    uint32_t epoch = 1000;
    for (int rep = 0; rep < 2; rep++)
        for (int grid : {16, 256}) {
            run<0, 0>("A: HBM replicas sc1/sc1, single-lane mail", grid, mail, rh, rd, rw, dbg, hdbg, epoch);
            run<0, 1>("A: HBM replicas sc1/sc1, coalesced mail", grid, mail, rh, rd, rw, dbg, hdbg, epoch);
            run<1, 1>("B: HBM replicas sc0sc1, coalesced mail", grid, mail, rh, rd, rw, dbg, hdbg, epoch);
            run<2, 1>("C: one HBM word, atomic load agent", grid, mail, rh, rd, rw, dbg, hdbg, epoch);
            run<3, 1>("D: HBM replicas, atomic acquire/release", grid, mail, rh, rd, rw, dbg, hdbg, epoch);
        }
    return 0;
}
