// Experiment: host-side Fiat-Shamir over a pinned-memory round channel.
//
// Question ("round channel"): how long is the serial spine of a sumcheck round when
// the transcript runs on a host core and the GPU and the host talk through fine-grained pinned
// memory — device -> host "mailbox" records written by the kernels, host -> device challenge slots
// polled by kernels that are ALREADY launched — compared with one launch per round and a D2H copy
// + stream synchronize?
//
//   records are 16-byte chunks {12 B payload, 4 B tag}: a chunk is one store / one PCIe write, so
//   a reader that sees the tag sees the payload (no fences, no L2 write-back on the GPU side).
//
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/exp_channel tools/exp_channel.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Chunk { uint32_t d[3]; uint32_t tag; };
static_assert(sizeof(Chunk) == 16, "chunk");

__device__ __forceinline__ void store16(void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    v4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ uint4 load16(const void* p) {
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    v4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_uint4(v.x, v.y, v.z, v.w);
}

// wait until the slot carries `tag`; returns the payload; gives up after ~2 s (s_memrealtime = 100 MHz)
__device__ __forceinline__ bool slot_wait(const Chunk* slot, uint32_t tag, uint32_t out[3]) {
    const uint64_t t0 = wall_clock64();
    for (;;) {
        uint4 v = load16(slot);
        if (v.w == tag) { out[0] = v.x; out[1] = v.y; out[2] = v.z; return true; }
        if (wall_clock64() - t0 > 200000000ull) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// ---- experiment 1: one resident workgroup ping-pongs with the host
__global__ void k_pingpong(Chunk* mail, const Chunk* rslot, int rounds, uint32_t base) {
    uint32_t acc = 1;
    for (int i = 0; i < rounds; i++) {
        if (threadIdx.x == 0) store16(&mail[i & 63], acc, i, 0, base + i);
        uint32_t r[3];
        if (threadIdx.x == 0) { if (!slot_wait(&rslot[i & 63], base + i, r)) return; acc = acc * 1664525u + r[0]; }
    }
}

// ---- experiment 2: a chain of pre-launched kernels, one per round; every workgroup waits for the
// previous round's challenge, does `work` multiply-adds, optionally streams `bytes` through HBM, and
// reports a partial
__global__ __launch_bounds__(256) void k_round(Chunk* mail /*[grid]*/, const Chunk* rslot_prev, uint32_t tag_prev, uint32_t tag,
                                               int work, uint32_t* buf, size_t words) {
    __shared__ uint32_t s_r;
    if (threadIdx.x == 0) {
        uint32_t r[3] = {0, 0, 0};
        if (rslot_prev) slot_wait(rslot_prev, tag_prev, r);
        s_r = r[0];
    }
    __syncthreads();
    uint32_t x = s_r + threadIdx.x;
    for (int k = 0; k < work; k++) x = x * 1664525u + 1013904223u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) buf[i] = buf[i] * 3u + x;
    // block "sum"
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    if (threadIdx.x == 0) store16(&mail[blockIdx.x], x, blockIdx.x, 0, tag);
}

// baseline for experiment 2: same kernel chain without host interaction
__global__ __launch_bounds__(256) void k_round_plain(uint32_t* partial, int work, uint32_t* buf, size_t words) {
    uint32_t x = threadIdx.x;
    for (int k = 0; k < work; k++) x = x * 1664525u + 1013904223u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) buf[i] = buf[i] * 3u + x;
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    if (threadIdx.x == 0) partial[blockIdx.x] = x;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static inline bool host_chunk_ready(const volatile Chunk* c, uint32_t tag) { return c->tag == tag; }

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Chunk *mail, *rslot;
    CK(hipHostMalloc(&mail, 4096 * sizeof(Chunk), hipHostMallocDefault));
    CK(hipHostMalloc(&rslot, 4096 * sizeof(Chunk), hipHostMallocDefault));
    memset(mail, 0, 4096 * sizeof(Chunk)); memset(rslot, 0, 4096 * sizeof(Chunk));
    uint32_t* buf; const size_t max_words = (size_t)64 << 20;   // 256 MB
    CK(hipMalloc(&buf, max_words * 4)); CK(hipMemset(buf, 1, max_words * 4));
    uint32_t* d_partial; CK(hipMalloc(&d_partial, 4096 * 4));
    uint32_t epoch = 1000;

    // 1. ping-pong
    for (int rep = 0; rep < 3; rep++) {
        const int rounds = 2000;
        const uint32_t base = epoch; epoch += rounds + 8;
        double t0 = now_us();
        k_pingpong<<<1, 64, 0, st>>>(mail, rslot, rounds, base);
        for (int i = 0; i < rounds; i++) {
            volatile Chunk* m = &mail[i & 63];
            while (!host_chunk_ready(m, base + i)) {}
            volatile Chunk* r = &rslot[i & 63];
            r->d[0] = m->d[0] + 7; r->d[1] = 0; r->d[2] = 0;
            __atomic_thread_fence(__ATOMIC_RELEASE);
            r->tag = base + i;
        }
        CK(hipStreamSynchronize(st));
        double t1 = now_us();
        printf("pingpong: %.2f us per round trip (device->host->device, %d rounds)\n", (t1 - t0) / rounds, rounds);
    }

    // 2. chain of pre-launched kernels vs plain chain vs launch+memcpy+sync per round
    const int grids[] = {1, 16, 256};
    const int works[] = {0, 2000};
    const size_t bytes_list[] = {0, (size_t)16 << 20, (size_t)128 << 20};
    for (int grid : grids) for (int work : works) for (size_t bytes : bytes_list) {
        const size_t words = bytes / 4;
        const int rounds = 22;
        double best_chan = 1e30, best_plain = 1e30, best_sync = 1e30;
        for (int rep = 0; rep < 5; rep++) {
            // channel: all kernels enqueued up front
            const uint32_t base = epoch; epoch += rounds + 8;
            double t0 = now_us();
            for (int i = 0; i < rounds; i++)
                k_round<<<grid, 256, 0, st>>>(mail + (size_t)(i & 7) * 256, i ? &rslot[i - 1] : nullptr, base + i - 1, base + i, work, buf, words);
            for (int i = 0; i < rounds; i++) {
                volatile Chunk* m = mail + (size_t)(i & 7) * 256;
                uint32_t s = 0;
                for (int b = 0; b < grid; b++) { while (m[b].tag != base + i) {} s += m[b].d[0]; }
                volatile Chunk* r = &rslot[i];
                r->d[0] = s; r->d[1] = 0; r->d[2] = 0;
                __atomic_thread_fence(__ATOMIC_RELEASE);
                r->tag = base + i;
            }
            CK(hipStreamSynchronize(st));
            double t1 = now_us();
            if (t1 - t0 < best_chan) best_chan = t1 - t0;
            // plain chain
            t0 = now_us();
            for (int i = 0; i < rounds; i++) k_round_plain<<<grid, 256, 0, st>>>(d_partial, work, buf, words);
            CK(hipStreamSynchronize(st));
            t1 = now_us();
            if (t1 - t0 < best_plain) best_plain = t1 - t0;
            // launch + D2H + sync per round (the round-1 host-stepped pattern)
            uint32_t* hp = (uint32_t*)mail;
            t0 = now_us();
            for (int i = 0; i < rounds; i++) {
                k_round_plain<<<grid, 256, 0, st>>>(d_partial, work, buf, words);
                CK(hipMemcpyAsync(hp, d_partial, 64, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
            }
            t1 = now_us();
            if (t1 - t0 < best_sync) best_sync = t1 - t0;
        }
        printf("grid %3d work %4d bytes %4zu MB: channel %.2f us/round, plain chain %.2f us/round, launch+D2H+sync %.2f us/round\n", grid, work,
               bytes >> 20, best_chan / rounds, best_plain / rounds, best_sync / rounds);
    }
    // 3. one launch per round, result through the mailbox (host polls), next launch carries r by value:
    //    the shape of the trait-driven C-ABI (compute_message / ingest_challenge called by the Rust driver)
    for (int grid : {1, 16}) {
        const int rounds = 200;
        const uint32_t base = epoch; epoch += rounds + 8;
        double t0 = now_us();
        for (int i = 0; i < rounds; i++) {
            k_round<<<grid, 256, 0, st>>>(mail, nullptr, 0, base + i, 500, buf, 0);
            volatile Chunk* m = mail;
            for (int b = 0; b < grid; b++) while (m[b].tag != base + i) {}
        }
        CK(hipStreamSynchronize(st));
        double t1 = now_us();
        printf("grid %3d: launch + mailbox poll per round %.2f us\n", grid, (t1 - t0) / rounds);
    }
    return 0;
}
