// Round channel, device side: how a launch talks to the host thread that runs the Fiat-Shamir
// transcript (joltworks/src/subprotocols/sumcheck.rs:578-590 crosses host <-> prover once per round).
//
// Why: the serial spine of a sumcheck round is five dependent BLAKE2b compressions plus a handful of
// field operations.  On one wavefront that is ~10 us (tools/exp_fs.hip) and a kernel boundary on each
// side; on a host core it is ~1 us.  The round trip device -> host -> device through fine-grained pinned
// memory is 2.8 us (tools/exp_channel.hip), so the transcript runs on the host and the kernels of ALL
// rounds are enqueued up front:
//   * device -> host "mail": per-workgroup partial sums written straight into pinned host memory as
//     16-byte chunks {3 payload words, tag}.  A chunk is one store and one PCIe write, so a reader that
//     sees the tag sees the payload: no fence and no L2 write-back on the GPU side, no copy engine, no
//     stream synchronize on the host side.
//   * host -> device "challenge slot": two tagged chunks per round in pinned host memory.  Workgroup 0 of
//     the waiting launch polls it over PCIe (one small read in flight) and fans the challenge out to one
//     HBM replica per workgroup (64-byte stride: different lines, different memory channels) with
//     agent-scope stores; every other workgroup polls its own replica with agent-scope loads.
//     Measured and rejected: all workgroups polling one host line (25 us per round); all workgroups
//     polling their own host line (the root complex serves ~20 small reads per us: 256 pollers see the
//     challenge 45 us late); one shared HBM copy (0.45 us per waiting workgroup: the uncached reads of
//     255 pollers queue on one memory channel).
// Tags are a 32-bit counter that never repeats within a process, so slots can be reused without
// clearing.  A wait gives up after the channel's device timeout (2 s by default; a RUNTIME setting — atlas_set_timeouts,
// ATLAS_DEVICE_WAIT_S — kept next to the abort word in HBM, ch_timeout_ticks) or when the abort word is set; the host
// publishes an abort record on its own error paths, so a failed call cannot leave a kernel spinning.
#pragma once
#include <hip/hip_runtime.h>

#include "f9.hip.h"

namespace atlas {

struct Chunk {
    uint32_t d[3];
    uint32_t tag;
};

struct RoundIo {              // kernel argument
    Chunk* mail;              // pinned host memory: records of this launch, 3 chunks per value
    const Chunk* r_host;      // pinned host slot (2 chunks) of the challenge this launch applies; null = none
    Chunk* r_dev;             // HBM replicas of that slot, CH_REPLICA_CHUNKS apart (workgroup b polls replica b % r_replicas)
    uint32_t r_replicas;
    uint32_t* abort_flag;     // HBM word; non-zero = some wait gave up, every later launch returns at once
    uint32_t tag_mail;        // tag of the records this launch writes
    uint32_t tag_r;           // tag of the challenge slot this launch waits for
    uint32_t tag_step;        // a RESIDENT launch that serves several rounds of its instance (the tails): the mail tag of the round after is
                              // tag_mail + tag_step, its challenge slot r_host + CH_SLOT_CHUNKS under tag_r + 1 (batched.hip hands out both in round order)
};
constexpr uint32_t CH_SLOT_CHUNKS = 4;                  // chunks per host challenge slot (Channel::SLOT_CHUNKS)

constexpr uint32_t CH_REPLICA_CHUNKS = 4;               // one 64-byte line per replica
constexpr uint32_t CH_MAX_REPLICAS = 256;

constexpr uint64_t CH_TIMEOUT_TICKS = 200000000ull;      // the DEFAULT: 2 s of s_memrealtime's 100 MHz
// The abort area (RoundIo::abort_flag, 64 bytes of HBM): word 0 = the abort flag, bytes 8..15 = how many ticks a wait may last.  An
// oversubscribed or time-sliced device (several ranks' processes on ONE GPU) needs more than the default; the host writes it
// (Channel::set_device_timeout) before the launches that read it.
__device__ __forceinline__ uint64_t ch_timeout_ticks(const uint32_t* abort_flag) {
    const uint64_t t = *reinterpret_cast<const volatile uint64_t*>(abort_flag + 2);
    return t ? t : CH_TIMEOUT_TICKS;
}

// Diagnosis (ATLAS_DEV_STAMPS=1, tools/dev_stamps.py): bytes 16..23 of the abort area hold a pointer to a stamp ring in HBM —
// {next index, capacity} then 16-byte records {event, tag, s_memrealtime} — or null; the host copies it out when asked (atlas_rt_stamps_dump).  ONE lane of a launch calls ch_stamp at the points
// of a round that matter (entry, challenge seen, partial rows complete, mail sent), so that a round's time can be split on the DEVICE's
// clock: S(challenge seen) - S(mail sent of the round before) = the host's share plus the two link crossings, the rest is the launch's.
enum : uint32_t { CH_EV_ENTRY = 1, CH_EV_CHALLENGE = 2, CH_EV_REDUCE_BEGIN = 3, CH_EV_ROWS_IN = 4, CH_EV_MAILED = 5, CH_EV_WORK_DONE = 6 };
__device__ __forceinline__ void ch_stamp(const uint32_t* abort_flag, uint32_t ev, uint32_t tag) {
    unsigned long long* ring = *reinterpret_cast<unsigned long long* const volatile*>(abort_flag + 4);
    if (!ring) return;
    const unsigned long long i = atomicAdd(ring, 1ull);
    if (i >= ring[1]) return;
    ring[2 + 2 * i] = ((unsigned long long)tag << 32) | ev;
    ring[3 + 2 * i] = wall_clock64();
}

typedef uint32_t ch_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ch_store_sys(void* p, ch_u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");      // (s_nop 1: the data registers are read a cycle after the issue; nothing is padded behind inline asm)
}
__device__ __forceinline__ void ch_store_dev(void* p, ch_u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ ch_u32x4 ch_load_dev(const void* p) {
    ch_u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ ch_u32x4 ch_load_sys(const void* p) {
    ch_u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// One polling thread.  Returns false on abort / timeout.
template <bool FROM_HOST = true>
__device__ __forceinline__ bool ch_poll_slot(const Chunk* slot, uint32_t tag_r, uint32_t* abort_flag, uint64_t& lo, uint64_t& hi) {
    const uint64_t t0 = wall_clock64(), t_max = ch_timeout_ticks(abort_flag);
    ch_u32x4 a, b;
    uint32_t spins = 0;
    for (;;) {
        if constexpr (FROM_HOST) { a = ch_load_sys(slot); b = ch_load_sys(slot + 1); }
        else { a = ch_load_dev(slot); b = ch_load_dev(slot + 1); }
        if (a.w == tag_r && b.w == tag_r) break;
        if ((++spins & 63u) == 0) {
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            if (wall_clock64() - t0 > t_max) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(4);
    }
    if (b.y != 0) {          // the host gave up on this proof
        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    lo = ((uint64_t)a.y << 32) | a.x;
    hi = ((uint64_t)b.x << 32) | a.z;
    return true;
}

// Every workgroup of a launch: wait for the challenge of the round being closed.  Uniform result.
__device__ __forceinline__ bool ch_wait_r(const RoundIo& io, uint64_t& lo, uint64_t& hi) {
    __shared__ uint64_t s_ch[3];
    const uint32_t wg = blockIdx.x + blockIdx.y * gridDim.x;
    if (threadIdx.x == 0) {
        uint64_t l = 0, h = 0;
        if (wg == 0) ch_stamp(io.abort_flag, CH_EV_ENTRY, io.tag_mail);
        const bool ok = wg == 0 || io.r_replicas == 0 ? ch_poll_slot<true>(io.r_host, io.tag_r, io.abort_flag, l, h)
                                : ch_poll_slot<false>(io.r_dev + (size_t)(wg % io.r_replicas) * CH_REPLICA_CHUNKS, io.tag_r, io.abort_flag, l, h);
        if (wg == 0) ch_stamp(io.abort_flag, CH_EV_CHALLENGE, io.tag_mail);
        s_ch[0] = l; s_ch[1] = h; s_ch[2] = ok ? 1 : 0;
    }
    __syncthreads();
    lo = s_ch[0]; hi = s_ch[1];
    const bool ok = s_ch[2] != 0;
    if (wg == 0 && ok && io.r_replicas && gridDim.x * gridDim.y > 1) {       // fan out (an abort travels through abort_flag)
        for (uint32_t t = threadIdx.x; t < io.r_replicas; t += blockDim.x) {
            Chunk* p = io.r_dev + (size_t)t * CH_REPLICA_CHUNKS;
            ch_store_dev(p, ch_u32x4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, io.tag_r});
            ch_store_dev(p + 1, ch_u32x4{(uint32_t)(hi >> 32), 0u, 0u, io.tag_r});
        }
    }
    return ok;
}

// The gate of a lane (batched.hip Pipeline::advance): ONE wavefront that waits for a round's challenge slot and exits.  It is
// enqueued on a lane stream in front of the wide launches of the next round when several lanes run on streams of their own, so
// that a wide launch only STARTS once its challenge exists: without it the workgroups of a fast lane's next round (thousands,
// all spinning in ch_wait_r) can occupy every workgroup slot of the chip while the slow lane's current round — which the
// challenge depends on — still has launches queued, and the proof stalls until the 2 s timeout (forward progress across streams
// is not guaranteed for spinning kernels).  With the gate at most one wavefront per lane spins.
static __global__ void k_ch_gate(RoundIo io) {
    if (threadIdx.x != 0) return;
    uint64_t lo, hi;
    (void)ch_poll_slot<true>(io.r_host, io.tag_r, io.abort_flag, lo, hi);
}

// Mail `n_vals` values of one workgroup (9 words each) with ONE store instruction: lane k < n_vals of the calling
// wavefront holds value k in w[]; the words are staged through LDS so that lane c < 3 n_vals stores chunk c.  One
// instruction = contiguous bytes = a few PCIe writes; six single-lane stores per workgroup arrive 30 us late when
// 256 workgroups mail at once (tools/exp_channel2.hip: the root complex takes ~50 M small writes per second).
// Every lane of the wavefront must call; `stage` = 9 n_vals words of LDS owned by this wavefront.
// A workgroup's record starts at chunk wg * ch_stride(n_vals); value k occupies chunks 3k .. 3k+2.
__host__ __device__ constexpr uint32_t ch_stride(uint32_t n_vals) { return (3 * n_vals + 3) & ~3u; }

__device__ __forceinline__ void ch_mail_wave(const RoundIo& io, uint32_t chunk0, uint32_t n_vals, const uint32_t w[9], uint32_t* stage) {
    const uint32_t lane = threadIdx.x & 63;
    if (lane < n_vals) {
#pragma unroll
        for (int i = 0; i < 9; i++) stage[9 * lane + i] = w[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (lane < 3 * n_vals) {
        const ch_u32x4 v = {stage[3 * lane], stage[3 * lane + 1], stage[3 * lane + 2], io.tag_mail};
        ch_store_sys(io.mail + chunk0 + lane, v);
    }
}
__device__ __forceinline__ void ch_mail_wave_f9(const RoundIo& io, uint32_t chunk0, uint32_t n_vals, const F9& v, uint32_t* stage) {
    ch_mail_wave(io, chunk0, n_vals, v.l, stage);
}
__device__ __forceinline__ void ch_mail_wave_fe(const RoundIo& io, uint32_t chunk0, uint32_t n_vals, const Fe& v, uint32_t* stage) {
    uint32_t w[9];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = v.v[i];
    w[8] = 0;
    ch_mail_wave(io, chunk0, n_vals, w, stage);
}

}  // namespace atlas
