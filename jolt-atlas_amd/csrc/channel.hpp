// Round channel, host side (see channel.hip.h for the protocol and the measurements behind it).
// The host thread that owns a sumcheck runs the Blake2b transcript (sumcheck.rs:578-590): it collects
// the per-workgroup partial sums the kernels mailed into pinned memory, finishes the round polynomial,
// hashes, and publishes the challenge in the round's slot, which launches enqueued earlier are polling.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>

#include "channel.hip.h"
#include "host_field.hpp"

namespace atlas_rt {

struct Channel {
    static constexpr size_t MAIL_CHUNKS = (size_t)1 << 19;  // 8 MB of mail, handed out round-robin (alloc): a launch's
                                                            // records are consumed before the allocator comes round again,
                                                            // because later rounds cannot run until the host has read them
    static constexpr size_t RING = 256;                     // challenge slots (a proof takes one per round: <= 89 for the 64-bit clamp lookup at T = 2^25)
    static constexpr size_t SLOT_CHUNKS = atlas::CH_SLOT_CHUNKS;                                                   // host slot: one line
    static constexpr size_t DEV_SLOT_CHUNKS = atlas::CH_MAX_REPLICAS * atlas::CH_REPLICA_CHUNKS;   // 256 HBM replicas of one line
    atlas::Chunk* mail = nullptr;       // pinned
    atlas::Chunk* rslots = nullptr;     // pinned, SLOT_CHUNKS per slot
    atlas::Chunk* d_rslots = nullptr;   // HBM, DEV_SLOT_CHUNKS per slot
    uint32_t* d_abort = nullptr;        // HBM
    uint32_t next_tag = 1;
    size_t next_chunk = 0, next_slot = 0;
    bool abort_dirty = false;
    // how long a wait may last before the proof is given up: the host's waits for mail (and for tables a launch publishes) and the device's
    // waits for a challenge.  Runtime settings (atlas_set_timeouts; ATLAS_HOST_WAIT_S / ATLAS_DEVICE_WAIT_S at init): a time-sliced device
    // — several ranks' processes sharing ONE GPU — needs more than the defaults, and atlas_prove_graph_sharded scales them with its world.
    double host_wait_s = 10.0, device_wait_s = 2.0;
    hipError_t set_device_timeout(double s, hipStream_t st = nullptr) {
        if (s > 0) device_wait_s = s;
        if (!d_abort) return hipSuccess;
        const uint64_t ticks = (uint64_t)(device_wait_s * 1e8);
        hipError_t e = hipMemcpyAsync(reinterpret_cast<char*>(d_abort) + 8, &ticks, 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        return e;
    }

    // diagnosis: the device stamp ring (channel.hip.h: ch_stamp), ATLAS_DEV_STAMPS=1.  Device records live in HBM and are copied out by dump;
    // the host's own marks (publish / collected, on ITS clock) are kept beside them.
    unsigned long long* d_stamps = nullptr;
    static constexpr size_t STAMP_CAP = (size_t)1 << 20;
    struct HostMark { uint32_t ev, tag; unsigned long long ns; };
    std::vector<HostMark> host_marks;
    hipError_t stamps_on() {
        if (d_stamps) return hipSuccess;
        hipError_t e = hipMalloc(&d_stamps, (2 + 2 * STAMP_CAP) * 8);
        if (e != hipSuccess) return e;
        const unsigned long long head[2] = {0, STAMP_CAP};
        e = hipMemcpy(d_stamps, head, 16, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(reinterpret_cast<char*>(d_abort) + 16, &d_stamps, 8, hipMemcpyHostToDevice);
        host_marks.reserve(1 << 20);
        return e;
    }
    void host_mark(uint32_t ev, uint32_t tag) {
        if (!d_stamps) return;
        host_marks.push_back(HostMark{ev, tag, (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count()});
    }
    // text dump: "D ev tag ticks" (device, 100 MHz) and "H ev tag ns" (host) lines; resets both
    hipError_t stamps_dump(const char* path) {
        if (!d_stamps) return hipSuccess;
        hipError_t e = hipDeviceSynchronize();
        unsigned long long head[2] = {0, 0};
        if (e == hipSuccess) e = hipMemcpy(head, d_stamps, 16, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        const size_t n = head[0] < STAMP_CAP ? (size_t)head[0] : STAMP_CAP;
        std::vector<unsigned long long> rec(2 * n);
        if (n) e = hipMemcpy(rec.data(), d_stamps + 2, 16 * n, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        if (FILE* f = fopen(path, "w")) {
            for (size_t i = 0; i < n; i++) fprintf(f, "D %u %u %llu\n", (unsigned)(rec[2 * i] & 0xffffffffu), (unsigned)(rec[2 * i] >> 32), rec[2 * i + 1]);
            for (auto& m : host_marks) fprintf(f, "H %u %u %llu\n", m.ev, m.tag, m.ns);
            fclose(f);
        }
        host_marks.clear();
        const unsigned long long zero = 0;
        return hipMemcpy(d_stamps, &zero, 8, hipMemcpyHostToDevice);
    }

    hipError_t init() {
        // fine-grained (coherent) whatever HIP_HOST_COHERENT says: the device polls these lines while the host writes them
        hipError_t e = hipHostMalloc(&mail, MAIL_CHUNKS * sizeof(atlas::Chunk), hipHostMallocCoherent);
        if (e == hipSuccess) e = hipHostMalloc(&rslots, RING * SLOT_CHUNKS * sizeof(atlas::Chunk), hipHostMallocCoherent);
        if (e == hipSuccess) e = hipMalloc(&d_rslots, RING * DEV_SLOT_CHUNKS * sizeof(atlas::Chunk));
        if (e == hipSuccess) e = hipMalloc(&d_abort, 64);
        if (e != hipSuccess) { release(); return e; }
        std::memset(mail, 0, MAIL_CHUNKS * sizeof(atlas::Chunk));
        std::memset(rslots, 0, RING * SLOT_CHUNKS * sizeof(atlas::Chunk));
        e = hipMemset(d_rslots, 0, RING * DEV_SLOT_CHUNKS * sizeof(atlas::Chunk));
        if (e == hipSuccess) e = hipMemset(d_abort, 0, 64);
        if (const char* v = getenv("ATLAS_HOST_WAIT_S")) { const double x = atof(v); if (x > 0) host_wait_s = x; }
        if (const char* v = getenv("ATLAS_DEVICE_WAIT_S")) { const double x = atof(v); if (x > 0) device_wait_s = x; }
        if (e == hipSuccess) e = set_device_timeout(0);
        if (e == hipSuccess && getenv("ATLAS_DEV_STAMPS")) e = stamps_on();
        return e;
    }
    void release() {
        if (mail) hipHostFree(mail);
        if (rslots) hipHostFree(rslots);
        if (d_rslots) hipFree(d_rslots);
        if (d_abort) hipFree(d_abort);
        if (d_stamps) { hipFree(d_stamps); d_stamps = nullptr; }
        mail = rslots = d_rslots = nullptr; d_abort = nullptr;
    }

    uint32_t tag() { uint32_t t = next_tag++; if (next_tag == 0) next_tag = 1; return t; }
    // mail area for one launch (n chunks, 64-byte aligned)
    atlas::Chunk* alloc(size_t n) {
        n = (n + 3) & ~(size_t)3;
        if (next_chunk + n > MAIL_CHUNKS - LONG_CHUNKS) next_chunk = 0;
        atlas::Chunk* p = mail + next_chunk;
        next_chunk += n;
        return p;
    }
    // Mail of a launch that outlives its round — a resident tail that mails a record per round, tables published at construction and read
    // dozens of rounds later: the round-robin above comes round after 128 launches (4 lanes x 32 rounds), so these take their areas from a
    // ring of their own at the top of the mail (1 MB: hundreds of tails; one or two are alive at a time).
    // Liveness: the ring does not know what is still read.  Everything taken from it lives inside ONE scope — a node of the graph prover
    // (instances built ahead, class tables, tails), the opening reduction, or one library-driven proof of a stand-alone caller — so the scope's
    // owner calls long_mark() when it starts and long_check() when it ends: more than the ring's size handed out in between (or a request the
    // ring cannot hold, which falls back to the short-lived area) means live mail may have been overwritten, and the proof is refused
    // (ATLAS_ESTATE) instead of returned with sums nobody can vouch for.
    static constexpr size_t LONG_CHUNKS = (size_t)1 << 16;
    size_t next_long = 0, long_used = 0;
    bool long_overflow = false, long_scoped = false;       // long_scoped: an outer owner (the graph prover) holds the scope
    void long_mark() { long_used = 0; long_overflow = false; }
    bool long_check() const { return !long_overflow; }
    atlas::Chunk* alloc_long(size_t n) {
        n = (n + 3) & ~(size_t)3;
        long_used += n;
        if (n > LONG_CHUNKS || long_used > LONG_CHUNKS) long_overflow = true;
        if (n > LONG_CHUNKS) return alloc(n);
        if (next_long + n > LONG_CHUNKS) next_long = 0;
        atlas::Chunk* p = mail + (MAIL_CHUNKS - LONG_CHUNKS) + next_long;
        next_long += n;
        return p;
    }
    // n consecutive challenge slots
    size_t take_slots(size_t n) { if (next_slot + n > RING) next_slot = 0; const size_t s = next_slot; next_slot += n; return s; }
    uint32_t take_tags(size_t n) { const uint32_t t = next_tag; next_tag += (uint32_t)n; if (next_tag < t) { next_tag = 1 + (uint32_t)n; return 1; } return t; }
    size_t slot() { return next_slot++ % RING; }

    // kernel argument of a launch of `waiters` workgroups that mails `tag_mail` records into `mail_region` and
    // (slot_r != npos) first waits for the challenge published in slot_r under tag_r
    static uint32_t replicas_for(size_t waiters) { return (uint32_t)(waiters < 1 ? 1 : waiters > atlas::CH_MAX_REPLICAS ? atlas::CH_MAX_REPLICAS : waiters); }
    atlas::RoundIo io(atlas::Chunk* mail_region, uint32_t tag_mail, size_t slot_r, uint32_t tag_r, size_t waiters = 1) const {
        atlas::RoundIo o;
        o.mail = mail_region;
        o.r_host = slot_r == (size_t)-1 ? nullptr : rslots + SLOT_CHUNKS * slot_r;
        o.r_dev = slot_r == (size_t)-1 ? nullptr : d_rslots + DEV_SLOT_CHUNKS * slot_r;
        static const bool host_poll = getenv("ATLAS_CH_HOST_POLL") != nullptr;     // diagnosis: every workgroup polls the host slot, no HBM replicas
        o.r_replicas = host_poll ? 0 : replicas_for(waiters);
        o.abort_flag = d_abort;
        o.tag_mail = tag_mail; o.tag_r = tag_r; o.tag_step = 1;
        return o;
    }

    // host -> device: challenge (or an abort record) for the launches waiting on `slot`.  A chunk is written with
    // one aligned 16-byte store, so its tag and payload arrive together.
    void publish(size_t slot, uint32_t tag, uint64_t lo, uint64_t hi, bool abort = false) {
        typedef uint32_t v4 __attribute__((vector_size(16)));
        const v4 c0 = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, tag};
        const v4 c1 = {(uint32_t)(hi >> 32), abort ? 1u : 0u, 0u, tag};
        atlas::Chunk* s = rslots + SLOT_CHUNKS * slot;
        *reinterpret_cast<volatile v4*>(s) = c0;
        *reinterpret_cast<volatile v4*>(s + 1) = c1;
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        if (abort) abort_dirty = true;
    }

    // device -> host: wait for n_blocks * n_vals records tagged `tag` and add word w of the k-th value of every
    // workgroup into acc[k][w].  false on timeout (the device is gone or a kernel gave up).
    bool collect(const atlas::Chunk* base, uint32_t tag, size_t n_blocks, int n_vals, uint64_t (*acc)[9]) {
        const double timeout_s = host_wait_s;
        for (int k = 0; k < n_vals; k++) for (int w = 0; w < 9; w++) acc[k][w] = 0;
        const size_t stride = atlas::ch_stride((uint32_t)n_vals);
        for (size_t b = 0; b < n_blocks; b++) {
            const volatile atlas::Chunk* c = base + b * stride;
            for (int k = 0; k < n_vals; k++) {
                for (int j = 0; j < 3; j++, c++) {
                    if (c->tag != tag && !spin(c, tag, timeout_s)) { abort_dirty = true; return false; }
                    // the tag was read first: x86 keeps load order, and the chunk arrived as one write
                    acc[k][3 * j] += c->d[0]; acc[k][3 * j + 1] += c->d[1]; acc[k][3 * j + 2] += c->d[2];
                }
            }
        }
        return true;
    }
    // single records (final claims, ...): raw 9 words each
    bool collect_raw(const atlas::Chunk* base, uint32_t tag, size_t n_rec, uint32_t (*out)[9]) {
        const double timeout_s = host_wait_s;
        const volatile atlas::Chunk* c = base;
        for (size_t r = 0; r < n_rec; r++)
            for (int j = 0; j < 3; j++, c++) {
                if (c->tag != tag && !spin(c, tag, timeout_s)) { abort_dirty = true; return false; }
                out[r][3 * j] = c->d[0]; out[r][3 * j + 1] = c->d[1]; out[r][3 * j + 2] = c->d[2];
            }
        return true;
    }

private:
    static bool spin(const volatile atlas::Chunk* c, uint32_t tag, double timeout_s) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            for (int i = 0; i < 4096; i++) {
                if (c->tag == tag) return true;
                __builtin_ia32_pause();
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
        }
    }
};

// sum of records -> Fr.  acc[w] = sum over records of word w; the value is sum_w acc[w] 2^(radix w), times 2^shl,
// reduced mod p.  radix 32 / shl 0: canonical 8 x u32 Montgomery residues (exact kernels).  radix 29 / shl 5: lazy
// 9 x 29-bit limbs of sums of f9_mul products, which carry 2^-5 relative to the Montgomery radix (f9.hip.h).
inline atlas_host::Fr sum_to_fr(const uint64_t acc[9], int radix, int shl) {
    typedef unsigned __int128 u128;
    uint64_t v[6] = {0, 0, 0, 0, 0, 0};                   // 384 bits
    for (int w = 0; w < 9; w++) {
        if (!acc[w]) continue;
        const int bit = radix * w + shl, limb = bit >> 6, off = bit & 63;
        u128 x = (u128)acc[w] << off;                     // acc[w] < 2^44: fits 128 bits
        u128 c = (u128)v[limb] + (uint64_t)x;
        v[limb] = (uint64_t)c; c >>= 64;
        c += (u128)v[limb + 1] + (uint64_t)(x >> 64);
        v[limb + 1] = (uint64_t)c; c >>= 64;
        for (int i = limb + 2; c && i < 6; i++) { c += v[i]; v[i] = (uint64_t)c; c >>= 64; }
    }
    // v < 2^300 = lo + hi 2^256.  With H::mul(a, b) = a b 2^-256 mod p (one operand < p suffices):
    // lo mod p = mul(lo, 2^256 mod p), hi 2^256 mod p = mul(hi, 2^512 mod p)
    namespace H = atlas_host;
    const H::Fr lo{{v[0], v[1], v[2], v[3]}}, hi{{v[4], v[5], 0, 0}};
    const H::Fr r2{{H::FR_R2[0], H::FR_R2[1], H::FR_R2[2], H::FR_R2[3]}};
    return H::add(H::mul(lo, H::one()), H::mul(hi, r2));
}

}  // namespace atlas_rt
