// One sumcheck instance over the GPUs of a node: the exchange between the per-GPU processes.
//
// The per-round payload is (degree) field elements per rank — 64 bytes — and the round is latency-bound (~10 us on one
// GPU).  An RCCL all-gather of that size costs more than the round it serves, and the partial sums are
// already in host memory (the round channel mails them there, channel.hip.h).  So the ranks exchange them through a
// POSIX shared-memory board: rank g publishes its record for exchange number `seq` in slot (seq mod RING, g) — payload,
// then the tag with release order — and reads the world records with acquire loads.  Every rank adds them in rank order
// and runs the same transcript step, so the challenge needs no broadcast.  Cross-core visibility is ~0.2 us.
// A slot is reused after RING exchanges; a rank cannot be more than one exchange ahead of the slowest (it needs
// everybody's record to finish one), so RING = 4 is ample.  Tags start at a random epoch chosen by rank 0, so a stale
// segment left by a crashed run cannot match.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <random>
#include <string>
#include <vector>

struct atlas_shard_group {
    static constexpr uint64_t MAGIC = 0x61746c6173736864ull;     // "atlasshd"
    static constexpr size_t RING = 4, PAYLOAD = 496, MAX_WORLD = 64;
    // `failed` != 0: the rank gave up on the call this exchange belongs to (fail_exchange) and carries its error code; written before the tag
    struct Slot { std::atomic<uint64_t> tag; uint64_t failed; uint8_t payload[PAYLOAD]; };     // 512 bytes
    struct Header { std::atomic<uint64_t> ready; uint64_t epoch; uint32_t world; uint32_t pad; std::atomic<uint32_t> joined; uint32_t pad2; uint64_t created_s; uint8_t fill[4096 - 40]; };
    int world = 0, rank = 0;
    std::string name;
    void* base = nullptr;
    size_t bytes = 0;
    uint64_t seq = 0;
    // the failure handshake: a rank whose part of a sharded call fails does not leave the others waiting for the board's timeout (and does not
    // let the exchange numbers drift apart): it posts ONE record marked `failed` in the place of the exchange the others are about to make
    // (fail_exchange) and returns; every other rank finds the mark in that exchange, finishes reading it, and returns an error too — all at
    // the same exchange number.  remote_failed / remote_code say that an allgather returned false for that reason (not a timeout).
    int remote_failed = -1;
    uint64_t remote_code = 0;
    double timeout_s_default = 30.0;             // atlas_set_timeouts: scaled with the world size by the caller

    Header* hdr() const { return reinterpret_cast<Header*>(base); }
    Slot* slot(uint64_t s, int r) const { return reinterpret_cast<Slot*>((uint8_t*)base + sizeof(Header)) + (s % RING) * world + r; }

    // rank 0 creates and initialises the board, the others wait for it (timeout_s)
    bool open(const char* nm, int world_, int rank_, double timeout_s = 60.0) {
        world = world_; rank = rank_; name = nm;
        bytes = sizeof(Header) + RING * (size_t)world * sizeof(Slot);
        const auto t0 = std::chrono::steady_clock::now();
        auto late = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s; };
        int fd = -1;
        if (rank == 0) {
            shm_unlink(nm);
            fd = shm_open(nm, O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { if (fd >= 0) ::close(fd); return false; }
        } else {
            // A segment left by a crashed run under the same name must not be joined: a fresh board still misses this rank (joined < world)
            // and was created within this call's patience; anything else is unmapped and looked up again until rank 0 has replaced it.
            for (;;) {
                fd = shm_open(nm, O_RDWR, 0600);
                struct stat st;
                if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) {
                    void* b = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                    ::close(fd); fd = -1;
                    if (b != MAP_FAILED) {
                        Header* h = reinterpret_cast<Header*>(b);
                        const uint64_t now_s = (uint64_t)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now().time_since_epoch()).count();
                        const bool ready = h->ready.load(std::memory_order_acquire) == MAGIC;
                        const bool stale = ready && (h->joined.load(std::memory_order_acquire) >= h->world || h->created_s + (uint64_t)timeout_s + 60 < now_s);
                        if (!stale) { base = b; break; }            // not ready yet (rank 0 is initialising it) or fresh
                        munmap(b, bytes);
                    }
                }
                if (fd >= 0) ::close(fd);
                if (late()) return false;
                usleep(200);
            }
        }
        if (rank == 0) {
            base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            ::close(fd);
            if (base == MAP_FAILED) { base = nullptr; return false; }
        }
        if (rank == 0) {
            std::memset(base, 0, bytes);
            std::random_device rd;
            hdr()->epoch = (((uint64_t)rd() << 32) | rd()) >> 2 | 1;
            hdr()->world = (uint32_t)world;
            hdr()->created_s = (uint64_t)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now().time_since_epoch()).count();
            hdr()->joined.store(1, std::memory_order_relaxed);
            hdr()->ready.store(MAGIC, std::memory_order_release);
        } else {
            while (hdr()->ready.load(std::memory_order_acquire) != MAGIC) { if (late()) return false; usleep(50); }
            if (hdr()->world != (uint32_t)world) return false;
            hdr()->joined.fetch_add(1, std::memory_order_acq_rel);
        }
        while (hdr()->joined.load(std::memory_order_acquire) < (uint32_t)world) { if (late()) return false; usleep(50); }
        seq = hdr()->epoch;
        return true;
    }
    void close() {
        if (base) munmap(base, bytes);
        base = nullptr;
        if (rank == 0 && !name.empty()) shm_unlink(name.c_str());
    }
    // all[r * n .. ) <- rank r's `n` bytes (n <= PAYLOAD).  false on timeout (a rank died).
    bool allgather(const void* mine, size_t n, void* all, double timeout_s = -1.0) {
        if (!base || n > PAYLOAD) return false;
        if (timeout_s < 0) timeout_s = timeout_s_default;
        remote_failed = -1;
        seq += 1;
        Slot* s = slot(seq, rank);
        std::memcpy(s->payload, mine, n);
        s->failed = 0;
        s->tag.store(seq, std::memory_order_release);
        bool ok = true;
        for (int r = 0; r < world; r++) {
            Slot* q = slot(seq, r);
            if (q->tag.load(std::memory_order_acquire) != seq) {
                const auto t0 = std::chrono::steady_clock::now();
                for (uint64_t spins = 0;; spins++) {
                    if (q->tag.load(std::memory_order_acquire) == seq) break;
                    __builtin_ia32_pause();
                    if ((spins & 0xffff) == 0xffff && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
                }
            }
            if (q->failed) { remote_failed = r; remote_code = q->failed; ok = false; continue; }     // (the others' records are still read: the exchange completes)
            std::memcpy((uint8_t*)all + (size_t)r * n, q->payload, n);
        }
        return ok;
    }
    // this rank gives up: one record in the place of the next exchange, no waiting
    void fail_exchange(uint64_t code) {
        if (!base) return;
        seq += 1;
        Slot* s = slot(seq, rank);
        s->failed = code ? code : 1;
        s->tag.store(seq, std::memory_order_release);
    }
    // all[r * n .. ) <- rank r's `n` bytes for records of any size (the witness commitments of a sharded whole proof: 64 B per committed
    // polynomial, hundreds of KB per graph): PAYLOAD-sized exchanges one after the other, ~1 us each.
    bool allgather_bulk(const void* mine, size_t n, void* all, double timeout_s = -1.0) {
        std::vector<uint8_t> part((size_t)world * PAYLOAD);
        for (size_t off = 0; off < n; off += PAYLOAD) {
            const size_t cnt = n - off < PAYLOAD ? n - off : PAYLOAD;
            if (!allgather((const uint8_t*)mine + off, cnt, part.data(), timeout_s)) return false;
            for (int r = 0; r < world; r++) std::memcpy((uint8_t*)all + (size_t)r * n + off, part.data() + (size_t)r * cnt, cnt);
        }
        return true;
    }
};
