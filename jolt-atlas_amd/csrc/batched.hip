// BatchedSumcheck::prove — front-loaded batching of several sumcheck instances
// (joltworks/src/subprotocols/sumcheck.rs:30-184): host driver over the trait-shaped round
// API of the device instances (compute_message / ingest_challenge).  The arithmetic per round
// is O(instances * degree) field operations on the host; all O(2^n) work stays in the
// instances' kernels.  Restated rules that decide the proof bytes:
//   * input claims are appended first, then one batching coefficient per instance is drawn with
//     challenge_vector (full 128-bit scalars, :40-47);
//   * an instance with fewer rounds contributes the constant polynomial
//     input_claim * 2^(remaining - rounds - 1) until its first round (:91-104), its running claim
//     starts at input_claim * 2^(max_rounds - rounds) (:58-66);
//   * poly * coeff trims trailing zero coefficients (UniPoly::from_coeff, unipoly.rs:39-52,
//     454-461); the sum keeps the longest length (:401-413); the batched polynomial is
//     compressed and absorbed like a single instance's (:118-121).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_field.hpp"
#include "shard_group.hpp"
#include <map>
#include <unordered_map>
#include "host_threads.hpp"
#include "instance.hpp"
#include "runtime.hpp"

namespace H = atlas_host;
using atlas_rt::fail;
using atlas_rt::rt;

namespace {

// adapters: the dot / Mul provers behind the generic instance interface
struct DotAdapter : atlas_instance {
    atlas_dot_prover_t p;
    explicit DotAdapter(atlas_dot_prover_t p_) : p(p_) {}
    size_t rounds() const override { return atlas_dot_num_rounds(p); }
    size_t degree() const override { return (size_t)atlas_dot_degree(p); }
    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& c) override {
        c.resize(4); size_t n = 0;
        int rc = atlas_dot_compute_message(p, round, (const atlas_fr_t*)&claim, (atlas_fr_t*)c.data(), &n);
        c.resize(rc ? 0 : n);
        return rc;
    }
    int ingest(const atlas_u128_t& r, size_t round) override { return atlas_dot_ingest_challenge(p, &r, round); }
    int finals(std::vector<H::Fr>& out) override { out.resize(3); return atlas_dot_final_claims(p, (atlas_fr_t*)out.data()); }
};

struct MulAdapter : atlas_instance {
    atlas_mul_prover_t p;
    explicit MulAdapter(atlas_mul_prover_t p_) : p(p_) {}
    size_t rounds() const override { return atlas_mul_num_rounds(p); }
    size_t degree() const override { return 3; }
    int message(size_t round, const H::Fr& claim, std::vector<H::Fr>& c) override {
        c.resize(4); size_t n = 0;
        int rc = atlas_mul_compute_message(p, round, (const atlas_fr_t*)&claim, (atlas_fr_t*)c.data(), &n);
        c.resize(rc ? 0 : n);
        return rc;
    }
    int ingest(const atlas_u128_t& r, size_t round) override { return atlas_mul_ingest_challenge(p, &r, round); }
    int finals(std::vector<H::Fr>& out) override { out.resize(3); return atlas_mul_final_claims(p, (atlas_fr_t*)out.data()); }
};

struct Instance {
    atlas_instance* inst;
    bool owned;               // adapter created here
    H::Fr input_claim;
    size_t rounds;
    bool remote = false;      // a member another rank of the batch's shard group steps (inst == nullptr): its claim and its coefficient are part of
                              // this rank's transcript, its round polynomials reach the sum through the group's exchange
};

H::Fr mul_pow2(H::Fr x, size_t pow) {      // JoltField::mul_pow_2 (field/mod.rs:274-284): same value, x * 2^pow — one multiplication by a tabled power
    static const std::vector<H::Fr> tab = [] { std::vector<H::Fr> t(129); t[0] = H::one(); for (size_t i = 1; i <= 128; i++) t[i] = H::add(t[i - 1], t[i - 1]); return t; }();
    if (pow == 0) return x;
    if (pow <= 128) return H::mul(x, tab[pow]);
    const H::Fr two = H::from_u64(2);
    for (size_t i = 0; i < pow; i++) x = H::mul(x, two);
    return x;
}

std::vector<H::Fr> trimmed_scale(const std::vector<H::Fr>& p, const H::Fr& c) {   // &UniPoly * F -> from_coeff
    std::vector<H::Fr> o(p.size());
    for (size_t i = 0; i < p.size(); i++) o[i] = H::mul(p[i], c);
    const H::Fr z = H::zero();
    while (!o.empty() && o.back() == z) o.pop_back();
    if (o.empty()) o.push_back(z);
    return o;
}

H::Fr eval_with_challenge(const std::vector<H::Fr>& c, const H::Fr& r) {           // UniPoly::evaluate (unipoly.rs:229-245)
    H::Fr ev = c[0], pw = r;
    for (size_t i = 1; i < c.size(); i++) { ev = H::add(ev, H::mul(pw, c[i])); pw = H::mul(pw, r); }
    return ev;
}

}  // namespace


// ---- library-driven proving over the round channel -------------------------------------------------------
// Every launch of every round is enqueued before the first challenge exists (instance.hpp: enqueue); the host
// then walks the rounds: collect the mailed sums, finish the round polynomial(s), run the transcript, publish the
// challenge in the round's slot.  Used when every instance of the proof supports it and the transcript is on the
// host (atlas_set_fs_mode); otherwise the host-stepped loops below run.
namespace {

struct Lane {                     // one instance inside a pipelined proof
    atlas_instance* inst;
    size_t rounds, offset;        // first global round it takes part in
    std::vector<atlas_mail_ref> mails;
    atlas_mail_ref fin;
};

struct Pipeline {
    atlas_rt::Channel& C = rt().chan;
    std::vector<Lane> lanes;
    size_t max_rounds = 0, slot0 = 0;
    uint32_t tag0 = 0;
    uint32_t rtag(size_t R) const { return tag0 + (uint32_t)R; }
    uint32_t mtag(size_t R, size_t lane) const { return tag0 + (uint32_t)(max_rounds + 1 + R * lanes.size() + lane); }   // R = max_rounds: finals

    void abort_from(size_t R) { for (size_t k = R; k < max_rounds; k++) C.publish(slot0 + k, rtag(k), 0, 0, true); }

    // The launches run ahead of the transcript by LOOKAHEAD rounds: a launch call costs the host ~4 us, so the
    // rounds are enqueued while the device works on earlier ones instead of all before the first collect.
    static constexpr size_t LOOKAHEAD = 3;
    size_t next_enqueue = 0;
    size_t next_enqueue_launched = 0;       // bumped by every enqueue that may have launched something (atlas_instance::silent_round says which do not)
    // The instances of a batch are independent until the host combines their round polynomials, and their kernels at
    // lookup sizes (T = 2^16) are low-occupancy launches: each lane gets a stream of its own so that RaVirtual's products
    // and Booleanity's folds overlap on the device instead of queueing behind each other.  Lane streams start behind
    // the library stream (the constructors ran there) and the library stream waits for them when the proof is done.
    static constexpr size_t N_SIDE = 4;
    bool side = false;
    static hipStream_t* side_streams() { static thread_local hipStream_t st[N_SIDE] = {nullptr, nullptr, nullptr, nullptr}; return st; }
    // ATLAS_LANE_EVENTS=1 (diagnosis only, tools/stress_lanes.py): order the lanes behind the library stream with events instead of host waits
    static hipEvent_t* side_events() { static thread_local hipEvent_t ev[N_SIDE + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr}; return ev; }
    static bool lane_events() { static const bool v = getenv("ATLAS_LANE_EVENTS") != nullptr; return v; }
    static bool one_side_stream() { static const bool v = getenv("ATLAS_LANE_ONE_STREAM") != nullptr; return v; }   // diagnosis: every lane on side stream 0
    hipStream_t lane_stream(size_t li) const { return side ? side_streams()[one_side_stream() ? 0 : li % N_SIDE] : rt().stream; }
    int begin() {               // the caller holds rt().mu
        for (auto& L : lanes) { max_rounds = L.rounds > max_rounds ? L.rounds : max_rounds; }
        for (auto& L : lanes) { L.offset = max_rounds - L.rounds; L.mails.resize(L.rounds); }
        host_only = max_rounds;
        for (auto& L : lanes) { const size_t hp = L.offset + (L.inst->host_prefix() < L.rounds ? L.inst->host_prefix() : L.rounds); host_only = hp < host_only ? hp : host_only; }
        if (lanes.size() < 2) host_only = 0;
        if (max_rounds == 0 || max_rounds > atlas_rt::Channel::RING / 2) return fail(ATLAS_EINVAL, "pipelined prove: round count");
        if (C.abort_dirty) { HIP_TRY(hipMemsetAsync(C.d_abort, 0, 4, rt().stream)); C.abort_dirty = false; }
        tag0 = C.take_tags((max_rounds + 2) * (lanes.size() + 1));
        slot0 = C.take_slots(max_rounds);
        static const bool no_side = getenv("ATLAS_NO_LANE_STREAMS") != nullptr;
        // (several runtimes of ONE process on one device — threads that share a GPU, a test configuration — share the process's few hardware queues
        // among twice the streams: the lanes stay on the library stream there, as they do for three and more ranks' processes on one device;
        // the two-thread proof of tests/test_gpu_sharded.py stalled once in ~15 runs with lane streams)
        const bool crowded = rt().device >= 0 && rt().device < 64 && __atomic_load_n(&atlas_rt::g_device_runtimes[rt().device], __ATOMIC_SEQ_CST) > 1;
        side = lanes.size() > 1 && !no_side && !rt().no_lane_streams && !crowded;
        if (side) {
            hipStream_t* st = side_streams();
            if (!st[0])
                for (size_t i = 0; i < N_SIDE; i++) HIP_TRY(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
            // The lanes start behind the library stream (the constructors ran there).  The host waits for it: an
            // event recorded on the library stream and waited for by the lane streams did NOT order them reliably —
            // in processes where a lane stream shares a hardware queue with the library stream (bench.py, after its
            // other legs) the first bind of a lane read rows the constructor had not finished, one proof in three.
            if (lane_events()) {
                hipEvent_t* ev = side_events();
                if (!ev[0]) for (size_t i = 0; i <= N_SIDE; i++) HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
                HIP_TRY(hipEventRecord(ev[N_SIDE], rt().stream));
                for (size_t i = 0; i < N_SIDE && i < lanes.size(); i++) HIP_TRY(hipStreamWaitEvent(st[i], ev[N_SIDE], 0));
            } else {
                HIP_TRY(hipStreamSynchronize(rt().stream));
                atlas_rt::dev_pool().retag_all();            // nothing is in flight: a block the library stream returned may go to a lane
            }
        }
        return advance(0);
    }
    // the library stream continues only after every lane stream has drained (called on every way out of a proof)
    void join() {
        if (!side) return;
        hipStream_t* st = side_streams();
        if (lane_events()) {
            hipEvent_t* ev = side_events();
            for (size_t i = 0; i < N_SIDE && i < lanes.size(); i++) { (void)hipEventRecord(ev[i], st[i]); (void)hipStreamWaitEvent(rt().stream, ev[i], 0); }
            side = false;
            return;
        }
        // (Measured and not adopted: no host wait when the proof ended in order — once the host holds every lane's final claims nothing on a
        // lane touches the instances' buffers any more.  ATLAS_JOIN_NO_WAIT=1 takes that path; the two waits meet idle lanes and cost nothing
        // measurable over whole proofs, profiles/r06f_join_ab.txt.)
        static const bool no_wait = getenv("ATLAS_JOIN_NO_WAIT") != nullptr;
        if (!finals_in || !no_wait)
            for (size_t i = 0; i < N_SIDE && i < lanes.size(); i++) (void)hipStreamSynchronize(st[i]);   // (host waits, for the reason given in begin; the finals have been mailed, the lanes are about idle)
        atlas_rt::dev_pool().retag_all();
        side = false;
    }
    bool finals_in = false;
    void query() { (void)hipStreamQuery(rt().stream); if (side) for (size_t i = 0; i < N_SIDE && i < lanes.size(); i++) (void)hipStreamQuery(side_streams()[i]); }
    void drain() { if (side) for (size_t i = 0; i < N_SIDE && i < lanes.size(); i++) (void)hipStreamSynchronize(side_streams()[i]); (void)hipStreamSynchronize(rt().stream); join(); }
    // make sure the launches of global rounds < R + LOOKAHEAD (and the final binds after the last) are enqueued
    // host_only = the leading global rounds in which no lane waits for the device (a one-hot batch: Booleanity's four address rounds, RaVirtual
    // not started yet).  Those rounds are ~3 us of host arithmetic each, so every launch made between them is on the batch's critical path:
    // everything up to the first device round is enqueued at once in begin() (RaVirtual's first product runs under the address rounds), nothing
    // more until they are over, and the usual lookahead from there on — when a round's launches hide in the wait for the round before.
    size_t host_only = 0;
    int advance(size_t R) {
        static const bool flat = getenv("ATLAS_LOOKAHEAD_FLAT") != nullptr;          // A-B
        const size_t limit = (!flat && R < host_only) ? host_only + 1 : R + LOOKAHEAD;
        while (next_enqueue <= max_rounds && next_enqueue < limit) {
            const size_t Q = next_enqueue++;
            for (size_t li = 0; li < lanes.size(); li++) {
                Lane& L = lanes[li];
                if (Q < L.offset) continue;
                const size_t local = Q - L.offset;
                const bool bind_prev = local > 0, wait = bind_prev || Q == max_rounds;
                atlas::Chunk* area = C.alloc(256 * atlas::ch_stride(4));
                atlas::RoundIo io = C.io(area, mtag(Q, li), wait ? slot0 + Q - 1 : (size_t)-1, wait ? rtag(Q - 1) : 0, 256);
                io.tag_step = (uint32_t)lanes.size();                // mtag(Q + 1, li) - mtag(Q, li)
                const hipStream_t lib_stream = rt().stream;          // the instance's launches go to its lane's stream
                rt().stream = lane_stream(li);
                atlas_rt::tl_lane_stream = side ? rt().stream : nullptr;
                static const bool no_gate = getenv("ATLAS_LANE_NO_GATE") != nullptr;                     // diagnosis only (tools/bisect_lanes.sh)
                if (side && wait && !no_gate && L.inst->wide_wait(local)) atlas::k_ch_gate<<<1, 64, 0, rt().stream>>>(io);      // the lane's wide launches start behind their challenge (channel.hip.h)
                int rc = Q < max_rounds ? L.inst->enqueue(local, io, bind_prev, L.mails[local]) : L.inst->enqueue_finals(io, L.fin);
                if (Q >= max_rounds || !L.inst->silent_round(local)) next_enqueue_launched++;
                rt().stream = lib_stream;
                atlas_rt::tl_lane_stream = nullptr;
                if (rc) { abort_from(0); drain(); return rc; }
            }
        }
        return ATLAS_OK;
    }
    // sums of lane li at its local round -> Fr
    bool collect(const atlas_mail_ref& m, uint32_t tag, H::Fr* out) {
        uint64_t acc[16][9];
        if (m.n_vals > 16 || !C.collect(m.base, tag, m.blocks, m.n_vals, acc)) return false;
        for (int k = 0; k < m.n_vals; k++) out[k] = atlas_rt::sum_to_fr(acc[k], m.radix, m.shl);
        return true;
    }
    int collect_finals() {
        for (size_t li = 0; li < lanes.size(); li++) {
            Lane& L = lanes[li];
            uint32_t raw[16][9];
            if (L.fin.n_vals > 16 || !C.collect_raw(L.fin.base, mtag(max_rounds, li), (size_t)L.fin.n_vals, raw)) { drain(); return fail(ATLAS_ENODEV, "round channel: no answer from the device"); }
            H::Fr v[16];
            for (int k = 0; k < L.fin.n_vals; k++) std::memcpy(&v[k], raw[k], 32);
            int rc = L.inst->set_finals(v, (size_t)L.fin.n_vals);
            if (rc) { join(); return rc; }
        }
        finals_in = true;
        for (auto& L : lanes) if (L.fin.n_vals == 0 && L.rounds && !L.inst->silent_lane()) finals_in = false;     // a lane whose last launch mails nothing: wait for it
        join();
        return ATLAS_OK;
    }
};

bool all_pipelined(const std::vector<atlas_instance*>& v) {
    if (rt().fs_mode != ATLAS_FS_HOST || getenv("ATLAS_NO_PIPELINE")) return false;
    for (auto* i : v) if (!i->pipelined()) return false;
    return !v.empty();
}

}  // namespace

struct atlas_batched {
    std::vector<Instance> inst;
    atlas_shard_group* shard = nullptr;      // != null: the members are split over the ranks (atlas_rt_batched_set_shard)
    ~atlas_batched() { for (auto& I : inst) if (I.owned) delete I.inst; }
};

// BatchedSumcheck::prove with the members split over the ranks of a shard group (the opening-reduction sumcheck of a sharded whole proof,
// reduced_openings.hip): every rank holds the SAME member list — input claims, round counts: the transcript absorbs all of them and draws
// all coefficients — but steps only its own members; the others are added with atlas_rt_batched_add_remote.  The batched round polynomial
// is linear in the members, so per round every rank sums its members' scaled polynomials (and the constant polynomials of its members that
// have not started), the ranks exchange {length, coefficients} through the board (one record each) and add them in rank order: the same
// polynomial, transcript step and challenge everywhere.
int atlas_rt_batched_set_shard(atlas_batched_t b, atlas_shard_group_t sh) {
    if (!b) return fail(ATLAS_EINVAL, "batched_set_shard");
    b->shard = sh;
    return ATLAS_OK;
}
int atlas_rt_batched_add_remote(atlas_batched_t b, size_t rounds, const atlas_fr_t* input_claim) {
    if (!b || !input_claim) return fail(ATLAS_EINVAL, "batched_add_remote");
    Instance I; I.inst = nullptr; I.owned = false; std::memcpy(&I.input_claim, input_claim, 32); I.rounds = rounds; I.remote = true;
    b->inst.push_back(I);
    return ATLAS_OK;
}
// sum over the ranks of a batched round polynomial; its length is the longest any rank holds (the reference's sum keeps the longest term)
static int shard_sum_poly(atlas_shard_group* sh, std::vector<H::Fr>& batched) {
    constexpr size_t MAXC = 8;
    if (batched.size() > MAXC) return fail(ATLAS_EINVAL, "batched_prove (sharded): more than 8 coefficients");
    struct Rec { uint64_t len; uint64_t pad[3]; H::Fr c[MAXC]; } mine{};
    mine.len = batched.size();
    for (size_t k = 0; k < batched.size(); k++) mine.c[k] = batched[k];
    for (size_t k = batched.size(); k < MAXC; k++) mine.c[k] = H::zero();
    std::vector<Rec> all((size_t)sh->world);
    if (!sh->allgather(&mine, sizeof(Rec), all.data()))
        return fail(ATLAS_ENODEV, sh->remote_failed >= 0 ? "batched_prove (sharded): another rank gave up (its members failed)" : "batched_prove (sharded): a rank did not answer");
    size_t len = 0;
    for (auto& r : all) len = r.len > len ? (size_t)r.len : len;
    batched.assign(len, H::zero());
    for (auto& r : all) for (size_t k = 0; k < len; k++) batched[k] = H::add(batched[k], r.c[k]);
    return ATLAS_OK;
}

static int add_instance(atlas_batched_t b, atlas_instance* inst, bool owned, const atlas_fr_t* input_claim) {
    Instance I; I.inst = inst; I.owned = owned; std::memcpy(&I.input_claim, input_claim, 32); I.rounds = inst->rounds();
    b->inst.push_back(I);
    return ATLAS_OK;
}

extern "C" {

int atlas_batched_new(atlas_batched_t* out) {
    PROF("atlas_batched_new");
    if (!out) return fail(ATLAS_EINVAL, "batched_new");
    *out = new atlas_batched();
    return ATLAS_OK;
}

int atlas_batched_free(atlas_batched_t b) {
    PROF("atlas_batched_free");
    delete b;          // instances stay owned by the caller
    return ATLAS_OK;
}

int atlas_batched_add_dot(atlas_batched_t b, atlas_dot_prover_t p, const atlas_fr_t* input_claim) {
    if (!b || !p || !input_claim) return fail(ATLAS_EINVAL, "batched_add_dot");
    return add_instance(b, new DotAdapter(p), true, input_claim);
}

int atlas_batched_add_mul(atlas_batched_t b, atlas_mul_prover_t p, const atlas_fr_t* input_claim) {
    if (!b || !p || !input_claim) return fail(ATLAS_EINVAL, "batched_add_mul");
    return add_instance(b, new MulAdapter(p), true, input_claim);
}

int atlas_batched_add_instance(atlas_batched_t b, atlas_instance_t inst, const atlas_fr_t* input_claim) {
    PROF("atlas_batched_add_instance");
    if (!b || !inst || !input_claim) return fail(ATLAS_EINVAL, "batched_add_instance");
    return add_instance(b, inst, false, input_claim);
}

// ---- the generic SumcheckInstanceProver surface
size_t atlas_instance_num_rounds(atlas_instance_t i) { return i ? i->rounds() : 0; }
size_t atlas_instance_degree(atlas_instance_t i) { return i ? i->degree() : 0; }

int atlas_instance_compute_message(atlas_instance_t i, size_t round, const atlas_fr_t* previous_claim, atlas_fr_t* coeffs_out,
                                   size_t cap, size_t* n_coeffs) {
    NEED_INIT();
    if (!i || !previous_claim || !coeffs_out || !n_coeffs) return fail(ATLAS_EINVAL, "instance_compute_message");
    std::vector<H::Fr> c;
    int rc = i->message(round, *reinterpret_cast<const H::Fr*>(previous_claim), c);
    if (rc) return rc;
    if (c.size() > cap) return fail(ATLAS_EINVAL, "instance_compute_message: coefficient buffer too small");
    std::memcpy(coeffs_out, c.data(), c.size() * 32);
    *n_coeffs = c.size();
    return ATLAS_OK;
}

int atlas_instance_ingest_challenge(atlas_instance_t i, const atlas_u128_t* r_j, size_t round) {
    NEED_INIT();
    if (!i || !r_j) return fail(ATLAS_EINVAL, "instance_ingest_challenge");
    return i->ingest(*r_j, round);
}

int atlas_instance_final_claims(atlas_instance_t i, atlas_fr_t* out, size_t cap, size_t* n) {
    PROF("atlas_instance_final_claims");
    NEED_INIT();
    if (!i || !out || !n) return fail(ATLAS_EINVAL, "instance_final_claims");
    std::vector<H::Fr> f;
    int rc = i->finals(f);
    if (rc) return rc;
    if (f.size() > cap) return fail(ATLAS_EINVAL, "instance_final_claims: buffer too small");
    std::memcpy(out, f.data(), f.size() * 32);
    *n = f.size();
    return ATLAS_OK;
}

int atlas_instance_free(atlas_instance_t i) {
    PROF("atlas_instance_free"); delete i; return ATLAS_OK; }

// Sumcheck::prove (sumcheck.rs:565-599) over one generic instance, host-stepped
static int instance_prove_impl(atlas_instance_t inst, const atlas_fr_t* input_claim, atlas_transcript_t* transcript,
                              atlas_fr_t* compressed, size_t row_stride, uint32_t* n_coeffs, atlas_u128_t* challenges);
int atlas_instance_prove(atlas_instance_t inst, const atlas_fr_t* input_claim, atlas_transcript_t* transcript,
                         atlas_fr_t* compressed, size_t row_stride, uint32_t* n_coeffs, atlas_u128_t* challenges) {
    PROF("atlas_instance_prove");
    NEED_INIT();
    // a stand-alone caller's proof is the scope of its long-lived mail (channel.hpp); inside the graph prover the node is
    if (rt().chan.long_scoped) return instance_prove_impl(inst, input_claim, transcript, compressed, row_stride, n_coeffs, challenges);
    rt().chan.long_used = 0;                                                 // (what the instance took at construction stays valid: the count restarts, the flag is the caller's)
    int rc = instance_prove_impl(inst, input_claim, transcript, compressed, row_stride, n_coeffs, challenges);
    if (!rc && !rt().chan.long_check()) rc = fail(ATLAS_ESTATE, "instance_prove: the long-lived mail ring wrapped inside one proof");
    rt().chan.long_overflow = false;
    return rc;
}
static int instance_prove_impl(atlas_instance_t inst, const atlas_fr_t* input_claim, atlas_transcript_t* transcript,
                              atlas_fr_t* compressed, size_t row_stride, uint32_t* n_coeffs, atlas_u128_t* challenges) {
    if (!inst || !input_claim || !transcript || !compressed || !n_coeffs || !challenges) return fail(ATLAS_EINVAL, "instance_prove");
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(transcript);
    H::Fr prev = *reinterpret_cast<const H::Fr*>(input_claim);
    H::tr_append_scalar(T, prev);
    const size_t n = inst->rounds();
    std::vector<H::Fr> c;
    if (n && all_pipelined({inst})) {
        std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
        Pipeline P;
        P.lanes.push_back(Lane{inst, n, 0, {}, {}});
        int rc;
        { PROF("instance_prove: Pipeline::begin"); rc = P.begin(); }
        if (rc) return rc;
        static const bool ptrace = getenv("ATLAS_TRACE") != nullptr;   // host time per part of a round, summed (ATLAS_TRACE=1)
        double tp[5] = {0, 0, 0, 0, 0};
        auto nowp = [] { return std::chrono::steady_clock::now(); };
        auto usp = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        size_t queried_at = 0;
        for (size_t round = 0; round < n; round++) {
            H::Fr sums[16];
            const auto q0 = nowp();
            inst->prepare(round);
            { PROF("instance_prove: collect (wait for the device)");
              if (!P.collect(P.lanes[0].mails[round], P.mtag(round, 0), sums)) { P.abort_from(round); (void)hipStreamSynchronize(rt().stream); return fail(ATLAS_ENODEV, "round channel: no answer from the device"); } }
            P.C.host_mark(101, P.mtag(round, 0));
            const auto q1 = nowp();
            { PROF("instance_prove: finish"); rc = inst->finish(round, prev, sums, c); }
            const auto q2 = nowp();
            if (rc) { P.abort_from(round); (void)hipStreamSynchronize(rt().stream); return rc; }
            std::vector<H::Fr> cc;
            if (c.size() < 2) cc = c;
            else { cc.push_back(c[0]); for (size_t k = 2; k < c.size(); k++) cc.push_back(c[k]); }
            if (cc.size() > row_stride) { P.abort_from(round); (void)hipStreamSynchronize(rt().stream); return fail(ATLAS_EINVAL, "instance_prove: row_stride below the degree"); }
            const double pt0 = atlas_rt::Prof::on() ? atlas_rt::Prof::now_us() : 0;
            H::tr_append_message(T, "UniPoly_begin");
            for (auto& x : cc) H::tr_append_scalar(T, x);
            H::tr_append_message(T, "UniPoly_end");
            n_coeffs[round] = (uint32_t)cc.size();
            std::memcpy(&compressed[round * row_stride], cc.data(), cc.size() * 32);
            uint64_t lo, hi;
            H::tr_challenge_u128(T, lo, hi);
            challenges[round].lo = lo; challenges[round].hi = hi;
            P.C.publish(P.slot0 + round, P.rtag(round), lo, hi);
            P.C.host_mark(100, P.rtag(round));
            if (atlas_rt::Prof::on()) atlas_rt::Prof::get().add("instance_prove: transcript + publish", atlas_rt::Prof::now_us() - pt0);
            // lets the runtime retire completed launches while the device works — when there are any: a query costs 2.5-6 us of this thread,
            // and 56 of the 64 address rounds of a 64-bit lookup (all of them in its pure phases) launch nothing
            if ((round & 7) == 7 && P.next_enqueue_launched != queried_at) { (void)hipStreamQuery(rt().stream); queried_at = P.next_enqueue_launched; }
            prev = eval_with_challenge(c, H::challenge_to_fr(lo, hi, rt().challenge_mode));
            const auto q3 = nowp();
            { PROF("instance_prove: host_ingest"); rc = inst->host_ingest(challenges[round], round); }
            const auto q4 = nowp();
            if (!rc) { PROF("instance_prove: advance (enqueue)"); rc = P.advance(round + 1); }
            if (rc) { P.abort_from(round + 1); (void)hipStreamSynchronize(rt().stream); return rc; }
            if (ptrace) {
                const auto q5 = nowp(); tp[0] += usp(q0, q1); tp[1] += usp(q1, q2); tp[2] += usp(q2, q3); tp[3] += usp(q3, q4); tp[4] += usp(q4, q5);
                static const bool per_round = getenv("ATLAS_TRACE_ROUNDS") != nullptr;
                if (per_round) fprintf(stderr, "[atlas trace]   round %3zu: wait %6.1f  finish %6.1f  transcript %5.1f  ingest %5.1f  enqueue %5.1f us\n", round, usp(q0, q1), usp(q1, q2), usp(q2, q3), usp(q3, q4), usp(q4, q5));
            }
        }
        if (ptrace)
            fprintf(stderr, "[atlas trace] instance_prove (round channel) %zu rounds: wait for sums %.1f us, finish %.1f, transcript + publish %.1f, host_ingest %.1f, enqueue %.1f\n",
                    n, tp[0], tp[1], tp[2], tp[3], tp[4]);
        PROF("instance_prove: collect_finals + join");
        return P.collect_finals();
    }
    const bool trace = getenv("ATLAS_TRACE") != nullptr;          // wall clock of the three parts of a round, summed
    double t_msg = 0, t_fs = 0, t_ing = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    for (size_t round = 0; round < n; round++) {
        const auto t0 = now();
        int rc = inst->message(round, prev, c);
        if (rc) return rc;
        const auto t1 = now();
        std::vector<H::Fr> cc;
        if (c.size() < 2) cc = c;
        else { cc.push_back(c[0]); for (size_t k = 2; k < c.size(); k++) cc.push_back(c[k]); }
        if (cc.size() > row_stride) return fail(ATLAS_EINVAL, "instance_prove: row_stride below the degree");
        H::tr_append_message(T, "UniPoly_begin");
        for (auto& x : cc) H::tr_append_scalar(T, x);
        H::tr_append_message(T, "UniPoly_end");
        n_coeffs[round] = (uint32_t)cc.size();
        std::memcpy(&compressed[round * row_stride], cc.data(), cc.size() * 32);
        uint64_t lo, hi;
        H::tr_challenge_u128(T, lo, hi);
        challenges[round].lo = lo; challenges[round].hi = hi;
        prev = eval_with_challenge(c, H::challenge_to_fr(lo, hi, rt().challenge_mode));
        const auto t2 = now();
        rc = inst->ingest(challenges[round], round);
        if (rc) return rc;
        if (trace) { const auto t3 = now(); t_msg += ms(t0, t1); t_fs += ms(t1, t2); t_ing += ms(t2, t3); }
    }
    if (trace)
        fprintf(stderr, "[atlas trace] instance_prove %zu rounds: compute_message %.3f ms, transcript + evaluate %.3f ms, ingest_challenge %.3f ms\n",
                n, t_msg, t_fs, t_ing);
    return ATLAS_OK;
}

// BatchedSumcheck::prove.  Outputs, per round i < max_rounds:
//   n_coeffs[i]                         number of compressed coefficients of round i
//   compressed[i*row_stride ..)         n_coeffs[i] coefficients: coeffs_except_linear_term
//   challenges[i]                       raw u128 draw
// max_rounds_out = number of rounds.  The instances are left fully bound (final claims are
// read with atlas_dot_final_claims / atlas_mul_final_claims; cache_openings is the caller's).
int atlas_batched_prove(atlas_batched_t b, atlas_transcript_t* transcript, atlas_fr_t* compressed, size_t row_stride,
                        uint32_t* n_coeffs, atlas_u128_t* challenges, size_t* max_rounds_out) {
    PROF("atlas_batched_prove");
    NEED_INIT();
    if (!b || !transcript || !compressed || !n_coeffs || !challenges || !max_rounds_out || b->inst.empty())
        return fail(ATLAS_EINVAL, "batched_prove");
    H::Transcript& T = *reinterpret_cast<H::Transcript*>(transcript);
    const size_t n = b->inst.size();
    size_t max_rounds = 0;
    for (auto& I : b->inst) max_rounds = I.rounds > max_rounds ? I.rounds : max_rounds;
    for (auto& I : b->inst) H::tr_append_scalar(T, I.input_claim);                    // :42-45
    std::vector<H::Fr> coeff(n);
    for (size_t i = 0; i < n; i++) coeff[i] = H::tr_challenge_scalar(T);               // challenge_vector :47
    std::vector<H::Fr> claim(n);
    for (size_t i = 0; i < n; i++) claim[i] = mul_pow2(b->inst[i].input_claim, max_rounds - b->inst[i].rounds);

    Pipeline PL;
    size_t batch_queried_at = 0;
    bool piped = false;
    {
        std::vector<atlas_instance*> v;
        for (auto& I : b->inst) if (!I.remote) v.push_back(I.inst);
        piped = !b->shard && all_pipelined(v);       // (a split batch is stepped by the host: its rounds cross the board)
    }
    std::unique_lock<atlas_rt::Mutex> pipe_lock(rt().mu, std::defer_lock);
    if (piped) {
        pipe_lock.lock();
        for (auto& I : b->inst) PL.lanes.push_back(Lane{I.inst, I.rounds, 0, {}, {}});
        PROF("batched_prove: Pipeline::begin");
        int rc = PL.begin();
        if (rc) return rc;
    }
    const bool trace = getenv("ATLAS_TRACE") != nullptr;          // per instance: wall clock of compute_message / ingest_challenge, summed
    std::vector<double> t_msg(n, 0.0), t_ing(n, 0.0), t_wait(n, 0.0);
    double t_fs = 0, t_enq = 0;
    // A batch of thousands of host-stepped instances (the opening-reduction sumcheck of a whole graph: one row per committed one-hot
    // polynomial) spends the round in per-instance arithmetic — ~1 us each: 3-9 ms per round on this thread.  The instances that allow it
    // (host_parallel: the rows of a OneHotPool) go to a few worker threads, after the first of them has run here (it does the shared
    // launches of the round); the reference runs the same loop under Rayon.  Field sums are exact: the split changes no value.
    H::HostThreads* HT = (!piped && n >= 1024) ? &H::HostThreads::get() : nullptr;
    if (HT && HT->threads() < 2) HT = nullptr;
    std::vector<uint8_t> par(n, 0);
    if (HT) for (size_t i = 0; i < n; i++) par[i] = !b->inst[i].remote && b->inst[i].inst->host_parallel() ? 1 : 0;
    // the members this thread visits every round: the serial ones, and ONE member per shared key of the parallel ones (shared_key(): a pool's
    // rows share their launches; the longest row stands for the pool) — a walk over all 8343 members of the GPT-2-shaped reduction, twice a
    // round, was ~0.4 ms of cache misses per round on the thread that runs the transcript
    std::vector<size_t> visit;
    if (HT) {
        std::unordered_map<const void*, size_t> rep;
        for (size_t i = 0; i < n; i++) {
            if (b->inst[i].remote) continue;
            if (!par[i]) { visit.push_back(i); continue; }
            const void* key = b->inst[i].inst->shared_key();
            auto it = rep.find(key);
            if (it == rep.end()) rep.emplace(key, i);
            else if (b->inst[i].rounds > b->inst[it->second].rounds) it->second = i;
        }
        for (auto& kv : rep) visit.push_back(kv.second);
        std::sort(visit.begin(), visit.end());
    }
    double tt_wait = 0;
    std::map<size_t, double> tt_visit;
    double tt[6] = {0, 0, 0, 0, 0, 0};                   // ATLAS_TRACE: message serial / parallel, combine + transcript, claim update, ingest serial / parallel
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    atlas_rt::Runtime* const rt_owner = atlas_rt::g_cur;          // the pool's workers run the members' arithmetic under THIS thread's runtime
    std::vector<std::vector<H::Fr>> ht_polys(HT ? n : 0);       // the members' round polynomials: allocated once, refilled every round (thousands of members:
    std::vector<std::vector<H::Fr>> ht_partial(HT ? HT->threads() : 0);   // a vector each per round was ~0.4 M allocations per GPT-2-shaped reduction)
    for (size_t round = 0; round < max_rounds && HT; round++) {
        const size_t remaining = max_rounds - round;
        std::vector<std::vector<H::Fr>>& polys = ht_polys;
        std::vector<int> rcs(HT->threads(), ATLAS_OK);
        const auto q0 = tnow();
        // compute_message: the constant members and the serial ones here; for the parallel ones first everything they share (a pool's launches
        // of the round — whichever of its rows is in its cycle phase, however many pools the batch holds) on THIS thread, then the per-member
        // arithmetic on the workers, which never touch the device
        for (size_t i : visit) {
            Instance& I = b->inst[i];
            if (remaining > I.rounds || I.remote) continue;       // (its constant polynomial: with the parallel part below)
            if (par[i]) {
                const auto v0 = tnow();
                int rc = I.inst->shared_message_step(round - (max_rounds - I.rounds));
                if (rc) return rc;
                if (trace) tt_visit[i] += tms(v0, tnow());
                continue;
            }
            int rc = I.inst->message(round - (max_rounds - I.rounds), claim[i], polys[i]);
            if (rc) return rc;
        }
        const auto q0b = tnow();
        if (rt().pending_async) {                                   // one wait for everything the shared steps launched without waiting
            std::lock_guard<atlas_rt::Mutex> lk(rt().mu);
            rt().pending_async = 0;
            HIP_TRY(hipStreamSynchronize(rt().stream));
        }
        const auto q1 = tnow();
        tt_wait += tms(q0b, q1);
        // the members that have not started (a constant polynomial each: thousands of them in the first rounds) and the parallel ones, in ranges of [0, n)
        HT->parallel_for(n, [&](size_t lo, size_t hi, size_t part) {
            const atlas_rt::RtScope rt_scope(rt_owner);
            for (size_t i = lo; i < hi && rcs[part] == ATLAS_OK; i++) {
                Instance& I = b->inst[i];
                if (I.remote) polys[i].assign(1, H::zero());
                else if (remaining > I.rounds) polys[i].assign(1, mul_pow2(I.input_claim, remaining - I.rounds - 1));
                else if (par[i]) rcs[part] = I.inst->message(round - (max_rounds - I.rounds), claim[i], polys[i]);
            }
        });
        for (int rc : rcs) if (rc) return fail(rc, "batched_prove: a member's compute_message failed on a worker thread");
        const auto q2 = tnow();
        // batched = sum coeff_i * poly_i (from_coeff trimming per term, the sum keeps the longest length): partial sums per thread
        std::vector<std::vector<H::Fr>>& partial = ht_partial;
        for (auto& acc : partial) acc.clear();
        HT->parallel_for(n, [&](size_t lo, size_t hi, size_t part) {
            const atlas_rt::RtScope rt_scope(rt_owner);
            std::vector<H::Fr>& acc = partial[part];
            H::Fr t[32];
            for (size_t i = lo; i < hi; i++) {                    // += trimmed_scale(polys[i], coeff[i]), without the temporary
                const std::vector<H::Fr>& p = polys[i];
                if (p.size() > 32) { const std::vector<H::Fr> tv = trimmed_scale(p, coeff[i]); for (size_t k = 0; k < tv.size(); k++) { if (k < acc.size()) acc[k] = H::add(acc[k], tv[k]); else acc.push_back(tv[k]); } continue; }
                size_t len = 0;
                for (size_t k = 0; k < p.size(); k++) { t[k] = H::mul(p[k], coeff[i]); if (!(t[k] == H::zero())) len = k + 1; }
                if (len == 0) { t[0] = H::zero(); len = 1; }
                for (size_t k = 0; k < len; k++) {
                    if (k < acc.size()) acc[k] = H::add(acc[k], t[k]);
                    else acc.push_back(t[k]);
                }
            }
        });
        std::vector<H::Fr> batched = {H::zero()};
        for (auto& acc : partial)
            for (size_t k = 0; k < acc.size(); k++) {
                if (k < batched.size()) batched[k] = H::add(batched[k], acc[k]);
                else batched.push_back(acc[k]);
            }
        if (b->shard) { const int rc = shard_sum_poly(b->shard, batched); if (rc) return rc; }
        std::vector<H::Fr> cc;
        if (batched.size() < 2) cc = batched;
        else { cc.push_back(batched[0]); for (size_t k = 2; k < batched.size(); k++) cc.push_back(batched[k]); }
        if (cc.size() > row_stride) return fail(ATLAS_EINVAL, "batched_prove: row_stride below the batched degree");
        H::tr_append_message(T, "UniPoly_begin");
        for (auto& x : cc) H::tr_append_scalar(T, x);
        H::tr_append_message(T, "UniPoly_end");
        n_coeffs[round] = (uint32_t)cc.size();
        std::memcpy(&compressed[round * row_stride], cc.data(), cc.size() * 32);
        uint64_t lo64, hi64;
        H::tr_challenge_u128(T, lo64, hi64);
        challenges[round].lo = lo64; challenges[round].hi = hi64;
        const H::Fr r = H::challenge_to_fr(lo64, hi64, rt().challenge_mode);
        const auto q3 = tnow();
        HT->parallel_for(n, [&](size_t lo, size_t hi, size_t) { for (size_t i = lo; i < hi; i++) claim[i] = eval_with_challenge(polys[i], r); });
        const auto q4 = tnow();
        // ingest_challenge, in the same steps
        for (size_t i : visit) {
            Instance& I = b->inst[i];
            if (remaining > I.rounds || I.remote) continue;
            if (par[i]) {
                int rc = I.inst->shared_ingest_step(challenges[round], round - (max_rounds - I.rounds));
                if (rc) return rc;
                continue;
            }
            int rc = I.inst->ingest(challenges[round], round - (max_rounds - I.rounds));
            if (rc) return rc;
        }
        std::fill(rcs.begin(), rcs.end(), ATLAS_OK);
        const auto q5 = tnow();
        HT->parallel_for(n, [&](size_t lo, size_t hi, size_t part) {
            const atlas_rt::RtScope rt_scope(rt_owner);
            for (size_t i = lo; i < hi && rcs[part] == ATLAS_OK; i++) {
                Instance& I = b->inst[i];
                if (remaining <= I.rounds && par[i]) rcs[part] = I.inst->ingest(challenges[round], round - (max_rounds - I.rounds));
            }
        });
        for (int rc : rcs) if (rc) return fail(rc, "batched_prove: a member's ingest_challenge failed on a worker thread");
        const auto q6 = tnow();
        tt[0] += tms(q0, q1); tt[1] += tms(q1, q2); tt[2] += tms(q2, q3); tt[3] += tms(q3, q4); tt[4] += tms(q4, q5); tt[5] += tms(q5, q6);
    }
    if (HT) {
        if (trace) fprintf(stderr, "[atlas trace] batched_prove (%zu instances on %zu host threads, %zu rounds): message first + serial %.3f ms, message parallel %.3f ms, combine + transcript %.3f ms, "
                                   "claim update %.3f ms, ingest first + serial %.3f ms, ingest parallel %.3f ms; of message first + serial: %.3f ms in the wait for the launches without their own (%zu members visited per round)\n", n, HT->threads(), max_rounds, tt[0], tt[1], tt[2], tt[3], tt[4], tt[5], tt_wait, visit.size());
        if (trace) for (auto& kv : tt_visit) fprintf(stderr, "[atlas trace] batched_prove: shared message steps of member %zu (%zu rounds): %.3f ms\n", kv.first, b->inst[kv.first].rounds, kv.second);
        *max_rounds_out = max_rounds;
        return ATLAS_OK;
    }
    for (size_t round = 0; round < max_rounds; round++) {
        const size_t remaining = max_rounds - round;
        const auto tr_round0 = std::chrono::steady_clock::now();
        std::vector<std::vector<H::Fr>> polys(n);
        for (size_t i = 0; i < n; i++) {
            Instance& I = b->inst[i];
            if (I.remote) polys[i] = {H::zero()};
            else if (remaining > I.rounds) {
                // constant polynomial, from_coeff (a zero claim stays [0])
                polys[i] = {mul_pow2(I.input_claim, remaining - I.rounds - 1)};
            } else if (piped) {
                const size_t local = round - (max_rounds - I.rounds);
                H::Fr sums[16];
                const auto tc0 = std::chrono::steady_clock::now();
                I.inst->prepare(local);
                int rc;
                { PROF("batched_prove: collect (wait for the device)"); rc = PL.collect(PL.lanes[i].mails[local], PL.mtag(round, i), sums) ? ATLAS_OK : fail(ATLAS_ENODEV, "round channel: no answer from the device"); }
                PL.C.host_mark(101, PL.mtag(round, i));
                const auto tc1 = std::chrono::steady_clock::now();
                if (!rc) { PROF("batched_prove: finish"); rc = I.inst->finish(local, claim[i], sums, polys[i]); }
                if (trace) { t_wait[i] += std::chrono::duration<double, std::milli>(tc1 - tc0).count(); t_msg[i] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc1).count(); }
                if (rc) { PL.abort_from(round); PL.drain(); return rc; }
            } else {
                const auto tm0 = std::chrono::steady_clock::now();
                int rc = I.inst->message(round - (max_rounds - I.rounds), claim[i], polys[i]);
                if (rc) return rc;
                if (trace) t_msg[i] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tm0).count();
            }
        }
        const auto tf0 = std::chrono::steady_clock::now();
        const double pf0 = atlas_rt::Prof::on() ? atlas_rt::Prof::now_us() : 0;
        // batched = sum coeff_i * poly_i, starting from UniPoly::from_coeff(vec![]) = [0]  (:109-116)
        std::vector<H::Fr> batched = {H::zero()};
        for (size_t i = 0; i < n; i++) {
            std::vector<H::Fr> t = trimmed_scale(polys[i], coeff[i]);
            for (size_t k = 0; k < t.size(); k++) {
                if (k < batched.size()) batched[k] = H::add(batched[k], t[k]);
                else batched.push_back(t[k]);
            }
        }
        if (b->shard) { const int rc = shard_sum_poly(b->shard, batched); if (rc) return rc; }
        // compress (unipoly.rs:307-318) + append_to_transcript (:550-558)
        std::vector<H::Fr> cc;
        if (batched.size() < 2) cc = batched;
        else { cc.push_back(batched[0]); for (size_t k = 2; k < batched.size(); k++) cc.push_back(batched[k]); }
        if (cc.size() > row_stride) { if (piped) { PL.abort_from(round); PL.drain(); } return fail(ATLAS_EINVAL, "batched_prove: row_stride below the batched degree"); }
        const double pf1 = atlas_rt::Prof::on() ? atlas_rt::Prof::now_us() : 0;
        H::tr_append_message(T, "UniPoly_begin");
        for (auto& x : cc) H::tr_append_scalar(T, x);
        H::tr_append_message(T, "UniPoly_end");
        n_coeffs[round] = (uint32_t)cc.size();
        std::memcpy(&compressed[round * row_stride], cc.data(), cc.size() * 32);
        uint64_t lo, hi;
        H::tr_challenge_u128(T, lo, hi);                                              // challenge_scalar_optimized :119
        challenges[round].lo = lo; challenges[round].hi = hi;
        if (piped) { PL.C.publish(PL.slot0 + round, PL.rtag(round), lo, hi); PL.C.host_mark(100, PL.rtag(round)); if ((round & 7) == 7 && PL.next_enqueue_launched != batch_queried_at) { PL.query(); batch_queried_at = PL.next_enqueue_launched; } }
        const H::Fr r = H::challenge_to_fr(lo, hi, rt().challenge_mode);
        const double pf2 = atlas_rt::Prof::on() ? atlas_rt::Prof::now_us() : 0;
        for (size_t i = 0; i < n; i++) claim[i] = eval_with_challenge(polys[i], r);    // :123-126
        if (atlas_rt::Prof::on()) {
            const double pf3 = atlas_rt::Prof::now_us();
            atlas_rt::Prof::get().add("batched_prove: combine (scale + sum)", pf1 - pf0);
            atlas_rt::Prof::get().add("batched_prove: transcript + publish", pf2 - pf1);
            atlas_rt::Prof::get().add("batched_prove: claim update", pf3 - pf2);
        }
        if (trace) t_fs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf0).count();
        for (size_t i = 0; i < n; i++) {
            Instance& I = b->inst[i];
            if (remaining <= I.rounds && !I.remote) {
                const auto ti0 = std::chrono::steady_clock::now();
                PROF("batched_prove: ingest");
                int rc = piped ? I.inst->host_ingest(challenges[round], round - (max_rounds - I.rounds))
                               : I.inst->ingest(challenges[round], round - (max_rounds - I.rounds));
                if (rc) { if (piped) { PL.abort_from(round + 1); PL.drain(); } return rc; }
                if (trace) t_ing[i] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti0).count();
            }
        }
        if (piped) {
            const auto te0 = std::chrono::steady_clock::now();
            PROF("batched_prove: advance (enqueue)");
            int rc = PL.advance(round + 1);       // (advance drains on its own failures)
            if (rc) return rc;
            if (trace) t_enq += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - te0).count();
            static const bool per_round = getenv("ATLAS_TRACE_ROUNDS") != nullptr;
            if (trace && per_round) {
                const auto te1 = std::chrono::steady_clock::now();
                fprintf(stderr, "[atlas trace]   batch round %3zu: collect+finish (all lanes) %6.1f  combine..ingest %5.1f  enqueue %5.1f us\n", round,
                        std::chrono::duration<double, std::micro>(tf0 - tr_round0).count(), std::chrono::duration<double, std::micro>(te0 - tf0).count(),
                        std::chrono::duration<double, std::micro>(te1 - te0).count());
            }
        }
    }
    *max_rounds_out = max_rounds;
    if (trace)
        for (size_t i = 0; i < n; i++)
            fprintf(stderr, "[atlas trace] batched_prove instance %zu (%zu rounds, degree %zu): wait for sums %.3f ms, compute_message / finish %.3f ms, ingest_challenge %.3f ms\n",
                    i, b->inst[i].rounds, b->inst[i].remote ? (size_t)0 : b->inst[i].inst->degree(), t_wait[i], t_msg[i], t_ing[i]);
    if (trace) fprintf(stderr, "[atlas trace] batched_prove: combine + transcript + publish %.3f ms, enqueue %.3f ms\n", t_fs, t_enq);
    if (piped) { PROF("batched_prove: collect_finals + join"); return PL.collect_finals(); }
    return ATLAS_OK;
}

}  // extern "C"
