// SumcheckInstanceProver (joltworks/src/subprotocols/sumcheck_prover.rs:10-68) as the host sees
// it: what Sumcheck::prove / BatchedSumcheck::prove call on an instance.
#pragma once
#include <vector>

#include "../../include/atlas_hip.h"
#include "channel.hpp"
#include "host_field.hpp"

// where the host finds the sums a pipelined round mailed (channel.hpp: Channel::collect)
struct atlas_mail_ref {
    atlas::Chunk* base = nullptr;
    size_t blocks = 0;          // workgroups that mail a record
    int n_vals = 0;             // values per workgroup
    int radix = 32, shl = 0;    // word radix / scaling of the records (sum_to_fr)
};

struct atlas_instance {
    virtual ~atlas_instance() {}
    virtual size_t rounds() const = 0;
    virtual size_t degree() const = 0;
    // compute_message(round, previous_claim) -> UniPoly coefficients
    virtual int message(size_t round, const atlas_host::Fr& claim, std::vector<atlas_host::Fr>& coeffs) = 0;
    virtual int ingest(const atlas_u128_t& r, size_t round) = 0;
    // the final_claim()s cache_openings reads
    virtual int finals(std::vector<atlas_host::Fr>& out) = 0;

    // ---- round-channel stepping (batched.hip): the library-driven provers enqueue the launches of EVERY round
    // before the first challenge exists; a launch takes the challenge it binds from the round's slot and mails
    // its sums to the host, which finishes the round polynomial and runs the transcript (channel.hip.h).
    // Nothing an enqueue reads on the host may depend on a challenge.
    virtual bool pipelined() const { return false; }
    // launches of ingest_challenge(round - 1) [bind_prev: challenge from io's slot] + compute_message(round);
    // io.mail / io.tag_mail name where and how the sums are mailed
    virtual int enqueue(size_t /*round*/, const atlas::RoundIo& /*io*/, bool /*bind_prev*/, atlas_mail_ref& /*mail*/) { return ATLAS_ESTATE; }
    // mailed sums (reduced by the driver) -> coefficients of the round polynomial; host arithmetic only
    virtual int finish(size_t /*round*/, const atlas_host::Fr& /*claim*/, const atlas_host::Fr* /*sums*/, std::vector<atlas_host::Fr>& /*coeffs*/) { return ATLAS_ESTATE; }
    // host half of ingest_challenge(r, round)
    virtual int host_ingest(const atlas_u128_t& /*r*/, size_t /*round*/) { return ATLAS_ESTATE; }
    // after the last round: bind the last challenge (from io's slot) and mail the final claims, one value per record
    virtual int enqueue_finals(const atlas::RoundIo& /*io*/, atlas_mail_ref& /*mail*/) { return ATLAS_ESTATE; }
    // the driver hands the mailed final claims back (canonical Montgomery residues)
    virtual int set_finals(const atlas_host::Fr* /*vals*/, size_t /*n*/) { return ATLAS_ESTATE; }
    // host work that does not need the sums of `round` (inversions, ...): called while the device computes them
    virtual void prepare(size_t /*round*/) {}
    // Do the launches of enqueue(round) [round == rounds(): enqueue_finals] that wait for a challenge spin with more than
    // WIDE_WAIT_WGS workgroups?  Such a round of a lane on a stream of its own goes behind a one-wavefront gate
    // (k_ch_gate, channel.hip.h); a round that launches nothing, or a few workgroups, does not need one (a gate is a
    // launch: ~3 us of the host thread and a kernel boundary on the lane per round).
    // May message() / ingest() of this instance run on a worker thread, concurrently with those of OTHER instances that say so too, once one
    // instance of the batch has made the call for the round on the driver's thread (it does the work the instances share: launches, copies)?
    // The rows of a OneHotPool do: their round arithmetic is per row.  (batched.hip: batches of thousands of instances)
    virtual bool host_parallel() const { return false; }
    // The work of message(round) / ingest(r, round) that a host_parallel instance SHARES with others (a pool's launches for the round), done
    // on the calling thread; idempotent per round.  The driver makes these calls for every member on its own thread before it hands the
    // members to the workers, so that no worker ever touches the device (batched.hip).
    virtual int shared_message_step(size_t /*round*/) { return ATLAS_OK; }
    // what the shared steps belong to (a pool: the same key for all its rows; default: the instance alone).  The driver calls the shared steps of ONE
    // member per key — the one with the most rounds, which is in its stepping phase whenever any member of the key is.
    virtual const void* shared_key() const { return this; }
    virtual int shared_ingest_step(const atlas_u128_t& /*r*/, size_t /*round*/) { return ATLAS_OK; }
    // does enqueue(round) launch nothing at all (host-only rounds: the driver then has nothing new for the runtime to retire)?
    virtual bool silent_round(size_t /*round*/) const { return false; }
    // how many of this instance's FIRST rounds wait for nothing from the device (Booleanity's address rounds; every round of HammingWeight):
    // while every lane of a batch is inside such a prefix the rounds are host arithmetic, and a launch made there is on the critical path
    virtual size_t host_prefix() const { return 0; }
    // does this instance never launch anything when it is stepped through the round channel (HammingWeight: host arithmetic only)?
    virtual bool silent_lane() const { return false; }
    static constexpr size_t WIDE_WAIT_WGS = 256;
    virtual bool wide_wait(size_t /*round*/) const { return true; }
};
