// SumcheckInstanceProver (joltworks/src/subprotocols/sumcheck_prover.rs:10-68) as the host sees
// it: what Sumcheck::prove / BatchedSumcheck::prove call on an instance.
#pragma once
#include <vector>

#include "../../include/atlas_hip.h"
#include "host_field.hpp"

struct atlas_instance {
    virtual ~atlas_instance() {}
    virtual size_t rounds() const = 0;
    virtual size_t degree() const = 0;
    // compute_message(round, previous_claim) -> UniPoly coefficients
    virtual int message(size_t round, const atlas_host::Fr& claim, std::vector<atlas_host::Fr>& coeffs) = 0;
    virtual int ingest(const atlas_u128_t& r, size_t round) = 0;
    // the final_claim()s cache_openings reads
    virtual int finals(std::vector<atlas_host::Fr>& out) = 0;
};
