// KZGProverKey::g1_powers resident in HBM (joltworks/src/poly/commitment/hyperkzg/kzg.rs:107-143): the handle behind atlas_srs_t.
#pragma once
#include <cstddef>

#include "curve.hip.h"

struct atlas_srs {
    atlas::G1Affine* d = nullptr;     // 64-byte affine points, Montgomery Fq; (0, 0) = infinity
    size_t len = 0;
};
