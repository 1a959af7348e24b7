// KZGProverKey::g1_powers resident in HBM (joltworks/src/poly/commitment/hyperkzg/kzg.rs:107-143): the handle behind atlas_srs_t.
#pragma once
#include <cstddef>

#include "curve.hip.h"

struct atlas_srs {
    atlas::G1Affine* d = nullptr;     // 64-byte affine points, Montgomery Fq; (0, 0) = infinity
    size_t len = 0;
    // fixed-base table (atlas_srs_precompute): level j of point i at tab[j * tab_len + i] = 2^(tab_c * j) * d[i];
    // level 0 is a copy of d[0 .. tab_len) so that one base pointer serves every level
    // The table may cover a RANGE of the powers only — points [tab_off, tab_off + tab_len) (atlas_srs_precompute_range): rank r of a sharded
    // proof commits the r-th range of every long vector (MSM by point range), so it holds 1 / world of the table; an MSM whose
    // points lie outside the range takes the variable-base path (same group element).
    atlas::G1Affine* tab = nullptr;
    size_t tab_len = 0, tab_off = 0;
    uint32_t tab_c = 0, tab_levels = 0;
};
