/* TEST INFRASTRUCTURE ONLY (oracle): build_materialized_rlc restated from
 * joltworks/src/poly/rlc_polynomial.rs:13-78.  dense_fr[j] or dense_i32[j] is non-NULL. */
#include <string.h>
#include "oracle.h"

void orc_rlc_build(const fr_t *const *dense_fr, const int32_t *const *dense_i32, const size_t *dense_len,
                   const fr_t *dense_coeff, size_t n_dense, const int32_t *const *oh_k, const size_t *oh_T,
                   const fr_t *oh_coeff, size_t n_oh, fr_t *joint, size_t joint_len) {
    for (size_t i = 0; i < joint_len; i++) {                      /* :42-58 */
        fr_t acc; fr_zero(&acc);
        for (size_t j = 0; j < n_dense; j++) {
            if (i >= dense_len[j]) continue;
            fr_t v, t;
            if (dense_fr[j]) v = dense_fr[j][i];
            else fr_from_i64(dense_i32[j][i], &v);                /* SmallScalar::field_mul, small_scalar.rs */
            fr_mul(&dense_coeff[j], &v, &t);
            fr_add(&acc, &t, &acc);
        }
        joint[i] = acc;
    }
    for (size_t j = 0; j < n_oh; j++) {                           /* :62-75, serial scatter */
        const size_t T = oh_T[j];
        for (size_t t = 0; t < T; t++) {
            const int32_t k = oh_k[j][t];
            if (k < 0) continue;
            const size_t idx = (size_t)k * T + t;
            fr_add(&joint[idx], &oh_coeff[j], &joint[idx]);
        }
    }
}
