/* ORACLE — TEST INFRASTRUCTURE ONLY (see bn254.h header note; parity unpinned).
 * EinsumLayout::fold for mk,kn->mn
 * (jolt-atlas-core/src/onnx_proof/ops/einsum/mk_kn_mn.rs:47-79):
 *   left[j]  = sum_i from_i32(A[i*k + j]) * eq_m[i]
 *   right[j] = sum_h from_i32(B[j*n + h]) * eq_n[h] */
#include "oracle.h"

void orc_fold_i32_cols(const int32_t *M, size_t rows, size_t cols, const fr_t *eq, fr_t *out) {
#pragma omp parallel for schedule(static) if (rows * cols >= 65536)
    for (size_t j = 0; j < cols; j++) {
        fr_t acc; fr_zero(&acc);
        for (size_t i = 0; i < rows; i++) {
            fr_t a, t; fr_from_i64(M[i * cols + j], &a); fr_mul(&a, &eq[i], &t); fr_add(&acc, &t, &acc);
        }
        out[j] = acc;
    }
}

void orc_fold_i32_rows(const int32_t *M, size_t rows, size_t cols, const fr_t *eq, fr_t *out) {
#pragma omp parallel for schedule(static) if (rows * cols >= 65536)
    for (size_t j = 0; j < rows; j++) {
        fr_t acc; fr_zero(&acc);
        for (size_t h = 0; h < cols; h++) {
            fr_t a, t; fr_from_i64(M[j * cols + h], &a); fr_mul(&a, &eq[h], &t); fr_add(&acc, &t, &acc);
        }
        out[j] = acc;
    }
}
