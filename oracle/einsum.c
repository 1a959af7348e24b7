/* ORACLE — TEST INFRASTRUCTURE ONLY (see bn254.h header note; parity unpinned).
 * EinsumLayout::fold for mk,kn->mn
 * (jolt-atlas-core/src/onnx_proof/ops/einsum/mk_kn_mn.rs:47-79):
 *   left[j]  = sum_i from_i32(A[i*k + j]) * eq_m[i]
 *   right[j] = sum_h from_i32(B[j*n + h]) * eq_n[h] */
#include <stdlib.h>
#include <string.h>
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

/* Folds of the other einsum layouts, loops as written in the reference
 * (jolt-atlas-core/src/onnx_proof/ops/einsum/): layout 0 = bmk,bkn->mbn, 1 = bmk,kbn->mbn
 * (bmk_rhs_mbn.rs:78-110 + transpose_flat_matrix, utils/dims.rs:658-690), 2 = mbk,bnk->bmn,
 * 3 = mbk,nbk->bmn (mbk_rhs_bmn.rs:78-119), 4 = k,nk->n (k_nk_n.rs:46-68, right only).
 * left_out / right_out: k*b Fr each (layout 4: right_out k Fr, left_out unused). */
static void transpose_flat(const fr_t *v, size_t num_rows, size_t num_cols, fr_t *out) {
    for (size_t j = 0; j < num_cols; j++) for (size_t i = 0; i < num_rows; i++) out[j * num_rows + i] = v[i * num_cols + j];
}

void orc_einsum_fold_layout(int layout, const int32_t *left, const int32_t *right, size_t b, size_t m, size_t k, size_t n,
                            const fr_t *eq_r_m, const fr_t *eq_r_n, fr_t *left_out, fr_t *right_out) {
    fr_t a, t;
    if (layout == 4) {
        for (size_t j = 0; j < k; j++) {
            fr_t acc; fr_zero(&acc);
            for (size_t h = 0; h < n; h++) { fr_from_i64(right[h * k + j], &a); fr_mul(&a, &eq_r_n[h], &t); fr_add(&acc, &t, &acc); }
            right_out[j] = acc;
        }
        return;
    }
    fr_t *lo = (fr_t *)malloc(k * b * sizeof(fr_t)), *ro = (fr_t *)malloc(k * b * sizeof(fr_t));
    if (layout <= 1) {
        for (size_t h = 0; h < b; h++) for (size_t j = 0; j < k; j++) {
            fr_t acc; fr_zero(&acc);
            for (size_t i = 0; i < m; i++) { fr_from_i64(left[h * (k * m) + i * k + j], &a); fr_mul(&a, &eq_r_m[i], &t); fr_add(&acc, &t, &acc); }
            lo[h * k + j] = acc;
        }
        for (size_t j = 0; j < k; j++) for (size_t h = 0; h < b; h++) {
            fr_t acc; fr_zero(&acc);
            for (size_t l = 0; l < n; l++) { fr_from_i64(right[j * (b * n) + h * n + l], &a); fr_mul(&a, &eq_r_n[l], &t); fr_add(&acc, &t, &acc); }
            ro[j * b + h] = acc;
        }
        transpose_flat(lo, b, k, left_out);
        if (layout == 0) transpose_flat(ro, b, k, right_out); else memcpy(right_out, ro, k * b * sizeof(fr_t));
    } else {
        for (size_t h = 0; h < b; h++) for (size_t j = 0; j < k; j++) {
            fr_t acc; fr_zero(&acc);
            for (size_t i = 0; i < m; i++) { fr_from_i64(left[i * (k * b) + h * k + j], &a); fr_mul(&a, &eq_r_m[i], &t); fr_add(&acc, &t, &acc); }
            left_out[h * k + j] = acc;
            fr_zero(&acc);
            for (size_t l = 0; l < n; l++) {
                const size_t idx = layout == 2 ? h * (n * k) + l * k + j : l * (k * b) + h * k + j;
                fr_from_i64(right[idx], &a); fr_mul(&a, &eq_r_n[l], &t); fr_add(&acc, &t, &acc);
            }
            right_out[h * k + j] = acc;
        }
    }
    free(lo); free(ro);
}

/* rbmk,rbnk->bmn family, loops as written in ops/einsum/rbmk_rbnk_bmn.rs: variant 0 = abmk,abnk->abmn
 * (build_abmk_abnk_abmn :163-217), 1 = acbmk,kcn->cbmn (build_acbmk_kcn_cbmn :219-290),
 * 2 = cbmk,cbkn->amn (build_cbmk_cbkn_amn :292-338).  Dimensions a, c, b, m, n, k as in the reference
 * (unused ones = 1).  left_out / right_out sized by the caller (batch * k, or c*b*a*k for variant 1). */
void orc_einsum_fold_rbmk(int variant, const int32_t *left, const int32_t *right, size_t a, size_t c, size_t b, size_t m,
                          size_t n, size_t k, const fr_t *eq_r_m, const fr_t *eq_r_n, fr_t *left_out, fr_t *right_out) {
    fr_t x, t;
    if (variant == 0) {
        const size_t batch = a * b;
        for (size_t hj = 0; hj < batch * k; hj++) {
            const size_t h = hj / k, j = hj % k;
            fr_t s; fr_zero(&s);
            for (size_t i = 0; i < m; i++) { fr_from_i64(left[(h * m + i) * k + j], &x); fr_mul(&x, &eq_r_m[i], &t); fr_add(&s, &t, &s); }
            left_out[hj] = s; fr_zero(&s);
            for (size_t l = 0; l < n; l++) { fr_from_i64(right[(h * n + l) * k + j], &x); fr_mul(&x, &eq_r_n[l], &t); fr_add(&s, &t, &s); }
            right_out[hj] = s;
        }
    } else if (variant == 1) {
        const size_t cb = c * b;
        fr_t *base = (fr_t *)malloc(c * k * sizeof(fr_t));
        for (size_t ck = 0; ck < c * k; ck++) {
            const size_t c_idx = ck / k, k_idx = ck % k;
            fr_t s; fr_zero(&s);
            for (size_t ni = 0; ni < n; ni++) { fr_from_i64(right[(k_idx * c + c_idx) * n + ni], &x); fr_mul(&x, &eq_r_n[ni], &t); fr_add(&s, &t, &s); }
            base[ck] = s;
        }
        for (size_t hak = 0; hak < cb * a * k; hak++) {
            const size_t h = hak / (a * k), rem = hak % (a * k), a_idx = rem / k, k_idx = rem % k, c_idx = h / b, b_idx = h % b;
            fr_t s; fr_zero(&s);
            for (size_t i = 0; i < m; i++) {
                fr_from_i64(left[((((a_idx * c + c_idx) * b + b_idx) * m + i) * k) + k_idx], &x); fr_mul(&x, &eq_r_m[i], &t); fr_add(&s, &t, &s);
            }
            left_out[hak] = s;
            right_out[hak] = base[c_idx * k + k_idx];
        }
        free(base);
    } else {
        const size_t cb = c * b;
        for (size_t hk = 0; hk < cb * k; hk++) {
            const size_t h = hk / k, k_idx = hk % k;
            fr_t s; fr_zero(&s);
            for (size_t i = 0; i < m; i++) { fr_from_i64(left[(h * m + i) * k + k_idx], &x); fr_mul(&x, &eq_r_m[i], &t); fr_add(&s, &t, &s); }
            left_out[hk] = s; fr_zero(&s);
            for (size_t l = 0; l < n; l++) { fr_from_i64(right[(h * k + k_idx) * n + l], &x); fr_mul(&x, &eq_r_n[l], &t); fr_add(&s, &t, &s); }
            right_out[hk] = s;
        }
    }
}
