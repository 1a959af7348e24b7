"""EinsumLayout::fold for the batched layouts, expressed with the two strided device folds
(jolt-atlas-core/src/onnx_proof/ops/einsum/bmk_rhs_mbn.rs, mbk_rhs_bmn.rs, k_nk_n.rs)."""
import ctypes as C

from . import MultilinearPolynomial, _check, lib


def _rows_batched(d, n0, n1, s0, s1, R, t0, t1, eq):
    h = C.c_void_p()
    _check(lib.atlas_fold_i32_rows_batched(d, *(C.c_size_t(x) for x in (n0, n1, s0, s1, R, t0, t1)), eq.h, C.byref(h)))
    return MultilinearPolynomial(h)


def _cols_batched(d, B, sB, R, sR, Cn, tB, tC, eq):
    h = C.c_void_p()
    _check(lib.atlas_fold_i32_cols_batched(d, *(C.c_size_t(x) for x in (B, sB, R, sR, Cn, tB, tC)), eq.h, C.byref(h)))
    return MultilinearPolynomial(h)


def fold_bmk_rhs_mbn(left, right, b, m, k, n, eq_r_m, eq_r_n, rhs):
    """bmk,{bkn|kbn}->mbn.  left/right: TensorI32 (flat). Returns (left, right) polynomials of k*b."""
    lo = _cols_batched(left.d, b, k * m, m, k, k, 1, b, eq_r_m)              # lo[h*k + j] then transpose(b, k)
    if rhs == "kbn":
        ro = _rows_batched(right.d, k, b, b * n, n, n, b, 1, eq_r_n)         # ro[j*b + h]
    else:
        ro = _rows_batched(right.d, b, k, k * n, n, n, 1, b, eq_r_n)         # transpose_flat_matrix(ro, b, k)
    return lo, ro


def fold_mbk_rhs_bmn(left, right, b, m, k, n, eq_r_m, eq_r_n, rhs):
    """mbk,{bnk|nbk}->bmn."""
    lo = _cols_batched(left.d, b, k, m, k * b, k, k, 1, eq_r_m)              # sum_i left[i*k*b + h*k + j]
    if rhs == "bnk":
        ro = _cols_batched(right.d, b, n * k, n, k, k, k, 1, eq_r_n)         # sum_l right[h*n*k + l*k + j]
    else:
        ro = _cols_batched(right.d, b, k, n, k * b, k, k, 1, eq_r_n)         # sum_l right[l*k*b + h*k + j]
    return lo, ro


def fold_k_nk_n(right, k, n, eq_r_node_output):
    """k,nk->n: right[j] = sum_h B[h*k + j] * eq[h] (the left operand is used as is)."""
    return _cols_batched(right.d, 1, 0, n, k, k, 0, 1, eq_r_node_output)


def _cols_batched2(d, B0, sB0, tB0, B1, sB1, tB1, R, sR, Cn, tC, eq):
    h = C.c_void_p()
    _check(lib.atlas_fold_i32_cols_batched2(d, *(C.c_size_t(x) for x in (B0, sB0, tB0, B1, sB1, tB1, R, sR, Cn, tC)), eq.h, C.byref(h)))
    return MultilinearPolynomial(h)


def _repeat_rows(base, rows, row_len, repeat):
    h = C.c_void_p()
    _check(lib.atlas_poly_repeat_rows(base.h, C.c_size_t(rows), C.c_size_t(row_len), C.c_size_t(repeat), C.byref(h)))
    return MultilinearPolynomial(h)


def fold_abmk_abnk_abmn(left, right, a, b, m, n, k, eq_r_m, eq_r_n):
    """abmk,abnk->abmn (rbmk_rbnk_bmn.rs:163-217): left[h*k + j] = sum_i L[(h*m + i)*k + j] eq_m[i], right likewise over n."""
    batch = a * b
    return (_cols_batched(left.d, batch, m * k, m, k, k, k, 1, eq_r_m),
            _cols_batched(right.d, batch, n * k, n, k, k, k, 1, eq_r_n))


def fold_acbmk_kcn_cbmn(left, right, a, c, b, m, n, k, eq_r_m, eq_r_n):
    """acbmk,kcn->cbmn (rbmk_rbnk_bmn.rs:219-290): the left fold transposes the (a, cb) batch axes; the right operand is
    folded once per (c, k) and broadcast over (b, a)."""
    cb = c * b
    lo = _cols_batched2(left.d, a, cb * m * k, k, cb, m * k, a * k, m, k, k, 1, eq_r_m)   # out[(h*a + a_idx)*k + j]
    base = _rows_batched(right.d, c, k, n, c * n, n, k, 1, eq_r_n)                         # base[c_idx*k + k_idx]
    ro = _repeat_rows(base, c, k, b * a)
    base.free()
    return lo, ro


def fold_cbmk_cbkn_amn(left, right, cb, m, n, k, eq_r_m, eq_r_n):
    """cbmk,cbkn->amn (rbmk_rbnk_bmn.rs:292-338)."""
    return (_cols_batched(left.d, cb, m * k, m, k, k, k, 1, eq_r_m),
            _rows_batched(right.d, cb * k, 1, n, 0, n, 1, 0, eq_r_n))
