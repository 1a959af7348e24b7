"""EinsumLayout::fold for every registered layout: ctypes marshalling over atlas_einsum_fold (the index maps live in
the library, csrc/einsum.hip; jolt-atlas-core/src/onnx_proof/ops/einsum/{mk_kn_mn,bmk_rhs_mbn,mbk_rhs_bmn,k_nk_n,
rbmk_rbnk_bmn}.rs)."""
import ctypes as C

from . import MultilinearPolynomial, _check, lib

MK_KN_MN, BMK_BKN_MBN, BMK_KBN_MBN, MBK_BNK_BMN, MBK_NBK_BMN, K_NK_N, ABMK_ABNK_ABMN, ACBMK_KCN_CBMN, CBMK_CBKN_AMN = range(9)


def einsum_fold(layout, dims, left, right, eq_r_m, eq_r_n):
    """left / right: TensorI32 resident in HBM (left may be None for k,nk->n).  Returns (left_poly, right_poly)."""
    d = (C.c_size_t * len(dims))(*dims)
    lo, ro = C.c_void_p(), C.c_void_p()
    _check(lib.atlas_einsum_fold(C.c_int(layout), d, C.c_size_t(len(dims)), left.d if left is not None else None, right.d,
                                 eq_r_m.h if eq_r_m is not None else None, eq_r_n.h, C.byref(lo), C.byref(ro)))
    return (MultilinearPolynomial(lo) if lo.value else None), MultilinearPolynomial(ro)


def fold_mk_kn_mn(left, right, m, k, n, eq_r_m, eq_r_n):
    return einsum_fold(MK_KN_MN, (m, k, n), left, right, eq_r_m, eq_r_n)


def fold_bmk_rhs_mbn(left, right, b, m, k, n, eq_r_m, eq_r_n, rhs):
    return einsum_fold(BMK_KBN_MBN if rhs == "kbn" else BMK_BKN_MBN, (b, m, k, n), left, right, eq_r_m, eq_r_n)


def fold_mbk_rhs_bmn(left, right, b, m, k, n, eq_r_m, eq_r_n, rhs):
    return einsum_fold(MBK_BNK_BMN if rhs == "bnk" else MBK_NBK_BMN, (b, m, k, n), left, right, eq_r_m, eq_r_n)


def fold_k_nk_n(right, k, n, eq_r_node_output):
    return einsum_fold(K_NK_N, (k, n), None, right, None, eq_r_node_output)[1]


def fold_abmk_abnk_abmn(left, right, a, b, m, n, k, eq_r_m, eq_r_n):
    return einsum_fold(ABMK_ABNK_ABMN, (a, b, m, n, k), left, right, eq_r_m, eq_r_n)


def fold_acbmk_kcn_cbmn(left, right, a, c, b, m, n, k, eq_r_m, eq_r_n):
    return einsum_fold(ACBMK_KCN_CBMN, (a, c, b, m, n, k), left, right, eq_r_m, eq_r_n)


def fold_cbmk_cbkn_amn(left, right, cb, m, n, k, eq_r_m, eq_r_n):
    return einsum_fold(CBMK_CBKN_AMN, (cb, m, n, k), left, right, eq_r_m, eq_r_n)
