"""GPU parity: EinsumLayout::fold of the batched layouts through the strided device folds vs the
oracle's loop-for-loop restatement (oracle/einsum.c)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle(layout, L, R, b, m, k, n, eq_m, eq_n):
    from oracle import orc
    lo, ro = orc.fr_array(max(k * b, 1)), orc.fr_array(max(k * b, 1))
    orc.lib.orc_einsum_fold_layout(C.c_int(layout), L.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p),
                                   *(C.c_size_t(x) for x in (b, m, k, n)), orc._p(eq_m), orc._p(eq_n), orc._p(lo), orc._p(ro))
    return lo, ro


@pytest.mark.parametrize("b,m,k,n", [(2, 4, 8, 4), (4, 16, 64, 32), (8, 2, 16, 128), (1, 8, 4, 2)])
@pytest.mark.parametrize("layout", [0, 1, 2, 3])
def test_batched_einsum_folds(atlas, layout, b, m, k, n):
    from oracle import orc
    from jolt_atlas_amd import einsum_layouts as EL
    A = atlas
    rng = np.random.default_rng(layout * 100 + b + m + k + n)
    L = rng.integers(-(1 << 31), (1 << 31) - 1, size=b * m * k, dtype=np.int32)
    R = rng.integers(-(1 << 31), (1 << 31) - 1, size=b * k * n, dtype=np.int32)
    L[:3] = [-(1 << 31), (1 << 31) - 1, 0]
    r_m, r_n = orc.random_fr(int(np.log2(m)), 1), orc.random_fr(int(np.log2(n)), 2)
    eq_m_h, eq_n_h = orc.eq_evals(r_m), orc.eq_evals(r_n)
    lo_w, ro_w = _oracle(layout, L, R, b, m, k, n, eq_m_h, eq_n_h)
    eq_m, eq_n = A.EqPolynomial.evals(r_m), A.EqPolynomial.evals(r_n)
    tl, tr = A.TensorI32(L.reshape(1, -1)), A.TensorI32(R.reshape(1, -1))
    if layout <= 1:
        lo, ro = EL.fold_bmk_rhs_mbn(tl, tr, b, m, k, n, eq_m, eq_n, "bkn" if layout == 0 else "kbn")
    else:
        lo, ro = EL.fold_mbk_rhs_bmn(tl, tr, b, m, k, n, eq_m, eq_n, "bnk" if layout == 2 else "nbk")
    assert np.array_equal(lo.to_host(), lo_w[:k * b])
    assert np.array_equal(ro.to_host(), ro_w[:k * b])
    for x in (lo, ro, eq_m, eq_n):
        x.free()
    tl.free(); tr.free()


@pytest.mark.parametrize("k,n", [(8, 4), (256, 64), (1024, 2)])
def test_k_nk_n_fold(atlas, k, n):
    from oracle import orc
    from jolt_atlas_amd import einsum_layouts as EL
    A = atlas
    R = np.random.default_rng(k + n).integers(-(1 << 20), 1 << 20, size=n * k, dtype=np.int32)
    r = orc.random_fr(int(np.log2(n)), 3)
    eq_h = orc.eq_evals(r)
    _, ro_w = _oracle(4, R, R, 1, 1, k, n, eq_h, eq_h)
    eq = A.EqPolynomial.evals(r)
    t = A.TensorI32(R.reshape(1, -1))
    ro = EL.fold_k_nk_n(t, k, n, eq)
    assert np.array_equal(ro.to_host(), ro_w[:k])
    ro.free(); eq.free(); t.free()


@pytest.mark.parametrize("variant,a,c,b,m,n,k", [(0, 2, 1, 4, 8, 4, 16), (0, 1, 1, 2, 2, 64, 4), (1, 2, 4, 2, 4, 8, 8), (1, 4, 2, 1, 16, 2, 32),
                                                  (2, 1, 2, 4, 8, 16, 8), (2, 1, 1, 1, 4, 4, 2)])
def test_rbmk_rbnk_bmn_folds(atlas, variant, a, c, b, m, n, k):
    from oracle import orc
    from jolt_atlas_amd import einsum_layouts as EL
    A = atlas
    rng = np.random.default_rng(variant * 1000 + a + c + b + m + n + k)
    if variant == 0:
        nl, nr, nout = a * b * m * k, a * b * n * k, a * b * k
    elif variant == 1:
        nl, nr, nout = a * c * b * m * k, k * c * n, c * b * a * k
    else:
        nl, nr, nout = c * b * m * k, c * b * k * n, c * b * k
    L = rng.integers(-(1 << 31), (1 << 31) - 1, size=nl, dtype=np.int32)
    R = rng.integers(-(1 << 31), (1 << 31) - 1, size=nr, dtype=np.int32)
    r_m, r_n = orc.random_fr(int(np.log2(m)), 1), orc.random_fr(int(np.log2(n)), 2)
    eq_m_h, eq_n_h = orc.eq_evals(r_m), orc.eq_evals(r_n)
    lo_w, ro_w = orc.fr_array(nout), orc.fr_array(nout)
    orc.lib.orc_einsum_fold_rbmk(C.c_int(variant), L.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p),
                                 *(C.c_size_t(x) for x in (a, c, b, m, n, k)), orc._p(eq_m_h), orc._p(eq_n_h), orc._p(lo_w), orc._p(ro_w))
    eq_m, eq_n = A.EqPolynomial.evals(r_m), A.EqPolynomial.evals(r_n)
    tl, tr = A.TensorI32(L.reshape(1, -1)), A.TensorI32(R.reshape(1, -1))
    if variant == 0:
        lo, ro = EL.fold_abmk_abnk_abmn(tl, tr, a, b, m, n, k, eq_m, eq_n)
    elif variant == 1:
        lo, ro = EL.fold_acbmk_kcn_cbmn(tl, tr, a, c, b, m, n, k, eq_m, eq_n)
    else:
        lo, ro = EL.fold_cbmk_cbkn_amn(tl, tr, c * b, m, n, k, eq_m, eq_n)
    assert np.array_equal(lo.to_host(), lo_w)
    assert np.array_equal(ro.to_host(), ro_w)
    for x in (lo, ro, eq_m, eq_n):
        x.free()
    tl.free(); tr.free()
