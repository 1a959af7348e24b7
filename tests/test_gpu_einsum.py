"""GPU parity: einsum operand folds (i32 matrix x Fr vector) and a whole mk,kn->mn einsum
instance (fold -> dot-product sumcheck) against the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _orc_fold(orc, M, eq, which):
    rows, cols = M.shape
    out = orc.fr_array(rows if which == "rows" else cols)
    fn = orc.lib.orc_fold_i32_rows if which == "rows" else orc.lib.orc_fold_i32_cols
    fn(M.ctypes.data_as(orc.i32p), C.c_size_t(rows), C.c_size_t(cols), orc._p(np.ascontiguousarray(eq)), orc._p(out))
    return out


@pytest.mark.parametrize("rows,cols,full_range", [(1, 1, False), (4, 7, False), (16, 300, True), (64, 1024, False),
                                                  (256, 4096, True), (8, 65536, True)])
def test_fold_rows(atlas, rows, cols, full_range):
    from oracle import orc
    rng = np.random.default_rng(rows * 31 + cols)
    lim = 1 << 31 if full_range else 1 << 14
    M = rng.integers(-lim, lim, size=(rows, cols), dtype=np.int64).astype(np.int32)
    if full_range:
        M[0, 0] = -(2 ** 31); M[-1, -1] = 2 ** 31 - 1
        M[rows // 2, :] = -(2 ** 31)                  # a whole row of the most negative value
    eq = orc.random_fr(cols, 5 + cols)
    t = atlas.TensorI32(M)
    e = atlas.MultilinearPolynomial.from_fr(eq) if (cols & (cols - 1)) == 0 else None
    if e is None:                                      # non-power-of-two vector: wrap via a padded upload
        pad = 1 << (cols - 1).bit_length()
        e = atlas.MultilinearPolynomial.from_fr(np.concatenate([eq, np.zeros((pad - cols, 4), dtype=np.uint64)]))
        Mp = np.zeros((rows, pad), dtype=np.int32); Mp[:, :cols] = M
        t.free(); t = atlas.TensorI32(Mp)
    out = atlas.fold_rows(t, e)
    assert np.array_equal(out.to_host(), _orc_fold(orc, M, eq, "rows"))
    out.free(); e.free(); t.free()


@pytest.mark.parametrize("rows,cols,full_range", [(1, 1, False), (16, 64, True), (256, 1024, False), (4096, 256, True),
                                                  (16384, 16, True)])
def test_fold_cols(atlas, rows, cols, full_range):
    from oracle import orc
    rng = np.random.default_rng(rows * 17 + cols)
    lim = 1 << 31 if full_range else 1 << 14
    M = rng.integers(-lim, lim, size=(rows, cols), dtype=np.int64).astype(np.int32)
    if full_range:
        M[:, 0] = -(2 ** 31)
    eq = orc.random_fr(rows, 9 + rows)
    t = atlas.TensorI32(M)
    e = atlas.MultilinearPolynomial.from_fr(eq)
    out = atlas.fold_cols(t, e)
    assert np.array_equal(out.to_host(), _orc_fold(orc, M, eq, "cols"))
    out.free(); e.free(); t.free()


def test_einsum_mk_kn_mn_instance(atlas):
    """fold at the output point, then the degree-2 sumcheck over k: the input claim equals the
    MLE of the i64 product A*B at (r_m, r_n) — the reference's acc(r) (dot.rs:204-207)."""
    from oracle import orc
    from oracle.pymodel import field as F
    A_ = atlas
    m, k, n = 8, 2048, 16
    rng = np.random.default_rng(3)
    A = rng.integers(-(1 << 14), 1 << 14, size=(m, k), dtype=np.int64).astype(np.int32)
    B = rng.integers(-(1 << 14), 1 << 14, size=(k, n), dtype=np.int64).astype(np.int32)
    r_m, r_n = orc.random_fr(3, 1), orc.random_fr(4, 2)
    eq_m, eq_n = A_.EqPolynomial.evals(r_m), A_.EqPolynomial.evals(r_n)
    tA, tB = A_.TensorI32(A), A_.TensorI32(B)
    left, right = A_.fold_cols(tA, eq_m), A_.fold_rows(tB, eq_n)
    Lh, Rh = left.to_host(), right.to_host()
    from jolt_atlas_amd import einsum_layouts as EL           # the layout entry point gives the same operands
    l2, r2 = EL.fold_mk_kn_mn(tA, tB, m, k, n, eq_m, eq_n)
    assert np.array_equal(l2.to_host(), Lh) and np.array_equal(r2.to_host(), Rh)
    l2.free(); r2.free()
    with pytest.raises(A_.AtlasError):
        EL.einsum_fold(EL.MK_KN_MN, (m, k), tA, tB, eq_m, eq_n)          # wrong number of dims
    assert np.array_equal(Lh, _orc_fold(orc, A, orc.eq_evals(r_m), "cols"))
    assert np.array_equal(Rh, _orc_fold(orc, B, orc.eq_evals(r_n), "rows"))
    # claim = MLE of the integer matmul at (r_m | r_n)
    Cmat = (A.astype(np.int64) @ B.astype(np.int64)).reshape(-1)
    Cf = orc.from_ints([int(v) % F.FR for v in Cmat])
    want = orc.evaluate(Cf, np.concatenate([r_m, r_n]))
    prover = A_.EinsumDotProver(left, right, None, A_.EQ_NONE, 0, 0)
    claim = prover.input_claim()
    assert np.array_equal(claim, want)
    t_g = A_.Blake2bTranscript(b"einsum")
    proof_g, ch_g, fin_g = A_.Sumcheck.prove(prover, claim, t_g, 11)
    prover.free()
    t_o = orc.new_transcript(b"einsum")
    proof_o, ch_o, fin_o = orc.sumcheck_dot_prove(Lh, Rh, orc.dot_claim(Lh, Rh), t_o)
    assert ch_g == ch_o and np.array_equal(proof_g, proof_o) and np.array_equal(fin_g, fin_o)
    assert t_g.state == t_o.state_bytes()
    eq_m.free(); eq_n.free(); tA.free(); tB.free()
