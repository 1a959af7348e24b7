"""GPU parity: Shout lookup argument (dense tables) — weighted histograms G, per-chunk ra
evaluations, and the ReadRaf sumcheck — against the oracle's literal restatement (val, G and
IdentityPolynomial bound separately)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _orc_G(orc, idx, log_K, E):
    G = orc.fr_array(1 << log_K)
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    orc.lib.orc_shout_G(idx.ctypes.data_as(orc.u64p), C.c_size_t(len(idx)), C.c_size_t(log_K), orc._p(np.ascontiguousarray(E)), orc._p(G))
    return G


@pytest.mark.parametrize("log_T,log_K,skew", [(0, 1, False), (6, 4, False), (10, 8, True), (12, 4, False), (13, 14, False),
                                              (14, 18, True), (16, 4, True)])
def test_read_raf_G(atlas, log_T, log_K, skew):
    from oracle import orc
    T = 1 << log_T
    rng = np.random.default_rng(log_T * 100 + log_K)
    idx = rng.integers(0, 1 << log_K, size=T, dtype=np.uint64)
    if skew:
        idx[: T // 2] = idx[0]                     # half the lookups hit one entry (ReLU-like tables)
    r = orc.random_fr(log_T, 50 + log_T) if log_T else np.zeros((0, 4), dtype=np.uint64)
    Eg = atlas.EqPolynomial.evals(r)
    Eo = orc.eq_evals(r) if log_T else orc.from_ints([1])
    G = atlas.Shout.read_raf_G(idx, log_K, Eg)
    assert np.array_equal(G.to_host(), _orc_G(orc, idx, log_K, Eo))
    G.free(); Eg.free()


def test_read_raf_G_rejects_out_of_table_index(atlas):
    from oracle import orc
    Eg = atlas.EqPolynomial.evals(orc.random_fr(3, 1))
    with pytest.raises(atlas.AtlasError, match="outside the table"):
        atlas.Shout.read_raf_G(np.array([1, 2, 99, 0, 0, 0, 0, 0], dtype=np.uint64), 4, Eg)
    Eg.free()


@pytest.mark.parametrize("log_T,log_K,log_kc", [(8, 32, 4), (12, 64, 4), (10, 18, 4), (9, 16, 8)])
def test_ra_evals(atlas, log_T, log_K, log_kc):
    from oracle import orc
    T = 1 << log_T
    rng = np.random.default_rng(log_K)
    idx = rng.integers(0, 1 << 63, size=T, dtype=np.uint64) * 2 + rng.integers(0, 2, size=T, dtype=np.uint64)
    if log_K < 64:
        idx &= np.uint64((1 << log_K) - 1)
    r = orc.random_fr(log_T, 7)
    Eg = atlas.EqPolynomial.evals(r); Eo = orc.eq_evals(r)
    d = (log_K + log_kc - 1) // log_kc
    want = orc.fr_array(d << log_kc)
    orc.lib.orc_shout_ra_evals(idx.ctypes.data_as(orc.u64p), C.c_size_t(T), C.c_size_t(log_K), C.c_size_t(log_kc),
                               orc._p(Eo), orc._p(want))
    G = atlas.Shout.ra_evals(idx, log_K, log_kc, Eg)
    assert np.array_equal(G.to_host(), want)
    # each chunk's histogram sums to sum_j E[j] = 1 (eq table sums to one)
    from oracle.pymodel import field as F
    rows = np.array(orc.to_ints(G.to_host()), dtype=object).reshape(d, 1 << log_kc)
    assert all(int(sum(row)) % F.FR == 1 for row in rows)
    G.free(); Eg.free()


@pytest.mark.parametrize("log_T,log_K", [(6, 4), (12, 8), (13, 12), (14, 18)])
def test_read_raf_sumcheck_bit_exact(atlas, log_T, log_K):
    """ReadRafProver: the device proves sum G*(val + gamma*int) as a dot product over (G, W);
    the oracle binds val, G and the identity polynomial separately as the reference does."""
    from oracle import orc
    A = atlas
    T, K = 1 << log_T, 1 << log_K
    rng = np.random.default_rng(log_T + 31 * log_K)
    idx = rng.integers(0, K, size=T, dtype=np.uint64)
    table = rng.integers(-(1 << 20), 1 << 20, size=K, dtype=np.int64).astype(np.int32)
    r = orc.random_fr(log_T, 3)
    gamma = orc.random_fr(1, 4)[0]
    Eg = A.EqPolynomial.evals(r)
    G = A.Shout.read_raf_G(idx, log_K, Eg)
    Go = _orc_G(orc, idx, log_K, orc.eq_evals(r))
    claim = orc.fr_array(1)
    orc.lib.orc_readraf_claim(orc._p(Go), table.ctypes.data_as(orc.i32p), C.c_size_t(log_K), orc._p(np.ascontiguousarray(gamma).reshape(1, 4)), orc._p(claim))
    prover = A.Shout.read_raf_prover(G, table, log_K, gamma)
    assert np.array_equal(prover.input_claim(), claim[0])
    t_g = A.Blake2bTranscript(b"shout")
    proof_g, ch_g, fin_g = A.Sumcheck.prove(prover, claim[0], t_g, log_K)
    prover.free()
    t_o = orc.new_transcript(b"shout")
    proof_o = orc.fr_array(2 * log_K); ch = np.zeros(2 * log_K, dtype=np.uint64); fG = orc.fr_array(1)
    orc.lib.orc_sumcheck_readraf_prove(orc._p(Go.copy()), table.ctypes.data_as(orc.i32p), C.c_size_t(log_K),
                                       orc._p(np.ascontiguousarray(gamma).reshape(1, 4)), orc._p(claim), C.byref(t_o),
                                       orc._p(proof_o), orc._p(ch), orc._p(fG))
    ch_o = [int(ch[2 * i]) | (int(ch[2 * i + 1]) << 64) for i in range(log_K)]
    assert ch_g == ch_o
    assert np.array_equal(proof_g.reshape(-1, 4), proof_o)
    assert np.array_equal(fin_g[0], fG[0])
    assert t_g.state == t_o.state_bytes()
    Eg.free()
