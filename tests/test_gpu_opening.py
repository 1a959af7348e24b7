"""GPU parity: opening-reduction sumcheck provers (opening_reduction.rs:355-425, 532-723) vs
oracle/opening.c, and a batched reduction over openings of different sizes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows_equal(a, b):
    return len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))


def _onehot(T, K, seed, none_frac=0.1):
    rng = np.random.default_rng(seed)
    k = rng.integers(0, K, size=T).astype(np.int32)
    k[rng.random(T) < none_frac] = -1
    return k


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 5, 10, 11, 14, 17])
def test_dense_opening_bit_exact(atlas, n, mode):
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    A.set_challenge_mode(mode); orc.lib.orc_set_challenge_mode(mode)
    try:
        poly, point = orc.random_fr(1 << n, 3 + n), orc.random_fr(n, 4 + n)
        claim = orc.evaluate(poly, point)
        t_o = orc.new_transcript(b"dense_opening")
        rows_o, ch_o = OR.dense_opening(poly, point).prove(claim, t_o)
        inst = I.dense_opening(A.MultilinearPolynomial.from_fr(poly), point)
        assert inst.num_rounds() == n and inst.degree() == 2
        t_g = A.Blake2bTranscript(b"dense_opening")
        rows_g, ch_g = inst.prove(claim, t_g)
        assert ch_g == ch_o and _rows_equal(rows_g, rows_o) and t_g.state == t_o.state_bytes()
        # final claim = P(r_sumcheck) (HighToLow: challenges in order are the big-endian point)
        fin = inst.final_claims()
        assert np.array_equal(fin[0], orc.evaluate(poly, np.ascontiguousarray(orc.challenges_to_fr(ch_g))))
        inst.free()
    finally:
        A.set_challenge_mode(0); orc.lib.orc_set_challenge_mode(0)


@pytest.mark.parametrize("n", [3, 12])
def test_dense_opening_i32_polynomial(atlas, n):
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    zi = np.random.default_rng(n).integers(-(1 << 20), 1 << 20, size=1 << n, dtype=np.int32)
    zf = orc.fr_array(1 << n)
    orc.lib.orc_i32_to_fr(zi.ctypes.data_as(orc.i32p), C.c_size_t(1 << n), orc._p(zf))
    point = orc.random_fr(n, 9)
    claim = orc.evaluate(zf, point)
    t_o = orc.new_transcript(b"dense_i32")
    rows_o, ch_o = OR.dense_opening(zf, point).prove(claim, t_o)
    inst = I.dense_opening(A.MultilinearPolynomial.from_i32(zi), point)
    t_g = A.Blake2bTranscript(b"dense_i32")
    rows_g, ch_g = inst.prove(claim, t_g)
    assert ch_g == ch_o and _rows_equal(rows_g, rows_o) and t_g.state == t_o.state_bytes()
    inst.free()


@pytest.mark.parametrize("log_K,log_T", [(1, 1), (4, 3), (4, 10), (4, 13), (8, 9), (2, 16)])
def test_onehot_opening_bit_exact(atlas, log_K, log_T):
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    K, T = 1 << log_K, 1 << log_T
    idx = _onehot(T, K, 7 * log_K + log_T)
    r_address, r_cycle = orc.random_fr(log_K, 5), orc.random_fr(log_T, 6)
    # claim = OneHotPolynomial::evaluate (one_hot_polynomial.rs:65-90)
    Fa = orc.eq_evals(r_address)
    vec = np.stack([Fa[k] if k >= 0 else np.zeros(4, dtype=np.uint64) for k in idx])
    claim = orc.evaluate(vec, r_cycle)
    t_o = orc.new_transcript(b"onehot_opening")
    rows_o, ch_o = OR.onehot_opening(idx, log_K, r_address, r_cycle).prove(claim, t_o)
    inst = I.onehot_opening(idx, log_K, r_address, r_cycle)
    assert inst.num_rounds() == log_K + log_T
    t_g = A.Blake2bTranscript(b"onehot_opening")
    rows_g, ch_g = inst.prove(claim, t_g)
    assert ch_g == ch_o and _rows_equal(rows_g, rows_o) and t_g.state == t_o.state_bytes()
    # final claim = the one-hot polynomial at r_sumcheck
    rs = orc.challenges_to_fr(ch_g)
    Fs = orc.eq_evals(np.ascontiguousarray(rs[:log_K]))
    vec = np.stack([Fs[k] if k >= 0 else np.zeros(4, dtype=np.uint64) for k in idx])
    assert np.array_equal(inst.final_claims()[0], orc.evaluate(vec, np.ascontiguousarray(rs[log_K:])))
    inst.free()


def test_batched_opening_reduction(atlas):
    """prove_batch_opening_sumcheck shape: dense and one-hot openings of different sizes in one
    BatchedSumcheck (poly/opening_proof.rs:447-532)."""
    from oracle import orc, orc_ra as OR, orc_batched as OB
    from jolt_atlas_amd import instances as I
    A = atlas
    n1, n2, log_K, log_T = 12, 9, 4, 10
    p1, pt1 = orc.random_fr(1 << n1, 1), orc.random_fr(n1, 2)
    p2, pt2 = orc.random_fr(1 << n2, 3), orc.random_fr(n2, 4)
    idx = _onehot(1 << log_T, 1 << log_K, 5)
    ra, rc = orc.random_fr(log_K, 6), orc.random_fr(log_T, 7)
    Fa = orc.eq_evals(ra)
    vec = np.stack([Fa[k] if k >= 0 else np.zeros(4, dtype=np.uint64) for k in idx])
    claims = [orc.evaluate(p1, pt1), orc.evaluate(vec, rc), orc.evaluate(p2, pt2)]
    o = [OB.ra_instance(OR.dense_opening(p1, pt1), claims[0]), OB.ra_instance(OR.onehot_opening(idx, log_K, ra, rc), claims[1]),
         OB.ra_instance(OR.dense_opening(p2, pt2), claims[2])]
    t_o = orc.new_transcript(b"opening_reduction")
    rows_o, ch_o, _ = OB.batched_prove(o, t_o)
    P = A.MultilinearPolynomial.from_fr
    gi = [I.dense_opening(P(p1), pt1), I.onehot_opening(idx, log_K, ra, rc), I.dense_opening(P(p2), pt2)]
    t_g = A.Blake2bTranscript(b"opening_reduction")
    rows_g, ch_g = A.BatchedSumcheck.prove(gi, claims, t_g)
    assert ch_g == ch_o and _rows_equal(rows_g, rows_o) and t_g.state == t_o.state_bytes()
    for x in gi:
        x.free()
