"""GPU parity: EQ tables, MLE evaluation and the Gruen split-eq Mul sumcheck vs the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [0, 1, 5, 11, 12, 13, 16])
def test_eq_evals(atlas, n):
    from oracle import orc
    r = orc.random_fr(n, 40 + n) if n else np.zeros((0, 4), dtype=np.uint64)
    t = atlas.EqPolynomial.evals(r)
    want = orc.eq_evals(r) if n else orc.from_ints([1])
    assert np.array_equal(t.to_host(), want)
    t.free()
    if n == 5:
        sc = orc.random_fr(1, 3)[0]
        t = atlas.EqPolynomial.evals(r, scaling=sc)
        got = t.to_host()
        assert all(np.array_equal(got[i], orc.fr_mul_arr(want[i], sc)) for i in range(32))
        t.free()


@pytest.mark.parametrize("n", [1, 2, 7, 12, 15])
def test_evaluate(atlas, n):
    from oracle import orc
    z = orc.random_fr(1 << n, 60 + n)
    r = orc.random_fr(n, 61 + n)
    p = atlas.MultilinearPolynomial.from_fr(z)
    assert np.array_equal(atlas.evaluate(p, r), orc.evaluate(z, r))
    assert np.array_equal(p.to_host(), z)
    p.free()
    zi = np.random.default_rng(n).integers(-(1 << 20), 1 << 20, size=1 << n, dtype=np.int32)
    zf = orc.fr_array(1 << n)
    orc.lib.orc_i32_to_fr(zi.ctypes.data_as(orc.i32p), C.c_size_t(1 << n), orc._p(zf))
    p = atlas.MultilinearPolynomial.from_i32(zi)
    assert np.array_equal(atlas.evaluate(p, r), orc.evaluate(zf, r))
    p.free()


@pytest.mark.parametrize("n", [0, 3, 13])
def test_batch_evaluate(atlas, n):
    """Mixed i32 / Fr polynomials at one point, one call."""
    from oracle import orc
    r = orc.random_fr(n, 31 + n) if n else np.zeros((0, 4), dtype=np.uint64)
    polys, want = [], []
    for k in range(5):
        if k % 2:
            zi = np.random.default_rng(n * 7 + k).integers(-(1 << 30), 1 << 30, size=1 << n, dtype=np.int32)
            z = orc.fr_array(1 << n)
            orc.lib.orc_i32_to_fr(zi.ctypes.data_as(orc.i32p), C.c_size_t(1 << n), orc._p(z))
            polys.append(atlas.MultilinearPolynomial.from_i32(zi))
        else:
            z = orc.random_fr(1 << n, 90 + n + k)
            polys.append(atlas.MultilinearPolynomial.from_fr(z))
        want.append(orc.evaluate(z, r) if n else z[0])
    got = atlas.batch_evaluate(polys, r)
    for k in range(5):
        assert np.array_equal(got[k], want[k])
        assert np.array_equal(atlas.evaluate(polys[k], r), want[k])
        polys[k].free()
    with pytest.raises(atlas.AtlasError):
        atlas.batch_evaluate([atlas.MultilinearPolynomial.from_fr(orc.random_fr(2, 1))], orc.random_fr(2, 2))


@pytest.mark.parametrize("n", [1, 2, 3, 5, 9, 10, 11, 12, 13, 15])
def test_mul_sumcheck_bit_exact(atlas, n):
    """The device accumulates q(1) directly where the reference divides (split_eq_poly.rs:410);
    proofs must still be byte-identical."""
    from oracle import orc
    A = atlas
    L = orc.random_fr(1 << n, 700 + n); R = orc.random_fr(1 << n, 800 + n)
    w = orc.random_fr(n, 900 + n)
    claim = orc.mul_claim(L, R, w)
    t_o = orc.new_transcript(b"mul_sc")
    proof_o, ch_o, fin_o = orc.sumcheck_mul_prove(L, R, w, claim, t_o)
    prover = A.MulProver(A.MultilinearPolynomial.from_fr(L), A.MultilinearPolynomial.from_fr(R), w)
    assert np.array_equal(prover.input_claim(), claim[0])
    t_g = A.Blake2bTranscript(b"mul_sc")
    proof_g, ch_g, fin_g = prover.prove(claim[0], t_g)
    prover.free()
    assert ch_g == ch_o
    assert np.array_equal(proof_g, proof_o)
    assert np.array_equal(fin_g, fin_o)
    assert t_g.state == t_o.state_bytes() and t_g.n_rounds == t_o.n_rounds
    # final claim relation: s_n(r_n) = eq(w, r) * L(r) * R(r)
    tv = orc.new_transcript(b"mul_sc")
    orc.lib.orc_transcript_append_scalar(C.byref(tv), orc._p(claim))
    e, _ = orc.sumcheck_verify(proof_g, claim, tv)
    assert np.array_equal(e[0], orc.fr_mul_arr(orc.fr_mul_arr(fin_g[0], fin_g[1]), fin_g[2]))


@pytest.mark.parametrize("n", [4, 12])
def test_mul_sumcheck_i32_operands(atlas, n):
    from oracle import orc
    A = atlas
    rng = np.random.default_rng(n)
    Li = rng.integers(-(1 << 14), 1 << 14, size=1 << n, dtype=np.int32)
    Ri = rng.integers(-(1 << 14), 1 << 14, size=1 << n, dtype=np.int32)
    L = orc.fr_array(1 << n); R = orc.fr_array(1 << n)
    orc.lib.orc_i32_to_fr(Li.ctypes.data_as(orc.i32p), C.c_size_t(1 << n), orc._p(L))
    orc.lib.orc_i32_to_fr(Ri.ctypes.data_as(orc.i32p), C.c_size_t(1 << n), orc._p(R))
    w = orc.random_fr(n, 5)
    claim = orc.mul_claim(L, R, w)
    t_o = orc.new_transcript(b"mul_sc")
    proof_o, ch_o, fin_o = orc.sumcheck_mul_prove(L, R, w, claim, t_o)
    prover = A.MulProver(A.MultilinearPolynomial.from_i32(Li), A.MultilinearPolynomial.from_i32(Ri), w)
    t_g = A.Blake2bTranscript(b"mul_sc")
    proof_g, ch_g, fin_g = prover.prove(claim[0], t_g)
    prover.free()
    assert ch_g == ch_o and np.array_equal(proof_g, proof_o) and np.array_equal(fin_g, fin_o)
    assert t_g.state == t_o.state_bytes()
