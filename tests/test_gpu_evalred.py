"""GPU parity: N-to-1 evaluation reduction (evaluation_reduction.rs:91-147, 213-249) vs the oracle's
coefficient-folding restatement."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n,N", [(1, 2), (4, 2), (6, 3), (9, 4), (10, 2), (12, 3), (3, 6)])
def test_eval_reduction_bit_exact(atlas, n, N, mode):
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    A.set_challenge_mode(mode); orc.lib.orc_set_challenge_mode(mode)
    try:
        mle = orc.random_fr(1 << n, 11 + n)
        pts = orc.random_fr(N * n, 12 + n).reshape(N, n, 4)
        claims = np.stack([orc.evaluate(mle, pts[j]) for j in range(N)])
        t_o = orc.new_transcript(b"eval_reduction")
        h_o, r_o, c_o = OR.eval_reduction_prove(mle, pts, claims, t_o)
        p = A.MultilinearPolynomial.from_fr(mle)
        t_g = A.Blake2bTranscript(b"eval_reduction")
        h_g, r_g, c_g = I.eval_reduction_prove(p, pts, claims, t_g)
        assert np.array_equal(h_g, h_o) and np.array_equal(r_g, r_o) and np.array_equal(c_g, c_o)
        assert t_g.state == t_o.state_bytes() and t_g.n_rounds == t_o.n_rounds
        assert np.array_equal(p.to_host(), mle)                      # witness not consumed
        assert np.array_equal(orc.evaluate(mle, r_g), c_g)           # reduced claim holds
        p.free()
    finally:
        A.set_challenge_mode(0); orc.lib.orc_set_challenge_mode(0)


def test_eval_reduction_i32_witness_structured_points_and_short_path(atlas):
    """EvalReductionWitness::from_tensor: zero-padded i32 tensor; points sharing coordinates."""
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    n, N = 8, 3
    zi = np.zeros(1 << n, dtype=np.int32)
    zi[:150] = np.random.default_rng(1).integers(-(1 << 14), 1 << 14, size=150, dtype=np.int32)
    zf = orc.fr_array(1 << n)
    orc.lib.orc_i32_to_fr(zi.ctypes.data_as(orc.i32p), C.c_size_t(1 << n), orc._p(zf))
    pts = np.repeat(orc.random_fr(n, 2)[None], N, axis=0).copy()
    pts[1, 3] = orc.random_fr(1, 3)[0]; pts[2, 3] = orc.random_fr(1, 4)[0]; pts[2, 5] = orc.random_fr(1, 5)[0]
    claims = np.stack([orc.evaluate(zf, pts[j]) for j in range(N)])
    t_o = orc.new_transcript(b"er")
    h_o, r_o, c_o = OR.eval_reduction_prove(zf, pts, claims, t_o)
    p = A.MultilinearPolynomial.from_i32(zi)
    t_g = A.Blake2bTranscript(b"er")
    h_g, r_g, c_g = I.eval_reduction_prove(p, pts, claims, t_g)
    assert len(h_g) <= 2 * (N - 1) + 1
    assert np.array_equal(h_g, h_o) and np.array_equal(r_g, r_o) and np.array_equal(c_g, c_o) and t_g.state == t_o.state_bytes()
    # one opening: nothing to reduce, transcript untouched
    t1 = A.Blake2bTranscript(b"er1")
    h1, r1, c1 = I.eval_reduction_prove(p, pts[:1], claims[:1], t1)
    assert np.array_equal(h1, claims[:1]) and np.array_equal(r1, pts[0]) and t1.n_rounds == 0
    with pytest.raises(A.AtlasError):
        I.eval_reduction_prove(p, pts[:, :n - 1], claims, t1)
    p.free()
