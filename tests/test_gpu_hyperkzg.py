"""GPU parity: HyperKZG::open through the C-ABI against the oracle — every commitment,
witness point, evaluation and the transcript state must be identical; the oracle's verifier
(trapdoor form of the pairing check) must accept the GPU proof."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAU_SEED = 0x51250001


def _same_points(a, b):
    from oracle import orc
    return len(a) == len(b) and all(orc.g1_eq(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("ell", [1, 2, 3, 5, 8, 12, 13, 14])
def test_open_matches_oracle_and_verifies(atlas, ell):
    from oracle import orc
    A = atlas
    n = 1 << ell
    tau = orc.random_fr(1, TAU_SEED)[0]
    srs_g = A.SRS.generate(tau, n)
    srs_o = srs_g.download()
    if ell <= 10:
        ref = orc.srs_powers(tau, n)
        assert np.array_equal(srs_o["x"], ref["x"]) and np.array_equal(srs_o["y"], ref["y"])
    poly = orc.random_fr(n, 500 + ell)
    rng = np.random.default_rng(ell)
    pt = [int(rng.integers(0, 1 << 62)) << 66 | int(rng.integers(0, 1 << 62)) for _ in range(ell)]
    t_o = orc.new_transcript(b"TestEval")
    com_o, w_o, v_o = orc.hyperkzg_open(srs_o, poly, pt, t_o)
    p = A.MultilinearPolynomial.from_fr(poly)
    t_g = A.Blake2bTranscript(b"TestEval")
    com_g, w_g, v_g = A.HyperKZG.open(srs_g, p, pt, t_g)
    assert np.array_equal(v_g, v_o)
    assert _same_points(com_g, com_o) and _same_points(w_g, w_o)
    assert t_g.state == t_o.state_bytes() and t_g.n_rounds == t_o.n_rounds
    assert np.array_equal(p.to_host(), poly)            # poly is not consumed
    C_ = A.HyperKZG.commit(srs_g, p)
    assert orc.g1_eq(C_, orc.msm(srs_o, poly))
    y = orc.evaluate(poly, orc.challenges_to_fr(pt))
    assert orc.hyperkzg_verify_trapdoor(srs_o, tau, C_, pt, y, com_g, w_g, v_g, orc.new_transcript(b"TestEval"))
    # HyperKZG::verify through the pairing (no trapdoor): accept, the transcript of the prover, a wrong evaluation rejected
    vk = A.HyperKZG.vk_from_trapdoor(tau, srs_g.download(0, 1)[0])
    tv = A.Blake2bTranscript(b"TestEval")
    assert A.HyperKZG.verify(vk, C_, pt, y, com_g, w_g, v_g, tv) and tv.state == t_g.state
    assert not A.HyperKZG.verify(vk, C_, pt, orc.fr_add_arr(y, orc.from_ints([1])[0]), com_g, w_g, v_g, A.Blake2bTranscript(b"TestEval"))
    p.free(); srs_g.free()


@pytest.mark.parametrize("case", range(4))
def test_open_matches_golden_fixture(atlas, case):
    """Same proof as the Python-model fixture (tests/golden/hyperkzg.json)."""
    from oracle import orc
    from oracle.pymodel import field as F
    A = atlas
    v = json.load(open(os.path.join(G, "hyperkzg.json")))[case]
    ell = v["ell"]
    tau = orc.from_ints([int(v["tau"], 16)])[0]
    srs = A.SRS.generate(tau, 1 << ell)
    p = A.MultilinearPolynomial.from_fr(orc.from_ints([int(x, 16) for x in v["poly"]]))
    t = A.Blake2bTranscript(b"TestEval")
    com, w, vv = A.HyperKZG.open(srs, p, [int(c, 16) for c in v["point_c128"]], t)
    assert t.state.hex() == v["final_state"]
    conv = lambda pt: (F.from_mont(F.from_limbs64(pt["x"]), F.FQ), F.from_mont(F.from_limbs64(pt["y"]), F.FQ))
    assert [conv(x) for x in com] == [(int(a, 16), int(b, 16)) for a, b in v["com"]]
    assert [conv(x) for x in w] == [(int(a, 16), int(b, 16)) for a, b in v["w"]]
    assert [[orc.to_ints(x)[0] for x in row] for row in vv] == [[int(x, 16) for x in row] for row in v["v"]]
    p.free(); srs.free()


def test_open_errors(atlas):
    from oracle import orc
    A = atlas
    srs = A.SRS.generate(orc.random_fr(1, 1)[0], 8)
    p = A.MultilinearPolynomial.from_fr(orc.random_fr(16, 2))
    with pytest.raises(A.AtlasError, match="KeyLengthError"):
        A.HyperKZG.open(srs, p, [1, 2, 3, 4], A.Blake2bTranscript(b"x"))
    with pytest.raises(A.AtlasError):
        A.HyperKZG.open(srs, p, [1, 2, 3], A.Blake2bTranscript(b"x"))       # 2^ell != len
    p.free(); srs.free()
