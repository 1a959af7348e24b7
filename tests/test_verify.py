"""The product's sumcheck verifier (csrc/verify.hip, host arithmetic: no GPU needed) against the oracle's verifier and
against the protocol's soundness relation: honest proofs of the oracle's provers are accepted with the same final claim,
challenges and transcript state; a tampered coefficient changes the final claim; a round polynomial above the degree
bound is rejected (ProofVerifyError::InvalidInputLength); BatchedSumcheck's output claim equals
sum coeff_i * expected_output_claim_i (sumcheck.rs:187-259, 653-686)."""
import numpy as np
import pytest

import jolt_atlas_amd as A
from oracle import orc, orc_batched as OB


@pytest.mark.parametrize("n", [1, 2, 5, 9])
def test_single_instance_verify_matches_oracle(n):
    L, R = orc.random_fr(1 << n, 10 + n), orc.random_fr(1 << n, 20 + n)
    claim = orc.dot_claim(L, R)
    t_p = orc.new_transcript(b"verify")
    proof, ch_p, fin = orc.sumcheck_dot_prove(L, R, claim, t_p)
    # the verifier starts from the same transcript as the prover: Sumcheck::prove appends the input claim first
    # (sumcheck.rs:573-574), the caller of verify does the same
    t_o = orc.new_transcript(b"verify"); orc.lib.orc_transcript_append_scalar(orc.C.byref(t_o), orc._p(claim))
    e_o, ch_o = orc.sumcheck_verify(proof, claim, t_o)
    t_g = A.Blake2bTranscript(b"verify"); t_g.append_scalar(claim[0])
    e_g, ch_g = A.Sumcheck.verify(proof, claim[0], t_g, 2)
    assert ch_g == ch_o == ch_p
    assert np.array_equal(e_g, e_o.reshape(4))
    assert t_g.state == t_o.state_bytes() == t_p.state_bytes()
    # the oracle check the reference leaves to the caller: e == L(r) * R(r)
    assert np.array_equal(e_g, orc.fr_mul_arr(fin[0], fin[1]))
    # a tampered coefficient is not rejected by verify itself, but it no longer meets the oracle
    bad = proof.copy(); bad[0, 0, 0] ^= np.uint64(1)
    t_b = A.Blake2bTranscript(b"verify"); t_b.append_scalar(claim[0])
    e_b, _ = A.Sumcheck.verify(bad, claim[0], t_b, 2)
    assert not np.array_equal(e_b, e_g)
    # degree bound
    with pytest.raises(A.AtlasError, match="InvalidInputLength"):
        A.Sumcheck.verify(proof, claim[0], A.Blake2bTranscript(b"verify"), 1)


def test_batched_verify_matches_oracle_and_accepts():
    shapes = [(6, None), (4, None), (6, "mul")]
    insts, claims, rounds, degrees = [], [], [], []
    for j, (n, kind) in enumerate(shapes):
        L, R = orc.random_fr(1 << n, 100 + j), orc.random_fr(1 << n, 200 + j)
        if kind == "mul":
            w = orc.random_fr(n, 300 + j)
            c = orc.mul_claim(L, R, w)
            insts.append(OB.mul_instance(L, R, w, c)); degrees.append(3)
        else:
            c = orc.dot_claim(L, R)
            insts.append(OB.dot_instance(L, R, c)); degrees.append(2)
        claims.append(np.asarray(c).reshape(4)); rounds.append(n)
    t_p = orc.new_transcript(b"batched")
    rows, ch_p, finals = OB.batched_prove(insts, t_p)
    t_o = orc.new_transcript(b"batched")
    e_o, co_o, ch_o = OB.batched_verify(rows, claims, rounds, t_o)
    t_g = A.Blake2bTranscript(b"batched")
    e_g, ch_g, co_g = A.BatchedSumcheck.verify(rows, claims, rounds, degrees, t_g)
    assert ch_g == ch_o == ch_p
    assert np.array_equal(e_g, e_o) and np.array_equal(co_g, co_o)
    assert t_g.state == t_o.state_bytes()
    # expected output claims: dot = L(r) R(r); Mul = eq(w, r) L(r) R(r) with r the instance's slice (LowToHigh binds: reversed)
    expected = []
    for (n, kind), inst, fin in zip(shapes, insts, finals):
        v = orc.fr_mul_arr(fin[0], fin[1])
        if kind == "mul":
            v = orc.fr_mul_arr(v, fin[2])            # the oracle's Mul prover reports eq(w, r) as its third final claim
        expected.append(v)
    A.BatchedSumcheck.check(co_g, expected, e_g)
    wrong = [expected[0], expected[1], orc.fr_add_arr(expected[2], orc.from_ints([1])[0])]
    with pytest.raises(A.AtlasError, match="SumcheckVerificationError"):
        A.BatchedSumcheck.check(co_g, wrong, e_g)
    # a proof with the wrong number of rounds
    with pytest.raises(A.AtlasError):
        A.BatchedSumcheck.verify(rows[:-1], claims, rounds, degrees, A.Blake2bTranscript(b"batched"))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 12, 16])
def test_device_proofs_are_accepted(atlas, n):
    """A proof made by the device prover is accepted by the product's verifier, and its final claim meets the oracle
    e = L(r) R(r) computed from the prover's final claims."""
    L, R = orc.random_fr(1 << n, 31 + n), orc.random_fr(1 << n, 41 + n)
    p = atlas.EinsumDotProver(atlas.MultilinearPolynomial.from_fr(L), atlas.MultilinearPolynomial.from_fr(R))
    claim = p.input_claim()
    t = atlas.Blake2bTranscript(b"accept")
    proof, ch, fin = atlas.Sumcheck.prove(p, claim, t, n)
    p.free()
    tv = atlas.Blake2bTranscript(b"accept"); tv.append_scalar(claim)
    e, chv = atlas.Sumcheck.verify(proof, claim, tv, 2)
    assert chv == ch and tv.state == t.state
    assert np.array_equal(e, orc.fr_mul_arr(fin[0], fin[1]))


# ---------------------------------------------------------------- pairing + HyperKZG::verify (host arithmetic: no GPU needed)
def _mont_to_int(limbs, modulus):
    return sum(int(x) << (64 * i) for i, x in enumerate(limbs)) * pow(1 << 256, -1, modulus) % modulus


def _g2_ints(p):
    from oracle.pymodel import pairing as PR
    if int(p["infinity"]):
        return None
    return tuple(tuple(_mont_to_int(p[c][k], PR.P) for k in range(2)) for c in ("x", "y"))


def _g1_ints(p):
    from oracle.pymodel import pairing as PR
    return None if int(p["inf"] if "inf" in p.dtype.names else p["infinity"]) else (_mont_to_int(p["x"], PR.P), _mont_to_int(p["y"], PR.P))


def test_g2_and_pairing_match_the_integer_model():
    """atlas_g2_mul / atlas_pairing_check against oracle/pymodel/pairing.py: the same G2 multiples, bilinearity e(aG, bH) e(-abG, H) = 1,
    rejection of a wrong exponent, points at infinity"""
    from oracle.pymodel import pairing as PR
    Hg = A.g2_generator()
    assert _g2_ints(Hg) == PR.G2_GEN and PR.g2_on_curve(_g2_ints(Hg))
    for seed in (1, 2):
        a, b = orc.random_fr(1, 100 + seed)[0], orc.random_fr(1, 200 + seed)[0]
        ai, bi = orc.to_ints(a.reshape(1, 4))[0], orc.to_ints(b.reshape(1, 4))[0]
        bH = A.g2_mul(Hg, b)
        assert _g2_ints(bH) == PR.g2_mul(PR.G2_GEN, bi)
        aG = orc.g1_mul_generator(a)
        abG = orc.g1_mul_generator(orc.fr_mul_arr(a, b))
        neg = abG.copy(); neg["y"] = orc.from_ints([(PR.P - _g1_ints(abG)[1]) % PR.P])[0] if False else neg["y"]
        # -abG through the Fq image: y -> p - y (Montgomery form of the negated integer)
        yi = (PR.P - _g1_ints(abG)[1]) % PR.P
        ym = yi * (1 << 256) % PR.P
        neg["y"] = np.array([(ym >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
        g1 = np.zeros(2, dtype=A.G1_DTYPE)
        for k, src in enumerate((aG, neg)):
            g1[k]["x"] = src["x"]; g1[k]["y"] = src["y"]; g1[k]["infinity"] = 0
        assert A.pairing_check(g1, np.array([bH, Hg], dtype=A.G2_DTYPE))
        assert PR.multi_pairing_is_one([(_g1_ints(aG), _g2_ints(bH)), ((_g1_ints(abG)[0], yi), PR.G2_GEN)])
        wrong = A.g2_mul(Hg, orc.fr_add_arr(b, orc.from_ints([1])[0]))
        assert not A.pairing_check(g1, np.array([wrong, Hg], dtype=A.G2_DTYPE))
    inf1 = np.zeros(1, dtype=A.G1_DTYPE); inf1[0]["infinity"] = 1
    assert A.pairing_check(inf1, np.array([Hg], dtype=A.G2_DTYPE))


def _as_atlas_g1(p):
    o = np.zeros(1, dtype=A.G1_DTYPE)[0]
    o["x"] = p["x"]; o["y"] = p["y"]; o["infinity"] = p["inf"] if "inf" in p.dtype.names else p["infinity"]
    return o


@pytest.mark.parametrize("ell", [1, 2, 4, 6])
def test_hyperkzg_verify_accepts_and_rejects(ell):
    """HyperKZG::verify through the pairing on proofs of the oracle's HyperKZG::open: accepted, transcript in step with the prover's,
    the same decisions as the oracle's trapdoor check; a wrong evaluation, a tampered v, a tampered witness commitment, a wrong point and a
    verifier key for another trapdoor are rejected"""
    tau = orc.random_fr(1, 0x51250001)[0]
    srs = orc.srs_powers(tau, 1 << ell)
    poly = orc.random_fr(1 << ell, 7 + ell)
    pt = [int(x) for x in np.random.default_rng(ell).integers(1, 1 << 62, size=ell)]
    pt = [(p << 60) | 12345 + i for i, p in enumerate(pt)]
    Cm = orc.msm(srs, poly)
    y = orc.evaluate(poly, orc.challenges_to_fr(pt))
    t = orc.new_transcript(b"TestEval")
    com, w, v = orc.hyperkzg_open(srs, poly, pt, t)
    vk = A.HyperKZG.vk_from_trapdoor(tau, _as_atlas_g1(srs[0]))
    g1 = lambda arr: np.array([_as_atlas_g1(p) for p in arr], dtype=A.G1_DTYPE)
    tv = A.Blake2bTranscript(b"TestEval")
    assert A.HyperKZG.verify(vk, _as_atlas_g1(Cm), pt, y, g1(com), g1(w), v, tv)
    assert tv.state == t.state_bytes()
    assert orc.hyperkzg_verify_trapdoor(srs, tau, Cm, pt, y, com, w, v, orc.new_transcript(b"TestEval"))
    bad_y = orc.fr_add_arr(y, orc.from_ints([1])[0])
    assert not A.HyperKZG.verify(vk, _as_atlas_g1(Cm), pt, bad_y, g1(com), g1(w), v, A.Blake2bTranscript(b"TestEval"))
    vbad = v.copy(); vbad[0, 0] = orc.fr_add_arr(vbad[0, 0], orc.from_ints([1])[0])
    assert not A.HyperKZG.verify(vk, _as_atlas_g1(Cm), pt, y, g1(com), g1(w), vbad, A.Blake2bTranscript(b"TestEval"))
    wbad = g1(w); wbad[1] = _as_atlas_g1(srs[0])                  # (w[1] = w[0] would be no tampering at ell = 1: the three quotients of a linear polynomial coincide)
    assert not A.HyperKZG.verify(vk, _as_atlas_g1(Cm), pt, y, g1(com), wbad, v, A.Blake2bTranscript(b"TestEval"))
    pbad = list(pt); pbad[0] ^= 1 << 70
    assert not A.HyperKZG.verify(vk, _as_atlas_g1(Cm), pbad, y, g1(com), g1(w), v, A.Blake2bTranscript(b"TestEval"))
    vk2 = A.HyperKZG.vk_from_trapdoor(orc.fr_add_arr(tau, orc.from_ints([1])[0]), _as_atlas_g1(srs[0]))
    assert not A.HyperKZG.verify(vk2, _as_atlas_g1(Cm), pt, y, g1(com), g1(w), v, A.Blake2bTranscript(b"TestEval"))
