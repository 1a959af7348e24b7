"""GPU parity: BatchedSumcheck::prove (sumcheck.rs:30-184) over device instances of differing
round counts vs the oracle's restatement, plus the per-round Mul instance API."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(A, shapes, seed):
    from oracle import orc, orc_batched as OB
    gpu, oc, claims = [], [], []
    P = A.MultilinearPolynomial.from_fr
    for j, sh in enumerate(shapes):
        n = sh[1]
        L, R = orc.random_fr(1 << n, seed + 10 * j), orc.random_fr(1 << n, seed + 10 * j + 1)
        if sh[0] == "mul":
            w = orc.random_fr(n, seed + 10 * j + 2)
            c = orc.mul_claim(L, R, w)[0]
            gpu.append(A.MulProver(P(L), P(R), w)); oc.append(OB.mul_instance(L, R, w, c))
        elif sh[0] == "dot":
            c = orc.dot_claim(L, R)[0]
            gpu.append(A.EinsumDotProver(P(L), P(R))); oc.append(OB.dot_instance(L, R, c))
        elif sh[0] == "high":
            a = sh[2]; b = n - a
            E = orc.random_fr(1 << a, seed + 10 * j + 2)
            c = orc.dot_claim(L, R, E, 1, a, b)[0]
            gpu.append(A.EinsumDotProver(P(L), P(R), P(E), A.EQ_HIGH, a, b)); oc.append(OB.dot_instance(L, R, c, E, 1, a, b))
        else:
            k = sh[2]; b = n - k
            E = orc.random_fr(1 << b, seed + 10 * j + 2)
            c = orc.dot_claim(L, R, E, 2, k, b)[0]
            gpu.append(A.EinsumDotProver(P(L), P(R), P(E), A.EQ_LOW, k, b)); oc.append(OB.dot_instance(L, R, c, E, 2, k, b))
        claims.append(c)
    return gpu, oc, claims


CASES = [
    [("dot", 4)],
    [("dot", 12), ("dot", 7)],
    [("mul", 11), ("dot", 13)],
    [("dot", 2), ("mul", 12), ("high", 10, 4), ("low", 9, 3)],
    [("mul", 1), ("mul", 3), ("dot", 1)],
    [("mul", 14), ("mul", 14), ("dot", 14), ("high", 14, 7)],
]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_batched_sumcheck_bit_exact(atlas, ci, mode):
    from oracle import orc, orc_batched as OB
    A = atlas
    shapes = CASES[ci]
    A.set_challenge_mode(mode); orc.lib.orc_set_challenge_mode(mode)
    try:
        gpu, oc, claims = _build(A, shapes, 1000 * ci + 3)
        t_o = orc.new_transcript(b"batched")
        rows_o, ch_o, fin_o = OB.batched_prove(oc, t_o)
        t_g = A.Blake2bTranscript(b"batched")
        rows_g, ch_g = A.BatchedSumcheck.prove(gpu, claims, t_g)
        assert ch_g == ch_o
        assert len(rows_g) == len(rows_o)
        for a, b in zip(rows_g, rows_o):
            assert np.array_equal(a, b)
        assert t_g.state == t_o.state_bytes() and t_g.n_rounds == t_o.n_rounds
        for p, f in zip(gpu, fin_o):
            assert np.array_equal(p.final_claims(), f)
            p.free()
        # verifier relation
        tv = orc.new_transcript(b"batched")
        e, co, _ = OB.batched_verify(rows_g, claims, [s[1] for s in shapes], tv)
        exp = orc.fr_array(1)[0]
        for f, c in zip(fin_o, co):
            term = orc.fr_mul_arr(orc.fr_mul_arr(orc.fr_mul_arr(f[0], f[1]), f[2]), c)
            exp = orc.fr_add_arr(exp, term)
        assert np.array_equal(e, exp)
    finally:
        A.set_challenge_mode(0); orc.lib.orc_set_challenge_mode(0)


@pytest.mark.parametrize("n", [1, 4, 11, 13])
def test_mul_round_api_matches_fused_driver(atlas, n):
    """compute_message / ingest_challenge stepped from the host give the proof the fused
    device-transcript driver gives."""
    from oracle import orc
    import ctypes as C
    A = atlas
    L = orc.random_fr(1 << n, 5 + n); R = orc.random_fr(1 << n, 6 + n); w = orc.random_fr(n, 7 + n)
    claim = orc.mul_claim(L, R, w)
    t_o = orc.new_transcript(b"mul_rounds")
    proof_o, ch_o, fin_o = orc.sumcheck_mul_prove(L, R, w, claim, t_o)
    P = A.MulProver(A.MultilinearPolynomial.from_fr(L), A.MultilinearPolynomial.from_fr(R), w)
    t = A.Blake2bTranscript(b"mul_rounds")
    t.append_scalar(claim[0])
    prev = claim[0]
    for rnd in range(n):
        co = P.compute_message(rnd, prev)
        assert len(co) == 4
        assert np.array_equal(co[[0, 2, 3]], proof_o[rnd])
        t.append_message(b"UniPoly_begin")
        for k in (0, 2, 3):
            t.append_scalar(co[k])
        t.append_message(b"UniPoly_end")
        r = t.challenge_u128()
        assert r == ch_o[rnd]
        rf = A.challenge_to_fr(r)
        acc = co[3]
        for k in (2, 1, 0):
            acc = orc.fr_add_arr(orc.fr_mul_arr(acc, rf), co[k])
        prev = acc
        P.ingest_challenge(r, rnd)
    assert np.array_equal(P.final_claims(), fin_o)
    assert t.state == t_o.state_bytes()
    P.free()
