"""CPU: the C oracle's BatchedSumcheck restatement against the independent plain-int model
(oracle/pymodel/batched.py) and against the protocol's own soundness relation
(BatchedSumcheck::verify, sumcheck.rs:186-262)."""
import numpy as np
import pytest

from oracle import orc, orc_batched as OB
from oracle.pymodel import batched as PB, field as F, poly as P, sumcheck as PS
from oracle.pymodel.transcript import Blake2bTranscript


def _rand(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [int.from_bytes(rng.bytes(40), "little") % F.FR for _ in range(n)]


def _case(shapes, seed):
    """shapes: list of ("dot", n) | ("high", n, log_eq) | ("low", n, log_k) | ("mul", n)."""
    py, oc, claims = [], [], []
    for j, sh in enumerate(shapes):
        n = sh[1]
        L, R = _rand(1 << n, seed + 10 * j), _rand(1 << n, seed + 10 * j + 1)
        if sh[0] == "mul":
            w = _rand(n, seed + 10 * j + 2)
            E = P.eq_evals(w)
            c = sum(l * r * e for l, r, e in zip(L, R, E)) % F.FR
            py.append(PB.MulModel(L, R, w))
            oc.append(OB.mul_instance(orc.from_ints(L), orc.from_ints(R), orc.from_ints(w), orc.from_ints([c])))
        elif sh[0] == "dot":
            c = sum(l * r for l, r in zip(L, R)) % F.FR
            py.append(PS.DotProver(L, R))
            oc.append(OB.dot_instance(orc.from_ints(L), orc.from_ints(R), orc.from_ints([c])))
        elif sh[0] == "high":
            a = sh[2]; b = n - a
            E = _rand(1 << a, seed + 10 * j + 2)
            c = sum(l * r * E[i >> b] for i, (l, r) in enumerate(zip(L, R))) % F.FR
            py.append(PS.DotProver(L, R, E, ("high", a, b)))
            oc.append(OB.dot_instance(orc.from_ints(L), orc.from_ints(R), orc.from_ints([c]), orc.from_ints(E), 1, a, b))
        else:
            k = sh[2]; b = n - k
            E = _rand(1 << b, seed + 10 * j + 2)
            c = sum(l * r * E[i & ((1 << b) - 1)] for i, (l, r) in enumerate(zip(L, R))) % F.FR
            py.append(PS.DotProver(L, R, E, ("low", k, b)))
            oc.append(OB.dot_instance(orc.from_ints(L), orc.from_ints(R), orc.from_ints([c]), orc.from_ints(E), 2, k, b))
        claims.append(c)
    return py, oc, claims


CASES = [
    [("dot", 4)],
    [("dot", 5), ("dot", 3)],
    [("mul", 4), ("dot", 4)],
    [("dot", 2), ("mul", 5), ("high", 4, 2), ("low", 3, 1)],
    [("mul", 1), ("mul", 3), ("dot", 1)],
]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_batched_oracle_matches_python_model(ci, mode):
    shapes = CASES[ci]
    orc.lib.orc_set_challenge_mode(mode)
    old = F.CHALLENGE_MODE
    F.CHALLENGE_MODE = "mont" if mode == 0 else "canonical"
    try:
        py, oc, claims = _case(shapes, 100 * ci + 7)
        tp = Blake2bTranscript(b"batched")
        rows_p, raw_p, coeffs_p = PB.prove(PB.prepare(py), claims, tp)
        to = orc.new_transcript(b"batched")
        rows_o, raw_o, finals_o = OB.batched_prove(oc, to)
        assert raw_o == raw_p
        assert [orc.to_ints(r) for r in rows_o] == rows_p
        assert bytes(to.state) == tp.state
        # verifier side: e == sum coeff_i * expected_output_i  (sumcheck.rs:234-258)
        tv = orc.new_transcript(b"batched")
        e, co, raw_v = OB.batched_verify(rows_o, [orc.from_ints([c])[0] for c in claims], [s[1] for s in shapes], tv)
        assert raw_v == raw_o and orc.to_ints(co) == coeffs_p
        exp = 0
        for f, c in zip(finals_o, coeffs_p):
            l, r, q = orc.to_ints(f)
            exp = (exp + l * r * q * c) % F.FR
        assert orc.to_ints(e)[0] == exp
        # final claims = MLE evaluations at the instance's slice of the challenges
        rs = [F.challenge_to_fr(c) for c in raw_o]
        mx = max(s[1] for s in shapes)
        for sh, inst, f in zip(shapes, py, finals_o):
            sl = rs[mx - sh[1]:]
            if sh[0] == "mul":
                assert tuple(orc.to_ints(f)) == inst.finals()
            else:
                assert orc.to_ints(f)[0] == inst.left[0] and orc.to_ints(f)[1] == inst.right[0]
            assert len(sl) == sh[1]
    finally:
        orc.lib.orc_set_challenge_mode(0)
        F.CHALLENGE_MODE = old
