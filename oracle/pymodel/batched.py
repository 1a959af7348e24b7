"""BatchedSumcheck::prove / verify, plain-int model (second, independent restatement used to
pin oracle/sumcheck.c's orc_batched_prove).

TEST INFRASTRUCTURE ONLY (oracle).  Restates joltworks/src/subprotocols/sumcheck.rs:30-262 and
the UniPoly arithmetic it leans on (unipoly.rs:39-52 from_coeff trimming, :401-413 AddAssign,
:454-461 Mul<F>).  MulModel is the MulProver instance (jolt-atlas-core/src/onnx_proof/ops/
mul.rs:125-185) WITHOUT the Gruen split: the round polynomial sum_x eq(w,(x,X)) L R is computed
from the full eq table, which must give the same coefficients as gruen_poly_deg_3.
"""
from .field import FR, challenge_to_fr
from . import poly as P


class MulModel:
    def __init__(self, left, right, w):
        self.left, self.right = list(left), list(right)
        self.eq = P.eq_evals(w)

    def degree(self):
        return 3

    def num_rounds(self):
        return len(self.left).bit_length() - 1

    def compute_message(self, rnd, previous_claim):
        e = [0, 0, 0]
        L = P.LOW_TO_HIGH
        for i in range(len(self.left) // 2):
            l = P.sumcheck_evals(self.left, i, 3, L)
            r = P.sumcheck_evals(self.right, i, 3, L)
            q = P.sumcheck_evals(self.eq, i, 3, L)
            for k in range(3):
                e[k] = (e[k] + l[k] * r[k] * q[k]) % FR
        return P.unipoly_from_evals_and_hint(previous_claim, e)

    def ingest_challenge(self, r, rnd):
        L = P.LOW_TO_HIGH
        self.left, self.right, self.eq = P.bind(self.left, r, L), P.bind(self.right, r, L), P.bind(self.eq, r, L)

    def finals(self):
        return self.left[0], self.right[0], self.eq[0]


def _from_coeff(c):
    c = list(c)
    while c and c[-1] == 0:
        c.pop()
    return c or [0]


def prove(instances, input_claims, transcript):
    """Returns (compressed rows, raw challenges, batching coeffs)."""
    max_rounds = max(s.num_rounds() for s in instances)
    for c in input_claims:
        transcript.append_scalar(c)
    coeffs = transcript.challenge_vector(len(instances))
    claims = [c * pow(2, max_rounds - s.num_rounds(), FR) % FR for s, c in zip(instances, input_claims)]
    rows, raw = [], []
    for rnd in range(max_rounds):
        remaining = max_rounds - rnd
        polys = []
        for s, c, prev in zip(instances, input_claims, claims):
            n = s._n
            if remaining > n:
                polys.append(_from_coeff([c * pow(2, remaining - n - 1, FR) % FR]))
            else:
                polys.append(s.compute_message(rnd - (max_rounds - n), prev))
        batched = _from_coeff([])
        for p, co in zip(polys, coeffs):
            t = _from_coeff([x * co % FR for x in p])
            for k, x in enumerate(t):
                if k < len(batched):
                    batched[k] = (batched[k] + x) % FR
                else:
                    batched.append(x)
        cc = P.unipoly_compress(batched)
        P.append_compressed(transcript, cc)
        c128 = transcript.challenge_u128()
        r = challenge_to_fr(c128)
        raw.append(c128); rows.append(cc)
        claims = [P.unipoly_eval(p, r) for p in polys]
        for s in instances:
            if remaining <= s._n:
                s.ingest_challenge(r, rnd - (max_rounds - s._n))
    return rows, raw, coeffs


def prepare(instances):
    """num_rounds() is read before any binding (the reference's instances keep it in params)."""
    for s in instances:
        s._n = s.num_rounds()
    return instances


def verify(rows, input_claims, rounds, transcript):
    """BatchedSumcheck::verify up to the expected-output comparison: (e, r, coeffs)."""
    max_rounds = max(rounds)
    for c in input_claims:
        transcript.append_scalar(c)
    coeffs = transcript.challenge_vector(len(rounds))
    e = sum(c * pow(2, max_rounds - n, FR) % FR * co for c, n, co in zip(input_claims, rounds, coeffs)) % FR
    rs = []
    for cc in rows:
        P.append_compressed(transcript, cc)
        r = challenge_to_fr(transcript.challenge_u128())
        rs.append(r)
        e = P.compressed_eval_from_hint(cc, e, r)
    return e, rs, coeffs
