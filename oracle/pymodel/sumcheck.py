"""Sumcheck driver + the einsum dot-product instance, plain-int model.

TEST INFRASTRUCTURE ONLY (oracle).  Restates (paths under /root/reference):
  Sumcheck::prove / SumcheckInstanceProof::verify
        joltworks/src/subprotocols/sumcheck.rs:565-599,653-686
  EinsumDotProver::{compute_message,ingest_challenge}
        jolt-atlas-core/src/onnx_proof/ops/einsum/dot.rs:290-375
"""
from .field import FR
from . import poly as P


class DotProver:
    """claim = sum_h L(h) R(h) [EQ(h)] bound HighToLow.

    schedule: ("none",) | ("high", log_eq, low_bits) | ("low", log_k, log_b)
    (EqSchedule, dot.rs:70-95)."""

    def __init__(self, left, right, eq=None, schedule=("none",)):
        self.left = list(left)
        self.right = list(right)
        self.eq = None if eq is None else list(eq)
        self.schedule = schedule
        self.eq_bound_claim = None
        if schedule[0] == "high" and schedule[1] == 0:
            self.eq_bound_claim = self.eq[0]

    def degree(self):
        return 2 if self.schedule[0] == "none" else 3

    def num_rounds(self):
        return (len(self.left)).bit_length() - 1

    def compute_message(self, rnd, previous_claim):
        half = len(self.left) // 2
        H = P.HIGH_TO_LOW
        kind = self.schedule[0]
        if kind == "none":
            e = [0, 0]
            for i in range(half):
                l = P.sumcheck_evals(self.left, i, 2, H)
                r = P.sumcheck_evals(self.right, i, 2, H)
                e[0] = (e[0] + l[0] * r[0]) % FR
                e[1] = (e[1] + l[1] * r[1]) % FR
            return P.unipoly_from_evals_and_hint(previous_claim, e)
        e = [0, 0, 0]
        for i in range(half):
            l = P.sumcheck_evals(self.left, i, 3, H)
            r = P.sumcheck_evals(self.right, i, 3, H)
            if kind == "high":
                _, log_eq, low_bits = self.schedule
                if rnd < log_eq:
                    q = P.sumcheck_evals(self.eq, i >> low_bits, 3, H)
                else:
                    q = [self.eq_bound_claim] * 3
            else:
                _, log_k, log_b = self.schedule
                if rnd < log_k:
                    q = [self.eq[i & ((1 << log_b) - 1)]] * 3
                else:
                    q = P.sumcheck_evals(self.eq, i, 3, H)
            for k in range(3):
                e[k] = (e[k] + l[k] * r[k] * q[k]) % FR
        return P.unipoly_from_evals_and_hint(previous_claim, e)

    def ingest_challenge(self, r, rnd):
        H = P.HIGH_TO_LOW
        self.left = P.bind(self.left, r, H)
        self.right = P.bind(self.right, r, H)
        kind = self.schedule[0]
        if kind == "high":
            log_eq = self.schedule[1]
            if rnd < log_eq:
                self.eq = P.bind(self.eq, r, H)
                if rnd == log_eq - 1:
                    self.eq_bound_claim = self.eq[0]
        elif kind == "low":
            if rnd >= self.schedule[1]:
                self.eq = P.bind(self.eq, r, H)


def prove(instance, input_claim, transcript):
    """Sumcheck::prove (sumcheck.rs:565-599). Returns (compressed_polys, r, claims)."""
    transcript.append_scalar(input_claim)
    prev = input_claim
    rs, polys, raw = [], [], []
    for rnd in range(instance.num_rounds()):
        up = instance.compute_message(rnd, prev)
        cc = P.unipoly_compress(up)
        P.append_compressed(transcript, cc)
        c128 = transcript.challenge_u128()
        from .field import challenge_to_fr
        r = challenge_to_fr(c128)
        rs.append(r)
        raw.append(c128)
        prev = P.unipoly_eval(up, r)
        instance.ingest_challenge(r, rnd)
        polys.append(cc)
    return polys, rs, raw, prev


def verify(polys, claim, num_rounds, degree_bound, transcript):
    """SumcheckInstanceProof::verify (sumcheck.rs:653-686); caller appended the claim."""
    from .field import challenge_to_fr
    e = claim
    rs = []
    assert len(polys) == num_rounds
    for cc in polys:
        if len(cc) > degree_bound:
            raise ValueError("degree bound")
        P.append_compressed(transcript, cc)
        r = challenge_to_fr(transcript.challenge_u128())
        rs.append(r)
        e = P.compressed_eval_from_hint(cc, e, r)
    return e, rs


def serialize_proof(polys):
    """ark CanonicalSerialize of SumcheckInstanceProof = Vec<CompressedUniPoly>:
    u64 LE length, then each Vec<Fr> as u64 LE length + 32-byte LE scalars
    (SURVEY App. A.3)."""
    out = len(polys).to_bytes(8, "little")
    for cc in polys:
        out += len(cc).to_bytes(8, "little")
        for a in cc:
            out += int(a).to_bytes(32, "little")
    return out
