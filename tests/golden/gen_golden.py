#!/usr/bin/env python3
"""Generates tests/golden/*.json from the independent Python big-int model
(oracle/pymodel).  Run from the repo root:  python tests/golden/gen_golden.py

The reference (Rust) cannot be executed in this image and holds no known-answer vectors
for this path, so these fixtures pin the C oracle and the HIP path to a
second, independently written model plus hashlib's BLAKE2b — not to a reference run."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pymodel import curve as EC            # noqa: E402
from oracle.pymodel import field as F             # noqa: E402
from oracle.pymodel import hyperkzg as HK        # noqa: E402
from oracle.pymodel import poly as P              # noqa: E402
from oracle.pymodel import sumcheck as S          # noqa: E402
from oracle.pymodel import transcript as T        # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
rnd = random.Random(0xA71A5)


def hx(v):
    return "%064x" % v


def mont_hex(v, p=F.FR):
    return hx(F.to_mont(v % p, p))


def field_vectors():
    vs = []
    specials = [0, 1, 2, F.FR - 1, F.FR - 2, (1 << 128) - 1, 1 << 253]
    pairs = [(a, b) for a in specials for b in specials[:4]]
    pairs += [(rnd.randrange(F.FR), rnd.randrange(F.FR)) for _ in range(40)]
    for a, b in pairs:
        vs.append({"a": hx(a), "b": hx(b), "a_mont": mont_hex(a), "b_mont": mont_hex(b),
                   "mul_mont": mont_hex(a * b), "add_mont": mont_hex(a + b), "sub_mont": mont_hex(a - b),
                   "inv_a_mont": mont_hex(pow(a, -1, F.FR)) if a else None,
                   "a_le_bytes": F.fr_le_bytes(a).hex()})
    ch = []
    for _ in range(16):
        c = rnd.getrandbits(128)
        F.CHALLENGE_MODE = "mont"
        m0 = hx(F.to_mont(F.challenge_to_fr(c)))
        F.CHALLENGE_MODE = "plain"
        m1 = hx(F.to_mont(F.challenge_to_fr(c)))
        F.CHALLENGE_MODE = "mont"
        ch.append({"c128": "%032x" % c, "mont_limbs_mode0": m0, "mont_limbs_mode1": m1})
    return {"fr_modulus": hx(F.FR), "fq_modulus": hx(F.FQ), "pairs": vs, "challenges": ch,
            "small": [{"v": v, "mont": mont_hex(v)} for v in (-(2 ** 31), -5, -1, 0, 1, 7, 2 ** 31 - 1, 2 ** 63 - 1, -(2 ** 63))]}


def transcript_vectors():
    t = T.Blake2bTranscript(b"ONNXProof")
    ops = [("new", "ONNXProof", t.state.hex())]
    x = [rnd.randrange(F.FR) for _ in range(3)]
    t.append_message(b"hello"); ops.append(("append_message", "hello", t.state.hex()))
    t.append_u64(0xdeadbeef12345678); ops.append(("append_u64", "deadbeef12345678", t.state.hex()))
    t.append_scalar(x[0]); ops.append(("append_scalar", hx(x[0]), t.state.hex()))
    t.append_scalars(x); ops.append(("append_scalars", [hx(v) for v in x], t.state.hex()))
    t.append_bytes(b"\x01" * 200); ops.append(("append_bytes", ("01" * 200), t.state.hex()))
    c = t.challenge_u128(); ops.append(("challenge_u128", "%032x" % c, t.state.hex()))
    s = t.challenge_scalar(); ops.append(("challenge_scalar", hx(s), t.state.hex()))
    pt = EC.mul(EC.G, 12345)
    t.append_point(pt); ops.append(("append_point", [hx(pt[0]), hx(pt[1])], t.state.hex()))
    t.append_point(None); ops.append(("append_point", None, t.state.hex()))
    return {"ops": ops, "n_rounds": t.n_rounds}


def sumcheck_vectors():
    out = []
    for n, sched in [(4, ("none",)), (5, ("high", 2, 3)), (5, ("low", 2, 3)), (3, ("high", 0, 3)), (6, ("none",))]:
        L = [rnd.randrange(F.FR) for _ in range(1 << n)]
        R = [rnd.randrange(F.FR) for _ in range(1 << n)]
        eq = None
        if sched[0] == "high":
            eq = P.eq_evals([rnd.randrange(F.FR) for _ in range(sched[1])])
            claim = sum(L[i] * R[i] * eq[i >> sched[2]] for i in range(1 << n)) % F.FR
        elif sched[0] == "low":
            eq = P.eq_evals([rnd.randrange(F.FR) for _ in range(sched[2])])
            claim = sum(L[i] * R[i] * eq[i & ((1 << sched[2]) - 1)] for i in range(1 << n)) % F.FR
        else:
            claim = sum(a * b for a, b in zip(L, R)) % F.FR
        t = T.Blake2bTranscript(b"synthetic_sc")
        inst = S.DotProver(L, R, eq, sched)
        polys, rs, raw, final = S.prove(inst, claim, t)
        out.append({"n_vars": n, "schedule": list(sched), "left": [hx(v) for v in L], "right": [hx(v) for v in R],
                    "eq": None if eq is None else [hx(v) for v in eq], "claim": hx(claim),
                    "proof_bytes": S.serialize_proof(polys).hex(), "challenges": ["%032x" % c for c in raw],
                    "final_claim": hx(final), "left_final": hx(inst.left[0]), "right_final": hx(inst.right[0]),
                    "transcript_history": [s.hex() for s in t.state_history]})
    return out


def eq_vectors():
    r = [rnd.randrange(F.FR) for _ in range(5)]
    return {"r": [hx(v) for v in r], "evals": [hx(v) for v in P.eq_evals(r)]}


def curve_vectors():
    tau = rnd.randrange(F.FR)
    n = 12
    srs = EC.srs_powers(tau, n)
    sc = [rnd.randrange(F.FR) for _ in range(n)]
    sc[2], sc[5], sc[7] = 0, 1, F.FR - 1
    res = EC.msm(srs, sc)
    idx = [0, 3, 3, 7, 11]
    acc = None
    for i in idx:
        acc = EC.add(acc, srs[i])
    return {"tau": hx(tau), "srs": [[hx(p[0]), hx(p[1])] for p in srs], "scalars": [hx(v) for v in sc],
            "msm": [hx(res[0]), hx(res[1])], "two_G": [hx(v) for v in EC.add(EC.G, EC.G)],
            "sum_indexed": {"idx": idx, "point": [hx(acc[0]), hx(acc[1])]}}


def hyperkzg_vectors():
    out = []
    tau = rnd.randrange(F.FR)
    for ell in (1, 2, 3, 5):
        n = 1 << ell
        srs = EC.srs_powers(tau, n)
        poly = [rnd.randrange(F.FR) for _ in range(n)]
        pt = [rnd.getrandbits(128) for _ in range(ell)]
        C = HK.commit(srs, poly)
        y = P.evaluate(poly, [F.challenge_to_fr(c) for c in pt])
        t = T.Blake2bTranscript(b"TestEval")
        com, w, v = HK.open_(srs, poly, pt, t)
        tv = T.Blake2bTranscript(b"TestEval")
        assert HK.verify_trapdoor(srs, tau, C, pt, y, (com, w, v), tv) and tv.state == t.state
        ser_len = 8 + (ell - 1) * 32 + 8 + 3 * 32 + 8 + 3 * (8 + ell * 32)      # ark compressed sizes
        out.append({"ell": ell, "tau": hx(tau), "poly": [hx(c) for c in poly], "point_c128": ["%032x" % c for c in pt],
                    "commitment": [hx(C[0]), hx(C[1])], "eval": hx(y),
                    "com": [[hx(p[0]), hx(p[1])] for p in com], "w": [[hx(p[0]), hx(p[1])] for p in w],
                    "v": [[hx(x) for x in row] for row in v], "final_state": t.state.hex(),
                    "serialized_len": ser_len})
    assert out[1]["serialized_len"] == 368          # reference: hyperkzg/tests.rs:107-109
    return out


def main():
    data = {"hyperkzg": hyperkzg_vectors(), "field": field_vectors(), "transcript": transcript_vectors(), "sumcheck": sumcheck_vectors(),
            "eq": eq_vectors(), "curve": curve_vectors()}
    for k, v in data.items():
        with open(os.path.join(OUT, k + ".json"), "w") as f:
            json.dump(v, f, indent=1)
        print("wrote", k)


if __name__ == "__main__":
    main()
