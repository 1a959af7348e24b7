"""GPU: the device replays the operator-prover golden vectors (tests/golden/instances.json) — no oracle involved."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _golden_instances import build, cases, check, ints   # noqa: E402


@pytest.mark.parametrize("case", cases(), ids=lambda c: c.get("name", c["family"]) + "_" + c["state"][:6])
def test_device_replays_golden_instance(atlas, case):
    from oracle import orc                                  # value conversion only
    from jolt_atlas_amd import instances as I
    A = atlas
    fam = case["family"]
    polys = []
    if fam == "elementwise":
        polys = [A.MultilinearPolynomial.from_fr(orc.from_ints(ints(o))) for o in case["operands"]]
        k = orc.from_ints(ints(case["constants"])) if case["constants"] else None
        inst, label = I.elementwise(case["op"], polys, orc.from_ints(ints(case["r"])), k), b"golden_ew"
    elif fam == "softmax":
        polys = [A.MultilinearPolynomial.from_fr(orc.from_ints(ints(case["a"])))]
        if case["b"] is not None:
            polys.append(A.MultilinearPolynomial.from_fr(orc.from_ints(ints(case["b"]))))
        inst = I.softmax_instance(case["kind"], polys[0], polys[1] if len(polys) > 1 else None, case["log_K"], case["log_N"],
                                  orc.from_ints(ints(case["r"])) if case["r"] else None)
        label = b"golden_sm"
    else:
        class Dev:                                           # the device constructors under the oracle wrapper's names
            ra_virtual, booleanity, onehot_opening = I.ra_virtual, I.booleanity, I.onehot_opening
            hamming = staticmethod(lambda G, log_k, gp: I.hamming_weight(G, log_k, gp))
            dense_opening = staticmethod(lambda poly, pt: I.dense_opening(A.MultilinearPolynomial.from_fr(poly), pt))
            ps_relu, ps_clamp, ps_ult, ps_rshift = I.ps_shout_relu, I.ps_shout_clamp, I.ps_shout_ult, I.ps_shout_rshift
            ps_identity = I.identity_range_check
        inst, label = build(case, Dev, orc)
    t = A.Blake2bTranscript(label)
    rows, raw = inst.prove(orc.from_ints([int(case["claim"], 16)])[0], t)
    finals = inst.final_claims()                             # against the MODEL's final claims
    check(orc, case, rows, raw, t.state, finals)
    inst.free()
    for p_ in polys:
        p_.free()
