"""GPU: the device replays the operator-prover golden vectors (tests/golden/instances.json) — no oracle involved."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _golden_instances import cases, check, ints          # noqa: E402


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
        inst = I.ps_shout_rshift(np.array(ints(case["idx"]), dtype=np.uint64), case["N"], case["shift"], orc.from_ints(ints(case["r"])),
                                 orc.from_ints([int(case["gamma"], 16)])[0])
        label = b"golden_rs"
    t = A.Blake2bTranscript(label)
    rows, raw = inst.prove(orc.from_ints([int(case["claim"], 16)])[0], t)
    finals = inst.final_claims() if fam != "ps_rshift" else []
    check(orc, case, rows, raw, t.state, finals)
    inst.free()
    for p_ in polys:
        p_.free()
