"""CPU: the C oracle replays the operator-prover golden vectors (tests/golden/instances.json, generated from the
dense-table Python models by tests/golden/gen_golden_instances.py)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _golden_instances import build, cases, check, ints   # noqa: E402
from oracle import orc, orc_ra as OR                       # noqa: E402


@pytest.mark.parametrize("case", cases(), ids=lambda c: c.get("name", c["family"]) + "_" + c["state"][:6])
def test_oracle_replays_golden_instance(case):
    fam = case["family"]
    if fam == "elementwise":
        k = orc.from_ints(ints(case["constants"])) if case["constants"] else None
        inst = OR.elementwise(case["op"], [orc.from_ints(ints(o)) for o in case["operands"]], orc.from_ints(ints(case["r"])), k)
        label = b"golden_ew"
    elif fam == "softmax":
        b = orc.from_ints(ints(case["b"])) if case["b"] is not None else None
        inst = OR.softmax(case["kind"], orc.from_ints(ints(case["a"])), b, case["log_K"], case["log_N"],
                          orc.from_ints(ints(case["r"])) if case["r"] else None)
        label = b"golden_sm"
    else:
        inst, label = build(case, OR, orc)
    t = orc.new_transcript(label)
    rows, raw = inst.prove(orc.from_ints([int(case["claim"], 16)])[0], t)
    finals = list(inst.finals()) if hasattr(inst, "finals") else []
    check(orc, case, rows, raw, t.state, finals)
