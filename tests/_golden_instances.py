"""Shared replay of tests/golden/instances.json: `make(case)` builds the instance (oracle or device), `prove` runs it."""
import json
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "instances.json")


def cases():
    with open(PATH) as f:
        return json.load(f)["cases"]


def ints(hexes):
    return [int(h, 16) for h in hexes]


def check(orc, case, rows, raw_challenges, state_bytes, finals):
    assert [int(c, 16) for c in case["challenges"]] == list(raw_challenges)
    assert [orc.to_ints(r) for r in rows] == [ints(r) for r in case["rows"]]
    assert bytes(state_bytes).hex() == case["state"]
    if len(finals):          # the oracle keeps no final-claim accessor for some families
        assert orc.to_ints(np.stack(finals)) == ints(case["finals"])[:len(finals)]
