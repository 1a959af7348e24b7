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


def build(case, B, orc):
    """B: backend adapter with the constructor names of oracle.orc_ra / jolt_atlas_amd.instances; returns (inst, label)."""
    fam = case["family"]
    fr = lambda hs: orc.from_ints(ints(hs))
    u64 = lambda hs: np.array(ints(hs), dtype=np.uint64)
    H = lambda rows: [np.array(r, dtype=np.int32) for r in rows]
    if fam == "ra_virtual":
        return B.ra_virtual(H(case["H"]), case["log_k"], np.stack([fr(c) for c in case["chunks"]]), fr(case["r_cycle"])), b"golden_ra"
    if fam == "booleanity":
        return B.booleanity(np.stack([fr(g) for g in case["G"]]), H(case["H"]), case["log_k"], fr(case["gammas"]), fr(case["r_address"]),
                            fr(case["r_cycle"])), b"golden_bool"
    if fam == "hamming":
        return B.hamming(np.stack([fr(g) for g in case["G"]]), case["log_k"], fr(case["gamma_powers"])), b"golden_hw"
    if fam == "dense_opening":
        return B.dense_opening(fr(case["poly"]), fr(case["point"])), b"golden_do"
    if fam == "onehot_opening":
        return B.onehot_opening(np.array(case["idx"], dtype=np.int32), case["log_K"], fr(case["r_address"]), fr(case["r_cycle"])), b"golden_oh"
    g = orc.from_ints([int(case["gamma"], 16)])[0] if "gamma" in case else None
    if fam == "ps_relu":
        return B.ps_relu(u64(case["idx"]), case["N"], fr(case["r"]), g), b"golden_relu"
    if fam == "ps_clamp":
        return B.ps_clamp(u64(case["idx"]), case["N"], case["bound"], bool(case["symmetric"]), fr(case["r"]), g), b"golden_clamp"
    if fam == "ps_identity":
        return B.ps_identity(u64(case["idx"]), case["log_K"], case["phases"], fr(case["r"])), b"golden_id"
    if fam == "ps_ult":
        return B.ps_ult(u64(case["idx"]), fr(case["r"]), g), b"golden_ult"
    if fam == "ps_rshift":
        return B.ps_rshift(u64(case["idx"]), case["N"], case["shift"], fr(case["r"]), g), b"golden_rs"
    raise KeyError(fam)
