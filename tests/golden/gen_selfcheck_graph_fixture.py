"""tests/golden/selfcheck_graph_fixtures.json: the layout tools/export_ref_graph_fixtures.rs prints, generated from THIS repository's
oracle composition (oracle/graph.py) — pins nothing against the reference, keeps the replay code of tests/test_reference_fixtures.py
exercised (and is a regression guard for the oracle's proof bytes).  Run from the repository root:
    python tests/golden/gen_selfcheck_graph_fixture.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import graph as OG, orc  # noqa: E402

FQ = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47


def fq_canon(m):
    return sum(int(x) << (64 * i) for i, x in enumerate(m)) * pow(1 << 256, -1, FQ) % FQ


def g1_hex(p):
    if int(p["infinity"]):
        return (bytes(31) + b"\x40").hex()
    x, y = fq_canon(p["x"]), fq_canon(p["y"])
    b = bytearray(x.to_bytes(32, "little"))
    if y > FQ - y:
        b[31] |= 0x80
    return bytes(b).hex()


rng = np.random.default_rng(0xA71A5)
models = []
# the first four models of the exporter, same shapes and operators (other random data: this file is not the reference's)
models.append(("relu16", [{"idx": 0, "op": "Input", "inputs": [], "dims": [16]}, {"idx": 1, "op": "ReLU", "inputs": [0], "dims": [16]}], [1],
               [rng.integers(-(1 << 12), 1 << 12, size=16).astype(np.int32)]))
c = rng.integers(-(1 << 20), 1 << 20, size=16).astype(np.int32)
models.append(("add4x4", [{"idx": 0, "op": "Input", "inputs": [], "dims": [4, 4]}, {"idx": 1, "op": "Constant", "inputs": [], "dims": [4, 4], "data": c},
                          {"idx": 2, "op": "Add", "inputs": [0, 1], "dims": [4, 4]}], [2], [rng.integers(-(1 << 20), 1 << 20, size=16).astype(np.int32)]))
w = rng.integers(-(1 << 12), 1 << 12, size=128).astype(np.int32)
models.append(("einsum_relu", [{"idx": 0, "op": "Input", "inputs": [], "dims": [4, 8]}, {"idx": 1, "op": "Constant", "inputs": [], "dims": [8, 16], "data": w},
                               {"idx": 2, "op": "Einsum", "inputs": [0, 1], "dims": [4, 16], "layout": "mk,kn->mn", "scale": 14, "shape": [4, 8, 16]},
                               {"idx": 3, "op": "ReLU", "inputs": [2], "dims": [4, 16]}], [3], [rng.integers(-(1 << 14), 1 << 14, size=32).astype(np.int32)]))
k = rng.integers(-(1 << 14), 1 << 14, size=32).astype(np.int32)
models.append(("mul4x8", [{"idx": 0, "op": "Input", "inputs": [], "dims": [4, 8]}, {"idx": 1, "op": "Constant", "inputs": [], "dims": [4, 8], "data": k},
                          {"idx": 2, "op": "Mul", "inputs": [0, 1], "dims": [4, 8], "scale": 14}], [2], [rng.integers(-(1 << 14), 1 << 14, size=32).astype(np.int32)]))
# round 5: the seven models the exporter gained (SoftmaxLastAxis, Tanh, GatherSmall, Div, Rsqrt, a one-element Div, a LayerNorm-shaped chain)
models.append(("softmax2x4x8", [{"idx": 0, "op": "Input", "inputs": [], "dims": [2, 4, 8]}, {"idx": 1, "op": "SoftmaxLastAxis", "inputs": [0], "dims": [2, 4, 8], "scale": 14}], [1],
               [rng.integers(-(1 << 15), 1 << 15, size=64).astype(np.int32)]))
models.append(("tanh4x4", [{"idx": 0, "op": "Input", "inputs": [], "dims": [4, 4]}, {"idx": 1, "op": "Tanh", "inputs": [0], "dims": [4, 4], "scale": 14}], [1],
               [rng.integers(-(1 << 18), 1 << 18, size=16).astype(np.int32)]))
dict_ = rng.integers(-(1 << 14), 1 << 14, size=64).astype(np.int32)
models.append(("gather8of16", [{"idx": 0, "op": "Input", "inputs": [], "dims": [8]}, {"idx": 1, "op": "Constant", "inputs": [], "dims": [16, 4], "data": dict_},
                               {"idx": 2, "op": "GatherSmall", "inputs": [1, 0], "dims": [8, 4], "axis": 0, "dict_len": 16}], [2], [rng.integers(0, 16, size=8).astype(np.int32)]))
den = rng.integers(1, 1 << 10, size=16).astype(np.int32)
models.append(("div4x4", [{"idx": 0, "op": "Input", "inputs": [], "dims": [4, 4]}, {"idx": 1, "op": "Constant", "inputs": [], "dims": [4, 4], "data": den},
                          {"idx": 2, "op": "Div", "inputs": [0, 1], "dims": [4, 4]}], [2], [rng.integers(-(1 << 20), 1 << 20, size=16).astype(np.int32)]))
models.append(("rsqrt4x4", [{"idx": 0, "op": "Input", "inputs": [], "dims": [4, 4]}, {"idx": 1, "op": "Rsqrt", "inputs": [0], "dims": [4, 4], "scale": 14}], [1],
               [rng.integers(1, 1 << 20, size=16).astype(np.int32)]))
models.append(("div1", [{"idx": 0, "op": "Input", "inputs": [], "dims": [2, 2]}, {"idx": 1, "op": "Sum", "inputs": [0], "dims": [2, 1], "axes": [1]},
                        {"idx": 2, "op": "Sum", "inputs": [1], "dims": [1, 1], "axes": [0]}, {"idx": 3, "op": "Constant", "inputs": [], "dims": [1, 1], "data": np.array([70000], dtype=np.int32)},
                        {"idx": 4, "op": "Div", "inputs": [3, 2], "dims": [1, 1]}, {"idx": 5, "op": "Broadcast", "inputs": [4], "dims": [1, 2]},
                        {"idx": 6, "op": "Constant", "inputs": [], "dims": [1, 2], "data": np.array([3, -4], dtype=np.int32)}, {"idx": 7, "op": "Add", "inputs": [5, 6], "dims": [1, 2]}], [7],
               [np.array([5000, 6000, 7000, 8000], dtype=np.int32)]))
models.append(("layernorm4x8", [{"idx": 0, "op": "Input", "inputs": [], "dims": [4, 8]}, {"idx": 1, "op": "Sum", "inputs": [0], "dims": [4, 1], "axes": [1]},
                                {"idx": 2, "op": "ScalarConstDiv", "inputs": [1], "dims": [4, 1], "divisor": 8}, {"idx": 3, "op": "Broadcast", "inputs": [2], "dims": [4, 8]},
                                {"idx": 4, "op": "Sub", "inputs": [0, 3], "dims": [4, 8]}, {"idx": 5, "op": "MeanOfSquares", "inputs": [4], "dims": [4, 1], "axes": [1], "scale": 14, "count": 8},
                                {"idx": 6, "op": "Rsqrt", "inputs": [5], "dims": [4, 1], "scale": 14}, {"idx": 7, "op": "Broadcast", "inputs": [6], "dims": [4, 8]},
                                {"idx": 8, "op": "Mul", "inputs": [4, 7], "dims": [4, 8], "scale": 14}], [8], [rng.integers(-(1 << 15), 1 << 15, size=32).astype(np.int32)]))

tau = orc.random_fr(1, 0x51250001)[0]
out = {"graphs": []}
for name, nodes, outputs, inputs in models:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_graphs as BG
    nv = max(BG.max_vars(nodes), 4 + max(int(np.log2(np.prod(nd["dims"]))) for nd in nodes))      # the reference's max_num_vars: a one-hot chunk, 16 addresses x T cycles
    srs = orc.srs_powers(tau, 1 << nv)
    P = OG.Prover(nodes, outputs, srs)
    proof = P.prove(inputs)
    jn = []
    for nd in nodes:
        d = dict(nd)
        if "data" in d:
            d["data"] = [int(x) for x in d["data"]]
        jn.append(d)
    out["graphs"].append(dict(name=name, pretty="(selfcheck: generated by oracle/graph.py)", nodes=jn, outputs=outputs, inputs=[[int(x) for x in a] for a in inputs],
                              output=[int(x) for x in P.trace[outputs[0]]], srs_g1=[g1_hex(p) for p in srs], proof=proof.hex()))
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "selfcheck_graph_fixtures.json"), "w"))
print("wrote tests/golden/selfcheck_graph_fixtures.json")
