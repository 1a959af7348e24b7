"""Reference-HELD vectors under `Model::trace` (f1 / f3): the literal inputs and expected tensors of the doc tests in
/root/reference/atlas-onnx-tracer/src/tensor/ops.rs and of Tensor::move_axis / Tensor::expand in tensor/mod.rs (tests/golden/ref_tensor_ops.json; data only, extracted by
tools/extract_ref_doctests.py in the build container), replayed

  * on the CPU through oracle/graph.py:execute and the oracle's table formulas (this pins the ORACLE against the reference), and
  * on the GPU through atlas_graph_trace (`-m gpu`; this pins graph_exec.hip against the reference without the oracle in between).

Every record is either green or listed with the reason it does not map (`tests/_ref_tensor_ops.py`), and the counts are asserted so
that a record cannot fall out silently."""
import json
import os

import numpy as np
import pytest

import _ref_tensor_ops as M

HERE = os.path.dirname(os.path.abspath(__file__))
DOC = json.load(open(os.path.join(HERE, "golden", "ref_tensor_ops.json")))
CASES = DOC["cases"]
PLANS = [M.plan(c) for c in CASES]
IDS = [f"{c['fn']}@{c['line']}#{i}" for i, c in enumerate(CASES)]


def test_fixture_shape():
    assert DOC["n_cases"] == len(CASES) == 105
    kinds = [p[0] for p in PLANS]
    # 34 records map onto operators / table formulas of the path; the other 71 are tensor functions no ONNXProof operator executes
    assert kinds.count("graph") + kinds.count("formula") == 34, (kinds.count("graph"), kinds.count("formula"))
    mapped_fns = {c["fn"] for c, p in zip(CASES, PLANS) if p[0] != "unmapped"}
    assert mapped_fns == {"add", "sub", "iff", "and", "neg", "einsum", "sum_axes", "gather", "concat", "slice", "sra", "div", "const_div",
                          "const_rem", "mean_of_squares_axes", "max_axes", "tanh", "erffunc", "sigmoid", "sin", "cos", "Tensor::move_axis", "Tensor::expand"}


def test_unmapped_records_are_listed_with_a_reason():
    for c, p in zip(CASES, PLANS):
        if p[0] == "unmapped":
            assert isinstance(p[1], str) and len(p[1]) > 10, c["fn"]


GRAPH = [(i, p) for i, p in enumerate(PLANS) if p[0] == "graph"]
FORMULA = [(i, p) for i, p in enumerate(PLANS) if p[0] == "formula"]


@pytest.mark.parametrize("i,p", GRAPH, ids=[IDS[i] for i, _ in GRAPH])
def test_oracle_executor_matches_reference_doctest(i, p):
    from oracle import graph as OG
    _, nodes, inputs, out_idx, wkey = p
    out, wit = OG.execute(nodes, inputs)
    got = wit[out_idx][wkey] if wkey else out[out_idx]
    want = np.array(CASES[i]["expected"]["data"], dtype=np.int64)
    assert np.array_equal(np.asarray(got, dtype=np.int64).reshape(-1), want), (CASES[i]["fn"], got, want)


@pytest.mark.parametrize("i,p", FORMULA, ids=[IDS[i] for i, _ in FORMULA])
def test_oracle_formula_matches_reference_doctest(i, p):
    from oracle import graph as OG
    c = CASES[i]
    x = c["args"][0]["tensor"]
    want = list(c["expected"]["data"])
    if p[1] == "softmax_max":                          # max_axes over the last axis = SoftmaxLastAxisTrace::max_k (ops/softmax.rs:121-123)
        assert c["args"][1]["list"] == [len(x["dims"]) - 1]
        N = x["dims"][-1]
        _, tr = OG.softmax_trace(np.array(x["data"], dtype=np.int32), len(x["data"]) // N, N, 1 << OG.MODEL_SCALE)
        assert list(tr["max_k"]) == want
        return
    sc = float(c["args"][1]["num"])
    got = [OG.nonlinearity_value(p[1], v, sc) for v in x["data"]]
    assert got == want, (c["fn"], sc, got, want)


# ---- the same records through the device executor
@pytest.mark.gpu
@pytest.mark.parametrize("i,p", GRAPH, ids=[IDS[i] for i, _ in GRAPH])
def test_device_trace_matches_reference_doctest(atlas, i, p):
    from jolt_atlas_amd import graph as GG
    _, nodes, inputs, out_idx, wkey = p
    if wkey:
        pytest.skip("the remainder witness of ScalarConstDiv stays in HBM (no C-ABI read-back); its value enters the proof bytes "
                    "that test_gpu_graph.py compares")
    padded, why = M.pad_plan(nodes, inputs, out_idx)
    if padded is None:
        pytest.skip(why)
    pnodes, pinputs, raw = padded
    G = GG.Graph(pnodes, [out_idx])
    try:
        G.trace(pinputs)
        got = M.crop(G.node_output(out_idx), next(n for n in pnodes if n["idx"] == out_idx)["dims"], raw)
    finally:
        G.free()
    want = np.array(CASES[i]["expected"]["data"], dtype=np.int64)
    assert np.array_equal(got.astype(np.int64), want), (CASES[i]["fn"], got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("op", ["Tanh", "Erf", "Sigmoid", "Sin", "Cos"])
def test_device_tables_equal_the_formula_the_doctests_pin(atlas, op):
    """The device's tables are compiled for MODEL_SCALE; the doc tests pin the FORMULA at other multipliers (above).  This closes the
    chain: the device's node output over every table index == the formula at the model's multiplier."""
    from jolt_atlas_amd import graph as GG
    from oracle import graph as OG
    S = OG.MODEL_SCALE
    if op in ("Sin", "Cos"):
        x = (np.arange(1 << 12, dtype=np.int64) * 997 - (1 << 21)).astype(np.int32)          # several periods of either sign
        q = np.floor_divide(x.astype(np.int64), OG.TRIG_PERIOD_MODULUS)
        down = (x.astype(np.int64) - q * OG.TRIG_PERIOD_MODULUS) >> OG.TRIG_DOWNSCALE_BITS
        sc = float(1 << (S - OG.TRIG_DOWNSCALE_BITS))
        want = np.array([OG.nonlinearity_value(op, int(v), sc) << OG.TRIG_DOWNSCALE_BITS for v in down], dtype=np.int64)
    else:
        x = np.arange(-(1 << 17), 1 << 17, 1 << 5, dtype=np.int32)                            # 2^13 points across the clamped domain
        want = np.array([OG.nonlinearity_value(op, int(v), float(1 << S)) for v in x], dtype=np.int64)
    nodes = [{"idx": 0, "op": "Input", "inputs": [], "dims": [len(x)]}, {"idx": 1, "op": op, "inputs": [0], "dims": [len(x)], "scale": S}]
    G = GG.Graph(nodes, [1])
    try:
        G.trace([x])
        got = G.node_output(1).astype(np.int64)
    finally:
        G.free()
    assert np.array_equal(got, want)
