"""Quantised model execution (f3) on the reference's OWN model files: the tensors of atlas-onnx-tracer/models/{microgpt,nanoGPT}/network.onnx,
quantised at 2^14 by the loader's rule, and the models' example token ids (tests/golden/ref_models.npz, data only, tools/extract_ref_model.py)
go through the graphs of tools/build_graphs.py — the operator decomposition `Model::load_onnx_model`'s handlers would produce, as read by the
builder — and the dequantised integer logits are compared with

  * microgpt: the float logits the exporting script stored beside the model (models/microgpt/input.json): REFERENCE-HELD expected output;
  * nanoGPT: a float64 forward pass over the file's tensors (the extractor computes it: the logits stored in models/nanoGPT/input.json are
    not this file's — correlation 0.32 with its own forward pass, where microgpt's agree to 3e-8).

Tolerance: the integer pipeline rounds at 2^-14 per operator and takes tanh / exp / rsqrt from the prover's tables; logits are O(0.1 .. 1).
The bound below (2 % of the logit range) is ~20 x the observed error and far below what a wrong operator order, a transposed weight or a
missing rescale produces (those give correlations below 0.9).  CPU: oracle executor.  GPU: atlas_graph_trace, node for node == the oracle's."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
Z = np.load(os.path.join(ROOT, "tests", "golden", "ref_models.npz"))
CASES = {"microgpt": ("microgpt_model", 32, "logits_f32"), "nanogpt": ("nanogpt_model", 65, "logits_float_forward")}


def _check(key, logits_i32, dims):
    builder, vocab, want_key = CASES[key]
    got = (logits_i32.astype(np.float64) / (1 << 14)).reshape(dims)[:, :vocab]
    want = Z[f"{key}/{want_key}"].astype(np.float64).reshape(-1, vocab)
    err = np.abs(got - want)
    span = want.max() - want.min()
    assert err.max() < 0.02 * span, (key, err.max(), span)
    assert np.corrcoef(got.ravel(), want.ravel())[0, 1] > 0.9995
    assert (got.argmax(1) == want.argmax(1)).mean() >= 0.95            # near-ties may flip under 2^-14 rounding
    return err.max(), span


def test_stored_nanogpt_logits_are_not_this_files():
    """documents the finding: the expected output of the nanoGPT case is the float forward pass, not input.json's output_data"""
    a, b = Z["nanogpt/logits_float_forward"].ravel(), Z["nanogpt/logits_f32"].ravel()
    assert np.corrcoef(a, b)[0, 1] < 0.5
    assert np.abs(Z["microgpt/logits_float_forward"].ravel() - Z["microgpt/logits_f32"].ravel()).max() < 1e-6


@pytest.mark.parametrize("key", ["microgpt", "nanogpt"])
def test_oracle_execution_of_the_model_file_matches_its_float_logits(key):
    import build_graphs as BG
    from oracle import graph as OG
    nodes, outputs, inputs = getattr(BG, CASES[key][0])()
    assert np.array_equal(inputs[0], Z[f"{key}/tokens"])
    out, _ = OG.execute(nodes, inputs)
    _check(key, out[outputs[0]], next(n for n in nodes if n["idx"] == outputs[0])["dims"])


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["microgpt", "nanogpt"])
def test_device_execution_of_the_model_file_matches_its_float_logits(atlas, key):
    import build_graphs as BG
    from oracle import graph as OG
    from jolt_atlas_amd import graph as GG
    nodes, outputs, inputs = getattr(BG, CASES[key][0])()
    G = GG.Graph(nodes, outputs)
    try:
        G.trace(inputs)
        got = G.node_output(outputs[0])
        _check(key, got, next(n for n in nodes if n["idx"] == outputs[0])["dims"])
        want, _ = OG.execute(nodes, inputs)                            # and bit for bit the oracle executor's trace
        for nd in nodes:
            assert np.array_equal(G.node_output(nd["idx"]), want[nd["idx"]]), (nd["idx"], nd["op"])
    finally:
        G.free()
