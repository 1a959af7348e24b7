"""GPU: ONNXProof::prove at the model shapes of BASELINE.json (config 1: microgpt-shaped; config 3: nanoGPT-shaped; one GPT-2 layer)
(microgpt_model / nanogpt_model: over the model files' own tensors and example inputs, tests/golden/ref_models.npz) against COMMITTED oracle results (tests/golden/graph_proofs.json, made in the build container by tests/golden/gen_graph_proofs.py:
oracle/graph.py takes minutes on these graphs, so the GPU box does not recompute it): per-node trace hashes, sha256 of the proof bytes,
the final transcript state, the number of committed polynomials — then ONNXProof::verify of the device's proof.

What is pinned: the device against the in-repo oracle composition (not against a run of the reference)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "graph_proofs.json")))


def _build(name):
    import build_graphs as BG
    return {"microgpt": BG.microgpt, "microgpt_model": BG.microgpt_model, "nanogpt_model": BG.nanogpt_model, "gpt2_layer": BG.gpt2_layer, "tiny4": lambda: BG.tiny(layers=4), "tiny2": lambda: BG.tiny(layers=2),
            "node_einsum": BG.node_einsum, "node_relu": BG.node_relu, "node_mul": BG.node_mul}[name]()


def _h(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int32).tobytes()).hexdigest()[:16]


@pytest.mark.parametrize("name", ["microgpt", "microgpt_model", "tiny2", "tiny4", "nanogpt_model", "gpt2_layer", "node_einsum", "node_relu", "node_mul"])
def test_model_shaped_proof_matches_committed_oracle_result(atlas, name):
    import build_graphs as BG
    from oracle import orc
    from jolt_atlas_amd import graph as GG
    if name not in GOLD["graphs"]:
        pytest.fail(f"tests/golden/graph_proofs.json has no entry for {name}: run tests/golden/gen_graph_proofs.py {name}")
    want = GOLD["graphs"][name]
    nodes, outputs, inputs = _build(name)
    assert len(nodes) == want["n_nodes"] and [_h(x) for x in inputs] == want["input_sha256"], "the builder no longer yields the graph the fixture was made from"
    nv = BG.max_vars(nodes)
    assert nv == want["max_vars"]
    tau = orc.random_fr(1, GOLD["tau_seed"])[0]
    srs = atlas.SRS.generate(tau, 1 << nv)
    G = GG.Graph(nodes, outputs)
    G.trace(inputs)
    for nd, hw in zip(nodes, want["trace"]):
        assert _h(G.node_output(nd["idx"])) == hw, f"trace of node {nd['idx']} ({nd['op']})"
    proof, state, tm = G.prove(srs, inputs)
    assert tm["n_committed"] == want["n_committed"]
    assert state.hex() == want["state"], "final transcript state"
    assert len(proof) == want["proof_len"] and hashlib.sha256(proof).hexdigest() == want["proof_sha256"], "ONNXProof bytes"
    # ONNXProof::verify on a graph that was never traced
    vk = atlas.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
    out = G.node_output(outputs[0])
    V = GG.Graph(nodes, outputs)
    ok, vstate = V.verify(vk, inputs, out, proof)
    assert ok and vstate == state
    bad = out.copy(); bad[len(bad) // 2] ^= 1
    assert not V.verify(vk, inputs, bad, proof)[0]
    G.free(); V.free(); srs.free()


def test_gpt2_12_layers_config4(atlas):
    """BASELINE config 4 at size on one GPU: the 12-layer GPT-2-shaped graph (842 nodes, 8343 committed polynomials, max_num_vars 24;
    jolt-atlas-core/examples/gpt2.rs:88-118 in shape — c_attn as ONE 768 -> 2304 MatMul padded to 1024 x 4096 and split three ways, as the HF export
    the example downloads has it; there is no model file in the reference's tree).  The oracle EXECUTED this graph in the build container (per-node
    trace hashes) and PROVED it there (tests/golden/gen_graph_proofs.py gpt2: about two hours on 4 threads): the device's proof is held to the
    oracle's sha256 and final transcript state (the device's own pin recorded before that, device_proof_sha256, must be the same value), and
    atlas_verify_graph — written from the reference's verifier side — accepts it and rejects a flipped output and a flipped proof byte."""
    import build_graphs as BG
    from oracle import orc
    from jolt_atlas_amd import graph as GG
    want = GOLD["graphs"].get("gpt2")
    if not want:
        pytest.fail("tests/golden/graph_proofs.json has no entry for gpt2: run tests/golden/gen_graph_proofs.py --trace-only gpt2")
    nodes, outputs, inputs = BG.gpt2()
    assert len(nodes) == want["n_nodes"] and [_h(x) for x in inputs] == want["input_sha256"], "the builder no longer yields the graph the fixture was made from"
    nv = BG.max_vars(nodes)
    assert nv == want["max_vars"] == 24
    tau = orc.random_fr(1, GOLD["tau_seed"])[0]
    srs = atlas.SRS.generate(tau, 1 << nv)
    srs.precompute()
    G = GG.Graph(nodes, outputs)
    G.trace(inputs)
    for nd, hw in zip(nodes, want["trace"]):
        if nd["op"] == "Constant":
            continue                                   # (1.1 GB of weights: their upload is covered by every node that reads them)
        assert _h(G.node_output(nd["idx"])) == hw, f"trace of node {nd['idx']} ({nd['op']})"
    proof, state, tm = G.prove(srs, inputs)
    assert tm["n_nodes"] == want["n_nodes"] == 842 and tm["n_committed"] == want["n_committed"] == 8343
    assert "proof_sha256" in want, "the fixture holds the device's pin only: merge the oracle's proof of this graph (gen_graph_proofs.py gpt2) before trusting it"
    assert want["proof_sha256"] == want["device_proof_sha256"] and want["state"] == want["device_state"]
    assert len(proof) == want["proof_len"] and hashlib.sha256(proof).hexdigest() == want["proof_sha256"], "ONNXProof bytes (oracle's proof of the same graph)"
    assert state.hex() == want["state"]
    vk = atlas.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
    out = G.node_output(outputs[0])
    V = GG.Graph(nodes, outputs)
    ok, vstate = V.verify(vk, inputs, out, proof)
    assert ok and vstate == state
    bad = out.copy(); bad[len(bad) // 2] ^= 1
    assert not V.verify(vk, inputs, bad, proof)[0]
    tampered = bytearray(proof); tampered[len(tampered) // 3] ^= 0x10
    assert not V.verify(vk, inputs, out, bytes(tampered))[0]
    G.free(); V.free(); srs.free()
