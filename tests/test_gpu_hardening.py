"""Round 6 hardening (`-m gpu`): a Gather index outside its dictionary fails the trace (the reference's tensor indexing panics) instead of
reading — and, for a one-index gather, writing — outside a buffer; the waits of the round channel are runtime settings; whole proofs do not
depend on how HIP maps streams onto hardware queues (GPU_MAX_HW_QUEUES = 1 / 2: the lane streams, their gates and the resident tails must
make progress on ONE queue) and give the bytes the default mapping gives."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "graph_proofs.json")))


def _gather_graph(op, idx, n_idx):
    rng = np.random.default_rng(3)
    return [{"idx": 0, "op": "Constant", "inputs": [], "dims": [8, 4], "data": rng.integers(-50, 50, size=32).astype(np.int32)},
            {"idx": 1, "op": "Input", "inputs": [], "dims": [n_idx]},
            {"idx": 2, "op": op, "inputs": [0, 1], "dims": [n_idx, 4], "axis": 0, "dict_len": 8}], [2], [np.asarray(idx, dtype=np.int32)]


@pytest.mark.parametrize("op", ["GatherSmall", "GatherLarge"])
@pytest.mark.parametrize("idx", [[1, 8], [-1, 3], [2 ** 31 - 1, 0], [9]])
def test_gather_index_outside_the_dictionary_fails_the_trace(atlas, op, idx):
    from jolt_atlas_amd import graph as GG
    nodes, outputs, inputs = _gather_graph(op, idx, len(idx))
    G = GG.Graph(nodes, outputs)
    with pytest.raises(Exception, match="outside its dictionary"):
        G.trace(inputs)
    good = [np.asarray([i % 8 for i in range(len(idx))], dtype=np.int32)]
    G.trace(good)                                            # the same graph object traces a valid input afterwards
    want = np.asarray(nodes[0]["data"]).reshape(8, 4)[good[0]].reshape(-1)
    assert np.array_equal(G.node_output(2)[:want.size], want)
    G.free()


def test_timeouts_are_runtime_settings(atlas):
    import jolt_atlas_amd as A
    d0, h0, b0 = A.get_timeouts()
    assert (d0, h0) == (2.0, 10.0) or os.environ.get("ATLAS_DEVICE_WAIT_S") or os.environ.get("ATLAS_HOST_WAIT_S")
    A.set_timeouts(5.0, 20.0, 45.0)
    assert A.get_timeouts() == (5.0, 20.0, 45.0)
    A.set_timeouts(0, 0, 0)                                  # 0 keeps
    assert A.get_timeouts() == (5.0, 20.0, 45.0)
    # a proof under the new settings is the same proof
    import build_graphs as BG
    from jolt_atlas_amd import graph as GG
    from oracle import orc
    nodes, outputs, inputs = BG.tiny(layers=2)
    tau = orc.random_fr(1, GOLD["tau_seed"])[0]
    srs = A.SRS.generate(tau, 1 << BG.max_vars(nodes))
    G = GG.Graph(nodes, outputs)
    proof, state, _ = G.prove(srs, inputs)
    assert hashlib.sha256(proof).hexdigest() == GOLD["graphs"]["tiny2"]["proof_sha256"]
    G.free(); srs.free()
    A.set_timeouts(d0, h0, b0)


_CHILD = r'''
import hashlib, json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tools"))
import build_graphs as BG
import jolt_atlas_amd as A
from jolt_atlas_amd import graph as GG
from oracle import orc
A.init(0)
out = {{}}
for name in {names!r}:
    nodes, outputs, inputs = getattr(BG, name)()
    nv = BG.max_vars(nodes)
    tau = orc.random_fr(1, {tau_seed})[0]
    srs = A.SRS.generate(tau, 1 << nv)
    if nv >= 16: srs.precompute()
    G = GG.Graph(nodes, outputs)
    for rep in range(2):
        proof, state, tm = G.prove(srs, inputs)
    out[name] = [hashlib.sha256(proof).hexdigest(), state.hex()]
    G.free(); srs.free()
print("RESULT " + json.dumps(out))
'''


@pytest.mark.parametrize("queues", [1, 2])
def test_proofs_do_not_depend_on_the_hardware_queue_mapping(queues):
    """GPU_MAX_HW_QUEUES caps the hardware queues HIP spreads its streams over: with ONE queue every lane stream, gate and resident tail of
    a proof shares it.  nanoGPT (the model file's tensors: 271 nodes, every operator family) and the ReLU node at T = 2^16."""
    names = ["node_relu", "nanogpt_model"]
    env = dict(os.environ, GPU_MAX_HW_QUEUES=str(queues))
    code = _CHILD.format(root=ROOT, names=names, tau_seed=GOLD["tau_seed"])
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    got = json.loads(line[len("RESULT "):])
    for name in names:
        want = GOLD["graphs"][name]
        assert got[name] == [want["proof_sha256"], want["state"]], (queues, name)
