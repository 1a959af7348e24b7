"""On an MI355X: prove a named graph of tools/build_graphs.py with the tau of tests/golden/graph_proofs.json and print what the DEVICE produced
(sha256 of the proof bytes, final transcript state, committed polynomials, proof length) as JSON — for the graphs whose oracle proof is out of
reach (the 12-layer GPT-2-shaped graph): merged into the fixture's entry as device_* by tests/golden/merge_device_proof.py.  A regression pin."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import build_graphs as BG
import jolt_atlas_amd as A
from jolt_atlas_amd import graph as GG
from oracle import orc

name = sys.argv[1] if len(sys.argv) > 1 else "gpt2"
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "graph_proofs.json")))
A.init(0)
nodes, outputs, inputs = getattr(BG, name)()
nv = BG.max_vars(nodes)
tau = orc.random_fr(1, gold["tau_seed"])[0]
srs = A.SRS.generate(tau, 1 << nv)
if nv >= 16: srs.precompute()
G = GG.Graph(nodes, outputs)
out = {}
for rep in range(2):
    proof, state, tm = G.prove(srs, inputs)
    rec = {"device_proof_sha256": hashlib.sha256(proof).hexdigest(), "device_state": state.hex(), "n_committed": tm["n_committed"], "proof_len": len(proof)}
    assert not out or out == rec, "two proofs of the same inputs differ"
    out = rec
vk = A.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
V = GG.Graph(nodes, outputs)
ok, vstate = V.verify(vk, inputs, G.node_output(outputs[0]), proof)
assert ok and vstate == state
out["graph"] = name
print(json.dumps(out))
