"""Wall clock of atlas_prove_graph on the synthetic transformer graphs (tools/build_graphs.py), stage split like the
reference's tracing spans (README: commit / iop / reduction / HyperKZG)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import build_graphs as BG  # noqa: E402
import jolt_atlas_amd as A  # noqa: E402
from jolt_atlas_amd import graph as GG  # noqa: E402


def run(name, level, reps):
    nodes, outputs, inputs = getattr(BG, name)() if name.endswith("_model") or name.startswith("node_") else getattr(BG, name)(level=level)
    nv = BG.max_vars(nodes)
    tau = np.array([0x1234567, 0, 0, 0], dtype=np.uint64)
    t0 = time.time()
    srs = A.SRS.generate(tau, 1 << nv)
    if os.environ.get("ATLAS_GRAPH_TAB", "1") != "0" and nv >= 16:
        srs.precompute()
    setup_s = time.time() - t0
    G = GG.Graph(nodes, outputs)
    best = None
    for _ in range(reps):
        t0 = time.time()
        proof, state, tm = G.prove(srs, inputs)
        tm["wall_ms"] = (time.time() - t0) * 1e3
        if best is None or tm["total_ms"] < best["total_ms"]:
            best = tm
    best.update(graph=name, level=level, proof_bytes=len(proof), max_vars=nv, setup_s=setup_s, state=state.hex()[:16])
    if os.environ.get("ATLAS_GRAPH_VERIFY", "1") != "0":      # ONNXProof::verify of the proof just made (host arithmetic + the pairing)
        vk = A.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
        out = G.node_output(outputs[0])
        V = GG.Graph(nodes, outputs)
        t0 = time.time()
        ok, vstate = V.verify(vk, inputs, out, proof)
        best.update(verify_ms=(time.time() - t0) * 1e3, verified=bool(ok and vstate == state))
        V.free()
    print(json.dumps(best))
    G.free(); srs.free()


if __name__ == "__main__":
    A.init(0)
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["tiny", "nanogpt"]
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    for n in names:
        run(n, level, reps)
