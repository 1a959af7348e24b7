#!/usr/bin/env python3
"""Whole-proof fixtures for the graphs the GPU suite proves: what oracle/graph.py (the CPU composition of ONNXProof::prove) yields for
each named graph — sha256 of the proof bytes, the final transcript state, a hash per node of the trace, the number of committed
polynomials.  Generated HERE (build container, minutes of CPU for the model-shaped graphs) so that the GPU box compares
atlas_prove_graph against committed values instead of recomputing the oracle (tests/test_gpu_graph_golden.py):

    python tests/golden/gen_graph_proofs.py [name ...]      # -> tests/golden/graph_proofs.json (merges into the existing file)

What this pins: the DEVICE against the in-repo ORACLE at the model shapes (BASELINE configs 1 and 3, one GPT-2 layer).  It does not pin
either against a run of the reference (parity unpinned)."""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.environ.get("GRAPH_PROOFS_OUT", os.path.join(HERE, "graph_proofs.json"))   # a long run writes a file of its own (merged afterwards)
TAU_SEED = 0x51250001


def cases():
    """name -> (nodes, outputs, inputs); the same builders and seeds the GPU tests use"""
    import build_graphs as BG
    return {
        "microgpt": lambda: BG.microgpt(),
        # the model files' own tensors and example inputs (tests/golden/ref_models.npz)
        "microgpt_model": BG.microgpt_model, "nanogpt_model": BG.nanogpt_model,
        "gpt2_layer": lambda: BG.gpt2_layer(),
        "tiny4": lambda: BG.tiny(layers=4),
        "tiny2": lambda: BG.tiny(layers=2),
        # the one-operator graphs bench.py times (T = 2^16): the sizes of the k-sliced accumulation kernel, k_ra_prod_f9 at d = 16 and the
        # 96 KB-LDS Q build of the 64-bit clamp lookup
        "node_einsum": BG.node_einsum, "node_relu": BG.node_relu, "node_mul": BG.node_mul,
        # BASELINE config 4 at size: the 12-layer GPT-2-shaped graph (--trace-only)
        "gpt2": lambda: BG.gpt2(),
    }


def node_hash(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int32).tobytes()).hexdigest()[:16]


def main():
    import build_graphs as BG
    from oracle import graph as OG, orc
    trace_only = "--trace-only" in sys.argv          # execution only (oracle/graph.py:execute): the 12-layer GPT-2-shaped graph, whose oracle PROOF takes hours
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(cases())
    doc = json.load(open(OUT)) if os.path.exists(OUT) else {"generator": "tests/golden/gen_graph_proofs.py", "tau_seed": TAU_SEED,
                                                            "pins": "device vs in-repo oracle (not the reference)", "graphs": {}}
    for name in names:
        nodes, outputs, inputs = cases()[name]()
        nv = BG.max_vars(nodes)
        if trace_only:
            t0 = time.time()
            trace = OG.execute(nodes, inputs)
            trace = trace[0] if isinstance(trace, tuple) else trace
            prev = doc["graphs"].get(name, {})
            doc["graphs"][name] = {
                "n_nodes": len(nodes), "max_vars": nv,
                "input_sha256": [hashlib.sha256(np.ascontiguousarray(x, dtype=np.int32).tobytes()).hexdigest()[:16] for x in inputs],
                "trace": [node_hash(trace[nd["idx"]]) for nd in nodes], "oracle_seconds": round(time.time() - t0, 1),
                "trace_only": "the oracle EXECUTED this graph (per-node hashes); its proof was not computed — hours for 854 nodes at max_num_vars 24. "
                              "device_proof_sha256 / device_state / n_committed / proof_len are the DEVICE's own (tools/record_device_proof.py on an MI355X): a regression pin, not an oracle pin",
                **{k: prev[k] for k in ("device_proof_sha256", "device_state", "n_committed", "proof_len") if k in prev},
            }
            print(name, {k: v for k, v in doc["graphs"][name].items() if k != "trace"}, flush=True)
            with open(OUT, "w") as f:
                json.dump(doc, f, indent=1)
                f.write("\n")
            continue
        tau = orc.random_fr(1, TAU_SEED)[0]
        t0 = time.time()
        srs_h = orc.srs_powers(tau, 1 << nv)
        P = OG.Prover(nodes, outputs, srs_h)
        proof = P.prove(inputs)
        dt = time.time() - t0
        doc["graphs"][name] = {
            "n_nodes": len(nodes), "max_vars": nv, "n_committed": len(P.committed), "proof_len": len(proof),
            "proof_sha256": hashlib.sha256(proof).hexdigest(), "state": P.t.state().hex(),
            "input_sha256": [hashlib.sha256(np.ascontiguousarray(x, dtype=np.int32).tobytes()).hexdigest()[:16] for x in inputs],
            "trace": [node_hash(P.trace[nd["idx"]]) for nd in nodes], "oracle_seconds": round(dt, 1),
        }
        print(name, {k: v for k, v in doc["graphs"][name].items() if k != "trace"}, flush=True)
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
