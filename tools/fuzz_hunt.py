"""One-off bug hunt: N more seeds of each draw of tests/test_gpu_graph_fuzz.py (narrow / wide / tiny shapes) through the device prover, the oracle
composition and the verifier.  usage: python tools/fuzz_hunt.py 40 [seed offset]   (on a GPU box)"""
import sys, os, numpy as np, traceback
sys.path.insert(0, "/root/repo")
import importlib.util
spec = importlib.util.spec_from_file_location("tf", "/root/repo/tests/test_gpu_graph_fuzz.py"); tf = importlib.util.module_from_spec(spec); spec.loader.exec_module(tf)
import jolt_atlas_amd as A
A.init(0)
class AtlasNS: pass
atlas = A
bad = 0
off = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for kind, base, kw in (("narrow", 20000, {}), ("wide", 30000, dict(steps=11, wide=True)), ("tiny", 40000, dict(steps=8, sizes=(1, 2)))):
    for seed in range(int(sys.argv[1])):
        nodes, outputs, inputs = tf.random_chain(base + off + seed, **kw)
        if any(nd["op"] == "Einsum" and (int(np.prod(nd["dims"])) == 1 or nd["shape"][1] == 1) for nd in nodes): continue
        if any(nd["op"] in ("ReLU", "Clamp", "Tanh", "Erf", "Sigmoid", "Sin", "Cos", "Rsqrt", "Div") and int(np.prod(nd["dims"])) == 1 for nd in nodes): continue
        try:
            tf._run(atlas, nodes, outputs, inputs, seed)
        except BaseException as e:
            bad += 1
            print("FAIL", kind, seed, repr(e)[:300], [(nd["op"], nd["dims"]) for nd in nodes if nd["op"] != "Constant"], flush=True)
print("done, failures:", bad)
