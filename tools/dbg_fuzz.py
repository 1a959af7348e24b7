import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("tf", os.path.join(ROOT, "tests/test_gpu_graph_fuzz.py")); tf = importlib.util.module_from_spec(spec); spec.loader.exec_module(tf)
import jolt_atlas_amd as A
from jolt_atlas_amd import graph as GG
from oracle import orc
A.init(0)
for seed in [int(x) for x in sys.argv[1:]]:
    nodes, outputs, inputs = tf.random_chain(1000 + seed)
    nv = 4 + max(int(np.log2(max(int(np.prod(nd["dims"])), 1))) for nd in nodes)
    tau = orc.random_fr(1, 0x51250002)[0]
    srs = A.SRS.generate(tau, 1 << nv)
    G = GG.Graph(nodes, outputs)
    try:
        got, state, _ = G.prove(srs, inputs)
    except A.AtlasError as e:
        print(seed, "prove:", e); continue
    vk = A.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
    V = GG.Graph(nodes, outputs)
    ok, vs = V.verify(vk, inputs, G.node_output(outputs[0]), got)
    print(seed, "verify:", ok, A.lib.atlas_last_error().decode() if not ok else "")
