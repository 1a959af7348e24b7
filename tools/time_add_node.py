"""One Add node (the clamp lookup + its one-hot checks: what 137 of the nanoGPT-shaped graph's 302 nodes pay) at T = seq x d; ATLAS_TRACE=1
prints the phases of the lookup and of the one-hot checks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import graph as GG

A.init(0)
m, n = int(os.environ.get("M", "64")), int(os.environ.get("N", "64"))
rng = np.random.default_rng(3)
nodes = [
    {"idx": 0, "op": "Input", "inputs": [], "dims": [m, n]},
    {"idx": 1, "op": "Constant", "inputs": [], "dims": [m, n], "data": rng.integers(-1 << 20, 1 << 20, size=m * n).astype(np.int32)},
    {"idx": 2, "op": "Add", "inputs": [0, 1], "dims": [m, n]},
]
x = rng.integers(-1 << 20, 1 << 20, size=m * n).astype(np.int32)
nv = 4 + int(np.log2(m * n))
srs = A.SRS.generate(np.array([0x1234567, 0, 0, 0], dtype=np.uint64), 1 << nv)
G = GG.Graph(nodes, [2])
for rep in range(int(os.environ.get("REPS", "3"))):
    t0 = time.time()
    proof, state, tm = G.prove(srs, [x])
    print("prove %.3f ms" % ((time.time() - t0) * 1e3), {k: round(v, 3) for k, v in tm.items() if k.endswith("_ms")}, state.hex()[:16], flush=True)
