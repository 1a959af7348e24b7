import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jolt_atlas_amd as A
from jolt_atlas_amd import graph as GG
from oracle import orc
A.init(0)
tau = orc.random_fr(1, 0x51250002)[0]
srs = A.SRS.generate(tau, 1 << 10)
vk = A.HyperKZG.vk_from_trapdoor(tau, srs.download(0, 1)[0])
rng = np.random.default_rng(1)
for dims, scale, lim in (([1, 8], 14, 1 << 11), ([1, 8], 7, 1 << 11), ([2, 4], 7, 1 << 11), ([2, 4], 7, 1 << 6), ([2, 4], 5, 1 << 14)):
    nodes = [{"idx": 0, "op": "Input", "inputs": [], "dims": dims}, {"idx": 1, "op": "Cube", "inputs": [0], "dims": dims, "scale": scale}]
    x = rng.integers(-lim, lim, size=8).astype(np.int32)
    G = GG.Graph(nodes, [1]); got, state, _ = G.prove(srs, [x])
    V = GG.Graph(nodes, [1]); ok, vs = V.verify(vk, [x], G.node_output(1), got)
    print(dims, scale, lim, ok, G.node_output(1)[:4], A.lib.atlas_last_error().decode() if not ok else "")
