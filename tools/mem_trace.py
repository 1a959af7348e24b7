import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
import build_graphs as BG, jolt_atlas_amd as A
from jolt_atlas_amd import graph as GG
A.init(0)
nodes, outputs, inputs = BG.gpt2()
nv = BG.max_vars(nodes)
srs = A.SRS.generate(np.array([0x1234567, 0, 0, 0], dtype=np.uint64), 1 << nv)
print("after SRS", [round(x / 2**30, 2) for x in A.device_memory()])
srs.precompute_range(0, (1 << nv) // 4)
print("after table/4", [round(x / 2**30, 2) for x in A.device_memory()])
G = GG.Graph(nodes, outputs)
for rep in range(2):
    G.prove(srs, inputs)
print("end", [round(x / 2**30, 2) for x in A.device_memory()])
