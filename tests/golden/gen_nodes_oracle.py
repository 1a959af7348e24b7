#!/usr/bin/env python3
"""The oracle side of the two largest cases of tests/test_gpu_nodes.py (the bench's smaller node size, T = 2^12 at scale 2^14), computed in the
build container from the tests' own functions and written to tests/golden/nodes_oracle.json (sha256 of every serialized proof and of the claims,
the final transcript state).

    python tests/golden/gen_nodes_oracle.py"""
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
spec = importlib.util.spec_from_file_location("nodes", os.path.join(os.path.dirname(HERE), "test_gpu_nodes.py"))
T = importlib.util.module_from_spec(spec); spec.loader.exec_module(T)
doc = {"einsum[4-64-1024-14]": T._node_digest(*T.einsum_node_oracle(4, 64, 1024, 14)), "mul[12-14]": T._node_digest(*T.mul_node_oracle(12, 14))}
with open(os.path.join(HERE, "nodes_oracle.json"), "w") as f:
    json.dump(doc, f, indent=0, sort_keys=True)
    f.write("\n")
print("wrote", list(doc))
