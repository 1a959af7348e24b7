"""Einsum node (16x1024 . 1024x4096, S = 14) wall clock and stage split; ATLAS_TRACE=1 prints the phases of the one-hot checks."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import node as NODE
A.init(0)
rng = np.random.default_rng(14)
m, k, n, S = [int(x) for x in os.environ.get("DIMS", "16,1024,4096,14").split(",")]
tA = A.TensorI32(rng.integers(-(1 << 14), 1 << 14, size=(m, k), dtype=np.int64).astype(np.int32))
tB = A.TensorI32(rng.integers(-(1 << 14), 1 << 14, size=(k, n), dtype=np.int64).astype(np.int32))
r0 = A.random_fr(16, 0xE1)
names = ("witness", "clamp_ps_shout", "ra_checks", "matmul", "range_check", "rem_ra_checks")
states = set()
for rep in range(int(os.environ.get("REPS", "4"))):
    tn = A.Blake2bTranscript(b"einsum_node")
    A.sync(); t0 = time.perf_counter()
    pf, cl, st = NODE.prove_einsum_node(tA, tB, m, k, n, S, r0, tn)
    A.sync(); dt = time.perf_counter() - t0
    states.add(tn.state)
    if os.environ.get("PRINT_STATE"): print("state", bytes(tn.state).hex()[:32] if not isinstance(tn.state, str) else tn.state[:32])
    print("node %.3f ms  " % (1e3 * dt) + "  ".join("%s %.2f" % (a, b) for a, b in zip(names, st)), flush=True)
assert len(states) == 1
