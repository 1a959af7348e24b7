"""Forward progress of the lane streams at sizes where the waiting grids exceed the chip's resident workgroups (a fast
lane's pollers must not starve the slow lane): ReLU nodes at T = 2^18 .. 2^20 (three lanes: RaVirtual d = 8, HammingWeight, Booleanity)
and the clamp one-hot checks of an Add node (d = 16), each proved with lane streams and on one stream; the proofs must agree."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jolt_atlas_amd as A  # noqa: E402
from jolt_atlas_amd import node as NODE  # noqa: E402

A.init(0)
out = {}
for log_T in (18, 20):
    rng = np.random.default_rng(log_T)
    tX = A.TensorI32(rng.integers(-(1 << 14), 1 << 14, size=1 << log_T, dtype=np.int64).astype(np.int32))
    r0 = A.random_fr(log_T, 0xE1)
    states = []
    for rep in range(3):
        t = A.Blake2bTranscript(b"relu_node")
        t0 = time.time()
        NODE.prove_relu_node(tX, log_T, r0, t)
        states.append((t.state, time.time() - t0))
    out[log_T] = (len({s for s, _ in states}), states[0][0].hex()[:16], ["%.1f ms" % (1e3 * d) for _, d in states])
    tX.free()
print("stress_big_lanes", "NO_LANE_STREAMS" if os.environ.get("ATLAS_NO_LANE_STREAMS") else "lanes", out)
