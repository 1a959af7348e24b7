"""PS-Shout ReLU-32 read-raf at T = 2^20 under rocprofv3 (kernel time vs wall-clock per prove)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import instances as I
A.init(0)
log_t = int(os.environ.get("LOG_T", "20"))
rng = np.random.default_rng(1)
idx = rng.integers(0, 1 << 32, size=1 << log_t, dtype=np.uint64)
rn = A.random_fr(log_t, 3); gam = A.random_fr(1, 4)[0]
for _ in range(3):
    t0 = time.perf_counter(); inst = I.ps_shout_relu(idx, 32, rn, gam); A.sync(); t1 = time.perf_counter()
    inst.prove(A.random_fr(1, 1)[0], A.Blake2bTranscript(b"t")); t2 = time.perf_counter()
    print("new ms", 1e3 * (t1 - t0), "prove ms", 1e3 * (t2 - t1))
    inst.free()
