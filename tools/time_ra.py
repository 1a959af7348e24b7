import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import instances as I
A.init(0)
log_t, d = 20, 16
rng = np.random.default_rng(1)
H = [rng.integers(0, 16, size=1 << log_t, dtype=np.int32) for _ in range(d)]
chunks = A.random_fr(d * 4, 6).reshape(d, 4, 4); rc = A.random_fr(log_t, 7)
gam = A.random_fr(d, 8); radr = A.random_fr(4, 9)
for _ in range(2):
    inst = I.ra_virtual(H, 4, chunks, rc)
    t0 = time.perf_counter(); inst.prove(A.random_fr(1, 1)[0], A.Blake2bTranscript(b"t")); print("ra", time.perf_counter() - t0)
    inst.free()
    inst = I.booleanity(np.zeros((d, 16, 4), dtype=np.uint64), H, 4, gam, radr, rc)
    t0 = time.perf_counter(); inst.prove(np.zeros(4, dtype=np.uint64), A.Blake2bTranscript(b"t")); print("bool", time.perf_counter() - t0)
    inst.free()
