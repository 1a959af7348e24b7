"""One RaVirtual (d = 16) and one Booleanity proof at T = 2^LOG_T for a dispatch trace (rocprofv3 --kernel-trace)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import instances as I
A.init(0)
n = int(os.environ.get("LOG_T", "10"))
rng = np.random.default_rng(1)
H = [rng.integers(0, 16, size=1 << n, dtype=np.int32) for _ in range(16)]
ch = A.random_fr(64, 3).reshape(16, 4, 4); rc = A.random_fr(n, 4)
for rep in range(3):
    inst = I.ra_virtual(H, 4, ch, rc); A.sync()
    t0 = time.perf_counter(); inst.prove(np.zeros(4, dtype=np.uint64), A.Blake2bTranscript(b"t")); dt = time.perf_counter() - t0
    inst.free()
    print("ra_virtual d16 2^%d: %.1f us/round" % (n, 1e6 * dt / n), flush=True)
    inst = I.booleanity(np.zeros((16, 16, 4), dtype=np.uint64), H, 4, A.random_fr(16, 5), A.random_fr(4, 6), rc); A.sync()
    t0 = time.perf_counter(); inst.prove(np.zeros(4, dtype=np.uint64), A.Blake2bTranscript(b"t")); dt = time.perf_counter() - t0
    inst.free()
    print("booleanity d16 2^%d: %.1f us/round" % (n, 1e6 * dt / (n + 4)), flush=True)
