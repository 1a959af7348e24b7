"""RaVirtual d = 16 at T = 2^LOG_T (default 20): prove wall clock (median of 5), as tools/time_components.py measures it; run with
ATLAS_RA_LAZY_LOG=31 for the gathered-rows path (A/B of the lazy rounds 0-1, csrc/ra.hip)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import instances as I
A.init(0)
for lt in [int(x) for x in os.environ.get("LOG_T", "18,20").split(",")]:
    rng = np.random.default_rng(1)
    H = [rng.integers(0, 16, size=1 << lt, dtype=np.int32) for _ in range(16)]
    chunks = A.random_fr(64, 6).reshape(16, 4, 4); rc = A.random_fr(lt, 7)
    ts, st = [], set()
    for rep in range(6):
        inst = I.ra_virtual(H, 4, chunks, rc)
        tr = A.Blake2bTranscript(b"t")
        A.sync(); t0 = time.perf_counter()
        inst.prove(A.random_fr(1, 1)[0], tr)
        ts.append(time.perf_counter() - t0)
        st.add(bytes(tr.state))
        inst.free()
    assert len(st) == 1
    print("ra_virtual d16 T=2^%d: %.3f ms (median of 5 after a warm-up)  state %s" % (lt, 1e3 * float(np.median(ts[1:])), bytes(tr.state).hex()[:16]), flush=True)
