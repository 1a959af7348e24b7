"""RaVirtual d = 16 with the split product (k_ra_prod16_split) against the chain kernels: the same instance proved in this process with the
library's default switch; run once more under ATLAS_RA_SPLIT_MIN=31 (never split) and compare the sha256 of the proof rows and the times.

    python tools/time_ra_split.py [log_T ...]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import instances as I
A.init(0)
d = 16
for log_t in [int(x) for x in sys.argv[1:]] or [13, 14, 16, 18, 20]:
    rng = np.random.default_rng(log_t)
    H = [rng.integers(0, 16, size=1 << log_t, dtype=np.int32) for _ in range(d)]
    chunks = A.random_fr(d * 4, 6).reshape(d, 4, 4); rc = A.random_fr(log_t, 7)
    ts, sha = [], None
    for _ in range(5):
        inst = I.ra_virtual(H, 4, chunks, rc); A.sync()
        t0 = time.perf_counter(); rows, ch = inst.prove(A.random_fr(1, 1)[0], A.Blake2bTranscript(b"t")); ts.append(time.perf_counter() - t0)
        inst.free()
        h = hashlib.sha256(b"".join(r.tobytes() for r in rows)).hexdigest()[:16]
        assert sha in (None, h); sha = h
    print(f"ra_virtual d16 T=2^{log_t} split_min={os.environ.get('ATLAS_RA_SPLIT_MIN', 'default')}: median {1e3 * float(np.median(ts)):.3f} ms  min {1e3 * min(ts):.3f} ms  proof {sha}", flush=True)
