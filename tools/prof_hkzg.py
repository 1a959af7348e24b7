"""HyperKZG::open at 2^20 under rocprofv3: `rocprofv3 --kernel-trace --stats -d out -- python tools/prof_hkzg.py`."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
A.init(0)
rng = np.random.default_rng(1)
ell = int(os.environ.get("ELL", "20"))
srs = A.SRS.generate(A.random_fr(1, 1)[0], 1 << ell)
P = A.MultilinearPolynomial.from_fr(A.random_fr(1 << ell, 32))
pt = [int.from_bytes(rng.bytes(16), "little") for _ in range(ell)]
for _ in range(5):
    t0 = time.perf_counter(); A.HyperKZG.open(srs, P, pt, A.Blake2bTranscript(b"t")); A.sync()
    print("open ms", 1e3 * (time.perf_counter() - t0))
