import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import jolt_atlas_amd as A
A.init(0)
ell = int(os.environ.get("LOG_N", "24")); n = 1 << ell
srs = A.SRS.generate(A.random_fr(1, 1)[0], n)
p = A.MultilinearPolynomial.from_fr(A.random_fr(n, 3))
rng = np.random.default_rng(ell)
pt = [int(rng.integers(0, 1 << 62)) << 66 | int(rng.integers(0, 1 << 62)) for _ in range(ell)]
srs.precompute()
A.HyperKZG.open(srs, p, pt, A.Blake2bTranscript(b"t")); A.sync()
os.environ["ATLAS_TRACE"] = "1"
t0 = time.perf_counter(); A.HyperKZG.open(srs, p, pt, A.Blake2bTranscript(b"t")); A.sync(); print("open ms", 1e3 * (time.perf_counter() - t0))
