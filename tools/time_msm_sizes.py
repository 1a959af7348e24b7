import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
A.init(0)
srs = A.SRS.generate(A.random_fr(1, 1)[0], 1 << 20)
for ln in range(1, 21):
    p = A.MultilinearPolynomial.from_fr(A.random_fr(1 << ln, ln))
    srs.msm(p); A.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); srs.msm(p); ts.append(time.perf_counter() - t0)
    print(ln, round(1e3 * float(np.median(ts)), 3))
    p.free()
