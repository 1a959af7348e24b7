"""2^22-point MSM wall clock (median of 7); knobs through the environment: ATLAS_MSM_C, ATLAS_MSM_SEG."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
A.init(0)
ln = int(os.environ.get("LOG_N", "22"))
srs = A.SRS.generate(A.random_fr(1, 1)[0], 1 << ln)
p = A.MultilinearPolynomial.from_fr(A.random_fr(1 << ln, 3))
srs.msm(p); A.sync()
ts = []
for _ in range(7):
    t0 = time.perf_counter(); srs.msm(p); ts.append(time.perf_counter() - t0)
print("msm 2^%d" % ln, "seg", os.environ.get("ATLAS_MSM_SEG", "default"), round(1e3 * float(np.median(ts)), 3), "ms")
