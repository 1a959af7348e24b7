import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
A.init(0)
n = 1 << 22
rng = np.random.default_rng(1)
srs = A.SRS.generate(A.random_fr(1, 1)[0], n)
u8 = rng.integers(0, 256, size=n, dtype=np.uint8)
i14 = A.MultilinearPolynomial.from_i32(rng.integers(-(1 << 14), 1 << 14, size=n, dtype=np.int32))
u20 = rng.integers(0, 1 << 20, size=n, dtype=np.uint32)
for _ in range(3):
    srs.msm_small(u8)
for _ in range(3):
    srs.msm(i14)
for _ in range(3):
    srs.msm_small(u20)
A.sync()
