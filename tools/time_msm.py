import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
A.init(0)
n = 1 << 22
srs = A.SRS.generate(A.random_fr(1, 1)[0], n)
p = A.MultilinearPolynomial.from_fr(A.random_fr(n, 2))
for _ in range(4):
    srs.msm(p)
A.sync()
