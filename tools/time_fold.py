import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import jolt_atlas_amd as A
A.init(0)
k, n = 768, 65536
rng = np.random.default_rng(0)
B = rng.integers(-(1 << 14), 1 << 14, size=(k, n), dtype=np.int64).astype(np.int32)
Bp = np.zeros((1024, n), dtype=np.int32); Bp[:k] = B        # k padded to a power of two rows
t = A.TensorI32(Bp)
eq = A.EqPolynomial.evals(A.random_fr(16, 1))
for _ in range(2):
    A.fold_rows(t, eq).free()
t0 = time.perf_counter()
for _ in range(10):
    A.fold_rows(t, eq).free()
dt = (time.perf_counter() - t0) / 10
print("fold_rows 1024x65536 i32 (268 MB): %.3f ms  %.0f GB/s" % (dt * 1e3, Bp.nbytes / dt / 1e9))
Am = rng.integers(-(1 << 14), 1 << 14, size=(65536, 1024), dtype=np.int64).astype(np.int32)
t2 = A.TensorI32(Am); eq2 = A.EqPolynomial.evals(A.random_fr(16, 2))
for _ in range(2):
    A.fold_cols(t2, eq2).free()
t0 = time.perf_counter()
for _ in range(10):
    A.fold_cols(t2, eq2).free()
dt = (time.perf_counter() - t0) / 10
print("fold_cols 65536x1024 i32 (268 MB): %.3f ms  %.0f GB/s" % (dt * 1e3, Am.nbytes / dt / 1e9))
