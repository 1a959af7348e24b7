"""MSM wall clock on skewed LargeScalars inputs at LOG_N (default 22): uniform, all ones, values < 256, 64-bit values."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
A.init(0)
ln = int(os.environ.get("LOG_N", "22"))
n = 1 << ln
srs = A.SRS.generate(A.random_fr(1, 1)[0], n)
rng = np.random.default_rng(1)
def fr_small(v):
    a = np.zeros((n, 4), dtype=np.uint64); a[:, 0] = v
    return A.MultilinearPolynomial.from_fr(A.to_montgomery(a)) if hasattr(A, "to_montgomery") else None
cases = {"uniform": A.MultilinearPolynomial.from_fr(A.random_fr(n, 3))}
try:
    from oracle import orc
    cases["all ones"] = A.MultilinearPolynomial.from_fr(np.repeat(orc.from_ints([1]), n, axis=0))
    small = orc.from_ints(list(range(256)))
    cases["< 256"] = A.MultilinearPolynomial.from_fr(small[rng.integers(0, 256, size=n)])
except Exception as e:
    print("oracle not available:", e)
for tab in (False, True):
    if tab:
        srs.precompute()
    for name, p in cases.items():
        srs.msm(p); A.sync()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); srs.msm(p); ts.append(time.perf_counter() - t0)
        print("2^%d %-10s %-14s %8.3f ms" % (ln, name, "table" if tab else "variable-base", 1e3 * float(np.median(ts))), flush=True)
